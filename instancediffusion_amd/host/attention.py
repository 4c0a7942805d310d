"""Host mirror of ``ldm/modules/attention.py`` -- parameter containers only (the HIP engine does the math).

Class names and attribute names follow the reference so that
  * state-dict keys match (``...transformer_blocks.0.{attn1,attn2,ff,fuser,norm1-3}...``),
  * ``utils/model.py:78-81 set_alpha_scale`` finds fusers by EXACT type and writes ``.scale``.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .params import Affine, Conv, Dense, Slots


class FeedForward(nn.Module):
    """GEGLU feed-forward (attention.py:36-63): net.0.proj [8C, C], net.2 [C, 4C]."""

    def __init__(self, dim: int, mult: int = 4):
        super().__init__()
        inner = dim * mult
        geglu = nn.Module()
        geglu.proj = Dense(dim, inner * 2)
        self.net = Slots({0: geglu, 2: Dense(inner, dim)})


class _AttnParams(nn.Module):
    def __init__(self, query_dim: int, key_dim: int, heads: int, dim_head: int):
        super().__init__()
        inner = heads * dim_head
        self.heads, self.dim_head = heads, dim_head
        self.scale = dim_head ** -0.5
        self.to_q = Dense(query_dim, inner, bias=False)
        self.to_k = Dense(key_dim, inner, bias=False)
        self.to_v = Dense(key_dim, inner, bias=False)
        self.to_out = Slots({0: Dense(inner, query_dim)})


class SelfAttention(_AttnParams):
    """attention.py:160-282."""

    def __init__(self, query_dim, heads=8, dim_head=64, dropout=0.0, efficient_attention=False):
        super().__init__(query_dim, query_dim, heads, dim_head)
        self.efficient_attention = efficient_attention


class CrossAttention(_AttnParams):
    """attention.py:98-157."""

    def __init__(self, query_dim, key_dim, value_dim, heads=8, dim_head=64, dropout=0, efficient_attention=False):
        assert key_dim == value_dim
        super().__init__(query_dim, key_dim, heads, dim_head)
        self.efficient_attention = efficient_attention


class GatedSelfAttentionDense(nn.Module):
    """UniFusion fuser (attention.py:285-311).  ``scale`` is the externally-set alpha gate."""

    def __init__(self, query_dim, context_dim, n_heads, d_head, efficient_attention=False):
        super().__init__()
        self.linear = Dense(context_dim, query_dim)
        self.attn = SelfAttention(query_dim, heads=n_heads, dim_head=d_head, efficient_attention=efficient_attention)
        self.ff = FeedForward(query_dim)
        self.norm1 = Affine(query_dim)
        self.norm2 = Affine(query_dim)
        self.alpha_attn = nn.Parameter(torch.tensor(0.0))
        self.alpha_dense = nn.Parameter(torch.tensor(0.0))
        self.scale = 1


class BasicTransformerBlock(nn.Module):
    """attention.py:314-338."""

    def __init__(self, query_dim, key_dim, value_dim, n_heads, d_head, fuser_type, use_checkpoint=True,
                 efficient_attention=False):
        super().__init__()
        self.attn1 = SelfAttention(query_dim, heads=n_heads, dim_head=d_head, efficient_attention=efficient_attention)
        self.ff = FeedForward(query_dim)
        self.attn2 = CrossAttention(query_dim, key_dim, value_dim, heads=n_heads, dim_head=d_head,
                                    efficient_attention=efficient_attention)
        self.norm1 = Affine(query_dim)
        self.norm2 = Affine(query_dim)
        self.norm3 = Affine(query_dim)
        self.fuser = GatedSelfAttentionDense(query_dim, key_dim, n_heads, d_head, efficient_attention)


class SpatialTransformer(nn.Module):
    """attention.py:341-379: GN(eps 1e-6) -> 1x1 conv -> block -> 1x1 conv (zero-init) -> + x_in."""

    def __init__(self, in_channels, key_dim, value_dim, n_heads, d_head, depth=1, fuser_type=None,
                 use_checkpoint=True, efficient_attention=False):
        super().__init__()
        assert depth == 1, "reference configs use transformer_depth=1"
        self.in_channels = in_channels
        self.n_heads, self.d_head = n_heads, d_head
        qd = n_heads * d_head
        self.norm = Affine(in_channels)
        self.proj_in = Conv(in_channels, qd, 1)
        self.transformer_blocks = nn.ModuleList(
            [BasicTransformerBlock(qd, key_dim, value_dim, n_heads, d_head, fuser_type,
                                   use_checkpoint=use_checkpoint, efficient_attention=efficient_attention)])
        self.proj_out = Conv(qd, in_channels, 1, zero=True)


# ---- masked gated self-attention (attention.py:187-255), SURVEY.md §8 row f-4 ------------------------------------
ALWAYS = -2 ** 31            # bit 31 as an int32: set in every query word and in every unconditionally visible key


def visibility_words(att_masks: torch.Tensor, n_groups: int = 4, n_seg_tokens: int = 64, force_masked: bool = False):
    """The reference's dense [B, 1, N, N] visibility mask of the fuser attention, as 32-bit membership words.

    ``att_masks`` [B, n_objs <= 30, h, w] (binary; ``utils/input.py:34-37``).  Visual token i carries one bit per
    instance whose box contains it.  Returns int32 tensors
        qbits  [B, h*w]    instance bits | bit 31,
        kbits0 [B, h*w]    instance bits                       (visual keys),
        kbits1 [B, ceil64(4 n_objs + 64)]  per grounding token: box token o -> bit o, point / scribble tokens -> all
                           ones, mask token o -> bit o, the 64 seg tokens -> all ones (padding -> all ones),
    such that query q sees key k iff ``(qbits[q] & kbits[k]) != 0`` or k is q itself -- exactly ``mask > 0`` of
    attention.py:206-253 for the visual query rows (the only rows attention.py:308 keeps).  The reference decides PER
    CALL whether to mask at all (``torch.sum(att_masks) > 0`` over the whole batch tensor, :200): an all-zero tensor (the
    null grounding input of the unconditional branch) gives all-ones words = no mask; otherwise every sample is masked,
    and a sample without any instance pixel sees only itself, the point / scribble tokens and the seg tokens.
    ``force_masked``: the caller already took the per-call decision on a LARGER tensor of which ``att_masks`` is a row
    subset (rank-sharded MIS, host/samplers.py) and it was "masked" -- so mask even if these rows are all zero."""
    B, n_objs = att_masks.shape[0], att_masks.shape[1]
    assert n_objs <= 31, "one bit per instance, bit 31 reserved"
    dev = att_masks.device
    m = (att_masks.reshape(B, n_objs, -1) > 0)
    weights = (torch.ones(n_objs, dtype=torch.int64, device=dev) << torch.arange(n_objs, device=dev)).view(1, n_objs, 1)
    inst = (m.to(torch.int64) * weights).sum(1)                                        # [B, hw], < 2^31
    unmasked = ((~m.any()) & (not force_masked)).expand(B)                             # [B], one decision per call
    full = torch.full_like(inst, 0xFFFFFFFF)
    q64 = torch.where(unmasked[:, None], full, inst | 0x80000000)
    k64 = torch.where(unmasked[:, None], full, inst)
    n1 = n_groups * n_objs + n_seg_tokens
    ld1 = (n1 + 63) // 64 * 64
    k1 = torch.full((B, ld1), 0xFFFFFFFF, dtype=torch.int64, device=dev)
    obj_bits = (torch.ones(n_objs, dtype=torch.int64, device=dev) << torch.arange(n_objs, device=dev))
    k1[:, :n_objs] = obj_bits                                                          # box tokens
    k1[:, (n_groups - 1) * n_objs:n_groups * n_objs] = obj_bits                        # mask (polygon) tokens
    k1 = torch.where(unmasked[:, None], torch.full_like(k1, 0xFFFFFFFF), k1)

    def i32(t):                                                                        # two's-complement reinterpretation
        return torch.where(t >= 2 ** 31, t - 2 ** 32, t).to(torch.int32).contiguous()
    return i32(q64), i32(k64), i32(k1)
