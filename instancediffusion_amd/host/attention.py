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
