"""CPU fp32 ORACLE for the InstanceDiffusion sampling hot path.  TEST INFRASTRUCTURE ONLY.

This file is a from-scratch functional restatement (plain PyTorch, CPU, fp32) of the reference
algorithm for the path named in BASELINE.json:north_star.  It is the *checker*: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.  Nothing under
``instancediffusion_amd/`` imports it, and the product path fails loudly without its HIP library.

Parity pin: the reference ships no tests / golden vectors -> "parity unpinned" by the
reference itself.  This oracle is pinned instead against outputs of the UNMODIFIED reference modules
run in the builder container (``oracle/make_golden.py`` -> ``tests/golden/*.pt``); see
``tests/test_oracle_golden.py``.

Each function cites the reference file:line (relative to the reference repo root) it follows.
All tensors use the reference's layouts (NCHW activations, [B, N, C] tokens) and state-dict key names.
"""
from __future__ import annotations

import math
from typing import Callable, Dict, List, Mapping, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

SD = Mapping[str, torch.Tensor]

# SD-1.5 InstanceDiffusion UNet hyper-parameters (configs/test_box.yaml:9-24)
DEFAULT_CFG = dict(
    in_channels=4, out_channels=4, model_channels=320, attention_resolutions=(4, 2, 1),
    num_res_blocks=2, channel_mult=(1, 2, 4, 4), num_heads=8, context_dim=768,
    # UniFusion (configs/test_box.yaml:26-40)
    in_dim=768, out_dim=768, mid_dim=3072,
    test_drop_boxes=False, test_drop_points=False, test_drop_scribbles=True, test_drop_masks=True,
)


# ------------------------------------------------------------------------------------------------
# small helpers
# ------------------------------------------------------------------------------------------------
def _lin(sd: SD, p: str, x: torch.Tensor, bias: bool = True) -> torch.Tensor:
    return F.linear(x, sd[p + ".weight"], sd[p + ".bias"] if bias else None)


def _conv(sd: SD, p: str, x: torch.Tensor, stride: int = 1, padding: int = 1, groups: int = 1) -> torch.Tensor:
    return F.conv2d(x, sd[p + ".weight"], sd[p + ".bias"], stride=stride, padding=padding, groups=groups)


def _gn32(sd: SD, p: str, x: torch.Tensor, eps: float) -> torch.Tensor:
    """GroupNorm32 forces fp32 statistics (util.py:223-226); Normalize uses eps=1e-6 (attention.py:75-76)."""
    return F.group_norm(x.float(), 32, sd[p + ".weight"], sd[p + ".bias"], eps).type(x.dtype)


def _ln(sd: SD, p: str, x: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], eps)


def silu(x: torch.Tensor) -> torch.Tensor:
    return x * torch.sigmoid(x)


def gelu_erf(x: torch.Tensor) -> torch.Tensor:
    """Exact (erf) GELU -- F.gelu default, attention.py:43."""
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


# ------------------------------------------------------------------------------------------------
# util.py
# ------------------------------------------------------------------------------------------------
def timestep_embedding(t: torch.Tensor, dim: int, max_period: float = 10000.0) -> torch.Tensor:
    """util.py:160-180 -- [cos(t f_k), sin(t f_k)], f_k = exp(-ln(max_period) k / half); cos FIRST."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


def fourier_embed(x: torch.Tensor, num_freqs: int = 16, temperature: float = 100.0) -> torch.Tensor:
    """util.py:12-26 -- for each freq f_j = T^(j/n): sin(f_j x), cos(f_j x), concatenated on the last dim."""
    bands = temperature ** (torch.arange(num_freqs) / num_freqs)
    parts = []
    for f in bands:
        parts.append(torch.sin(f * x))
        parts.append(torch.cos(f * x))
    return torch.cat(parts, dim=-1)


# ------------------------------------------------------------------------------------------------
# attention.py
# ------------------------------------------------------------------------------------------------
def mha(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, heads: int, mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """softmax(Q K^T / sqrt(d)) V per head (attention.py:120-157, 174-186, 257-282).  q [B,N,C], k/v [B,M,C] ->
    [B,N,C].  ``mask`` [B,1,N,M]: scores where mask <= 0 are filled with -inf (non-efficient path, attention.py:276-277)."""
    B, N, C = q.shape
    M = k.shape[1]
    d = C // heads
    qh = q.view(B, N, heads, d).permute(0, 2, 1, 3)
    kh = k.view(B, M, heads, d).permute(0, 2, 1, 3)
    vh = v.view(B, M, heads, d).permute(0, 2, 1, 3)
    if mask is None:
        # the reference's efficient path IS this call (attention.py:140,143,263,266); on CPU it is also ~2x faster than the
        # explicit form below, which matters only for bench.py's cpu_baseline (profiles/r02_cpu_reference_vs_port.json)
        o = F.scaled_dot_product_attention(qh, kh, vh)
    else:
        s = torch.matmul(qh, kh.transpose(-1, -2)) * (d ** -0.5)
        s = s.masked_fill(mask <= 0.0, float("-inf"))
        o = torch.matmul(torch.softmax(s, dim=-1), vh)
    return o.permute(0, 2, 1, 3).reshape(B, N, C)


def fuser_attention_mask(att_masks: torch.Tensor, n_tokens: int) -> Optional[torch.Tensor]:
    """The instance-visibility mask of the masked gated self-attention, attention.py:187-255 (reached only with
    ``efficient_attention=False``).  att_masks [B, n_objs, h, w] (binary, from utils/input.py:34-37); the sequence is
    [h*w visual tokens | n_objs box | n_objs point | n_objs scribble | n_objs mask | 64 seg tokens].  Returns
    [B, 1, N, N] or None when the reference would not mask (N - 4 n_objs - 64 != 64*64, or an all-zero mask)."""
    B, n_objs = att_masks.shape[0], att_masks.shape[1]
    N = n_tokens
    if N - n_objs * 4 - 64 != 64 * 64:                               # :195 "brute-force" resolution check
        return None
    if not float(att_masks.sum()) > 0.0:                             # :200
        return None
    w_h = att_masks.shape[2] * att_masks.shape[3]
    m = att_masks.reshape(B, n_objs, w_h).float()
    mask = torch.ones(B, 1, N, N)
    # :211-238 -- two visual tokens see each other iff at least one instance box contains both (the reference's
    # sum_o m_o m_o^T >= 1, written as one matmul); tokens outside every box see no other visual token
    ind = torch.einsum("bow,bov->bwv", m, m)
    mask[:, 0, :w_h, :w_h] = (ind >= 1.0).float()
    # :243-248 -- box tokens (first n_objs) and mask tokens (last n_objs of the 4 groups) are tied to their instance's
    # box in both directions; point and scribble tokens (the two middle groups) and the 64 seg tokens stay visible
    rep = m.repeat(1, 4, 1)                                          # [B, 4 n_objs, w_h]
    mask[:, 0, w_h:N - 64, :w_h] = rep
    mask[:, 0, w_h + n_objs:w_h + n_objs * 3, :w_h] = 1
    mask[:, 0, :w_h, w_h:N - 64] = rep.transpose(1, 2)
    mask[:, 0, :w_h, w_h + n_objs:w_h + n_objs * 3] = 1
    return mask + torch.eye(N).view(1, 1, N, N) * 1e-9               # :251-252: a token always sees itself


def self_attention(sd: SD, p: str, x: torch.Tensor, heads: int, mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """SelfAttention.forward, attention.py:174-282 (efficient path when ``mask`` is None, else :268-281)."""
    q = _lin(sd, p + ".to_q", x, bias=False)
    k = _lin(sd, p + ".to_k", x, bias=False)
    v = _lin(sd, p + ".to_v", x, bias=False)
    return _lin(sd, p + ".to_out.0", mha(q, k, v, heads, mask))


def cross_attention(sd: SD, p: str, x: torch.Tensor, ctx: torch.Tensor, heads: int) -> torch.Tensor:
    """CrossAttention.forward, attention.py:120-157."""
    q = _lin(sd, p + ".to_q", x, bias=False)
    k = _lin(sd, p + ".to_k", ctx, bias=False)
    v = _lin(sd, p + ".to_v", ctx, bias=False)
    return _lin(sd, p + ".to_out.0", mha(q, k, v, heads))


def feed_forward(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """FeedForward with GEGLU, attention.py:36-63: chunk(2) -> a * gelu(gate); then Linear(4C -> C)."""
    h = _lin(sd, p + ".net.0.proj", x)
    a, gate = h.chunk(2, dim=-1)
    return _lin(sd, p + ".net.2", a * gelu_erf(gate))


def gated_self_attention(sd: SD, p: str, x: torch.Tensor, objs: torch.Tensor, heads: int, scale: float,
                         att_masks: Optional[torch.Tensor] = None) -> torch.Tensor:
    """GatedSelfAttentionDense.forward, attention.py:304-311.  ``att_masks`` [B,n_objs,h,w]: masked variant
    (efficient_attention=False and ``grounding_input['att_masks']`` present and boxes not dropped)."""
    n_vis = x.shape[1]
    o = _lin(sd, p + ".linear", objs)
    seq = torch.cat([x, o], dim=1)
    mask = None if att_masks is None else fuser_attention_mask(att_masks, seq.shape[1])
    a = self_attention(sd, p + ".attn", _ln(sd, p + ".norm1", seq), heads, mask)
    x = x + scale * torch.tanh(sd[p + ".alpha_attn"]) * a[:, :n_vis]
    x = x + scale * torch.tanh(sd[p + ".alpha_dense"]) * feed_forward(sd, p + ".ff", _ln(sd, p + ".norm2", x))
    return x


def transformer_block(sd: SD, p: str, x, ctx, objs, heads: int, scale: float, att_masks=None):
    """BasicTransformerBlock._forward, attention.py:333-338."""
    x = self_attention(sd, p + ".attn1", _ln(sd, p + ".norm1", x), heads) + x
    x = gated_self_attention(sd, p + ".fuser", x, objs, heads, scale, att_masks)
    x = cross_attention(sd, p + ".attn2", _ln(sd, p + ".norm2", x), ctx, heads) + x
    x = feed_forward(sd, p + ".ff", _ln(sd, p + ".norm3", x)) + x
    return x


def spatial_transformer(sd: SD, p: str, x, ctx, objs, heads: int, scale: float, att_masks=None):
    """SpatialTransformer.forward, attention.py:366-379."""
    b, c, h, w = x.shape
    x_in = x
    y = _gn32(sd, p + ".norm", x, 1e-6)
    y = _conv(sd, p + ".proj_in", y, padding=0)
    y = y.permute(0, 2, 3, 1).reshape(b, h * w, c)
    y = transformer_block(sd, p + ".transformer_blocks.0", y, ctx, objs, heads, scale, att_masks)
    y = y.reshape(b, h, w, c).permute(0, 3, 1, 2)
    y = _conv(sd, p + ".proj_out", y, padding=0)
    return y + x_in


# ------------------------------------------------------------------------------------------------
# openaimodel.py
# ------------------------------------------------------------------------------------------------
def fourier_filter(x_in: torch.Tensor, threshold: int, scale: torch.Tensor) -> torch.Tensor:
    """Fourier_filter, openaimodel.py:25-48 (FFT-based, as the reference does it)."""
    x = x_in
    B, C, H, W = x.shape
    if (W & (W - 1)) != 0 or (H & (H - 1)) != 0:
        x = x.to(torch.float32)
    xf = torch.fft.fftshift(torch.fft.fftn(x, dim=(-2, -1)), dim=(-2, -1))
    mask = torch.ones((B, C, H, W))
    cr, cc = H // 2, W // 2
    mask[..., cr - threshold:cr + threshold, cc - threshold:cc + threshold] = scale
    xf = xf * mask
    out = torch.fft.ifftn(torch.fft.ifftshift(xf, dim=(-2, -1)), dim=(-2, -1)).real
    return out.to(x_in.dtype)


def lowfreq_4bin(x: torch.Tensor) -> torch.Tensor:
    """The exact identity the HIP ScaleU kernel uses:
    Fourier_filter(x, 1, s) == x + (s-1) * lowfreq_4bin(x), where lowfreq is the real part of the inverse DFT
    restricted to the 2x2 frequency window (u,v) in {-1,0}^2.  Verified against ``fourier_filter`` in tests."""
    B, C, H, W = x.shape
    hh = torch.arange(H, dtype=torch.float64)
    ww = torch.arange(W, dtype=torch.float64)
    xd = x.double()
    out = torch.zeros_like(xd)
    for u in (-1, 0):
        for v in (-1, 0):
            ph = 2.0 * math.pi * (u * hh[:, None] / H + v * ww[None, :] / W)  # [H,W]
            # X[u,v] = sum x * exp(-i ph)
            xr = (xd * torch.cos(ph)).sum(dim=(-2, -1), keepdim=True)
            xi = -(xd * torch.sin(ph)).sum(dim=(-2, -1), keepdim=True)
            # Re{ X exp(+i ph) } = xr cos(ph) - xi sin(ph)
            out = out + (xr * torch.cos(ph) - xi * torch.sin(ph))
    return (out / (H * W)).to(x.dtype)


def res_block(sd: SD, p: str, x: torch.Tensor, emb: torch.Tensor) -> torch.Tensor:
    """ResBlock._forward, openaimodel.py:237-257 (no up/down, use_scale_shift_norm=False, dropout 0)."""
    h = _conv(sd, p + ".in_layers.2", silu(_gn32(sd, p + ".in_layers.0", x, 1e-5)))
    e = _lin(sd, p + ".emb_layers.1", silu(emb))
    h = h + e[:, :, None, None]
    h = _conv(sd, p + ".out_layers.3", silu(_gn32(sd, p + ".out_layers.0", h, 1e-5)))
    if (p + ".skip_connection.weight") in sd:
        x = _conv(sd, p + ".skip_connection", x, padding=0)
    return x + h


def unet_layout(cfg) -> Dict[str, list]:
    """Block structure derived from UNetModel.__init__, openaimodel.py:371-464.
    Returns lists of (kind, ...) per input/middle/output block."""
    mc = cfg["model_channels"]
    inp: List[list] = [[("conv", cfg["in_channels"], mc)]]
    chans = [mc]
    ch, ds = mc, 1
    cm = list(cfg["channel_mult"])
    for level, mult in enumerate(cm):
        for _ in range(cfg["num_res_blocks"]):
            layers = [("res", ch, mult * mc)]
            ch = mult * mc
            if ds in cfg["attention_resolutions"]:
                layers.append(("st", ch))
            inp.append(layers)
            chans.append(ch)
        if level != len(cm) - 1:
            inp.append([("down", ch)])
            chans.append(ch)
            ds *= 2
    mid = [("res", ch, ch), ("st", ch), ("res", ch, ch)]
    out: List[list] = []
    scaleu_ch: List[int] = []
    for level, mult in list(enumerate(cm))[::-1]:
        for i in range(cfg["num_res_blocks"] + 1):
            ich = chans.pop()
            layers = [("res", ch + ich, mc * mult)]
            scaleu_ch.append(ch)
            ch = mc * mult
            if ds in cfg["attention_resolutions"]:
                layers.append(("st", ch))
            if level and i == cfg["num_res_blocks"]:
                layers.append(("up", ch))
                ds //= 2
            out.append(layers)
    return dict(input=inp, middle=mid, output=out, scaleu_ch=scaleu_ch, final_ch=ch)


def _run_layers(sd: SD, prefix: str, layers, h, emb, ctx, objs, heads, scale, att_masks=None):
    """TimestepEmbedSequential.forward, openaimodel.py:62-79."""
    for j, layer in enumerate(layers):
        p = f"{prefix}.{j}"
        kind = layer[0]
        if kind == "conv":
            h = _conv(sd, p, h)
        elif kind == "res":
            h = res_block(sd, p, h, emb)
        elif kind == "st":
            h = spatial_transformer(sd, p, h, ctx, objs, heads, scale, att_masks)
        elif kind == "down":  # Downsample, openaimodel.py:115-141 (3x3 stride-2 pad-1 conv)
            h = _conv(sd, p + ".op", h, stride=2)
        elif kind == "up":    # Upsample, openaimodel.py:82-110 (nearest x2 then 3x3 conv)
            h = F.interpolate(h, scale_factor=2, mode="nearest")
            h = _conv(sd, p + ".conv", h)
        else:
            raise ValueError(kind)
    return h


def unet_forward(sd: SD, cfg, x: torch.Tensor, timesteps: torch.Tensor, context: torch.Tensor,
                 objs: torch.Tensor, fuser_scale: float = 1.0,
                 first_conv: Optional[Mapping[str, torch.Tensor]] = None,
                 probes: Optional[dict] = None, att_masks: Optional[torch.Tensor] = None) -> torch.Tensor:
    """UNetModel.forward_single_input, openaimodel.py:482-563, with ``objs`` (UniFusion tokens) precomputed.
    ``first_conv``: replacement {weight,bias} of input_blocks.0.0 (restore_first_conv_from_SD, :469-480).
    ``att_masks``: ``grounding_input['att_masks']`` for a model built with efficient_attention=False and boxes not
    dropped (openaimodel.py:508-518,558-561 pass grounding_input down only then); None = the efficient path."""
    lay = unet_layout(cfg)
    heads = cfg["num_heads"]
    if first_conv is not None:
        sd = dict(sd)
        sd["input_blocks.0.0.weight"] = first_conv["weight"]
        sd["input_blocks.0.0.bias"] = first_conv["bias"]
    t_emb = timestep_embedding(timesteps, cfg["model_channels"])
    emb = _lin(sd, "time_embed.2", silu(_lin(sd, "time_embed.0", t_emb)))
    h = x
    hs = []
    for i, layers in enumerate(lay["input"]):
        h = _run_layers(sd, f"input_blocks.{i}", layers, h, emb, context, objs, heads, fuser_scale, att_masks)
        hs.append(h)
        if probes is not None:
            probes[f"input_blocks.{i}"] = h
    h = _run_layers(sd, "middle_block", lay["middle"], h, emb, context, objs, heads, fuser_scale, att_masks)
    if probes is not None:
        probes["middle_block"] = h
    for i, layers in enumerate(lay["output"]):
        skip = hs.pop()
        b = torch.tanh(sd[f"scaleu_b_{i}"]) + 1          # openaimodel.py:524
        s = torch.tanh(sd[f"scaleu_s_{i}"]) + 1          # :525
        h = h * b[None, :, None, None]                   # :536
        skip = fourier_filter(skip, 1, s)                # :537
        h = torch.cat([h, skip], dim=1)                  # :539
        h = _run_layers(sd, f"output_blocks.{i}", layers, h, emb, context, objs, heads, fuser_scale, att_masks)
        if probes is not None:
            probes[f"output_blocks.{i}"] = h
    h = silu(_gn32(sd, "out.0", h, 1e-5))
    return _conv(sd, "out.2", h)


# ------------------------------------------------------------------------------------------------
# convnext.py + text_grounding_net.py (UniFusion)
# ------------------------------------------------------------------------------------------------
def _ln_cf(sd: SD, p: str, x: torch.Tensor, eps: float = 1e-6) -> torch.Tensor:
    """channels_first LayerNorm, convnext.py:128-136."""
    u = x.mean(1, keepdim=True)
    s = (x - u).pow(2).mean(1, keepdim=True)
    x = (x - u) / torch.sqrt(s + eps)
    return sd[p + ".weight"][:, None, None] * x + sd[p + ".bias"][:, None, None]


def convnext_tiny(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """ConvNeXt.forward_features, convnext.py:101-110; Block.forward :38-50. depths [3,3,9,3], dims [96,192,384,768]."""
    depths = [3, 3, 9, 3]
    for i in range(4):
        d = f"{p}.downsample_layers.{i}"
        if i == 0:
            x = F.conv2d(x, sd[d + ".0.weight"], sd[d + ".0.bias"], stride=4)
            x = _ln_cf(sd, d + ".1", x)
        else:
            x = _ln_cf(sd, d + ".0", x)
            x = F.conv2d(x, sd[d + ".1.weight"], sd[d + ".1.bias"], stride=2)
        for j in range(depths[i]):
            b = f"{p}.stages.{i}.{j}"
            y = F.conv2d(x, sd[b + ".dwconv.weight"], sd[b + ".dwconv.bias"], padding=3, groups=x.shape[1])
            y = y.permute(0, 2, 3, 1)
            y = F.layer_norm(y, (y.shape[-1],), sd[b + ".norm.weight"], sd[b + ".norm.bias"], 1e-6)
            y = _lin(sd, b + ".pwconv1", y)
            y = gelu_erf(y)
            y = _lin(sd, b + ".pwconv2", y)
            y = sd[b + ".gamma"] * y
            x = x + y.permute(0, 3, 1, 2)
    return x


def _mlp3(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """linears_list[i]: Linear -> SiLU -> Linear -> SiLU -> Linear, text_grounding_net.py:73-81."""
    return _lin(sd, p + ".4", silu(_lin(sd, p + ".2", silu(_lin(sd, p + ".0", x)))))


def unifusion(sd: SD, cfg, g: Mapping[str, torch.Tensor], p: str = "position_net"):
    """UniFusion.forward in eval mode, text_grounding_net.py:185-313.  Returns (objs [B,184,768], drop_box_mask)."""
    boxes, masks, pos = g["boxes"], g["masks"], g["positive_embeddings"]
    scribbles, polygons, segs = g["scribbles"], g["polygons"], g["segs"]
    points = g.get("points")
    B, N, _ = boxes.shape
    m = masks.unsqueeze(-1)
    # eval-mode drops come from the YAML test_drop_* flags (:104-116, :206-207)
    drop_box = bool(cfg["test_drop_boxes"]); drop_point = bool(cfg["test_drop_points"])
    drop_scribble = bool(cfg["test_drop_scribbles"]); drop_polygons = bool(cfg["test_drop_masks"])
    drop_segs = bool(cfg["test_drop_masks"])
    if drop_point and drop_box and drop_scribble and drop_polygons and drop_segs:   # :211-213
        drop_box = False
    xyxy = fourier_embed(boxes)                                                        # :216-217
    if points is None:
        points = (boxes[:, :, :2] + boxes[:, :, 2:]) / 2.0                            # :219-220
    pt = fourier_embed(points)
    sc = fourier_embed(scribbles)
    pg = fourier_embed(polygons)
    segs_r = F.interpolate(segs, 512, mode="nearest")                                 # :227
    sf = F.conv2d(segs_r, sd[p + ".in_conv.weight"], sd[p + ".in_conv.bias"], padding=1)
    sf = convnext_tiny(sd, p + ".convnext_tiny_backbone", sf)
    sf = sf.reshape(B, -1, 64).permute(0, 2, 1)                                       # :230-231
    pos = pos * m + (1 - m) * sd[p + ".null_positive_feature"].view(1, 1, -1)         # :248
    zeros = torch.zeros_like(m)
    bm = zeros if drop_box else m
    xyxy = xyxy * bm + (1 - bm) * sd[p + ".null_position_feature"].view(1, 1, -1)     # :253-254
    pm = zeros if drop_point else m
    pt = pt * pm + (1 - pm) * sd[p + ".null_point_feature"].view(1, 1, -1)            # :259-260
    sm = zeros if drop_scribble else ((scribbles.sum(-1, keepdim=True) + m) > 0).float()   # :267
    sc = sc * sm + (1 - sm) * sd[p + ".null_scribble_feature"].view(1, 1, -1)
    gm = zeros if drop_polygons else ((polygons.sum(-1, keepdim=True) + m) > 0).float()    # :272
    pg = pg * gm + (1 - gm) * sd[p + ".null_polygon_feature"].view(1, 1, -1)
    segm = torch.zeros(B) if drop_segs else (segs_r.sum(dim=(1, 2, 3)) > 0).float()        # :279
    segm = segm.view(-1, 1, 1)
    se = sf * segm + (1 - segm) * sd[p + ".null_seg_feature"].view(1, 1, -1).repeat(B, 64, 1)
    se = se + sd[p + ".pos_embedding"]                                                # :285
    toks = []
    for i, emb in enumerate([xyxy, pt, sc, pg]):                                      # :291-298
        toks.append(_mlp3(sd, f"{p}.linears_list.{i}", torch.cat([pos, emb], dim=-1)))
    toks.append(_mlp3(sd, f"{p}.linears_list.4", se))
    objs = torch.cat(toks, dim=1)
    return objs, bool(drop_box and drop_polygons)


def null_grounding(g: Mapping[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """GroundingNetInput.get_null_input, grounding_input/text_grounding_tokinzer_input.py:59-94."""
    return {k: torch.zeros_like(v) for k, v in g.items()}


def prepare_grounding(batch: Mapping[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """GroundingNetInput.prepare (:13-56): passes tensors through, renaming text_embeddings."""
    return dict(boxes=batch["boxes"], masks=batch["masks"], positive_embeddings=batch["text_embeddings"],
                scribbles=batch["scribbles"], polygons=batch["polygons"], segs=batch["segs"], points=batch["points"])


# ------------------------------------------------------------------------------------------------
# ddpm.py / util.py schedules, utils/model.py alpha schedule
# ------------------------------------------------------------------------------------------------
def alphas_cumprod(linear_start=0.00085, linear_end=0.012, timesteps=1000) -> np.ndarray:
    """make_beta_schedule('linear') util.py:30-34 + DDPM.register_schedule ddpm.py:19-24 (float64 numpy)."""
    betas = (torch.linspace(linear_start ** 0.5, linear_end ** 0.5, timesteps, dtype=torch.float64) ** 2).numpy()
    return np.cumprod(1.0 - betas, axis=0)


def alpha_generator(length: int, type: Optional[Sequence[float]] = None) -> list:
    """utils/model.py:83-117."""
    if type is None:
        type = [1, 0, 0]
    assert len(type) == 3 and type[0] + type[1] + type[2] == 1
    n0 = int(type[0] * length)
    n1 = int(type[1] * length)
    n2 = length - n0 - n1
    decay = list(np.arange(start=0, stop=1, step=1 / n1)[::-1]) if n1 != 0 else []
    out = [1] * n0 + decay + [0] * n2
    assert len(out) == length
    return out


# ------------------------------------------------------------------------------------------------
# plms.py / plms_instance.py
# ------------------------------------------------------------------------------------------------
class OracleModel:
    """Stateful wrapper with the reference model's mutable state: fuser scale (utils/model.py:78-81) and the
    swapped first conv (openaimodel.py:469-480, never undone).  ``__call__(input)`` mirrors UNetModel.forward."""

    def __init__(self, sd: SD, cfg, first_conv_sd: Optional[Mapping[str, torch.Tensor]] = None):
        self.sd, self.cfg = sd, cfg
        self.scale = 1.0
        self.first_conv_sd = first_conv_sd
        self.first_conv_active = None
        self._objs_cache: Dict[int, torch.Tensor] = {}
        self.null_g: Optional[Dict[str, torch.Tensor]] = None
        self.n_forward = 0

    def set_alpha_scale(self, a: float):
        self.scale = float(a)

    def restore_first_conv_from_SD(self):
        assert self.first_conv_sd is not None
        self.first_conv_active = self.first_conv_sd

    def objs_for(self, g: Mapping[str, torch.Tensor]) -> torch.Tensor:
        key = id(g)
        if key not in self._objs_cache:   # UniFusion is x/t independent & deterministic in eval mode
            self._objs_cache[key] = unifusion(self.sd, self.cfg, g)[0]
        return self._objs_cache[key]

    def __call__(self, inp: Mapping) -> torch.Tensor:
        if "grounding_input" in inp:
            g = inp["grounding_input"]
            if self.null_g is None:
                self.null_g = null_grounding(g)
        else:
            assert self.null_g is not None, "null grounding needs a prior prepared input (get_null_input :65)"
            g = self.null_g
        self.n_forward += 1
        return unet_forward(self.sd, self.cfg, inp["x"], inp["timesteps"], inp["context"], self.objs_for(g),
                            fuser_scale=self.scale, first_conv=self.first_conv_active)


def _p_sample_plms(model: Callable, inp: dict, t, index: int, a, a_prev, uc, g_scale, old_eps, t_next):
    """p_sample_plms, plms.py:117-167 (identical in plms_instance.py:162-212).  sigma == 0 (eta = 0)."""
    x = inp["x"].clone()
    b = x.shape[0]

    def model_out(i):
        e = model(i)
        if uc is not None and g_scale != 1:
            e_uc = model(dict(x=i["x"], timesteps=i["timesteps"], context=uc))
            e = e_uc + g_scale * (e - e_uc)
        return e

    def x_prev_of(e):
        a_t = torch.full((b, 1, 1, 1), float(a[index]))
        a_p = torch.full((b, 1, 1, 1), float(a_prev[index]))
        # ddim_sqrt_one_minus_alphas = sqrt(1 - a_t) evaluated in float32 (plms.py:55)
        s1m = torch.full((b, 1, 1, 1), float(torch.sqrt(1.0 - torch.tensor(a[index], dtype=torch.float32))))
        pred_x0 = (x - s1m * e) / a_t.sqrt()
        dir_xt = (1.0 - a_p).sqrt() * e
        return a_p.sqrt() * pred_x0 + dir_xt

    inp["timesteps"] = t
    e_t = model_out(inp)
    if len(old_eps) == 0:
        inp["x"] = x_prev_of(e_t)
        inp["timesteps"] = t_next
        e_next = model_out(inp)
        e_p = (e_t + e_next) / 2
    elif len(old_eps) == 1:
        e_p = (3 * e_t - old_eps[-1]) / 2
    elif len(old_eps) == 2:
        e_p = (23 * e_t - 16 * old_eps[-1] + 5 * old_eps[-2]) / 12
    else:
        e_p = (55 * e_t - 59 * old_eps[-1] + 37 * old_eps[-2] - 9 * old_eps[-3]) / 24
    return x_prev_of(e_p), e_t


def _schedule(S: int):
    ac = alphas_cumprod()
    n = ac.shape[0]
    c = n // S
    steps = np.asarray(list(range(0, n, c))) + 1
    ac32 = torch.tensor(ac, dtype=torch.float32)       # ddpm.py:33-36 registers float32 buffers
    a = ac32[steps].numpy()
    a_prev = np.asarray([ac32[0].item()] + ac32[steps[:-1]].tolist(), dtype=np.float32)
    return steps, a, a_prev


def _step_common(model: OracleModel, alphas, i):
    """set alpha + first-conv swap, plms.py:90-94."""
    if alphas is not None:
        model.set_alpha_scale(alphas[i])
        if alphas[i] == 0:
            model.restore_first_conv_from_SD()


def q_sample(x_start: torch.Tensor, t: torch.Tensor, noise: torch.Tensor) -> torch.Tensor:
    """LatentDiffusion.q_sample, ldm.py:17-20 (buffers registered in float32, ddpm.py:33-36)."""
    ac = torch.tensor(alphas_cumprod(), dtype=torch.float32)
    shape = (x_start.shape[0],) + (1,) * (x_start.dim() - 1)
    return ac.sqrt()[t].reshape(shape) * x_start + (1.0 - ac).sqrt()[t].reshape(shape) * noise


def plms_sample(model: OracleModel, S: int, inp: dict, uc, guidance_scale: float,
                alpha_type: Optional[Sequence[float]] = None, trace: Optional[list] = None,
                mask: Optional[torch.Tensor] = None, x0: Optional[torch.Tensor] = None,
                noises: Optional[torch.Tensor] = None) -> torch.Tensor:
    """PLMSSampler.sample / plms_sampling, plms.py:66-113.  ``mask`` / ``x0``: the inpainting blend of plms.py:99-104 in
    front of every step; ``noises[i]`` is the noise q_sample draws at step i (the reference draws it from the global RNG)."""
    steps, a, a_prev = _schedule(S)
    time_range = np.flip(steps)
    total = steps.shape[0]
    b = inp["x"].shape[0]
    alphas = alpha_generator(len(time_range), alpha_type) if alpha_type is not None else None
    old_eps: list = []
    img = inp["x"]
    for i, step in enumerate(time_range):
        _step_common(model, alphas, i)
        index = total - i - 1
        ts = torch.full((b,), int(step), dtype=torch.long)
        ts_next = torch.full((b,), int(time_range[min(i + 1, len(time_range) - 1)]), dtype=torch.long)
        if mask is not None:                                   # plms.py:99-104
            assert x0 is not None and noises is not None
            img = q_sample(x0, ts, noises[i]) * mask + (1.0 - mask) * img
            inp["x"] = img
        img, e_t = _p_sample_plms(model, inp, ts, index, a, a_prev, uc, guidance_scale, old_eps, ts_next)
        inp["x"] = img
        old_eps.append(e_t)
        if len(old_eps) >= 4:
            old_eps.pop(0)
        if trace is not None:
            trace.append(img.clone())
    return img


def crop_paste(target: torch.Tensor, source: torch.Tensor, box_xyxy, latent_size: int) -> torch.Tensor:
    """crop_and_paste_tensor, plms_instance.py:112-126.  NOTE the reference slices dim-2 with the x-coords and
    dim-3 with the y-coords (bbox[0]:bbox[2] on dim 2), as written; reproduced verbatim."""
    bb = [int(v * latent_size) for v in box_xyxy]
    target = target.clone()
    target[:, :, bb[0]:bb[2], bb[1]:bb[3]] = source[:, :, bb[0]:bb[2], bb[1]:bb[3]]
    return target


def plms_sample_mis(model: OracleModel, S: int, inputs: List[dict], uc, guidance_scale: float, mis: float,
                    alpha_type: Optional[Sequence[float]] = None, crop_and_paste: bool = False) -> torch.Tensor:
    """PLMSSamplerInst.sample / plms_sampling, plms_instance.py:59-158."""
    steps, a, a_prev = _schedule(S)
    time_range = np.flip(steps)
    total = steps.shape[0]
    b = inputs[0]["x"].shape[0]
    latent_size = inputs[0]["x"].shape[2]
    alphas = alpha_generator(len(time_range), alpha_type) if alpha_type is not None else None
    mis_step = int(total * mis)
    eps_hist: Dict[int, list] = {}
    for idx, inp in enumerate(inputs):
        eps_hist[idx] = []
        for i, step in enumerate(time_range[:mis_step]):
            _step_common(model, alphas, i)
            index = total - i - 1
            ts = torch.full((b,), int(step), dtype=torch.long)
            ts_next = torch.full((b,), int(time_range[min(i + 1, len(time_range) - 1)]), dtype=torch.long)
            img, e_t = _p_sample_plms(model, inp, ts, index, a, a_prev, uc, guidance_scale, eps_hist[idx], ts_next)
            inp["x"] = img
            eps_hist[idx].append(e_t)
            if len(eps_hist[idx]) >= 4:
                eps_hist[idx].pop(0)
    base = inputs[0]
    old_eps = eps_hist[0]
    if crop_and_paste:
        for src in inputs[1:]:
            box = src["grounding_input"]["boxes"][0][0].tolist()
            base["x"] = crop_paste(base["x"], src["x"], box, latent_size)
    else:
        base["x"] = torch.mean(torch.stack([i["x"] for i in inputs]), dim=0)          # :135
    img = base["x"]
    for i, step in enumerate(time_range):
        if i < mis_step:
            continue
        _step_common(model, alphas, i)
        index = total - i - 1
        ts = torch.full((b,), int(step), dtype=torch.long)
        ts_next = torch.full((b,), int(time_range[min(i + 1, len(time_range) - 1)]), dtype=torch.long)
        img, e_t = _p_sample_plms(model, base, ts, index, a, a_prev, uc, guidance_scale, old_eps, ts_next)
        base["x"] = img
        old_eps.append(e_t)
        if len(old_eps) >= 4:
            old_eps.pop(0)
    return img


# ------------------------------------------------------------------------------------------------
# VAE decoder: ldm/models/autoencoder.py + ldm/modules/diffusionmodules/model.py
# ------------------------------------------------------------------------------------------------
# SD-1.5 KL-f8 autoencoder (configs/test_box.yaml:42-61)
DEFAULT_VAE_CFG = dict(scale_factor=0.18215, embed_dim=4, z_channels=4, ch=128, ch_mult=(1, 2, 4, 4), num_res_blocks=2,
                       attn_resolutions=(), resolution=256, out_ch=3)


def vae_resnet_block(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """model.py:121-143 with temb None: GN(32, eps 1e-6) -> swish -> conv3x3, twice; 1x1 nin_shortcut if Cin != Cout."""
    h = _conv(sd, p + ".conv1", silu(_gn32(sd, p + ".norm1", x, 1e-6)))
    h = _conv(sd, p + ".conv2", silu(_gn32(sd, p + ".norm2", h, 1e-6)))
    if (p + ".nin_shortcut.weight") in sd:
        x = _conv(sd, p + ".nin_shortcut", x, padding=0)
    return x + h


def vae_attn_block(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """model.py:178-202: one head over the H*W positions, scale = C^-0.5, q/k/v/proj_out are 1x1 convs WITH bias."""
    h = _gn32(sd, p + ".norm", x, 1e-6)
    q, k, v = (_conv(sd, f"{p}.{n}", h, padding=0) for n in ("q", "k", "v"))
    b, c, hh, ww = q.shape
    q = q.reshape(b, c, hh * ww).permute(0, 2, 1)                       # b, hw, c
    k = k.reshape(b, c, hh * ww)                                        # b, c, hw
    w_ = torch.softmax(torch.bmm(q, k) * (int(c) ** (-0.5)), dim=2)     # w_[b,i,j] over keys j
    v = v.reshape(b, c, hh * ww)
    h = torch.bmm(v, w_.permute(0, 2, 1)).reshape(b, c, hh, ww)         # h[b,c,i] = sum_j v[b,c,j] w_[b,i,j]
    return x + _conv(sd, p + ".proj_out", h, padding=0)


def vae_decode(sd: SD, cfg, z: torch.Tensor) -> torch.Tensor:
    """AutoencoderKL.decode (autoencoder.py:32-36) + Decoder.forward (model.py:534-568): latent [B,4,H,W] ->
    image [B,3,8H,8W] (for the 4-level config).  ``sd`` uses the reference key names (``decoder.*``,
    ``post_quant_conv.*``)."""
    nres = len(cfg["ch_mult"])
    z = (1.0 / cfg["scale_factor"]) * z
    z = _conv(sd, "post_quant_conv", z, padding=0)
    h = _conv(sd, "decoder.conv_in", z)
    h = vae_resnet_block(sd, "decoder.mid.block_1", h)
    h = vae_attn_block(sd, "decoder.mid.attn_1", h)
    h = vae_resnet_block(sd, "decoder.mid.block_2", h)
    curr_res = cfg["resolution"] // 2 ** (nres - 1)
    for i_level in reversed(range(nres)):
        for i_block in range(cfg["num_res_blocks"] + 1):
            h = vae_resnet_block(sd, f"decoder.up.{i_level}.block.{i_block}", h)
            if curr_res in cfg["attn_resolutions"]:
                h = vae_attn_block(sd, f"decoder.up.{i_level}.attn.{i_block}", h)
        if i_level != 0:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")      # model.py:53
            h = _conv(sd, f"decoder.up.{i_level}.upsample.conv", h)
            curr_res *= 2
    h = silu(_gn32(sd, "decoder.norm_out", h, 1e-6))
    return _conv(sd, "decoder.conv_out", h)
