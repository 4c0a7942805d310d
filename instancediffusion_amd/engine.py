"""UNet denoise-step executor for MI355X: weight packing, HBM-resident workspace, op sequencing, hipGraph replay.

Replaces the reference's ``UNetModel.forward_single_input`` (openaimodel.py:482-563) and everything below it.
MI355X-first design decisions (vs the reference's PyTorch module tree):
  * activations are NHWC / token-major 16-bit matrices for the whole forward -- the reference's NCHW<->token
    permutes (attention.py:371,376; 13 % of its CPU time) do not exist;
  * every step-invariant quantity is hoisted into a ``Cond`` object built once per conditioning: UniFusion tokens,
    cross-attention K/V of the text context, and the fuser's K/V rows of the 184 grounding tokens (LayerNorm is
    row-wise, so those rows do not depend on x or t);
  * the gated self-attention attends over two KV segments (visual, grounding) and computes only the visual query rows
    that attention.py:308 keeps; when the gate scale is 0 the fuser is skipped (exactly 0 * tanh(a) * y);
  * one forward = a fixed sequence of C-ABI launches on one stream over preallocated buffers, so it is captured
    once into a hipGraph per (batch, resolution, fuser on/off) and replayed.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from .host.attention import SpatialTransformer
from .host.unet import Downsample, ResBlock, UNetModel, Upsample

# LayerNorm statistics of the folded GEMMs (A/B switch, env IDF_LN_SELF): 0 = every producer of the residual stream emits them
# (out_stats pass), 2 = every LN_ROW consumer sums its own A rows, 1 (default) = q/k and cross-q sum their own, the GEGLU GEMMs
# take them from the producer.
LN_SELF_MODE = int(os.environ.get("IDF_LN_SELF", "1"))
if LN_SELF_MODE not in (0, 1, 2):          # any other value would leave LN_ROW consumers reading statistics nobody wrote
    raise ValueError(f"IDF_LN_SELF={LN_SELF_MODE}: must be 0, 1 or 2")
# Round 6: the fused q | k | v projection of the C = 320 level runs on the row-resident kernel (csrc/qkv_fused.hip), which takes
# the LayerNorm statistics from the caller: there the producers of the self-attention / gated self-attention input (proj_in,
# attn1's out-projection: plain GEMMs, whose epilogue partials make the statistics almost free) emit them even in mode 1, while
# the cross-attention query -- fed by the fused feed-forward kernel, which cannot emit them -- keeps summing its own.
QKV_ROW = os.environ.get("IDF_QKV_ROW", "1") != "0"
OBJ_TOKENS = 184
MASK_RES = 64                  # the reference applies the fuser mask only when H*W == 64*64 (attention.py:195)


def _round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


# GEGLU weight-row interleave period (A/B switch, env IDF_GEGLU_PERIOD): 64 = [32 value | 32 gate] (value and gate of an output in
# the two MFMA fragments of a pair: even fragment count per wave, 256-wide tiles), 32 = [16 value | 16 gate] (both in ONE
# fragment: the GEGLU GEMMs run on the 320-wide persistent tiles -- 9 LDS-DMA instructions per 40 MFMAs instead of 8 per 32,
# and 20 % fewer tiles re-reading the activation rows).
GEGLU_PERIOD = int(os.environ.get("IDF_GEGLU_PERIOD", "32"))
if GEGLU_PERIOD not in (32, 64):
    raise ValueError(f"IDF_GEGLU_PERIOD={GEGLU_PERIOD}: must be 32 or 64")
# Fused GEGLU feed-forward (idf_mlp_geglu: one launch per MLP, the 4C-wide intermediate never reaches HBM) where the kernel
# exists -- the C = 320 blocks -- and the statistics of its LayerNorm are at hand; IDF_MLP_FUSED=0: the two idf_gemm calls
# (A/B switch; results equal up to the fp32 summation order of the second product).
MLP_FUSED = os.environ.get("IDF_MLP_FUSED", "1")
if MLP_FUSED not in ("0", "1"):
    raise ValueError(f"IDF_MLP_FUSED={MLP_FUSED}: must be 0 or 1")
MLP_FUSED = MLP_FUSED == "1" and GEGLU_PERIOD == 32
# ... and only from IDF_MLP_MIN_M token rows up: the kernel's 128-row tiles run one per workgroup slot for ~110-175 us whatever
# their number, so below a few hundred tiles (a 2-row forward has 64) the two small GEMMs are faster
# (profiles/r04_mlp_small_m.log).
MLP_MIN_M = int(os.environ.get("IDF_MLP_MIN_M", "32768"))
# Paired forwards (classifier-free guidance: rows [n, 2n) carry the SAME latent and timestep as rows [0, n) and differ only in
# their conditioning): everything in front of the first block that reads the conditioning -- first conv, first ResBlock,
# GroupNorm, proj_in and the first 64 x 64 self-attention with its out-projection -- is computed ONCE for the n distinct
# rows and duplicated (exact: those layers see only (x, t)).  IDF_PAIR_HOIST=0 computes all 2n rows (A/B switch).
PAIR_HOIST = os.environ.get("IDF_PAIR_HOIST", "1")
if PAIR_HOIST not in ("0", "1"):
    raise ValueError(f"IDF_PAIR_HOIST={PAIR_HOIST}: must be 0 or 1")
PAIR_HOIST = PAIR_HOIST == "1"
# Round 5: every 3x3 conv whose output feeds a GroupNorm leaves that GroupNorm's partial statistics from its epilogue registers
# (idf_conv3x3 gn_partial), and the GroupNorm runs its normalise pass only.  IDF_GN_EPI=0: every GroupNorm runs both passes.
GN_EPI = os.environ.get("IDF_GN_EPI", "1")
if GN_EPI not in ("0", "1"):
    raise ValueError(f"IDF_GN_EPI={GN_EPI}: must be 0 or 1")
GN_EPI = GN_EPI == "1"
DEBUG_PAIRED = os.environ.get("IDF_DEBUG_PAIRED", "0") == "1"     # verify the paired=True guarantee on EVERY forward_cond call


def pack_geglu(w: torch.Tensor, b: torch.Tensor, period: int = 64) -> Tuple[torch.Tensor, torch.Tensor]:
    """attention.py:36-43: proj -> chunk(2) = (value, gate).  Interleave rows as [P/2 value | P/2 gate] per P so that value and
    gate of the SAME output columns sit in one lane's accumulators (in-register GEGLU): P = 64 -> the two 32-wide MFMA column
    tiles of a pair, P = 32 -> registers q and q + 2 of one tile."""
    n2, k = w.shape
    n = n2 // 2
    h = period // 2
    assert n % h == 0
    wv, wg = w[:n].reshape(n // h, h, k), w[n:].reshape(n // h, h, k)
    bv, bg = b[:n].reshape(n // h, h), b[n:].reshape(n // h, h)
    return torch.cat([wv, wg], dim=1).reshape(n2, k).contiguous(), torch.cat([bv, bg], dim=1).reshape(n2).contiguous()


def pack_conv3x3(w: torch.Tensor) -> torch.Tensor:
    """[Cout, Cin, 3, 3] -> [Cout, (ky*3+kx)*Cin + ci]  (K order of the implicit-GEMM gather)."""
    co, ci = w.shape[0], w.shape[1]
    return w.permute(0, 2, 3, 1).reshape(co, 9 * ci).contiguous()


class _Lin:
    __slots__ = ("w", "b")

    def __init__(self, w, b):
        self.w, self.b = w, b


class Cond:
    """Step-invariant conditioning state for a batch of B samples (built by ``UNetEngine.prepare_cond``)."""

    _next_uid = 0

    def __init__(self, B: int):
        Cond._next_uid += 1
        self.uid = Cond._next_uid                              # identity for the engine's static-slot binding
        self.B = B
        self.objs: Optional[torch.Tensor] = None            # [B, 184, 768] 16-bit
        self.k_ctx: List[torch.Tensor] = []                  # per ST layer [B, 77, C]
        self.vt_ctx: List[torch.Tensor] = []                 # per ST layer [B, C, 128]
        self.k_obj: List[torch.Tensor] = []                  # per ST layer [B, 184, C]
        self.vt_obj: List[torch.Tensor] = []                 # per ST layer [B, C, 192]
        self.n_ctx = 0
        # masked gated self-attention only (model built with efficient_attention=False): visibility words, int32
        self.vis: List[torch.Tensor] = []                    # [] or [qbits [B,4096], kbits0 [B,4096], kbits1 [B,192]]

    @staticmethod
    def cat(conds: Sequence["Cond"]) -> "Cond":
        out = Cond(sum(c.B for c in conds))
        out.n_ctx = conds[0].n_ctx
        out.objs = torch.cat([c.objs for c in conds], 0)
        for name in ("k_ctx", "vt_ctx", "k_obj", "vt_obj", "vis"):
            lists = [getattr(c, name) for c in conds]
            setattr(out, name, [torch.cat(ts, 0) for ts in zip(*lists)])
        return out

    def _tensors(self):
        return [self.objs] + self.k_ctx + self.vt_ctx + self.k_obj + self.vt_obj + self.vis

    def clone(self) -> "Cond":
        out = Cond(self.B)
        out.n_ctx = self.n_ctx
        out.objs = self.objs.clone()
        for name in ("k_ctx", "vt_ctx", "k_obj", "vt_obj", "vis"):
            setattr(out, name, [t.clone() for t in getattr(self, name)])
        return out

    def copy_from(self, other: "Cond"):
        for dst, src in zip(self._tensors(), other._tensors()):
            dst.copy_(src)

    def same_layout(self, other: "Cond") -> bool:
        a, b = self._tensors(), other._tensors()
        return len(a) == len(b) and all(x.shape == y.shape for x, y in zip(a, b)) and self.n_ctx == other.n_ctx

    def select(self, idx: torch.Tensor) -> "Cond":
        out = Cond(int(idx.numel()))
        out.n_ctx = self.n_ctx
        out.objs = self.objs[idx].contiguous()
        for name in ("k_ctx", "vt_ctx", "k_obj", "vt_obj", "vis"):
            setattr(out, name, [t[idx].contiguous() for t in getattr(self, name)])
        return out


class UNetEngine:
    def __init__(self, model: UNetModel, ops=None, dtype: torch.dtype = torch.bfloat16, use_graphs: bool = True):
        if ops is None:
            from .ops import HipOps          # raises when libidf_gfx950.so / the GPU is missing: no fallback
            ops = HipOps(dtype)
        self.ops = ops
        self.dtype = ops.dtype
        self.device = ops.device
        self.model = model
        self.heads = model.num_heads
        self.masked_fuser = not getattr(model, "efficient_attention", True)     # attention.py:189
        self.vt_global = os.environ.get("IDF_VT_GLOBAL", "1") != "0"            # A/B knob, see _self_attn
        # fewest tokens per sample for the fused q | k | v launch.  Round 5: 256 (was 1024) -- at the 16 x 16 level the fused launch
        # (M32768 N3840 K1280, y read once) replaces the q | k GEMM + the batched 62-%-padded V^T GEMM (2.05 ms per 128-row forward
        # at 523 TF): 165.9 -> 164.7 ms per forward on one box, 64 no further gain (profiles/r05_vt_min_n_ab.log)
        self.vt_min_n = int(os.environ.get("IDF_VT_MIN_N", "256"))
        if self.vt_min_n < 64 or self.vt_min_n % 64:
            raise ValueError(f"IDF_VT_MIN_N={self.vt_min_n}: a positive multiple of 64 (tokens per sample from which the fused "
                             "q | k | v launch is used; default 256)")
        self.use_graphs = use_graphs and self.device.type == "cuda"
        self._bufs: Dict[tuple, torch.Tensor] = {}
        self._graphs: Dict[tuple, tuple] = {}
        self._paired_checked: set = set()
        self._cond_cache: Dict[tuple, Cond] = {}
        self._slots: Dict[int, Cond] = {}                      # batch size -> static Cond the hipGraphs read from
        self._slot_bound: Dict[int, int] = {}
        self.fuser_scale = None
        self._pack(model)
        self.set_fuser_scale(1.0)

    # =================================================================================================
    # weight packing (once; HBM-resident 16-bit GEMM images + fp32 vectors)
    # =================================================================================================
    def _w16(self, t: torch.Tensor) -> torch.Tensor:
        return t.detach().to(device=self.device, dtype=torch.float32).to(self.dtype).contiguous()

    def _f32(self, t: torch.Tensor) -> torch.Tensor:
        return t.detach().to(device=self.device, dtype=torch.float32).contiguous()

    def _lin(self, m, with_bias=True) -> _Lin:
        w = m.weight
        if w.dim() == 4:
            w = w.reshape(w.shape[0], w.shape[1])            # 1x1 conv
        return _Lin(self._w16(w), self._f32(m.bias) if (with_bias and m.bias is not None) else None)

    def _conv(self, m) -> _Lin:
        return _Lin(self._w16(pack_conv3x3(m.weight.detach().float())), self._f32(m.bias))

    # LayerNorm -> Linear pairs are stored FOLDED (include/idf.h IDF_EPI_LN_ROW / LN_COL): the weight carries gamma,
    #   LN(x) W^T = rstd * (x (gamma*W)^T - mu * c) + d,   c = row sums of the 16-bit (gamma*W),   d = W beta (+ bias),
    # so the forward never materialises LN(x): a GEMM reads the raw residual stream and its epilogue applies (mu, rstd).
    def _fold_ln(self, w: torch.Tensor, bias, norm):
        """-> (16-bit gamma-folded weight, c [N] fp32 = its row sums as the MFMA sees them, d [N] fp32 = W beta + bias)."""
        w = w.detach().float()
        g, b = norm.weight.detach().float(), norm.bias.detach().float()
        w16 = self._w16(w * g[None, :])
        d = w @ b + (bias.detach().float() if bias is not None else 0.0)
        return w16, w16.float().sum(1).contiguous(), self._f32(d)

    def _pack_ff(self, ff, norm):
        """GEGLU feed-forward behind LayerNorm `norm`: proj rows interleaved value | gate per GEGLU_PERIOD (pack_geglu), gamma folded."""
        w = ff.net[0].proj.weight.detach().float()
        g, b = norm.weight.detach().float(), norm.bias.detach().float()
        wp, dp = pack_geglu(w * g[None, :], ff.net[0].proj.bias.detach().float() + w @ b, GEGLU_PERIOD)
        w16 = self._w16(wp)
        f = dict(w1=w16, b1=self._f32(dp), c1=w16.float().sum(1).contiguous(), l2=self._lin(ff.net[2]))
        C = w.shape[1]
        if MLP_FUSED and hasattr(self.ops, "mlp_geglu") and self.ops.mlp_supported(128, C):
            f["cd"], f["w2p"] = self.ops.mlp_pack(w16, f["c1"], f["b1"], f["l2"].w)
        return f

    def _attn(self, a, norm, self_attn: bool):
        """Attention projections behind LayerNorm `norm` (applied to the QUERY side input; for self-attention also to K / V)."""
        d = dict(out=self._lin(a.to_out[0]))
        if self_attn:
            # ONE gamma-folded [3C, C] image: rows [0, 2C) = q | k, rows [2C, 3C) = v.  The fused projection (tokens >= 1024:
            # V^T in the batch-interleaved global layout) runs on the whole image; the small levels use its two row ranges
            # as the separate q | k and transposed-V GEMMs (views, no second copy).
            w3, c3, d3 = self._fold_ln(torch.cat([a.to_q.weight.detach(), a.to_k.weight.detach(), a.to_v.weight.detach()], 0),
                                       None, norm)
            C2 = 2 * a.to_q.weight.shape[0]
            d["wqkv"], d["cqkv"], d["dqkv"] = w3, c3, d3
            d["wqk"], d["cqk"], d["dqk"] = w3[:C2], c3[:C2], d3[:C2]
            d["wv"], d["cv"], d["dv"] = w3[C2:], c3[C2:], d3[C2:]
        else:                                        # cross attention: K / V come from the (un-normalised) text context
            d["wq"], d["cq"], d["dq"] = self._fold_ln(a.to_q.weight, None, norm)
            d["wk"], d["wv"] = self._w16(a.to_k.weight), self._w16(a.to_v.weight)
        return d

    def _attn_kv_plain(self, a):
        """K / V projections of the fuser for the grounding-token rows (their LayerNorm runs once per conditioning)."""
        return dict(wk=self._w16(a.to_k.weight), wv=self._w16(a.to_v.weight))

    def _pack_res(self, rb: ResBlock, emb_w: list, emb_b: list, off: list):
        p = dict(kind="res", cin=rb.channels, cout=rb.out_channels,
                 gn1=(self._f32(rb.in_layers[0].weight), self._f32(rb.in_layers[0].bias)),
                 conv1=self._conv(rb.in_layers[2]),
                 gn2=(self._f32(rb.out_layers[0].weight), self._f32(rb.out_layers[0].bias)),
                 conv2=self._conv(rb.out_layers[3]),
                 skip=None if isinstance(rb.skip_connection, torch.nn.Identity) else self._lin(rb.skip_connection),
                 emb_off=off[0])
        emb_w.append(rb.emb_layers[1].weight.detach().float())
        emb_b.append(rb.emb_layers[1].bias.detach().float())
        off[0] += rb.out_channels
        return p

    def _pack_st(self, st: SpatialTransformer, tanh_alphas: list):
        blk = st.transformer_blocks[0]
        fz = blk.fuser
        idx = len(tanh_alphas)
        tanh_alphas.append(torch.stack([torch.tanh(fz.alpha_attn.detach().float()),
                                        torch.tanh(fz.alpha_dense.detach().float())]))
        return dict(kind="st", c=st.in_channels, idx=idx,
                    norm=(self._f32(st.norm.weight), self._f32(st.norm.bias)),
                    proj_in=self._lin(st.proj_in), proj_out=self._lin(st.proj_out),
                    n1=(self._f32(blk.norm1.weight), self._f32(blk.norm1.bias)),
                    n2=(self._f32(blk.norm2.weight), self._f32(blk.norm2.bias)),
                    n3=(self._f32(blk.norm3.weight), self._f32(blk.norm3.bias)),
                    attn1=self._attn(blk.attn1, blk.norm1, True), attn2=self._attn(blk.attn2, blk.norm2, False),
                    ff=self._pack_ff(blk.ff, blk.norm3),
                    f_lin=self._lin(fz.linear), f_attn=self._attn(fz.attn, fz.norm1, True), f_kv=self._attn_kv_plain(fz.attn),
                    f_ff=self._pack_ff(fz.ff, fz.norm2),
                    f_n1=(self._f32(fz.norm1.weight), self._f32(fz.norm1.bias)),
                    f_n2=(self._f32(fz.norm2.weight), self._f32(fz.norm2.bias)))

    def _pack_layers(self, seq, emb_w, emb_b, off, tanh_alphas):
        out = []
        for _, layer in seq.items():
            if isinstance(layer, ResBlock):
                out.append(self._pack_res(layer, emb_w, emb_b, off))
            elif isinstance(layer, SpatialTransformer):
                out.append(self._pack_st(layer, tanh_alphas))
            elif isinstance(layer, Downsample):
                out.append(dict(kind="down", conv=self._conv(layer.op)))
            elif isinstance(layer, Upsample):
                out.append(dict(kind="up", conv=self._conv(layer.conv)))
            else:                                             # first conv
                out.append(dict(kind="conv_in", w=self._f32(layer.weight), b=self._f32(layer.bias)))
        return out

    def _pack(self, model: UNetModel):
        emb_w, emb_b, off, tanh_alphas = [], [], [0], []
        self.te0 = self._lin(model.time_embed[0])
        self.te2 = self._lin(model.time_embed[2])
        self.in_blocks = [self._pack_layers(b, emb_w, emb_b, off, tanh_alphas) for b in model.input_blocks]
        self.mid_block = self._pack_layers(model.middle_block, emb_w, emb_b, off, tanh_alphas)
        self.out_blocks = [self._pack_layers(b, emb_w, emb_b, off, tanh_alphas) for b in model.output_blocks]
        self.emb_all = _Lin(self._w16(torch.cat(emb_w, 0)), self._f32(torch.cat(emb_b, 0)))
        self.emb_total = off[0]
        self.tanh_alphas = torch.stack(tanh_alphas).to(self.device)            # [n_st, 2] fp32
        self.gates = torch.zeros_like(self.tanh_alphas)                         # scale * tanh(alpha), read by kernels
        self.n_st = len(tanh_alphas)
        self.scaleu = []
        for i in range(len(model.output_blocks)):
            b = getattr(model, f"scaleu_b_{i}").detach().float()
            s = getattr(model, f"scaleu_s_{i}").detach().float()
            self.scaleu.append((self._f32(torch.tanh(b) + 1.0), self._f32(torch.tanh(s))))
        self.out_gn = (self._f32(model.out[0].weight), self._f32(model.out[0].bias))
        oc = model.out[2]
        wpad = torch.zeros(64, oc.weight.shape[1], 3, 3)
        wpad[: oc.weight.shape[0]] = oc.weight.detach().float().cpu()
        bpad = torch.zeros(64)
        bpad[: oc.bias.shape[0]] = oc.bias.detach().float().cpu()
        self.out_conv = _Lin(self._w16(pack_conv3x3(wpad)), self._f32(bpad))
        self.n_out = oc.weight.shape[0]
        self._pack_tokenizer(model.position_net)

    def repack_first_conv(self, conv):
        """restore_first_conv_from_SD (openaimodel.py:469-480): only the fp32 first-conv image changes."""
        p = self.in_blocks[0][0]
        p["w"].copy_(conv.weight.detach().float())
        p["b"].copy_(conv.bias.detach().float())

    def _pack_tokenizer(self, pn):
        self.pn = pn
        self.tok_mlps = []
        for i in range(5):
            seq = pn.linears_list[i]
            self.tok_mlps.append([self._lin(seq[0]), self._lin(seq[2]), self._lin(seq[4])])
        self.tok_null = dict(
            text=self._f32(pn.null_positive_feature), box=self._f32(pn.null_position_feature),
            point=self._f32(pn.null_point_feature), scribble=self._f32(pn.null_scribble_feature),
            polygon=self._f32(pn.null_polygon_feature), seg=self._f32(pn.null_seg_feature))
        self.tok_pos = self._f32(pn.pos_embedding)            # [1, 64, 3072]
        self.tok_freqs = (100.0 ** (torch.arange(16) / 16)).to(device=self.device, dtype=torch.float32)
        self._pack_convnext(pn)

    def _pack_convnext(self, pn):
        """ConvNeXt-T mask backbone (convnext.py:52-110) as GEMM images: stem 4x4/s4 and downsample 2x2/s2 convs are
        patch GEMMs; the layer-scale gamma (convnext.py:45-46) is folded into pwconv2; K is zero-padded to 64-multiples."""
        bb = pn.convnext_tiny_backbone
        cn = dict(in_w=self._f32(pn.in_conv.weight), in_b=self._f32(pn.in_conv.bias), stages=[], down=[])
        stem = bb.downsample_layers[0]
        w = stem[0].weight.detach().float().reshape(stem[0].weight.shape[0], -1)           # [96, 48] (c, ky, kx)
        wp = torch.zeros(w.shape[0], 64)
        wp[:, :w.shape[1]] = w
        cn["stem"] = _Lin(self._w16(wp), self._f32(stem[0].bias))
        cn["stem_ln"] = (self._f32(stem[1].weight), self._f32(stem[1].bias))
        for i in range(1, 4):
            dl = bb.downsample_layers[i]
            wd = dl[1].weight.detach().float()                                              # [Cout, Cin, 2, 2]
            cn["down"].append(dict(ln=(self._f32(dl[0].weight), self._f32(dl[0].bias)),
                                   conv=_Lin(self._w16(wd.permute(0, 2, 3, 1).reshape(wd.shape[0], -1)),
                                             self._f32(dl[1].bias))))
        for i in range(4):
            blocks = []
            for _, blk in bb.stages[i].items():
                C = blk.dwconv.weight.shape[0]
                kpad = _round_up(C, 64)
                w1 = torch.zeros(4 * C, kpad)
                w1[:, :C] = blk.pwconv1.weight.detach().float()
                g = blk.gamma.detach().float()
                blocks.append(dict(
                    C=C, kpad=kpad,
                    dw_w=self._f32(blk.dwconv.weight.detach().float().reshape(C, 49).t().contiguous()),   # [49][C]
                    dw_b=self._f32(blk.dwconv.bias), ln=(self._f32(blk.norm.weight), self._f32(blk.norm.bias)),
                    l1=_Lin(self._w16(w1), self._f32(blk.pwconv1.bias)),
                    l2=_Lin(self._w16(g[:, None] * blk.pwconv2.weight.detach().float()),
                            self._f32(g * blk.pwconv2.bias.detach().float()))))
            cn["stages"].append(blocks)
        self.cn = cn
        t = torch.arange(64)
        i = torch.arange(3072)
        # text_grounding_net.py:230-231: feat[B,768,16,16].reshape(B,3072,64).permute(0,2,1)
        #   token t, feature i  <-  feat[c = i//4, h = (i%4)*4 + t//16, w = t%16]
        self.seg_gather = (((i[None, :] % 4) * 4 + t[:, None] // 16) * 16 + (t[:, None] % 16)) * 768 + i[None, :] // 4
        self.seg_gather = self.seg_gather.to(self.device)

    def convnext_features(self, segs: torch.Tensor) -> torch.Tensor:
        """in_conv + ConvNeXt-T forward_features on the mask stack -> NHWC feature map [B, S/32, S/32, 768]."""
        ops, cn = self.ops, self.cn
        B, Cin, S, _ = segs.shape
        P = S // 4
        pm = ops.zeros((B * P * P, 64))
        ops.seg_in_conv(segs.contiguous(), cn["in_w"], cn["in_b"], pm)
        x = ops.gemm(pm, cn["stem"].w, ops.empty((B * P * P, cn["stem"].w.shape[0])), bias=cn["stem"].b)
        C = x.shape[1]
        x = ops.layernorm(x, ops.empty(x.shape), cn["stem_ln"][0], cn["stem_ln"][1], 1e-6).view(B, P, P, C)
        H = P
        for i in range(4):
            if i > 0:
                d = cn["down"][i - 1]
                pmat = ops.layernorm_patch2(x, ops.empty((B * (H // 2) * (H // 2), 4 * C)), d["ln"][0], d["ln"][1], 1e-6)
                H //= 2
                Cn = d["conv"].w.shape[0]
                x = ops.gemm(pmat, d["conv"].w, ops.empty((B * H * H, Cn)), bias=d["conv"].b).view(B, H, H, Cn)
                C = Cn
            M = B * H * H
            for blk in cn["stages"][i]:
                y = ops.dwconv7x7(x, blk["dw_w"], blk["dw_b"], ops.empty(x.shape))
                ln = ops.zeros((M, blk["kpad"])) if blk["kpad"] != C else ops.empty((M, C))
                ops.layernorm(y.view(M, C), ln[:, :C], blk["ln"][0], blk["ln"][1], 1e-6)
                hmid = ops.gemm(ln, blk["l1"].w, ops.empty((M, 4 * C)), bias=blk["l1"].b, act="gelu")
                ops.gemm(hmid, blk["l2"].w, x.view(M, C), bias=blk["l2"].b, res=x.view(M, C))
        return x

    # =================================================================================================
    # alpha gate (utils/model.py:78-81 set_alpha_scale semantics)
    # =================================================================================================
    def set_fuser_scale(self, scale: float):
        scale = float(scale)
        if scale != self.fuser_scale:
            self.fuser_scale = scale
            self.gates.copy_(self.tanh_alphas * scale)

    def sync_fuser_scale_from_modules(self):
        from .host.attention import GatedSelfAttentionDense
        for m in self.model.modules():
            if type(m) == GatedSelfAttentionDense:
                self.set_fuser_scale(m.scale)
                return

    # =================================================================================================
    # buffers
    # =================================================================================================
    def buf(self, role: str, shape, dtype=None, zero=False) -> torch.Tensor:
        dtype = dtype or self.dtype
        key = (role, tuple(int(s) for s in shape), dtype)
        t = self._bufs.get(key)
        if t is None:
            t = (self.ops.zeros if zero else self.ops.empty)(key[1], dtype)
            self._bufs[key] = t
        return t

    # =================================================================================================
    # conditioning (step-invariant): UniFusion tokens + per-layer K/V caches
    # =================================================================================================
    def tokens(self, g: Dict[str, torch.Tensor]) -> torch.Tensor:
        """UniFusion.forward in eval mode (text_grounding_net.py:185-313) -> objs [B, 184, 768] (16-bit)."""
        ops, pn = self.ops, self.pn
        dev = self.device
        boxes = g["boxes"].to(dev, torch.float32)
        B, N, _ = boxes.shape
        m = g["masks"].to(dev, torch.float32).reshape(B * N)
        text = g["positive_embeddings"].to(dev, torch.float32).reshape(B * N, -1).contiguous()
        points = g.get("points")
        points = ((boxes[..., :2] + boxes[..., 2:]) / 2.0) if points is None else points.to(dev, torch.float32)
        scribbles = g["scribbles"].to(dev, torch.float32)
        polygons = g["polygons"].to(dev, torch.float32)
        drop_point, drop_box, drop_scribble, drop_polygons, drop_segs = pn.eval_drops()
        zero = torch.zeros_like(m)
        masks = [
            zero if drop_box else m,
            zero if drop_point else m,
            zero if drop_scribble else ((scribbles.sum(-1).reshape(-1) + m) > 0).float(),
            zero if drop_polygons else ((polygons.sum(-1).reshape(-1) + m) > 0).float(),
        ]
        locs = [boxes.reshape(B * N, 4), points.reshape(B * N, 2), scribbles.reshape(B * N, -1),
                polygons.reshape(B * N, -1)]
        nulls = [self.tok_null["box"], self.tok_null["point"], self.tok_null["scribble"], self.tok_null["polygon"]]
        objs = ops.zeros((B, OBJ_TOKENS, pn.out_dim))
        for i in range(4):
            loc = locs[i].contiguous()
            inp = ops.empty((B * N, text.shape[1] + 32 * loc.shape[1]))
            ops.unifusion_embed(text, loc, m.contiguous(), masks[i].contiguous(), self.tok_null["text"], nulls[i],
                                self.tok_freqs, inp)
            l0, l1, l2 = self.tok_mlps[i]
            h0 = ops.gemm(inp, l0.w, ops.empty((B * N, l0.w.shape[0])), bias=l0.b, act="silu")
            h1 = ops.gemm(h0, l1.w, ops.empty((B * N, l1.w.shape[0])), bias=l1.b, act="silu")
            ops.gemm(h1.view(B, N, -1), l2.w, objs[:, i * N:(i + 1) * N, :], bias=l2.b)
        # --- segmentation tokens (text_grounding_net.py:226-231, 279-285)
        segs = g["segs"]
        if segs.dim() == 4 and segs.shape[0] > 1 and segs.stride(0) == 0:
            segs = segs[:1]               # batch-broadcast mask stack (host/input.py): tokenize it once, exact hoist
        Bs = segs.shape[0]
        use_segs = (not drop_segs) and bool((segs.reshape(Bs, -1).sum(1) > 0).any())
        l0, l1, l2 = self.tok_mlps[4]
        null_in = self.tok_null["seg"].view(1, 1, -1) + self.tok_pos                         # [1, 64, 3072]
        if use_segs:
            segs_r = segs.to(dev, torch.float32)
            if segs_r.shape[-1] != pn.resize_input:                                          # :227 nearest resize
                segs_r = torch.nn.functional.interpolate(segs_r, pn.resize_input, mode="nearest")
            feat = self.convnext_features(segs_r)                                            # [Bs, 16, 16, 768]
            sf = feat.reshape(Bs, -1).float()[:, self.seg_gather.reshape(-1)].reshape(Bs, 64, -1)  # :230-231 layout
            sm = (segs_r.reshape(Bs, -1).sum(1) > 0).float().view(Bs, 1, 1)                  # :279
            seg_in = sf * sm + (1 - sm) * self.tok_null["seg"].view(1, 1, -1) + self.tok_pos  # :282-285
            rows = Bs * 64
        else:
            seg_in, rows = null_in, 64                                                       # batch-independent
        seg_in = seg_in.reshape(rows, -1).contiguous()
        seg16 = ops.cast16(seg_in, ops.empty(seg_in.shape))
        h0 = ops.gemm(seg16, l0.w, ops.empty((rows, l0.w.shape[0])), bias=l0.b, act="silu")
        h1 = ops.gemm(h0, l1.w, ops.empty((rows, l1.w.shape[0])), bias=l1.b, act="silu")
        seg_tok = ops.gemm(h1, l2.w, ops.empty((rows, l2.w.shape[0])), bias=l2.b)
        objs[:, 4 * N:, :] = seg_tok.view(-1, 64, seg_tok.shape[-1])
        return objs

    def _st_layers(self):
        for blk in self.in_blocks + [self.mid_block] + self.out_blocks:
            for p in blk:
                if p["kind"] == "st":
                    yield p

    def prepare_cond(self, context: torch.Tensor, grounding: Dict[str, torch.Tensor]) -> Cond:
        ops = self.ops
        B, n_ctx, cd = context.shape
        c = Cond(B)
        c.n_ctx = n_ctx
        c.objs = self.tokens(grounding)
        if self.masked_fuser:
            # attention.py:187-255: instance-visibility mask of the fuser attention, as membership words.  A grounding
            # input (= one reference model call) without att_masks, with boxes dropped (eval-mode drop_box_mask) or
            # with an all-zero mask tensor is unmasked; conds of different calls are concatenated row-wise later.
            from .host.attention import visibility_words
            am = grounding.get("att_masks")
            # attention.py:200: unmasked when att_masks is absent, when drop_box_mask = drop_box AND drop_polygons
            # (text_grounding_net.py:312) or when the whole per-call tensor is zero.  ``att_masks_any`` carries the
            # per-call ``torch.sum(att_masks) > 0`` decision taken on the UN-sharded batch by a rank-sharding caller.
            any_flag = grounding.get("att_masks_any")
            if am is None or self.pn.drop_box_mask or any_flag is False:
                am = torch.zeros(B, 1, MASK_RES, MASK_RES, device=self.device)
            assert am.shape[-2:] == (MASK_RES, MASK_RES), "the reference masks at the 64x64 resolution only"
            am = am.to(self.device)
            if am.shape[0] != B:
                am = am.expand(B, *am.shape[1:])
            n_objs = (OBJ_TOKENS - 64) // 4
            if am.shape[1] < n_objs:
                am = torch.cat([am, torch.zeros(B, n_objs - am.shape[1], MASK_RES, MASK_RES, device=self.device)], 1)
            c.vis = list(visibility_words(am, force_masked=bool(any_flag) and not self.pn.drop_box_mask))
        ctx16 = ops.cast16(context.to(self.device, torch.float32).contiguous(), ops.empty((B, n_ctx, cd)))
        ld_ctx, ld_obj = _round_up(n_ctx, 64), _round_up(OBJ_TOKENS, 64)
        for p in self._st_layers():
            C = p["c"]
            a2, fa = p["attn2"], p["f_kv"]
            k = ops.gemm(ctx16.view(B * n_ctx, cd), a2["wk"], ops.empty((B * n_ctx, C))).view(B, n_ctx, C)
            vt = ops.zeros((B, C, ld_ctx))
            ops.gemm(a2["wv"], ctx16, vt[:, :, :n_ctx])
            c.k_ctx.append(k)
            c.vt_ctx.append(vt)
            o = ops.gemm(c.objs.view(B * OBJ_TOKENS, -1), p["f_lin"].w, ops.empty((B * OBJ_TOKENS, C)), bias=p["f_lin"].b)
            ln = ops.layernorm(o, ops.empty((B * OBJ_TOKENS, C)), *p["f_n1"])
            ko = ops.gemm(ln, fa["wk"], ops.empty((B * OBJ_TOKENS, C))).view(B, OBJ_TOKENS, C)
            vo = ops.zeros((B, C, ld_obj))
            ops.gemm(fa["wv"], ln.view(B, OBJ_TOKENS, C), vo[:, :, :OBJ_TOKENS])
            c.k_obj.append(ko)
            c.vt_obj.append(vo)
        return c

    # =================================================================================================
    # forward
    # =================================================================================================
    def _gnp(self, role, B, HW, C):
        """Buffer for the GroupNorm partial statistics a conv3x3 leaves for the GroupNorm that reads its output, or None."""
        shape = self.ops.gn_partial_shape(B, HW, C) if (GN_EPI and hasattr(self.ops, "gn_partial_shape")) else None
        return None if shape is None else self.buf(role, shape, torch.float32)

    def _res(self, p, x, emb_all, out_role, gnp=None, want_out=True):
        """ResBlock (openaimodel.py:237-257).  ``gnp``: partial GroupNorm statistics of x left by its producer, or None.
        ``want_out``: the layer that follows normalises this block's output first (ResBlock / SpatialTransformer), so conv2 leaves
        the statistics too.  Returns (h, partial statistics of h for the next layer's GroupNorm, or None)."""
        ops = self.ops
        B, H, W, Cin = x.shape
        Cout = p["cout"]
        g = ops.groupnorm(x, self.buf("gn", x.shape), p["gn1"][0], p["gn1"][1], 1e-5, True, partial=gnp)
        rb = emb_all[:, p["emb_off"]:p["emb_off"] + Cout]
        p1 = self._gnp("gnp.h1", B, H * W, Cout)
        h1 = ops.conv3x3(g, p["conv1"].w, self.buf("rb.h1", (B, H, W, Cout)), bias=p["conv1"].b, rowbias=rb, gn_partial=p1)
        g2 = ops.groupnorm(h1, self.buf("gn", h1.shape), p["gn2"][0], p["gn2"][1], 1e-5, True, partial=p1)
        if p["skip"] is not None:
            xs = ops.gemm(x.view(B * H * W, Cin), p["skip"].w, self.buf("rb.skip", (B * H * W, Cout)),
                          bias=p["skip"].b).view(B, H, W, Cout)
        else:
            xs = x
        p2 = self._gnp("gnp.out", B, H * W, Cout) if want_out else None
        return ops.conv3x3(g2, p["conv2"].w, self.buf(out_role, (B, H, W, Cout)), bias=p["conv2"].b, res=xs, gn_partial=p2), p2

    def _self_attn(self, a, y, st, B, N, C, kv_extra=None, vis=None):
        """y [B*N, C] residual stream (raw), st [B*N, 2] its LayerNorm statistics (mu, rstd); the LayerNorm itself is folded
        into the projections (``_fold_ln``).  Returns the attention output [B, N, C] (pre out-proj).
        ``vis``: (qbits, kbits0, kbits1) visibility words of the masked gated self-attention, or None."""
        ops = self.ops
        fused = self.vt_global and N % 64 == 0 and N >= self.vt_min_n
        if fused:
            # fused q | k | v projection (round 3): ONE GEMM over the [3C, C] image reads y once; its last C columns are
            # stored TRANSPOSED straight into the batch-interleaved V^T image [C][B][N] (the attention kernel reads sample b
            # through base + b*N, ld = B*N), replacing the second GEMM V^T = Wv . y^T (M = C: 62 % tile padding at C = 320)
            # that re-read y.  The LayerNorm statistics come from the K loop itself where that is free (C = 320), else `st`.
            vtg = self.buf("st.vtg", (C, B * N))
            own = self._qkv_self(C, N)
            qk = ops.gemm(y, a["wqkv"], self.buf("st.qk", (B * N, 2 * C)), bias=a["dqkv"],
                          ln_row=(None if own else st, a["cqkv"]), ln_stats_out=st if own else None,
                          vt_out=vtg).view(B, N, 2 * C)
            vt = vtg.view(C, B, N).permute(1, 0, 2)
        else:
            # the q/k projection computes the LayerNorm statistics of y's rows in its own K loop and leaves them in st for
            # the transposed-V projection below
            if self._ln_self(C):
                qk = ops.gemm(y, a["wqk"], self.buf("st.qk", (B * N, 2 * C)), bias=a["dqk"], ln_row=(None, a["cqk"]),
                              ln_stats_out=st).view(B, N, 2 * C)
            else:
                qk = ops.gemm(y, a["wqk"], self.buf("st.qk", (B * N, 2 * C)), bias=a["dqk"], ln_row=(st, a["cqk"])).view(B, N, 2 * C)
            ldv = _round_up(N, 64)
            vt = self.buf("st.vt", (B, C, ldv), zero=True)
            ops.gemm(a["wv"], y.view(B, N, C), vt[:, :, :N], ln_col=(st.view(B, N, 2), a["cv"], a["dv"]))
        att = self.buf("st.att", (B, N, C))
        if kv_extra is None:
            ops.attention(qk[:, :, :C], qk[:, :, C:], vt, N, att, self.heads)
        elif vis is None:
            ops.attention(qk[:, :, :C], qk[:, :, C:], vt, N, att, self.heads, k1=kv_extra[0], vt1=kv_extra[1],
                          n1=OBJ_TOKENS)
        else:
            ops.attention(qk[:, :, :C], qk[:, :, C:], vt, N, att, self.heads, k1=kv_extra[0], vt1=kv_extra[1],
                          n1=OBJ_TOKENS, qbits=vis[0], kbits0=vis[1], kbits1=vis[2])
        return att

    def _qkv_self(self, C, N):
        """Does the fused q | k | v projection sum its own statistics?  Not where the row-resident kernel takes the launch."""
        return self._ln_self(C) and not (QKV_ROW and C == 320 and self.vt_global and N % 64 == 0 and N >= self.vt_min_n)

    @staticmethod
    def _ln_self(C):
        """Do the q/k and cross-q projections of a C-channel level sum their own A rows?  Mode 1: only where the projection
        is HBM-bound (C = 320: the in-loop v_dot2c sums share the matrix pipe and cost 12 % on the MFMA-bound levels, more
        than the 8-B-per-row statistics pass they replace there)."""
        return LN_SELF_MODE == 2 or (LN_SELF_MODE == 1 and C <= 320)

    def _ff(self, f, y, st, M, C, gate=None, out_stats=None):
        """y += [gate *] GEGLU-FF(LN(y)); the LayerNorm is folded into the first GEMM (statistics st, left by the GEMM that
        produced y: a GEGLU GEMM has 8-16 column tiles per row block, each of which would repeat the in-loop row sums --
        measured +15 % on those launches -- so here the separate 8-B-per-row pass is the cheaper form)."""
        ops = self.ops
        if "w2p" in f and st is not None and LN_SELF_MODE != 2 and out_stats is None and M >= MLP_MIN_M and ops.mlp_supported(M, C):
            return ops.mlp_geglu(y, st, f["w1"], f["cd"], f["w2p"], f["l2"].b, y, gate=gate)
        mid = ops.gemm(y, f["w1"], self.buf("st.ffmid", (M, 4 * C)), bias=f["b1"], geglu=True, geglu_period=GEGLU_PERIOD,
                       ln_row=(None if LN_SELF_MODE == 2 else st, f["c1"]))
        return ops.gemm(mid, f["l2"].w, y, bias=f["l2"].b, res=y, gate=gate, out_stats=out_stats)

    def _dup(self, t: torch.Tensor, role: str) -> torch.Tensor:
        """[n, ...] -> [2n, ...]: both halves a copy of t (the second half of a paired forward)."""
        n = t.shape[0]
        out = self.buf(role, (2 * n,) + tuple(t.shape[1:]), t.dtype)
        out[:n].copy_(t)
        out[n:].copy_(t)
        return out

    def _st(self, p, x, cond: Cond, fuser_on: bool, dup: bool = False, gnp=None):
        """SpatialTransformer (attention.py:366-379).  No LayerNorm kernel runs: every GEMM that reads LN(y) reads y itself
        against gamma-folded weights and applies (mu, rstd) in its epilogue.  The q/k and cross-q projections sum their A
        rows in their own K loop (the q/k one hands the statistics to the transposed-V projection); the GEGLU GEMMs take
        them from the out-projection that wrote y (``out_stats``: an 8-B-per-row pass inside that idf_gemm call).
        ``dup``: x holds the n DISTINCT rows of a paired forward (see PAIR_HOIST); the block's input and its residual stream
        are duplicated to 2n rows behind the self-attention, the last layer that does not read the conditioning."""
        ops = self.ops
        B, H, W, C = x.shape
        N, M = H * W, B * H * W
        g = ops.groupnorm(x, self.buf("gn", x.shape), p["norm"][0], p["norm"][1], 1e-6, False, partial=gnp)
        st = self.buf("st.stats", (M, 2), torch.float32)
        own = self._ln_self(C)
        own_qkv = self._qkv_self(C, N)
        # which producers emit statistics: `pqkv` those in front of a q | k | v projection, `pre` the one in front of the
        # cross-attention query, `ffs` those in front of a feed-forward
        pre, ffs = (None if own else st), (st if LN_SELF_MODE <= 1 else None)
        pqkv = None if own_qkv else st
        y = ops.gemm(g.view(M, C), p["proj_in"].w, self.buf("st.x", (M, C)), bias=p["proj_in"].b, out_stats=pqkv)
        # --- self attention (attention.py:334): LN norm1
        att = self._self_attn(p["attn1"], y, st, B, N, C)
        ops.gemm(att.view(M, C), p["attn1"]["out"].w, y, bias=p["attn1"]["out"].b, res=y,
                 out_stats=pqkv if fuser_on else pre)
        if dup:
            x, y = self._dup(x, "st.x_in2"), self._dup(y.view(B, N, C), "st.x2").view(2 * M, C)
            if (pqkv if fuser_on else pre) is not None:
                st = self._dup(st.view(B, N, 2), "st.stats2").view(2 * M, 2)
                pre = st if pre is not None else None
                pqkv = st if pqkv is not None else None
                ffs = st if ffs is not None else None
            else:
                st = self.buf("st.stats", (2 * M, 2), torch.float32)
                ffs = st if ffs is not None else None
            B, M = 2 * B, 2 * M
        # --- gated self attention over [visual ; grounding tokens] (attention.py:304-311): LN fuser.norm1 / norm2
        i = p["idx"]
        if fuser_on:
            vis = cond.vis if (cond.vis and N == MASK_RES * MASK_RES) else None
            att = self._self_attn(p["f_attn"], y, st, B, N, C, kv_extra=(cond.k_obj[i], cond.vt_obj[i]), vis=vis)
            ops.gemm(att.view(M, C), p["f_attn"]["out"].w, y, bias=p["f_attn"]["out"].b, res=y, gate=self.gates[i, 0:1],
                     out_stats=ffs)
            self._ff(p["f_ff"], y, st, M, C, gate=self.gates[i, 1:2], out_stats=pre)
        # --- cross attention (attention.py:336): LN norm2 on the query side
        a2 = p["attn2"]
        q = ops.gemm(y, a2["wq"], self.buf("st.q", (M, C)), bias=a2["dq"],
                     ln_row=(None if own else st, a2["cq"])).view(B, N, C)
        att = self.buf("st.att", (B, N, C))
        ops.attention(q, cond.k_ctx[i], cond.vt_ctx[i], cond.n_ctx, att, self.heads)
        ops.gemm(att.view(M, C), a2["out"].w, y, bias=a2["out"].b, res=y, out_stats=ffs)
        # --- feed forward (attention.py:337): LN norm3
        self._ff(p["ff"], y, st, M, C)
        # --- proj_out + x_in (attention.py:378-379), in place on the block input
        ops.gemm(y, p["proj_out"].w, x.view(M, C), bias=p["proj_out"].b, res=x.view(M, C))
        return x

    def _run_block(self, layers, h, x_nchw, emb_all, cond, fuser_on, out_role, dup_st: bool = False, gnp=None, next_kind=None):
        """One TimestepEmbedSequential.  ``gnp``: partial GroupNorm statistics of h left by the conv that produced it (or None);
        they are handed from a producing conv to the layer that follows IMMEDIATELY and dropped by anything else -- a
        SpatialTransformer rewrites its input buffer in place.  ``next_kind``: kind of the first layer of the block that consumes
        this block's output directly (None: a ScaleU concat or nothing).  Returns (h, partial statistics of h or None)."""
        for j, p in enumerate(layers):
            k = p["kind"]
            nxt = layers[j + 1]["kind"] if j + 1 < len(layers) else next_kind
            norm_next = nxt in ("res", "st")                  # the consumer's first op is a GroupNorm of this layer's output
            if k == "conv_in":
                B, _, H, W = x_nchw.shape
                h = self.ops.conv_in(x_nchw, p["w"], p["b"], self.buf(out_role, (B, H, W, p["w"].shape[0])))
                gnp = None
            elif k == "res":
                h, gnp = self._res(p, h, emb_all, out_role, gnp, want_out=norm_next)
            elif k == "st":
                h = self._st(p, h, cond, fuser_on, dup=dup_st, gnp=gnp)
                gnp = None
            elif k == "down":
                B, H, W, C = h.shape
                Ho, Wo = (H + 1) // 2, (W + 1) // 2
                gnp = self._gnp("gnp.out", B, Ho * Wo, C) if norm_next else None
                h = self.ops.conv3x3(h, p["conv"].w, self.buf(out_role, (B, Ho, Wo, C)), bias=p["conv"].b, stride=2, gn_partial=gnp)
            elif k == "up":
                B, H, W, C = h.shape
                h = self.ops.conv3x3(h, p["conv"].w, self.buf(out_role + ".up", (B, 2 * H, 2 * W, C)),
                                     bias=p["conv"].b, upsample=1)
                gnp = None                                      # feeds the ScaleU concat, not a GroupNorm
        return h, gnp

    def _forward_ops(self, x: torch.Tensor, t_f32: torch.Tensor, cond: Cond, eps: torch.Tensor, fuser_on: bool,
                     paired: bool = False):
        """Enqueue one UNet forward (openaimodel.py:482-563).  x [B,4,H,W] fp32, t_f32 [B], eps [B,4,H,W] fp32.
        ``paired``: rows [B/2, B) repeat the latent and timestep of rows [0, B/2) (see PAIR_HOIST)."""
        ops = self.ops
        B = x.shape[0]
        # the hoist needs the reference layout: input block 0 = first conv, input block 1 = ResBlock + SpatialTransformer
        hoist = (paired and PAIR_HOIST and B % 2 == 0 and len(self.in_blocks) > 1
                 and [p["kind"] for p in self.in_blocks[0]] == ["conv_in"] and [p["kind"] for p in self.in_blocks[1]] == ["res", "st"])
        n = B // 2
        mc = self.model.model_channels
        te = ops.timestep_embedding(t_f32, self.buf("temb.sin", (B, mc)))
        e1 = ops.gemm(te, self.te0.w, self.buf("temb.1", (B, 4 * mc)), bias=self.te0.b, act="silu")
        es = ops.gemm(e1, self.te2.w, self.buf("temb.2", (B, 4 * mc)), bias=self.te2.b, act="silu")   # silu(emb)
        emb_all = ops.gemm(es, self.emb_all.w, self.buf("temb.all", (B, self.emb_total)), bias=self.emb_all.b)
        hs = []
        h, gnp = None, None
        for i, layers in enumerate(self.in_blocks):
            nk = (self.in_blocks[i + 1] if i + 1 < len(self.in_blocks) else self.mid_block)[0]["kind"]
            if hoist and i == 0:
                h, gnp = self._run_block(layers, h, x[:n], emb_all[:n], cond, fuser_on, "in0.half")
                hs.append(self._dup(h, "in0"))
                continue
            if hoist and i == 1:
                h, gnp = self._run_block(layers, h, x[:n], emb_all[:n], cond, fuser_on, "in1.half", dup_st=True, gnp=gnp, next_kind=nk)
                hs.append(h)
                continue
            h, gnp = self._run_block(layers, h, x, emb_all, cond, fuser_on, f"in{i}", gnp=gnp, next_kind=nk)
            hs.append(h)
        h, gnp = self._run_block(self.mid_block, h, x, emb_all, cond, fuser_on, "mid", gnp=gnp)
        for i, layers in enumerate(self.out_blocks):
            skip = hs.pop()
            Bq, H, W, Ch = h.shape
            cat = ops.scaleu_concat(h, skip, self.buf("cat", (Bq, H, W, Ch + skip.shape[-1])), *self.scaleu[i])
            h, gnp = self._run_block(layers, cat, x, emb_all, cond, fuser_on, f"out{i}")
        g = ops.groupnorm(h, self.buf("gn", h.shape), self.out_gn[0], self.out_gn[1], 1e-5, True)
        ops.conv3x3(g, self.out_conv.w, eps, bias=self.out_conv.b, n_valid=self.n_out)
        return eps

    def _bind(self, cond: Cond) -> Cond:
        """Copy ``cond`` into the static per-batch-size slot that captured graphs read (no-op if already bound)."""
        B = cond.B
        slot = self._slots.get(B)
        if slot is None or not slot.same_layout(cond):
            slot = cond.clone()
            self._slots[B] = slot
            self._slot_bound[B] = cond.uid
            for k in [k for k in self._graphs if k[0] == B]:
                del self._graphs[k]
        elif self._slot_bound[B] != cond.uid:
            slot.copy_from(cond)
            self._slot_bound[B] = cond.uid
        return slot

    def gather_cond(self, bank: Cond, idx: torch.Tensor) -> Cond:
        """Fill the static slot of batch size len(idx) with rows ``idx`` of ``bank`` (ONE gather per tensor, written
        in place) and return the slot; ``forward_cond(x, t, slot)`` then replays the captured graph without any further
        conditioning copy.  Used by the samplers to assemble [cond | uncond] / per-unit batches."""
        B = int(idx.numel())
        slot = self._slots.get(B)
        if slot is None or len(slot._tensors()) != len(bank._tensors()) or any(
                a.shape[1:] != b.shape[1:] for a, b in zip(slot._tensors(), bank._tensors())) or slot.n_ctx != bank.n_ctx:
            slot = bank.select(idx)
            self._slots[B] = slot
            for k in [k for k in self._graphs if k[0] == B]:
                del self._graphs[k]
        else:
            for dst, src in zip(slot._tensors(), bank._tensors()):
                torch.index_select(src, 0, idx, out=dst)
        self._slot_bound[B] = slot.uid
        return slot

    def forward_cond(self, x: torch.Tensor, t: torch.Tensor, cond: Cond, out: Optional[torch.Tensor] = None,
                     paired: bool = False) -> torch.Tensor:
        """eps = UNet(x, t | cond).  The launch sequence is captured once per (batch, resolution, fuser on/off, paired)
        into a hipGraph over static buffers and replayed; conditioning is copied into a static slot when it changes.
        ``paired``: a [cond | uncond] guidance batch -- rows [B/2, B) carry the latent and timestep of rows [0, B/2), and the
        conditioning-free prefix of the network runs once (PAIR_HOIST).  Pass the B/2 DISTINCT rows (x: [B/2, ...], t: [B/2]): the
        engine writes both halves of its static input itself, so the invariant holds by construction.  A full
        [B, ...] batch is still accepted; it is then VERIFIED the first time a launch configuration is used (one compare + host
        sync per (B, H, W, fuser, paired) key; on every call with IDF_DEBUG_PAIRED=1), not on later calls."""
        B = cond.B
        half = bool(paired) and x.shape[0] * 2 == B
        if not half:
            assert x.shape[0] == B
        _, Cx, H, W = x.shape
        fuser_on = self.fuser_scale != 0.0
        key = (B, H, W, fuser_on, bool(paired))
        if paired and not half and (DEBUG_PAIRED or key not in self._paired_checked):
            n = B // 2
            if B % 2 or not (torch.equal(x[:n], x[n:]) and torch.equal(t[:n], t[n:])):
                raise ValueError("forward_cond(paired=True): rows [B/2, B) must repeat the latent and timestep of rows [0, B/2)")
            self._paired_checked.add(key)
        x_s = self.buf("io.x", (B,) + tuple(x.shape[1:]), torch.float32)
        t_s = self.buf("io.t", (B,), torch.float32)
        eps_s = self.buf("io.eps", (B, self.n_out, H, W), torch.float32)
        if half:
            n = B // 2
            x_s[:n].copy_(x)
            x_s[n:].copy_(x)
            t_s[:n].copy_(t)
            t_s[n:].copy_(t)
        else:
            x_s.copy_(x)
            t_s.copy_(t)
        if not self.use_graphs:
            self._forward_ops(x_s, t_s, cond, eps_s, fuser_on, paired)
        else:
            slot = self._bind(cond)
            graph = self._graphs.get(key)
            if graph is None:
                # eager warm-up sizes every buffer, then capture the identical launch sequence
                self._forward_ops(x_s, t_s, slot, eps_s, fuser_on, paired)
                torch.cuda.synchronize()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    self._forward_ops(x_s, t_s, slot, eps_s, fuser_on, paired)
                self._graphs[key] = graph
            graph.replay()
        if out is None:
            return eps_s.clone()
        out.copy_(eps_s)
        return out

    # ---- reference-style single call: model(input_dict) -----------------------------------------------
    def forward(self, x, timesteps, context, grounding) -> torch.Tensor:
        self.sync_fuser_scale_from_modules()
        key = (context.data_ptr(), context._version, tuple(context.shape)) + tuple(
            (k, v.data_ptr(), v._version) for k, v in sorted(grounding.items()) if torch.is_tensor(v))
        cond = self._cond_cache.get(key)
        if cond is None:
            if len(self._cond_cache) > 64:
                self._cond_cache.clear()
            cond = self.prepare_cond(context, grounding)
            cond._keepalive = (context, grounding)           # data_ptr keys stay unique while these live
            self._cond_cache[key] = cond
        return self.forward_cond(x.to(self.device, torch.float32), timesteps.to(self.device), cond)
