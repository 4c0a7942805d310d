"""Tensor-level wrappers over the C ABI (``include/idf.h``): derive pointers / leading dimensions / strides from
(possibly strided) torch views and enqueue the HIP kernel on torch's current stream.

PyTorch here is plumbing only (device memory + stream); every numeric op is a hand-written gfx950 kernel.
``HipOps()`` raises if the shared library is missing or no GPU is visible -- there is no CPU path.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib
from ._lib import (EPI_BIAS, EPI_GATE, EPI_GEGLU, EPI_GEGLU_P32, EPI_GELU, EPI_LN_COL, EPI_LN_ROW, EPI_OUT_F32, EPI_OUT_NCHW,
                   EPI_RES, EPI_ROWBIAS, EPI_SILU)

_DT = {torch.bfloat16: _lib.IDF_BF16, torch.float16: _lib.IDF_F16}


def _p(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _rows2d(t: torch.Tensor):
    """(ld, batch_stride) of a [.., R, Ccols] view whose last dim is contiguous."""
    assert t.stride(-1) == 1 or t.shape[-1] == 1, "last dim must be contiguous"
    return t.stride(-2), (t.stride(0) if t.dim() == 3 else 0)


class HipOps:
    """The product backend.  One instance per compute dtype (bf16 default, fp16 supported by every kernel)."""

    def __init__(self, dtype: torch.dtype = torch.bfloat16, device: Optional[torch.device] = None):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError("no MI355X visible: the InstanceDiffusion HIP path needs a GPU (no CPU fallback)")
        self.dtype = dtype
        self.dt = _DT[dtype]
        self.device = device or torch.device("cuda", torch.cuda.current_device())
        self._ws = {}

    # ------------------------------------------------------------------------------------------------
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def _workspace(self, key, nfloats: int) -> torch.Tensor:
        """fp32 scratch keyed by (role, size): a buffer is NEVER reallocated or freed once handed out, because
        captured hipGraphs keep its raw device pointer."""
        k = (key, int(nfloats))
        w = self._ws.get(k)
        if w is None:
            w = torch.empty(int(nfloats), dtype=torch.float32, device=self.device)
            self._ws[k] = w
        return w

    SPLITK_WS_BYTES = 256 << 20

    def _splitk_ws(self) -> torch.Tensor:
        return self._workspace("splitk", self.SPLITK_WS_BYTES // 4)

    def empty(self, shape, dtype=None):
        return torch.empty(shape, dtype=dtype or self.dtype, device=self.device)

    def zeros(self, shape, dtype=None):
        return torch.zeros(shape, dtype=dtype or self.dtype, device=self.device)

    # ------------------------------------------------------------------------------------------------
    def gemm(self, a, w, out, *, bias=None, rowbias=None, rows_per_batch=0, res=None, gate=None,
             act: Optional[str] = None, geglu: bool = False, geglu_period: int = 64, ln_row=None, ln_col=None, out_stats=None, out_stats_eps=1e-5,
             ln_eps=1e-5, ln_stats_out=None, vt_out=None):
        """out[..,M,N] = epi(a[..,M,K] @ w[..,N,K]^T).  2-D or batched 3-D views; a/w may be shared (2-D) in a
        batched call.  geglu: ``w``/``bias`` are in the packed [P/2 value | P/2 gate] row order (``geglu_period`` P = 64 or 32,
        engine.pack_geglu), out has N/2 cols.
        ln_row = (stats [.., M, 2], c [N]): ``a`` is the RAW input of a LayerNorm whose gamma is folded into ``w`` and whose
        beta term travels in ``bias`` (include/idf.h IDF_EPI_LN_ROW).  ln_col = (stats [.., N, 2], c [M], d [M]): the same
        with the normalised operand on the ``w`` side (transposed-V projection).  out_stats [.., M, 2]: also emit (mu, rstd)
        of every output row for a LayerNorm that follows.  vt_out [N - out.shape[-1], >= M]: fused q | k | v projection --
        the product's columns beyond out's width are stored transposed there (include/idf.h)."""
        batched = out.dim() == 3
        M, K = a.shape[-2], a.shape[-1]
        N = w.shape[-2]
        assert w.shape[-1] == K
        lda, sA = _rows2d(a)
        ldw, sW = _rows2d(w)
        ldo, sO = _rows2d(out)
        epi = 0
        if bias is not None:
            epi |= EPI_BIAS
        if rowbias is not None:
            epi |= EPI_ROWBIAS
        ldr = sR = 0
        if res is not None:
            epi |= EPI_RES
            ldr, sR = _rows2d(res)
        if gate is not None:
            epi |= EPI_GATE
        if act == "silu":
            epi |= EPI_SILU
        elif act == "gelu":
            epi |= EPI_GELU
        elif act is not None:
            raise ValueError(act)
        if geglu:
            epi |= EPI_GEGLU
            assert geglu_period in (32, 64)
            if geglu_period == 32:
                epi |= EPI_GEGLU_P32
            assert out.shape[-1] == N // 2
        elif vt_out is not None:
            assert not batched and out.shape[-2] == M and vt_out.shape[0] == N - out.shape[-1] and vt_out.shape[1] >= M
            assert vt_out.stride(1) == 1 and vt_out.dtype == self.dtype
        else:
            assert out.shape[-1] == N and out.shape[-2] == M
        if out.dtype == torch.float32:
            epi |= EPI_OUT_F32
        ln_stats = ln_c = ln_d = None
        s_ln = 0
        if ln_row is not None:
            ln_stats, ln_c = ln_row                     # ln_stats None: the GEMM computes the row statistics itself
            assert bias is not None and ln_c.numel() == N
            assert ln_stats is None or (ln_stats.shape[-2] == M and ln_stats.dtype == torch.float32)
            if ln_stats is None:
                assert not batched
                if ln_stats_out is not None:
                    assert ln_stats_out.is_contiguous() and ln_stats_out.dtype == torch.float32 and ln_stats_out.numel() == 2 * M
            epi |= EPI_LN_ROW
        elif ln_col is not None:
            ln_stats, ln_c, ln_d = ln_col
            assert ln_stats.shape[-2] == N and ln_c.numel() == M and ln_d.numel() == M and ln_stats.dtype == torch.float32
            epi |= EPI_LN_COL
        if ln_stats is not None:
            assert ln_stats.is_contiguous() and ln_stats.shape[-1] == 2
            s_ln = ln_stats.stride(0) if (batched and ln_stats.dim() == 3) else 0
        if out_stats is not None:
            assert out_stats.is_contiguous() and out_stats.dtype == torch.float32 and out_stats.numel() == 2 * out.numel() // N
        args = _lib.GemmArgs(
            A=a.data_ptr(), W=w.data_ptr(), out=out.data_ptr(),
            bias=None if bias is None else bias.data_ptr(),
            rowbias=None if rowbias is None else rowbias.data_ptr(),
            res=None if res is None else res.data_ptr(),
            gate=None if gate is None else gate.data_ptr(),
            M=M, N=N, K=K, lda=lda, ldw=ldw, ldo=ldo, ldr=ldr,
            ld_rowbias=0 if rowbias is None else rowbias.stride(-2), rows_per_batch=rows_per_batch,
            batch=out.shape[0] if batched else 1, strideA=sA, strideW=sW, strideO=sO, strideR=sR,
            epi=epi, dtype=self.dt, ws=self._splitk_ws().data_ptr(), ws_bytes=self.SPLITK_WS_BYTES,
            ln_stats=None if ln_stats is None else ln_stats.data_ptr(), stride_ln_stats=s_ln,
            ln_c=None if ln_c is None else ln_c.data_ptr(), ln_d=None if ln_d is None else ln_d.data_ptr(),
            out_stats=None if out_stats is None else out_stats.data_ptr(), out_stats_eps=float(out_stats_eps),
            ln_eps=float(ln_eps), ln_stats_out=None if ln_stats_out is None else ln_stats_out.data_ptr(),
            vt_out=None if vt_out is None else vt_out.data_ptr(), ld_vt=0 if vt_out is None else vt_out.stride(0),
            vt_col0=0 if vt_out is None else out.shape[-1])
        _lib.check(self.lib.idf_gemm(C.byref(args), self._stream()), "idf_gemm")
        return out

    # permutation of the k index inside every 16-group of the fused MLP's second weight (include/idf.h idf_mlp_geglu): an
    # involution (it swaps positions 4..7 and 8..11)
    MLP_W2_PERM = (0, 1, 2, 3, 8, 9, 10, 11, 4, 5, 6, 7, 12, 13, 14, 15)

    @staticmethod
    def mlp_supported(M: int, C: int) -> bool:
        """Shapes idf_mlp_geglu takes (the 64 x 64-latent blocks: C = 320, whole 128-row tiles)."""
        return C == 320 and M % 128 == 0

    @classmethod
    def mlp_pack(cls, w1p16, c1, d1, w2_16):
        """Operand images of idf_mlp_geglu from the period-32-packed first projection (16-bit weight, fp32 row sums c1 and
        bias / beta term d1) and the 16-bit second weight [C, 4C]: (cd [8C/64][128] fp32, w2p)."""
        n2 = c1.numel()
        cd = torch.cat([c1.float().view(n2 // 64, 64), d1.float().view(n2 // 64, 64)], dim=1).contiguous()
        perm = torch.tensor(cls.MLP_W2_PERM, device=w2_16.device)
        n, k = w2_16.shape
        w2p = w2_16.view(n, k // 16, 16).index_select(2, perm).reshape(n, k).contiguous()
        return cd, w2p

    def mlp_geglu(self, x, stats, w1, cd, w2p, b2, out, *, gate=None):
        """out = x + [gate *] FF(LN(x)) in one launch (idf_mlp_geglu); x / out [M, 320] views (may alias), stats [M, 2]."""
        M, Cc = x.shape
        assert self.mlp_supported(M, Cc) and out.shape == x.shape and x.stride(-1) == 1 and out.stride(-1) == 1
        assert stats.is_contiguous() and stats.dtype == torch.float32 and stats.numel() == 2 * M
        assert w1.shape == (8 * Cc, Cc) and w2p.shape == (Cc, 4 * Cc) and cd.is_contiguous() and cd.numel() == 16 * Cc
        assert w1.stride(-1) == 1 and w2p.stride(-1) == 1 and b2.dtype == torch.float32 and cd.dtype == torch.float32
        args = _lib.MlpArgs(x=x.data_ptr(), ldx=x.stride(0), ln_stats=stats.data_ptr(), w1=w1.data_ptr(), ldw1=w1.stride(0),
                            cd=cd.data_ptr(), w2p=w2p.data_ptr(), ldw2=w2p.stride(0), b2=b2.data_ptr(),
                            gate=None if gate is None else gate.data_ptr(), out=out.data_ptr(), ldo=out.stride(0), M=M, C=Cc,
                            dtype=self.dt)
        _lib.check(self.lib.idf_mlp_geglu(C.byref(args), self._stream()), "idf_mlp_geglu")
        return out

    @staticmethod
    def gn_partial_shape(B, HW, C):
        """Shape of the GroupNorm partial-statistics buffer a conv3x3 can fill for its output ([B, HW/64, 32, 2] fp32), or
        None when the output does not qualify (whole 64-row chunks per sample, 32 groups)."""
        return (B, HW // 64, 32, 2) if (HW % 64 == 0 and C % 32 == 0) else None

    def conv3x3(self, x, w, out, *, bias=None, rowbias=None, res=None, stride=1, upsample=0, n_valid=0, gn_partial=None):
        """x [B,H,W,Cin] view (channel-contiguous), w [Cout, 9*Cin]; out [B,Ho,Wo,Cout] 16-bit, or fp32 NCHW
        [B,n_valid,Ho,Wo] when ``n_valid`` > 0 (final conv).  ``gn_partial`` (fp32 [B, Ho*Wo/64, 32, 2], see
        ``gn_partial_shape``): filled with the GroupNorm partial statistics of the output, for ``groupnorm(.., partial=)``."""
        B, H, W_, Cin = x.shape
        assert x.stride(-1) == 1 and x.stride(1) == W_ * x.stride(2) and x.stride(0) == H * x.stride(1)
        Cout = w.shape[0]
        epi = 0
        if bias is not None:
            epi |= EPI_BIAS
        if rowbias is not None:
            epi |= EPI_ROWBIAS
        if res is not None:
            epi |= EPI_RES
        if n_valid:
            epi |= EPI_OUT_NCHW
            assert out.dtype == torch.float32 and out.is_contiguous()
            ldo = 0
        else:
            ldo = out.stride(2)
        args = _lib.ConvArgs(
            x=x.data_ptr(), W=w.data_ptr(), out=out.data_ptr(),
            bias=None if bias is None else bias.data_ptr(),
            rowbias=None if rowbias is None else rowbias.data_ptr(),
            res=None if res is None else res.data_ptr(),
            B=B, Hin=H, Win=W_, Cin=Cin, Cout=Cout, stride=stride, upsample=upsample,
            ldx=x.stride(2), ldo=ldo, ldr=0 if res is None else res.stride(2),
            ld_rowbias=0 if rowbias is None else rowbias.stride(-2), n_valid=n_valid, epi=epi, dtype=self.dt,
            ws=self._splitk_ws().data_ptr(), ws_bytes=self.SPLITK_WS_BYTES,
            gn_partial=None if gn_partial is None else gn_partial.data_ptr())
        if gn_partial is not None:
            assert gn_partial.dtype == torch.float32 and gn_partial.is_contiguous() and out.is_contiguous()
            assert tuple(gn_partial.shape) == self.gn_partial_shape(out.shape[0], out.shape[1] * out.shape[2], Cout)
        _lib.check(self.lib.idf_conv3x3(C.byref(args), self._stream()), "idf_conv3x3")
        return out

    def conv_in(self, x_nchw, w, bias, out):
        B, Cin, H, W_ = x_nchw.shape
        assert x_nchw.dtype == torch.float32 and x_nchw.is_contiguous() and out.is_contiguous()
        _lib.check(self.lib.idf_conv_in(_p(x_nchw), _p(w), _p(bias), _p(out), B, Cin, H, W_, out.shape[-1], self.dt,
                                        self._stream()), "idf_conv_in")
        return out

    def attention(self, q, k0, vt0, n0, out, heads, *, k1=None, vt1=None, n1=0, qbits=None, kbits0=None, kbits1=None):
        """q [B,Nq,C] view, k [B,n,C] view, vt [B,C,ld>=ceil64(n)], out [B,Nq,C] view.  Optional visibility mask:
        int32 words qbits [B,Nq], kbits0 [B,>=n0], kbits1 [B,>=n1] (see include/idf.h)."""
        if qbits is not None:
            for t in (qbits, kbits0) + ((kbits1,) if n1 else ()):
                assert t.dtype == torch.int32 and t.stride(-1) == 1 and t.dim() == 2
        B, Nq, Cc = q.shape
        d = Cc // heads
        a = _lib.AttnArgs(
            q=q.data_ptr(), ldq=q.stride(1), strideQ=q.stride(0), nq=Nq,
            k0=k0.data_ptr(), ldk0=k0.stride(1), strideK0=k0.stride(0),
            vt0=vt0.data_ptr(), ldv0=vt0.stride(1), strideV0=vt0.stride(0), n0=n0,
            k1=None if k1 is None else k1.data_ptr(), ldk1=0 if k1 is None else k1.stride(1),
            strideK1=0 if k1 is None else k1.stride(0),
            vt1=None if vt1 is None else vt1.data_ptr(), ldv1=0 if vt1 is None else vt1.stride(1),
            strideV1=0 if vt1 is None else vt1.stride(0), n1=n1,
            out=out.data_ptr(), ldo=out.stride(1), strideO=out.stride(0),
            B=B, H=heads, d=d, scale=float(d) ** -0.5, dtype=self.dt,
            qbits=None if qbits is None else qbits.data_ptr(), strideQb=0 if qbits is None else qbits.stride(0),
            kbits0=None if qbits is None else kbits0.data_ptr(), strideKb0=0 if qbits is None else kbits0.stride(0),
            kbits1=None if (qbits is None or not n1) else kbits1.data_ptr(),
            strideKb1=0 if (qbits is None or not n1) else kbits1.stride(0))
        _lib.check(self.lib.idf_attention(C.byref(a), self._stream()), "idf_attention")
        return out

    def groupnorm(self, x, out, gamma, beta, eps, silu, partial=None):
        """x/out [B, HW, C] (or [B,H,W,C]) contiguous.  ``partial``: the statistics of x are already there ([B, nchunks, 32, 2]
        fp32 (mean, M2) per row chunk, as a conv3x3 leaves them): only the normalise pass runs."""
        assert x.is_contiguous() and out.is_contiguous()
        B, Cc = x.shape[0], x.shape[-1]
        HW = x.numel() // (B * Cc)
        if partial is not None:
            assert partial.dtype == torch.float32 and partial.is_contiguous() and partial.shape[0] == B and partial.shape[2:] == (32, 2)
            _lib.check(self.lib.idf_groupnorm_apply(_p(x), _p(out), _p(gamma), _p(beta), _p(partial), B, HW, Cc,
                                                    int(partial.shape[1]), float(eps), int(bool(silu)), self.dt,
                                                    self._stream()), "idf_groupnorm_apply")
            return out
        ws = self._workspace("gn", self.lib.idf_groupnorm_ws_floats(B, HW))
        _lib.check(self.lib.idf_groupnorm(_p(x), _p(out), _p(gamma), _p(beta), _p(ws), B, HW, Cc, float(eps),
                                          int(bool(silu)), self.dt, self._stream()), "idf_groupnorm")
        return out

    def layernorm(self, x, out, gamma, beta, eps=1e-5):
        """x/out [M, C] views (last dim contiguous)."""
        M, Cc = x.shape
        _lib.check(self.lib.idf_layernorm(_p(x), x.stride(0), _p(out), out.stride(0), _p(gamma), _p(beta), M, Cc,
                                          float(eps), self.dt, self._stream()), "idf_layernorm")
        return out

    def row_stats(self, x, stats, eps=1e-5):
        """stats[m] = (mean, rstd) of row m of x [M, C] (fp32 [M, 2]): the LayerNorm statistics a following gemm(ln_row /
        ln_col) applies in its epilogue."""
        M, Cc = x.shape
        assert stats.is_contiguous() and stats.dtype == torch.float32 and stats.numel() == 2 * M
        _lib.check(self.lib.idf_row_stats(_p(x), x.stride(0), _p(stats), M, Cc, float(eps), self.dt, self._stream()),
                   "idf_row_stats")
        return stats

    def layernorm_patch2(self, x, out, gamma, beta, eps):
        """x [B,H,W,C] contiguous; out [B*(H/2)*(W/2), >=4C] rows in 2x2 patch order."""
        B, H, W_, Cc = x.shape
        assert x.is_contiguous() and out.stride(-1) == 1
        _lib.check(self.lib.idf_layernorm_patch2(_p(x), _p(out), out.stride(0), _p(gamma), _p(beta), B, H, W_, Cc,
                                                 float(eps), self.dt, self._stream()), "idf_layernorm_patch2")
        return out

    def seg_in_conv(self, segs, w, bias, out):
        """segs fp32 [B,Cin,S,S]; out [B*(S/4)^2, ld>=48] 16-bit stem patch matrix."""
        B, Cin, S, _ = segs.shape
        assert segs.is_contiguous() and segs.dtype == torch.float32
        _lib.check(self.lib.idf_seg_in_conv(_p(segs), _p(w), _p(bias), _p(out), B, Cin, S, out.stride(0), self.dt,
                                            self._stream()), "idf_seg_in_conv")
        return out

    def dwconv7x7(self, x, w_tap_major, bias, out):
        B, H, W_, Cc = x.shape
        assert x.is_contiguous() and out.is_contiguous()
        _lib.check(self.lib.idf_dwconv7x7(_p(x), _p(w_tap_major), _p(bias), _p(out), B, H, W_, Cc, self.dt,
                                          self._stream()), "idf_dwconv7x7")
        return out

    def scaleu_concat(self, h, skip, out, hscale, sm1):
        B, H, W_, Ch = h.shape
        Cs = skip.shape[-1]
        assert h.is_contiguous() and skip.is_contiguous() and out.is_contiguous() and out.shape[-1] == Ch + Cs
        ws = self._workspace("scaleu", B * Cs * 64)
        _lib.check(self.lib.idf_scaleu_concat(_p(h), _p(skip), _p(out), _p(hscale), _p(sm1), _p(ws), B, H, W_, Ch, Cs,
                                              self.dt, self._stream()), "idf_scaleu_concat")
        return out

    def timestep_embedding(self, t_f32, out):
        B, dim = out.shape
        _lib.check(self.lib.idf_timestep_embedding(_p(t_f32), _p(out), B, dim, self.dt, self._stream()),
                   "idf_timestep_embedding")
        return out

    def unifusion_embed(self, text, loc, tmask, lmask, null_text, null_loc, freqs, out):
        rows, text_dim = text.shape
        D = loc.shape[-1]
        assert out.is_contiguous() and out.shape == (rows, text_dim + 32 * D)
        for t in (text, loc, tmask, lmask, null_text, null_loc, freqs):
            assert t.dtype == torch.float32 and t.is_contiguous()
        _lib.check(self.lib.idf_unifusion_embed(_p(text), _p(loc), _p(tmask), _p(lmask), _p(null_text), _p(null_loc),
                                                _p(freqs), _p(out), rows, text_dim, D, self.dt, self._stream()),
                   "idf_unifusion_embed")
        return out

    def cfg_combine(self, e_cond, e_uncond, guidance, out):
        _lib.check(self.lib.idf_cfg_combine(_p(e_cond), _p(e_uncond), float(guidance), _p(out), out.numel(),
                                            self._stream()), "idf_cfg_combine")
        return out

    def plms_update(self, x, e_t, old, e_next, mode, a_t, a_prev, sqrt_1m_at, out):
        e1 = old[-1] if len(old) >= 1 else None
        e2 = old[-2] if len(old) >= 2 else None
        e3 = old[-3] if len(old) >= 3 else None
        _lib.check(self.lib.idf_plms_update(_p(x), _p(e_t), _p(e1), _p(e2), _p(e3), _p(e_next), int(mode), float(a_t),
                                            float(a_prev), float(sqrt_1m_at), _p(out), out.numel(), self._stream()),
                   "idf_plms_update")
        return out

    def mis_merge(self, lat, boxes_i32, out, mode):
        n1, B, Cc, H, W_ = lat.shape
        assert lat.is_contiguous() and out.is_contiguous()
        _lib.check(self.lib.idf_mis_merge(_p(lat), _p(boxes_i32), _p(out), n1 - 1, B, Cc, H, W_, int(mode),
                                          self._stream()), "idf_mis_merge")
        return out

    def cast16(self, x_f32, out):
        assert x_f32.is_contiguous() and out.is_contiguous()
        _lib.check(self.lib.idf_cast_f32_to_16(_p(x_f32), _p(out), out.numel(), self.dt, self._stream()),
                   "idf_cast_f32_to_16")
        return out

    def softmax_rows(self, s_f32, out, scale):
        """s_f32 [.., R, n] fp32 (rows stacked contiguously), out same shape 16-bit: out = softmax(scale * s, -1)."""
        assert s_f32.dtype == torch.float32 and s_f32.is_contiguous() and out.is_contiguous() and out.shape == s_f32.shape
        n = s_f32.shape[-1]
        rows = s_f32.numel() // n
        _lib.check(self.lib.idf_softmax_rows(_p(s_f32), _p(out), rows, n, n, n, float(scale), self.dt, self._stream()),
                   "idf_softmax_rows")
        return out

    def pointwise_nchw(self, x, w, bias, out, in_scale=1.0):
        """fp32 NCHW 1x1 conv between small channel counts: out = w @ (in_scale * x) + bias."""
        B, Cin = x.shape[0], x.shape[1]
        Cout = w.shape[0]
        HW = x.numel() // (B * Cin)
        for t in (x, w, out) + ((bias,) if bias is not None else ()):
            assert t.dtype == torch.float32 and t.is_contiguous()
        _lib.check(self.lib.idf_pointwise_nchw(_p(x), _p(w), _p(bias), _p(out), B, Cin, Cout, HW, float(in_scale),
                                               self._stream()), "idf_pointwise_nchw")
        return out
