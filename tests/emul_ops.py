"""CPU emulation of the C-ABI op set -- TEST DOUBLE ONLY (lives under tests/, never imported by the product).

Purpose: run the *host logic* of the engine (weight packing, op sequencing, buffer aliasing, Cond caches,
sampler control flow) on a machine without a GPU, with the same 16-bit storage rounding points as the HIP kernels,
so that (a) orchestration bugs are caught by the CPU suite and (b) the expected bf16-vs-fp32 error is known before
going to the GPU.  The product path (`instancediffusion_amd.ops.HipOps`) has no such fallback and raises without
the HIP library.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


class EmulOps:
    def __init__(self, dtype=torch.bfloat16, batch_invariant=False):
        # batch_invariant: make every op's result for one batch row independent of which other rows share the call, the
        # property the HIP kernels have per launch configuration.  torch's CPU kernels do not have it for free (M < 64
        # matmuls take a gemv path with another summation order, batch-1 convolutions another algorithm), so matmuls
        # run in fixed 256-row blocks (the last one zero-padded) and convolutions sample by sample.  Used by the
        # world-size tests, which require W-rank output == 1-rank output bit for bit.
        self.batch_invariant = batch_invariant
        self.dtype = dtype
        self.device = torch.device("cpu")
        self.calls = {}
        self.rows = {}            # batch rows that went through an op (work measure: launches x batch)

    def _count(self, name):
        self.calls[name] = self.calls.get(name, 0) + 1

    def empty(self, shape, dtype=None):
        # poison fresh buffers so that reads of never-written memory are caught
        t = torch.empty(shape, dtype=dtype or self.dtype)
        if t.dtype.is_floating_point:
            t.fill_(float("nan"))
        return t

    def zeros(self, shape, dtype=None):
        return torch.zeros(shape, dtype=dtype or self.dtype)

    # ----------------------------------------------------------------------------------------------
    def gemm(self, a, w, out, *, bias=None, rowbias=None, rows_per_batch=0, res=None, gate=None, act=None, geglu=False,
             geglu_period=64, ln_row=None, ln_col=None, out_stats=None, out_stats_eps=1e-5, ln_eps=1e-5, ln_stats_out=None, vt_out=None):
        self._count("gemm")
        assert a.dtype == self.dtype and w.dtype == self.dtype
        assert a.shape[-1] % 64 == 0, "K % 64"
        af32 = a.float()
        if self.batch_invariant:                       # fixed 256-row blocks (the last one zero-padded), see __init__
            M, BLK = af32.shape[-2], 256
            wt = w.float().transpose(-1, -2)
            blocks = []
            for m0 in range(0, M, BLK):
                blk = af32[..., m0:m0 + BLK, :]
                if blk.shape[-2] < BLK:
                    blk = torch.cat([blk, torch.zeros(*blk.shape[:-2], BLK - blk.shape[-2], blk.shape[-1])], -2)
                blocks.append(torch.matmul(blk, wt)[..., :min(BLK, M - m0), :])
            acc = torch.cat(blocks, -2)
        else:
            acc = torch.matmul(af32, w.float().transpose(-1, -2))
        if ln_row is not None:                         # include/idf.h IDF_EPI_LN_ROW: rstd_m * (acc - mu_m c_n) (+ d as bias)
            st, c = ln_row
            assert bias is not None
            if st is None:                             # self-normalising: statistics of a's rows (the 16-bit values)
                af = a.float()
                st = torch.stack([af.mean(-1), torch.rsqrt(af.var(-1, unbiased=False) + ln_eps)], -1)
                if ln_stats_out is not None:
                    ln_stats_out.reshape(-1, 2).copy_(st.reshape(-1, 2))
            acc = st[..., 1:2] * (acc - st[..., 0:1] * c.reshape(1, -1))
        if ln_col is not None:                         # IDF_EPI_LN_COL: rstd_n * (acc - c_m mu_n) + d_m
            st, c, d = ln_col
            acc = st[..., 1].unsqueeze(-2) * (acc - c.reshape(-1, 1) * st[..., 0].unsqueeze(-2)) + d.reshape(-1, 1)
        if geglu:
            n2 = acc.shape[-1]
            acc = acc + bias
            half = geglu_period // 2
            blocks = acc.reshape(*acc.shape[:-1], n2 // geglu_period, 2, half)
            v = blocks[..., 0, :] * F.gelu(blocks[..., 1, :])
            out.copy_(v.reshape(*acc.shape[:-1], n2 // 2))
            return out
        if bias is not None:
            acc = acc + bias
        if vt_out is not None:                         # fused q | k | v: trailing columns stored transposed (include/idf.h)
            assert rowbias is None and res is None and act is None and out_stats is None and acc.dim() == 2
            n_out = out.shape[-1]
            out.copy_(acc[:, :n_out])
            vt_out[:, :acc.shape[0]] = acc[:, n_out:].t().to(vt_out.dtype)
            return out
        if rowbias is not None:
            M = acc.shape[-2]
            idx = torch.arange(M) // rows_per_batch
            acc = acc + rowbias.float()[idx]
        if act == "silu":
            acc = F.silu(acc)
        elif act == "gelu":
            acc = F.gelu(acc)
        if res is not None:
            r = res.float()
            acc = r + (gate.float() * acc if gate is not None else acc)
        out.copy_(acc)
        if out_stats is not None:                      # (mu, rstd) of the 16-bit-rounded output rows
            o = out.float().reshape(-1, out.shape[-1])
            mu = o.mean(-1)
            out_stats.reshape(-1, 2)[:, 0] = mu
            out_stats.reshape(-1, 2)[:, 1] = torch.rsqrt(o.var(-1, unbiased=False) + out_stats_eps)
        return out

    # ---- fused GEGLU feed-forward (include/idf.h idf_mlp_geglu): the double un-packs the operand images and runs the two
    # products the kernel fuses, rounding the intermediate to the storage type as the kernel does
    @staticmethod
    def mlp_supported(M, C):
        from instancediffusion_amd.ops import HipOps
        return HipOps.mlp_supported(M, C)

    @staticmethod
    def mlp_pack(w1p16, c1, d1, w2_16):
        from instancediffusion_amd.ops import HipOps
        return HipOps.mlp_pack(w1p16, c1, d1, w2_16)

    def mlp_geglu(self, x, stats, w1, cd, w2p, b2, out, *, gate=None):
        from instancediffusion_amd.ops import HipOps
        self._count("mlp_geglu")
        M, C = x.shape
        assert self.mlp_supported(M, C) and cd.shape == (8 * C // 64, 128)
        c1, d1 = cd[:, :64].reshape(-1), cd[:, 64:].reshape(-1)
        perm = torch.tensor(HipOps.MLP_W2_PERM)                  # an involution: applying it again restores W2
        w2 = w2p.view(C, 4 * C // 16, 16).index_select(2, perm).reshape(C, 4 * C)
        mid = self.gemm(x, w1, torch.empty((M, 4 * C), dtype=self.dtype), bias=d1, geglu=True, geglu_period=32, ln_row=(stats, c1))
        return self.gemm(mid, w2, out, bias=b2, res=x, gate=gate)

    @staticmethod
    def gn_partial_shape(B, HW, C):
        return (B, HW // 64, 32, 2) if (HW % 64 == 0 and C % 32 == 0) else None

    def conv3x3(self, x, w, out, *, bias=None, rowbias=None, res=None, stride=1, upsample=0, n_valid=0, gn_partial=None):
        self._count("conv3x3")
        B, H, W_, Cin = x.shape
        assert Cin % 64 == 0
        Cout = w.shape[0]
        wt = w.float().reshape(Cout, 3, 3, Cin).permute(0, 3, 1, 2)
        xi = x.float().permute(0, 3, 1, 2)
        if upsample:
            xi = F.interpolate(xi, scale_factor=2, mode="nearest")
        if self.batch_invariant:
            y = torch.cat([F.conv2d(xi[i:i + 1], wt, None, stride=stride, padding=1) for i in range(B)], 0)
        else:
            y = F.conv2d(xi, wt, None, stride=stride, padding=1)
        if bias is not None:
            y = y + bias.view(1, -1, 1, 1)
        if rowbias is not None:
            y = y + rowbias.float()[:, :, None, None]
        if res is not None:
            y = y + res.float().permute(0, 3, 1, 2)
        if n_valid:
            out.copy_(y[:, :n_valid])
        else:
            out.copy_(y.permute(0, 2, 3, 1))
        if gn_partial is not None:
            # (mean, M2) per (sample, 64-row chunk, group) of the STORED output, as the HIP conv epilogue leaves them
            Bo, Cc = out.shape[0], out.shape[-1]
            assert tuple(gn_partial.shape) == self.gn_partial_shape(Bo, out.shape[1] * out.shape[2], Cc)
            v = out.float().reshape(Bo, -1, 64, 32, Cc // 32).permute(0, 1, 3, 2, 4).reshape(Bo, gn_partial.shape[1], 32, -1)
            mean = v.mean(-1)
            gn_partial[..., 0] = mean
            gn_partial[..., 1] = ((v - mean[..., None]) ** 2).sum(-1)
            self._count("conv3x3_gn_partial")
        return out

    def conv_in(self, x_nchw, w, bias, out):
        self._count("conv_in")
        if self.batch_invariant:
            y = torch.cat([F.conv2d(x_nchw[i:i + 1], w, bias, padding=1) for i in range(x_nchw.shape[0])], 0)
        else:
            y = F.conv2d(x_nchw, w, bias, padding=1)
        out.copy_(y.permute(0, 2, 3, 1))
        return out

    def attention(self, q, k0, vt0, n0, out, heads, *, k1=None, vt1=None, n1=0, qbits=None, kbits0=None, kbits1=None):
        self._count("attention")
        self.rows["attention"] = self.rows.get("attention", 0) + int(q.shape[0])
        if qbits is not None:
            self._count("attention_masked")
        B, Nq, C = q.shape
        d = C // heads
        assert vt0.shape[-1] >= (n0 + 63) // 64 * 64 or vt0.stride(1) >= (n0 + 63) // 64 * 64
        k = k0.float()[:, :n0]
        v = vt0.float()[:, :, :n0].transpose(1, 2)
        if n1:
            assert vt1.shape[-1] >= (n1 + 63) // 64 * 64
            k = torch.cat([k, k1.float()[:, :n1]], 1)
            v = torch.cat([v, vt1.float()[:, :, :n1].transpose(1, 2)], 1)
        M = k.shape[1]
        qh = q.float().reshape(B, Nq, heads, d).permute(0, 2, 1, 3)
        kh = k.reshape(B, M, heads, d).permute(0, 2, 1, 3)
        vh = v.reshape(B, M, heads, d).permute(0, 2, 1, 3)
        s = torch.matmul(qh, kh.transpose(-1, -2)) * (d ** -0.5)
        if qbits is not None:                      # include/idf.h: visible iff words intersect, or own token (segment 0)
            vis = (qbits[:, :, None] & kbits0[:, None, :n0]) != 0
            idx = torch.arange(Nq)
            own = torch.zeros(Nq, n0, dtype=torch.bool)
            own[idx[idx < n0], idx[idx < n0]] = True
            vis = vis | own[None]
            if n1:
                vis = torch.cat([vis, (qbits[:, :, None] & kbits1[:, None, :n1]) != 0], -1)
            s = s.masked_fill(~vis[:, None], float("-inf"))
        mx = s.max(-1, keepdim=True).values
        p = torch.exp(s - mx)
        l = p.sum(-1, keepdim=True)
        p16 = p.to(self.dtype).float()             # the kernel packs P to 16-bit before P.V
        o = torch.matmul(p16, vh) / l
        out.copy_(o.permute(0, 2, 1, 3).reshape(B, Nq, C))
        return out

    def groupnorm(self, x, out, gamma, beta, eps, silu, partial=None):
        self._count("groupnorm")
        B, C = x.shape[0], x.shape[-1]
        xi = x.float().reshape(B, -1, C).permute(0, 2, 1)
        if partial is not None:
            # merge the chunk partials (equal counts n = rows_per_chunk * C / 32) like gn_apply_kernel and normalise with them
            self._count("groupnorm_from_partial")
            HW = xi.shape[2]
            nch = partial.shape[1]
            n = (HW // nch) * (C // 32)
            pm, pq = partial[..., 0].double(), partial[..., 1].double()            # [B, nch, 32]
            mu = pm.mean(1)
            var = (pq.sum(1) + n * ((pm - mu[:, None]) ** 2).sum(1)) / (n * nch)
            mu_c = mu.float().repeat_interleave(C // 32, 1)[:, :, None]
            rs_c = torch.rsqrt(var.float() + eps).repeat_interleave(C // 32, 1)[:, :, None]
            y = (xi - mu_c) * rs_c * gamma.view(1, -1, 1) + beta.view(1, -1, 1)
        else:
            y = F.group_norm(xi, 32, gamma, beta, eps)
        if silu:
            y = F.silu(y)
        out.copy_(y.permute(0, 2, 1).reshape(x.shape))
        return out

    def layernorm(self, x, out, gamma, beta, eps=1e-5):
        self._count("layernorm")
        out.copy_(F.layer_norm(x.float(), (x.shape[-1],), gamma, beta, eps))
        return out

    def row_stats(self, x, stats, eps=1e-5):
        self._count("row_stats")
        xf = x.float()
        stats.reshape(-1, 2)[:, 0] = xf.mean(-1)
        stats.reshape(-1, 2)[:, 1] = torch.rsqrt(xf.var(-1, unbiased=False) + eps)
        return stats

    def layernorm_patch2(self, x, out, gamma, beta, eps):
        self._count("layernorm_patch2")
        B, H, W_, C = x.shape
        y = F.layer_norm(x.float(), (C,), gamma, beta, eps)
        y = y.reshape(B, H // 2, 2, W_ // 2, 2, C).permute(0, 1, 3, 2, 4, 5).reshape(B * (H // 2) * (W_ // 2), 4 * C)
        out[:, :4 * C] = y.to(out.dtype)
        return out

    def seg_in_conv(self, segs, w, bias, out):
        self._count("seg_in_conv")
        B, Cin, S, _ = segs.shape
        y = F.conv2d(segs, w, bias, padding=1)                       # [B,3,S,S]
        P = S // 4
        y = y.reshape(B, 3, P, 4, P, 4).permute(0, 2, 4, 1, 3, 5).reshape(B * P * P, 48)
        out[:, :48] = y.to(out.dtype)
        return out

    def dwconv7x7(self, x, w_tap_major, bias, out):
        self._count("dwconv7x7")
        C = x.shape[-1]
        w = w_tap_major.t().reshape(C, 1, 7, 7)
        y = F.conv2d(x.float().permute(0, 3, 1, 2), w, bias, padding=3, groups=C)
        out.copy_(y.permute(0, 2, 3, 1))
        return out

    def scaleu_concat(self, h, skip, out, hscale, sm1):
        self._count("scaleu_concat")
        B, H, W_, Ch = h.shape
        out[..., :Ch] = (h.float() * hscale).to(self.dtype)
        x = skip.float().permute(0, 3, 1, 2).double()
        hh = torch.arange(H, dtype=torch.float64)[:, None]
        ww = torch.arange(W_, dtype=torch.float64)[None, :]
        low = x.sum((-2, -1), keepdim=True).expand_as(x).clone()
        for ph in (2 * math.pi * hh / H + 0 * ww, 2 * math.pi * ww / W_ + 0 * hh, 2 * math.pi * (hh / H + ww / W_)):
            c, s = torch.cos(ph), torch.sin(ph)
            low = low + (x * c).sum((-2, -1), keepdim=True) * c + (x * s).sum((-2, -1), keepdim=True) * s
        y = x + sm1.double() * low / (H * W_)
        out[..., Ch:] = y.permute(0, 2, 3, 1).to(self.dtype)
        return out

    def timestep_embedding(self, t_f32, out):
        self._count("timestep_embedding")
        dim = out.shape[1]
        half = dim // 2
        freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
        a = t_f32[:, None] * freqs[None]
        out.copy_(torch.cat([torch.cos(a), torch.sin(a)], -1))
        return out

    def unifusion_embed(self, text, loc, tmask, lmask, null_text, null_loc, freqs, out):
        self._count("unifusion_embed")
        parts = []
        for f in freqs:
            parts.append(torch.sin(f * loc))
            parts.append(torch.cos(f * loc))
        fe = torch.cat(parts, -1)
        tm, lm = tmask[:, None], lmask[:, None]
        out.copy_(torch.cat([text * tm + (1 - tm) * null_text, fe * lm + (1 - lm) * null_loc], -1))
        return out

    def cfg_combine(self, e_cond, e_uncond, guidance, out):
        out.copy_(e_uncond + guidance * (e_cond - e_uncond))
        return out

    def plms_update(self, x, e_t, old, e_next, mode, a_t, a_prev, sqrt_1m_at, out):
        if mode == 0:
            ep = e_t
        elif mode == 1:
            ep = (e_t + e_next) / 2
        elif mode == 2:
            ep = (3 * e_t - old[-1]) / 2
        elif mode == 3:
            ep = (23 * e_t - 16 * old[-1] + 5 * old[-2]) / 12
        else:
            ep = (55 * e_t - 59 * old[-1] + 37 * old[-2] - 9 * old[-3]) / 24
        a_t = torch.tensor(a_t, dtype=torch.float32)
        a_prev = torch.tensor(a_prev, dtype=torch.float32)
        pred = (x - sqrt_1m_at * ep) / a_t.sqrt()
        out.copy_(a_prev.sqrt() * pred + (1.0 - a_prev).sqrt() * ep)
        return out

    def mis_merge(self, lat, boxes_i32, out, mode):
        if mode == 0:
            out.copy_(lat.mean(0))
        else:
            v = lat[0].clone()
            for k in range(lat.shape[0] - 1):
                b = boxes_i32[k].tolist()
                v[:, :, b[0]:b[2], b[1]:b[3]] = lat[k + 1][:, :, b[0]:b[2], b[1]:b[3]]
            out.copy_(v)
        return out

    def cast16(self, x_f32, out):
        out.copy_(x_f32)
        return out

    def softmax_rows(self, s_f32, out, scale):
        self._count("softmax_rows")
        assert s_f32.dtype == torch.float32
        out.copy_(torch.softmax(s_f32 * scale, -1))
        return out

    def pointwise_nchw(self, x, w, bias, out, in_scale=1.0):
        self._count("pointwise_nchw")
        y = torch.einsum("oc,bc...->bo...", w, x * in_scale)
        if bias is not None:
            y = y + bias.view(1, -1, *([1] * (x.dim() - 2)))
        out.copy_(y)
        return out
