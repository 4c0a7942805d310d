"""VAE decoder executor for MI355X: ``AutoencoderKL.decode`` over the same C-ABI kernels.

Replaces ``ldm/models/autoencoder.py:32-36`` + ``Decoder.forward`` (``ldm/modules/diffusionmodules/model.py:534-568``).
Design, following the UNet engine:
  * activations NHWC 16-bit for the whole decoder; the latent enters as fp32 NCHW (``idf_pointwise_nchw`` applies
    ``1/scale_factor`` and ``post_quant_conv``, ``idf_conv_in`` reads it), the image leaves as fp32 NCHW straight from
    the last conv's epilogue (``IDF_EPI_OUT_NCHW``) -- no layout permutes anywhere;
  * every 3x3 conv / 1x1 shortcut / projection is the MFMA GEMM / implicit-GEMM conv kernel, ``nearest x2`` is folded
    into the following conv's gather, GroupNorm(eps 1e-6)+SiLU is the streaming two-launch kernel;
  * the mid-block attention is single-head with head dim = 512 (model.py:178-196), beyond the register budget of the
    flash kernels: scores = ONE batched MFMA GEMM with fp32 output, ``idf_softmax_rows`` normalises them into 16-bit
    probabilities, a second batched GEMM applies V^T.  The bias of the ``v`` 1x1 conv is added in that GEMM's epilogue
    (softmax rows sum to 1, so P.(V + 1 b^T) = P.V + 1 b^T exactly);
  * one decode = a fixed launch sequence over static buffers, captured into a hipGraph per (batch, H, W).

Work (SD-1.5 KL-f8 decoder, 64x64 latent -> 512x512 image): 2514.5 GFLOP per image.
"""
from __future__ import annotations

from typing import Dict

import torch

from .engine import _Lin, pack_conv3x3


class VAEDecoderEngine:
    def __init__(self, vae, ops=None, dtype: torch.dtype = torch.bfloat16, use_graphs: bool = True):
        if ops is None:
            from .ops import HipOps          # raises when libidf_gfx950.so / the GPU is missing: no fallback
            ops = HipOps(dtype)
        self.ops = ops
        self.dtype = ops.dtype
        self.device = ops.device
        self.use_graphs = use_graphs and self.device.type == "cuda"
        self._bufs: Dict[tuple, torch.Tensor] = {}
        self._graphs: Dict[tuple, object] = {}
        self._pack(vae)

    # ---- weight packing ---------------------------------------------------------------------------------------
    def _w16(self, t):
        return t.detach().to(device=self.device, dtype=torch.float32).to(self.dtype).contiguous()

    def _f32(self, t):
        return t.detach().to(device=self.device, dtype=torch.float32).contiguous()

    def _conv(self, m) -> _Lin:
        return _Lin(self._w16(pack_conv3x3(m.weight.detach().float())), self._f32(m.bias))

    def _lin(self, m) -> _Lin:
        w = m.weight.detach()
        return _Lin(self._w16(w.reshape(w.shape[0], w.shape[1])), self._f32(m.bias))

    def _pack_res(self, rb):
        return dict(cin=rb.in_channels, cout=rb.out_channels,
                    n1=(self._f32(rb.norm1.weight), self._f32(rb.norm1.bias)), conv1=self._conv(rb.conv1),
                    n2=(self._f32(rb.norm2.weight), self._f32(rb.norm2.bias)), conv2=self._conv(rb.conv2),
                    skip=self._lin(rb.nin_shortcut) if rb.in_channels != rb.out_channels else None)

    def _pack_attn(self, ab):
        C = ab.in_channels
        wq, wk = ab.q.weight.detach().reshape(C, C), ab.k.weight.detach().reshape(C, C)
        return dict(c=C, norm=(self._f32(ab.norm.weight), self._f32(ab.norm.bias)),
                    wqk=self._w16(torch.cat([wq, wk], 0)),
                    bqk=self._f32(torch.cat([ab.q.bias.detach(), ab.k.bias.detach()], 0)),
                    wv=self._w16(ab.v.weight.detach().reshape(C, C)), bv=self._f32(ab.v.bias),
                    proj=self._lin(ab.proj_out))

    def _pack(self, vae):
        dec = vae.decoder
        assert not dec.tanh_out, "tanh_out decoders are not used by any reference config"
        self.inv_scale = 1.0 / float(vae.scale_factor)
        pq = vae.post_quant_conv
        self.pq_w = self._f32(pq.weight.detach().reshape(pq.weight.shape[0], pq.weight.shape[1]))
        self.pq_b = self._f32(pq.bias)
        self.conv_in = (self._f32(dec.conv_in.weight), self._f32(dec.conv_in.bias))
        self.mid = [("res", self._pack_res(dec.mid.block_1)), ("attn", self._pack_attn(dec.mid.attn_1)),
                    ("res", self._pack_res(dec.mid.block_2))]
        self.levels = []
        for i_level in reversed(range(dec.num_resolutions)):
            up = dec.up[i_level]
            layers = []
            for i_block in range(dec.num_res_blocks + 1):
                layers.append(("res", self._pack_res(up.block[i_block])))
                if len(up.attn) > 0:
                    layers.append(("attn", self._pack_attn(up.attn[i_block])))
            if i_level != 0:
                layers.append(("up", self._conv(up.upsample.conv)))
            self.levels.append(layers)
        self.norm_out = (self._f32(dec.norm_out.weight), self._f32(dec.norm_out.bias))
        oc = dec.conv_out
        wpad = torch.zeros(64, oc.weight.shape[1], 3, 3)
        wpad[: oc.weight.shape[0]] = oc.weight.detach().float().cpu()
        bpad = torch.zeros(64)
        bpad[: oc.bias.shape[0]] = oc.bias.detach().float().cpu()
        self.conv_out = _Lin(self._w16(pack_conv3x3(wpad)), self._f32(bpad))
        self.n_out = oc.weight.shape[0]
        self.up_factor = 2 ** (dec.num_resolutions - 1)

    # ---- buffers ----------------------------------------------------------------------------------------------
    def buf(self, role: str, shape, dtype=None) -> torch.Tensor:
        dtype = dtype or self.dtype
        key = (role, tuple(int(s) for s in shape), dtype)
        t = self._bufs.get(key)
        if t is None:
            t = self.ops.empty(key[1], dtype)
            self._bufs[key] = t
        return t

    # ---- blocks -----------------------------------------------------------------------------------------------
    def _res(self, p, x, out_role):
        """ResnetBlock.forward (model.py:121-143), temb None."""
        ops = self.ops
        B, H, W, Cin = x.shape
        Cout = p["cout"]
        g = ops.groupnorm(x, self.buf("gn", x.shape), p["n1"][0], p["n1"][1], 1e-6, True)
        h1 = ops.conv3x3(g, p["conv1"].w, self.buf("h1", (B, H, W, Cout)), bias=p["conv1"].b)
        g2 = ops.groupnorm(h1, self.buf("gn", h1.shape), p["n2"][0], p["n2"][1], 1e-6, True)
        if p["skip"] is not None:
            xs = ops.gemm(x.view(B * H * W, Cin), p["skip"].w, self.buf("skip", (B * H * W, Cout)),
                          bias=p["skip"].b).view(B, H, W, Cout)
        else:
            xs = x
        return ops.conv3x3(g2, p["conv2"].w, self.buf(out_role, (B, H, W, Cout)), bias=p["conv2"].b, res=xs)

    def _attn(self, p, x, out_role):
        """AttnBlock.forward (model.py:178-202): softmax(q k^T / sqrt(C)) v over the H*W positions, one head."""
        ops = self.ops
        B, H, W, C = x.shape
        N, M = H * W, B * H * W
        assert N % 64 == 0, "the V^T GEMM needs H*W to be a multiple of 64"
        g = ops.groupnorm(x, self.buf("gn", x.shape), p["norm"][0], p["norm"][1], 1e-6, False)
        qk = ops.gemm(g.view(M, C), p["wqk"], self.buf("at.qk", (M, 2 * C)), bias=p["bqk"]).view(B, N, 2 * C)
        vt = ops.gemm(p["wv"], g.view(B, N, C), self.buf("at.vt", (B, C, N)))             # V^T[b] = Wv . X_b^T
        s = ops.gemm(qk[:, :, :C], qk[:, :, C:], self.buf("at.s", (B, N, N), torch.float32))
        pr = ops.softmax_rows(s, self.buf("at.p", (B, N, N)), float(C) ** -0.5)
        o = ops.gemm(pr, vt, self.buf("at.o", (B, N, C)), bias=p["bv"])                    # + b_v: rows of P sum to 1
        out = ops.gemm(o.view(M, C), p["proj"].w, self.buf(out_role, (M, C)), bias=p["proj"].b, res=x.view(M, C))
        return out.view(B, H, W, C)

    def _decode_ops(self, z: torch.Tensor, img: torch.Tensor):
        """Enqueue one decode.  z [B, zc, H, W] fp32, img [B, 3, f*H, f*W] fp32."""
        ops = self.ops
        B, zc, H, W = z.shape
        z2 = ops.pointwise_nchw(z, self.pq_w, self.pq_b, self.buf("z.pq", (B, self.pq_w.shape[0], H, W), torch.float32),
                                self.inv_scale)
        h = ops.conv_in(z2, self.conv_in[0], self.conv_in[1], self.buf("a", (B, H, W, self.conv_in[0].shape[0])))
        cur = "a"

        def other():
            return "b" if cur == "a" else "a"

        for layers in [self.mid] + self.levels:
            for kind, p in layers:
                if kind == "res":
                    h = self._res(p, h, other())
                elif kind == "attn":
                    h = self._attn(p, h, other())
                else:
                    Bq, Hq, Wq, C = h.shape
                    h = ops.conv3x3(h, p.w, self.buf(other(), (Bq, 2 * Hq, 2 * Wq, C)), bias=p.b, upsample=1)
                cur = other()
        g = ops.groupnorm(h, self.buf("gn", h.shape), self.norm_out[0], self.norm_out[1], 1e-6, True)
        ops.conv3x3(g, self.conv_out.w, img, bias=self.conv_out.b, n_valid=self.n_out)
        return img

    def decode(self, z: torch.Tensor) -> torch.Tensor:
        B, zc, H, W = z.shape
        f = self.up_factor
        z_s = self.buf("io.z", z.shape, torch.float32)
        img_s = self.buf("io.img", (B, self.n_out, f * H, f * W), torch.float32)
        z_s.copy_(z)
        if not self.use_graphs:
            self._decode_ops(z_s, img_s)
        else:
            key = (B, H, W)
            graph = self._graphs.get(key)
            if graph is None:
                self._decode_ops(z_s, img_s)               # eager warm-up sizes every buffer
                torch.cuda.synchronize()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    self._decode_ops(z_s, img_s)
                self._graphs[key] = graph
            graph.replay()
        return img_s.clone()

