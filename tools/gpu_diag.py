"""Per-kernel timing on the headline shapes (SD-1.5 UNet, 64x64 latent) -- run on the GPU box via gpurun.
Writes gpurun_out/diag.json.  Not a test: numbers guide kernel optimisation (fraction of MFMA / HBM roofline)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from instancediffusion_amd.ops import HipOps  # noqa: E402

PEAK_TF, PEAK_GBS = 2500.0, 8000.0
ops = HipOps(torch.bfloat16)
res = []


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def rnd(*shape):
    return (torch.randn(*shape, device="cuda") * 0.5).to(torch.bfloat16)


def rec(name, sec, flops=None, bytes_=None):
    d = dict(name=name, us=sec * 1e6)
    if flops:
        d["tflops"] = flops / sec / 1e12
        d["frac_mfma"] = d["tflops"] / PEAK_TF
    if bytes_:
        d["gbs"] = bytes_ / sec / 1e9
        d["frac_hbm"] = d["gbs"] / PEAK_GBS
    res.append(d)
    print(json.dumps(d), flush=True)


B = int(os.environ.get("DIAG_B", "2"))
SECTIONS = os.environ.get("DIAG_SECTIONS", "gemm,conv,attn,norm").split(",")
for (M, N, K, tag) in [] if "gemm" not in SECTIONS else [(B * 4096, 320, 320, "proj 64^2"), (B * 4096, 640, 320, "qk 64^2"), (B * 4096, 2560, 320, "geglu-in 64^2"),
                       (B * 4096, 320, 1280, "ff-out 64^2"), (B * 1024, 640, 640, "proj 32^2"), (B * 1024, 5120, 640, "geglu-in 32^2"),
                       (B * 256, 1280, 1280, "proj 16^2"), (B * 256, 10240, 1280, "geglu-in 16^2"), (8192, 8192, 8192, "square 8k")]:
    a, w, o = rnd(M, K), rnd(N, K), ops.empty((M, N))
    rec(f"gemm {tag} M{M} N{N} K{K}", timeit(lambda: ops.gemm(a, w, o)), flops=2.0 * M * N * K)

# epilogue cost on the short-K layers: same GEMM with bias+residual, and the packed GEGLU
if os.environ.get("DIAG_EPI", "1") == "1" and "gemm" in SECTIONS:
    for (M, N, K, tag) in [(B * 4096, 320, 320, "proj 64^2"), (B * 1024, 640, 640, "proj 32^2")]:
        a, w, o = rnd(M, K), rnd(N, K), ops.empty((M, N))
        bias, resid = torch.randn(N, device="cuda"), rnd(M, N)
        rec(f"gemm+bias+res {tag} M{M} N{N} K{K}", timeit(lambda: ops.gemm(a, w, o, bias=bias, res=resid)), flops=2.0 * M * N * K)
    for (M, C, tag) in [(B * 4096, 320, "64^2"), (B * 1024, 640, "32^2")]:
        a, w, o = rnd(M, C), rnd(8 * C, C), ops.empty((M, 4 * C))
        bias = torch.randn(8 * C, device="cuda")
        rec(f"gemm+geglu {tag} M{M} N{8 * C} K{C}", timeit(lambda: ops.gemm(a, w, o, bias=bias, geglu=True)), flops=2.0 * M * 8 * C * C)

for (H, Cin, Cout, tag) in [] if "conv" not in SECTIONS else [(64, 320, 320, "64^2 320"), (32, 640, 640, "32^2 640"), (16, 1280, 1280, "16^2 1280"),
                            (8, 2560, 1280, "8^2 2560->1280"), (64, 960, 320, "64^2 960->320"), (32, 1920, 640, "32^2 1920->640")]:
    x, w, o = rnd(B, H, H, Cin), rnd(Cout, 9 * Cin), ops.empty((B, H, H, Cout))
    rec(f"conv3x3 {tag} B{B}", timeit(lambda: ops.conv3x3(x, w, o)), flops=2.0 * B * H * H * Cout * 9 * Cin)

for (N, d, n1, tag) in [] if "attn" not in SECTIONS else [(4096, 40, 0, "self 64^2"), (4096, 40, 184, "gated 64^2"), (1024, 80, 184, "gated 32^2"),
                        (256, 160, 184, "gated 16^2"), (4096, 40, -77, "cross 64^2")]:
    C = 8 * d
    q = rnd(B, N, C)
    if n1 < 0:
        n0 = 77
        k0, vt0 = rnd(B, n0, C), torch.zeros(B, C, 128, device="cuda", dtype=torch.bfloat16)
        n1 = 0
    else:
        n0 = N
        k0, vt0 = rnd(B, N, C), rnd(B, C, N)
    o = ops.empty((B, N, C))
    kw = {}
    if n1:
        kw = dict(k1=rnd(B, 184, C), vt1=rnd(B, C, 192), n1=184)
    rec(f"attn {tag} B{B} d{d}", timeit(lambda: ops.attention(q, k0, vt0, n0, o, 8, **kw)), flops=4.0 * B * N * (n0 + n1) * C)

NORM = "norm" in SECTIONS
for (HW, C) in [(4096, 320), (4096, 960), (1024, 1920), (64, 2560)] if NORM else []:
    x, o = rnd(B, HW, C), ops.empty((B, HW, C))
    gm, bt = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    rec(f"groupnorm+silu HW{HW} C{C} B{B}", timeit(lambda: ops.groupnorm(x, o, gm, bt, 1e-5, True)), bytes_=4.0 * B * HW * C)
for (M, C) in [(B * 4096, 320), (B * 1024, 640), (B * 256, 1280)] if NORM else []:
    x, o = rnd(M, C), ops.empty((M, C))
    gm, bt = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    rec(f"layernorm M{M} C{C}", timeit(lambda: ops.layernorm(x, o, gm, bt)), bytes_=4.0 * M * C)
for (H, Ch, Cs) in [(64, 320, 320), (32, 640, 640), (8, 1280, 1280)] if NORM else []:
    h, s, o = rnd(B, H, H, Ch), rnd(B, H, H, Cs), ops.empty((B, H, H, Ch + Cs))
    hs, sm1 = torch.ones(Ch, device="cuda"), torch.full((1,), 0.2, device="cuda")
    rec(f"scaleu H{H} {Ch}+{Cs} B{B}", timeit(lambda: ops.scaleu_concat(h, s, o, hs, sm1)), bytes_=4.0 * B * H * H * (Ch + Cs))

os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open(f"gpurun_out/diag_B{B}.json", "w"), indent=1)
