#!/usr/bin/env python
"""A/B of the two V^T-projection forms at the headline forward batch (64 rows): B batched GEMMs  V^T[b] = Wv . X_b^T
(128x128-tile kernel) vs ONE unbatched GEMM over all B*N tokens writing the batch-interleaved image [C][B][N]
(persistent big-tile kernel).  Prints per-shape microseconds and checks both forms give the same numbers.
Usage: python tools/vt_gemm_ab.py [B=64]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from instancediffusion_amd.ops import HipOps  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ops = HipOps(torch.bfloat16)


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


for C, N in [(320, 4096), (640, 1024), (1280, 256), (1280, 64)]:
    wv = (torch.randn(C, C, device="cuda") * C ** -0.5).to(torch.bfloat16)
    x = torch.randn(B * N, C, device="cuda").to(torch.bfloat16)
    vt_b = ops.zeros((B, C, N))
    vt_g = ops.empty((C, B * N))
    t_b = timeit(lambda: ops.gemm(wv, x.view(B, N, C), vt_b))
    t_g = timeit(lambda: ops.gemm(wv, x, vt_g))
    same = torch.equal(vt_g.view(C, B, N).permute(1, 0, 2), vt_b)
    diff = float((vt_g.view(C, B, N).permute(1, 0, 2).float() - vt_b.float()).abs().max())
    gf = 2.0 * B * N * C * C / 1e9          # GFLOP; GFLOP / us = PFLOP/s
    print(f"C={C:5d} N={N:5d} B={B}: batched {t_b:8.1f} us ({gf / t_b * 1e3:6.1f} TF)   global {t_g:8.1f} us "
          f"({gf / t_g * 1e3:6.1f} TF)   bitwise equal={same} max|diff|={diff:.2e}")
