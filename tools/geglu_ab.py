"""A/B of the GEGLU projection at the C = 640 level: geglu640w_kernel (IDF_TUNE_GEGLU_ROW = 1) against the persistent GEMM kernel (0),
same operands, statistics handed in.  rel-RMS against fp32 LN -> Linear -> value * gelu(gate) on a row sample, the difference
between the two kernels on the whole output, HIP-event times.   python tools/geglu_ab.py [M=131072] [dtype=bf16] [reps=20]"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from instancediffusion_amd import _lib  # noqa: E402
from instancediffusion_amd.engine import pack_geglu  # noqa: E402
from instancediffusion_amd.ops import HipOps  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
dtype = {"bf16": torch.bfloat16, "fp16": torch.float16}[sys.argv[2] if len(sys.argv) > 2 else "bf16"]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
C, N = 640, 5120
ops = HipOps(dtype)
lib = _lib.load()
g = torch.Generator().manual_seed(7)
gamma, beta = 1 + 0.2 * torch.randn(C, generator=g), 0.3 * torch.randn(C, generator=g)
rows = 4096
x = (torch.randn(rows, C, generator=g) * 1.5 + 0.8 * torch.randn(rows, 1, generator=g)).to(dtype)
x = x.repeat(M // rows + 1, 1)[:M].contiguous().cuda()
w = torch.randn(N, C, generator=g) * C ** -0.5
b = 0.2 * torch.randn(N, generator=g)
wp, dp = pack_geglu(w * gamma[None, :], b + w @ beta, 32)
w16 = wp.to(dtype).cuda()
c = w16.float().sum(1).contiguous()
d = dp.cuda()
st = ops.empty((M, 2), torch.float32)
ops.row_stats(x, st, 1e-5)
h = F.layer_norm(x[:rows].float(), (C,), gamma.cuda(), beta.cuda(), 1e-5) @ w.cuda().t() + b.cuda()
want = h[:, :N // 2] * F.gelu(h[:, N // 2:])
outs = {}
for mode in (0, 1):
    lib.idf_set_tuning(_lib.IDF_TUNE_GEGLU_ROW, mode)
    out = ops.empty((M, N // 2)); out.fill_(7.0)
    n0 = lib.idf_get_stat(_lib.IDF_STAT_GEGLU_ROW_LAUNCHES)
    ops.gemm(x, w16, out, bias=d, geglu=True, geglu_period=32, ln_row=(st, c))
    torch.cuda.synchronize()
    served = lib.idf_get_stat(_lib.IDF_STAT_GEGLU_ROW_LAUNCHES) - n0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(3):
        e0.record()
        for _ in range(reps):
            ops.gemm(x, w16, out, bias=d, geglu=True, geglu_period=32, ln_row=(st, c))
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / reps * 1e3)
    err = float(((out[:rows].float() - want) ** 2).sum().sqrt() / (want ** 2).sum().sqrt())
    same = bool(torch.equal(out[:rows], out[rows:2 * rows])) if M >= 2 * rows else None
    fin = bool(torch.isfinite(out.float()).all())
    print(f"mode {mode}: served by geglu640w {served}; rel-rms {err:.3e}; copies of a row bitwise equal {same}; finite {fin}; "
          f"{min(ts):.1f} us ({2.0 * M * N * C / min(ts) * 1e-6:.1f} TF)  runs {['%.1f' % t for t in ts]}")
    outs[mode] = out
dd = (outs[0].float() - outs[1].float())
print(f"kernel 1 vs kernel 0 on the whole output: max |diff| {float(dd.abs().max()):.3e}; differing elements {float((outs[0] != outs[1]).float().mean()):.2e}; "
      f"rel-rms {float(dd.pow(2).sum().sqrt() / outs[0].float().pow(2).sum().sqrt()):.3e}")
