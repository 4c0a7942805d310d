"""A/B of the fused q | k | v projection at the C = 320 level: qkv320w_kernel (IDF_TUNE_QKV_ROW = 1) against the persistent GEMM
kernel (0), same operands, statistics handed in.  Prints rel-RMS against fp32 LayerNorm -> Linear on a row sample, the difference
between the two kernels on the whole output, and HIP-event times.   python tools/qkv_ab.py [M=524288] [dtype=bf16] [reps=20]"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from instancediffusion_amd import _lib  # noqa: E402
from instancediffusion_amd.ops import HipOps  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 524288
dtype = {"bf16": torch.bfloat16, "fp16": torch.float16}[sys.argv[2] if len(sys.argv) > 2 else "bf16"]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
C = int(os.environ.get("QKV_C", "320"))
ops = HipOps(dtype)
lib = _lib.load()
g = torch.Generator().manual_seed(5)
gamma, beta = 1 + 0.2 * torch.randn(C, generator=g), 0.3 * torch.randn(C, generator=g)
rows = 8192
x = (torch.randn(rows, C, generator=g) * 1.5 + 0.8 * torch.randn(rows, 1, generator=g)).to(dtype)
x = x.repeat(M // rows + 1, 1)[:M].contiguous().cuda()
w = torch.randn(3 * C, C, generator=g) * C ** -0.5
w16 = (w * gamma[None, :]).to(dtype)
c = w16.float().sum(1).cuda()
d = (w @ beta).cuda()
w16 = w16.cuda()
st = ops.empty((M, 2), torch.float32)
ops.row_stats(x, st, 1e-5)
want = (F.layer_norm(x[:rows].float(), (C,), gamma.cuda(), beta.cuda(), 1e-5) @ w.cuda().t())
outs = {}
for mode in (0, 1):
    lib.idf_set_tuning(_lib.IDF_TUNE_QKV_ROW, mode)
    qk, vt = ops.empty((M, 2 * C)), ops.empty((C, M))
    qk.fill_(7.0); vt.fill_(7.0)
    n0 = lib.idf_get_stat(6)
    ops.gemm(x, w16, qk, bias=d, ln_row=(st, c), vt_out=vt)
    torch.cuda.synchronize()
    served = lib.idf_get_stat(6) - n0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(3):
        e0.record()
        for _ in range(reps):
            ops.gemm(x, w16, qk, bias=d, ln_row=(st, c), vt_out=vt)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / reps * 1e3)
    def rr(a, b):
        return float(((a.float() - b.float()) ** 2).sum().sqrt() / (b.float() ** 2).sum().sqrt())
    eq, ev = rr(qk[:rows], want[:, :2 * C]), rr(vt[:, :rows].t(), want[:, 2 * C:])
    same_rows = bool(torch.equal(qk[:rows], qk[rows:2 * rows]) and torch.equal(vt[:, :rows], vt[:, rows:2 * rows])) if M >= 2 * rows else None
    fin = bool(torch.isfinite(qk.float()).all() and torch.isfinite(vt.float()).all())
    print(f"mode {mode}: served by qkv320w {served}; q|k rel-rms {eq:.3e}, V^T rel-rms {ev:.3e}; copies of a row bitwise equal {same_rows}; finite {fin}; "
          f"{min(ts):.1f} us ({2.0 * M * 3 * C * C / min(ts) * 1e-6:.1f} TF)  runs {['%.1f' % t for t in ts]}")
    outs[mode] = (qk, vt)
dq = float((outs[0][0].float() - outs[1][0].float()).abs().max()); dv = float((outs[0][1].float() - outs[1][1].float()).abs().max())
nq = float((outs[0][0] != outs[1][0]).float().mean()); nv = float((outs[0][1] != outs[1][1]).float().mean())
print(f"kernel 1 vs kernel 0 on the whole output: max |diff| q|k {dq:.3e} V^T {dv:.3e}; differing elements {nq:.2e} / {nv:.2e}")
