"""rocprofv3 target: the 64x64-latent gated self-attention launch as the engine issues it (B rows, 8 heads, d = 40,
4096 visual + 184 grounding keys; q / k = column slices of the fused projection buffer, V^T in the batch-interleaved
[C][B][N] image) a few times.  Usage: python tools/attn_only.py [batch=64] [iters=3] [mode=-1 (default kernel)] [n1=184]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from instancediffusion_amd import _lib  # noqa: E402
from instancediffusion_amd.ops import HipOps  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
mode = int(sys.argv[3]) if len(sys.argv) > 3 else -1
n1 = int(sys.argv[4]) if len(sys.argv) > 4 else 184
ops = HipOps(torch.bfloat16)
if mode >= 0:
    _lib.load().idf_set_tuning(1, mode)
N, C = 4096, 320
g = torch.Generator(device="cuda").manual_seed(0)
qk = (torch.randn(B, N, 2 * C, device="cuda", generator=g) * 0.5).bfloat16()
vt = (torch.randn(C, B, N, device="cuda", generator=g) * 0.5).bfloat16().permute(1, 0, 2)
kw = {}
if n1:
    kw = dict(k1=(torch.randn(B, n1, C, device="cuda", generator=g) * 0.5).bfloat16(),
              vt1=(torch.randn(B, C, 192, device="cuda", generator=g) * 0.5).bfloat16(), n1=n1)
o = ops.empty((B, N, C))
for _ in range(iters):
    ops.attention(qk[:, :, :C], qk[:, :, C:], vt, N, o, 8, **kw)
torch.cuda.synchronize()
print("ok", float(o.float().abs().mean()))
