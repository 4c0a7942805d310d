"""rocprofv3 target: the 64x64-latent self-attention launch (B rows, 8 heads, d = 40, 4096 keys) a few times.
Usage: [IDF_ATTN2=0|1] python tools/attn_only.py [batch=16] [iters=3]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from instancediffusion_amd.ops import HipOps  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ops = HipOps(torch.bfloat16)
N, C = 4096, 320
g = torch.Generator(device="cuda").manual_seed(0)
q = (torch.randn(B, N, C, device="cuda", generator=g) * 0.5).bfloat16()
k = (torch.randn(B, N, C, device="cuda", generator=g) * 0.5).bfloat16()
vt = (torch.randn(B, C, N, device="cuda", generator=g) * 0.5).bfloat16()
o = ops.empty((B, N, C))
for _ in range(iters):
    ops.attention(q, k, vt, N, o, 8)
torch.cuda.synchronize()
print("ok", float(o.float().abs().mean()))
