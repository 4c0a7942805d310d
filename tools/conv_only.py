"""rocprofv3 target: one big-tile conv (64x64, 320->320) and one long-K dense GEMM (8192^3), a few launches each.
Usage: python tools/conv_only.py [batch=64] [iters=3]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from instancediffusion_amd.ops import HipOps  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ops = HipOps(torch.bfloat16)


def rnd(*shape):
    return (torch.randn(*shape, device="cuda") * 0.5).to(torch.bfloat16)


x, w, o = rnd(B, 32, 32, 640), rnd(640, 9 * 640), ops.empty((B, 32, 32, 640))
a, w2, o2 = rnd(8192, 8192), rnd(8192, 8192), ops.empty((8192, 8192))
for _ in range(iters):
    ops.conv3x3(x, w, o)
    ops.gemm(a, w2, o2)
torch.cuda.synchronize()
print("ok")
