"""rocprofv3 target: a few eager UNet forwards of the FULL model at the MIS phase-1 batch (18 = 9 trajectories x
cond/uncond) through the C ABI.  Used for the per-kernel stats and the FETCH_SIZE / WRITE_SIZE PMC passes
(separate runs; see profiles/README.md).  Usage: python tools/profile_forward.py [batch] [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from instancediffusion_amd.host.config import SD15_BOX_CFG  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 18
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
cfg = dict(SD15_BOX_CFG)
model, sd = bench.build_model(cfg)
dev = torch.device("cuda", 0)
inputs, uc, gi, _ = bench.make_inputs(cfg, batch, dev)
model.grounding_tokenizer_input = gi
eng = model.engine
eng.use_graphs = False
cond = eng.prepare_cond(inputs[0]["context"], inputs[0]["grounding_input"])
x = torch.randn(batch, 4, 64, 64, device=dev)
t = torch.full((batch,), 500.0, device=dev)
for _ in range(iters):
    eps = eng.forward_cond(x, t, cond)
torch.cuda.synchronize()
print("ok", float(eps.abs().mean()))
