"""rocprofv3 target: a few eager UNet forwards of the FULL model at the MIS phase-1 batch (18 = 9 trajectories x
cond/uncond) through the C ABI.  Used for the per-kernel stats and the FETCH_SIZE / WRITE_SIZE PMC passes
(separate runs; see profiles/README.md).  Usage: python tools/profile_forward.py [batch] [iters] [graph]
With a third argument the forwards are hipGraph replays (as the samplers run them) and the wall time per replay is printed:
together with the kernel-trace sum of the same run that separates kernel time from launch / dependency gaps."""
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
if os.environ.get("PROFILE_DTYPE") == "fp16":          # (round 6: per-kernel fp16-vs-bf16 comparison, tools/dtype_kernel_diff.py)
    model.compute_dtype = torch.float16
dev = torch.device("cuda", 0)
inputs, uc, gi, _ = bench.make_inputs(cfg, batch, dev)
model.grounding_tokenizer_input = gi
eng = model.engine
graph = len(sys.argv) > 3
eng.use_graphs = graph
cond = eng.prepare_cond(inputs[0]["context"], inputs[0]["grounding_input"])
x = torch.randn(batch, 4, 64, 64, device=dev)
t = torch.full((batch,), 500.0, device=dev)
# the sampler's forwards are guidance pairs: rows [batch/2, batch) repeat the latent of rows [0, batch/2) and the engine computes
# their conditioning-free prefix once (engine.PAIR_HOIST); PROFILE_UNPAIRED=1 profiles a forward of `batch` distinct rows
paired = batch % 2 == 0 and os.environ.get("PROFILE_UNPAIRED") != "1"
if paired:
    x[batch // 2:] = x[:batch // 2]
if graph:
    import time
    for _ in range(3):
        eps = eng.forward_cond(x, t, cond, paired=paired)    # eager warm-up, capture, first replays
    torch.cuda.synchronize()
    t0 = time.perf_counter()
for _ in range(iters):
    eps = eng.forward_cond(x, t, cond, paired=paired)
torch.cuda.synchronize()
if graph:
    print(f"graph replay: {(time.perf_counter() - t0) / iters * 1e3:.3f} ms per {batch}-row forward ({iters} replays)")
print("ok", float(eps.abs().mean()))
