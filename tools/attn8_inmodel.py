"""In-model A/B of a library tuning knob: graph-replayed full-model forwards at a given row batch for each value of the knob
(the graphs are re-captured per value).  Usage: python tools/attn8_inmodel.py [batch=128] [knob=4] [values=0,1,2,5,6] [iters=10]
Prints ms per forward per value, interleaved over `rounds` passes (boxes drift; the first pass of a process runs slow)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from instancediffusion_amd import _lib  # noqa: E402
from instancediffusion_amd.host.config import SD15_BOX_CFG  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 128
knob = int(sys.argv[2]) if len(sys.argv) > 2 else _lib.IDF_TUNE_ATTN8
values = [int(v) for v in (sys.argv[3] if len(sys.argv) > 3 else "0,1,2,5,6").split(",")]
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 10
cfg = dict(SD15_BOX_CFG)
model, sd = bench.build_model(cfg)
dev = torch.device("cuda", 0)
inputs, uc, gi, _ = bench.make_inputs(cfg, batch, dev)
model.grounding_tokenizer_input = gi
eng = model.engine
eng.use_graphs = True
cond = eng.prepare_cond(inputs[0]["context"], inputs[0]["grounding_input"])
x = torch.randn(batch, 4, 64, 64, device=dev)
t = torch.full((batch,), 500.0, device=dev)
paired = batch % 2 == 0
if paired:
    x[batch // 2:] = x[:batch // 2]
lib = _lib.load()
res = {v: [] for v in values}
for rnd in range(3):
    for v in values:
        prev = lib.idf_set_tuning(knob, v)
        eng._graphs.clear()
        for _ in range(3):
            eps = eng.forward_cond(x, t, cond, paired=paired)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            eps = eng.forward_cond(x, t, cond, paired=paired)
        torch.cuda.synchronize()
        res[v].append((time.perf_counter() - t0) / iters * 1e3)
        lib.idf_set_tuning(knob, prev)
for v in values:
    print(f"knob {knob} = {v}: " + "  ".join(f"{m:.2f}" for m in res[v]) + f"  ms per {batch}-row forward (min {min(res[v]):.2f})  |eps| {float(eps.abs().mean()):.5f}")
