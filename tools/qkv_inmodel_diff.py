"""Full-model forward at a given row batch with the q | k | v row kernel off / on (same plumbing): relative difference of eps.
python tools/qkv_inmodel_diff.py [batch=16]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from instancediffusion_amd import _lib  # noqa: E402
from instancediffusion_amd.host.config import SD15_BOX_CFG  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 16
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
lib = _lib.load()
out = {}
for v in (0, 1, 0):
    lib.idf_set_tuning(_lib.IDF_TUNE_QKV_ROW, v)
    n0 = lib.idf_get_stat(6)
    e = eng.forward_cond(x, t, cond, paired=False).float().clone()
    torch.cuda.synchronize()
    print(f"knob {v}: served by qkv320w {lib.idf_get_stat(6) - n0}  |eps| {float(e.abs().mean()):.6f}")
    if v in out:
        print("  repeat of the same knob: max |diff|", float((e - out[v]).abs().max()))
    out[v] = e
d = out[1] - out[0]
print(f"rel-rms(knob 1 - knob 0) = {float(d.pow(2).sum().sqrt() / out[0].pow(2).sum().sqrt()):.3e}  max |diff| {float(d.abs().max()):.3e}")
