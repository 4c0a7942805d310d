"""Per-(op, shape) time table of ONE eager full-model UNet forward at a given row batch (HIP events around every C-ABI
call).  Guides kernel work: which shapes carry the time, at which fraction of the MFMA peak.
Usage: python tools/shape_profile.py [batch=72] ; writes gpurun_out/shape_profile_B<batch>.json"""
import json
import os
import sys
from collections import OrderedDict

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from instancediffusion_amd.host.config import SD15_BOX_CFG  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 72
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
# a guidance pair, as the samplers form their forwards (engine.PAIR_HOIST); PROFILE_UNPAIRED=1: `batch` distinct rows
paired = batch % 2 == 0 and os.environ.get("PROFILE_UNPAIRED") != "1"
if paired:
    x[batch // 2:] = x[:batch // 2]
for _ in range(2):
    eng.forward_cond(x, t, cond, paired=paired)
torch.cuda.synchronize()


class ShapeTimer(bench.OpTimer):
    def __getattr__(self, name):
        fn = getattr(self.ops, name)
        if name not in ("gemm", "conv3x3", "attention", "groupnorm", "layernorm", "scaleu_concat", "row_stats", "mlp_geglu", "conv_in"):
            return fn

        def timed(*a, **k):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            r = fn(*a, **k)
            e.record()
            if name == "gemm":
                A, W, out = a[0], a[1], a[2]
                nb = out.shape[0] if out.dim() == 3 and A.dim() == 3 and W.dim() == 3 else 1
                key = f"gemm M{A.numel() // A.shape[-1] // nb} N{W.shape[-2]} K{A.shape[-1]} b{nb}" + (" geglu" if k.get("geglu") else "")
            elif name == "conv3x3":
                xx, w, out = a[0], a[1], a[2]
                key = f"conv {tuple(xx.shape)}->{w.shape[0]} s{k.get('stride', 1)} u{k.get('upsample', 0)}"
            elif name == "attention":
                q = a[0]
                key = f"attn Nq{q.shape[1]} C{q.shape[2]} n0={a[3]} n1={k.get('n1', 0)}"
            elif name == "mlp_geglu":
                key = f"mlp_geglu (fused GEGLU-in + ff-out) M{a[0].shape[0]} C{a[0].shape[1]}"
            else:
                key = f"{name} {tuple(a[0].shape)}"
            self.records.append((key, self._work(name, a, k), s, e))
            return r
        return timed


timer = ShapeTimer(eng.ops)
real = eng.ops
eng.ops = timer
eng.forward_cond(x, t, cond, paired=paired)
torch.cuda.synchronize()
eng.ops = real
tab = OrderedDict()
for key, work, s, e in timer.records:
    d = tab.setdefault(key, dict(n=0, ms=0.0, gflop=0.0))
    d["n"] += 1
    d["ms"] += s.elapsed_time(e)
    d["gflop"] += work / 1e9
rows = sorted(tab.items(), key=lambda kv: -kv[1]["ms"])
tot = sum(d["ms"] for _, d in rows)
out = []
for key, d in rows:
    tf = d["gflop"] / d["ms"] if d["ms"] > 0 else 0.0
    out.append(dict(op=key, launches=d["n"], ms=round(d["ms"], 3), tflops=round(tf, 1)))
    print(f"{d['ms']:8.3f} ms  {100 * d['ms'] / tot:5.1f}%  x{d['n']:<3d} {tf:7.1f} TF  {key}")
print(f"total {tot:.2f} ms at batch {batch}")
os.makedirs("gpurun_out", exist_ok=True)
json.dump(dict(batch=batch, total_ms=tot, rows=out), open(f"gpurun_out/shape_profile_B{batch}.json", "w"), indent=1)
