"""The reference's OWN low-precision noise floor per golden case -- builder container only (TEST INFRASTRUCTURE).

SURVEY.md §8c anchors the parity tolerances on "the unmodified reference under torch.autocast vs its own fp32 output".
That anchor was measured on the C1 box inputs only; this script measures it for EVERY forward golden case (same seeded
weights and inputs as oracle/make_golden.py) and writes ``tests/golden/noise_floor.json``:
    {tag: {"bf16": rel-RMS, "fp16": rel-RMS, "bf16_maxabs_over_rms": ..., "fp16_maxabs_over_rms": ...}}
The GPU parity tests take ``max(SURVEY bar, 1.3 x this floor)`` as the per-case tolerance and print both.

    python oracle/noise_floor.py [tag ...]
"""
from __future__ import annotations

import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402

CASES = {
    "tiny_box": dict(cfg="test_box.yaml", variant="tiny", latent=16, n_boxes=3, batch=2),
    "tiny_mask": dict(cfg="test_mask.yaml", variant="tiny", latent=16, n_boxes=3, batch=1, with_polygons=True, with_segs=True),
    "tiny_point": dict(cfg="test_point.yaml", variant="tiny", latent=16, n_boxes=3, batch=1),
    "tiny_scribble": dict(cfg="test_scribble.yaml", variant="tiny", latent=16, n_boxes=3, batch=1, with_scribbles=True,
                          with_polygons=True, with_segs=True),
    "mid_box": dict(cfg="test_box.yaml", variant="mid", latent=16, n_boxes=3, batch=2),
    "full_box_c1": dict(cfg="test_box.yaml", variant="full", latent=64, n_boxes=4, batch=1, boxes="c1"),
    "full_mask_c4": dict(cfg="test_mask.yaml", variant="full", latent=96, n_boxes=12, batch=1, with_polygons=True,
                         with_segs=True),
}


def rel(a, b):
    a, b = a.double(), b.double()
    return float(((a - b).pow(2).mean() / b.pow(2).mean()).sqrt()), float((a - b).abs().max() / b.pow(2).mean().sqrt())


@torch.no_grad()
def floor(tag, c):
    cfg = mg.load_cfg(c["cfg"], c["variant"])
    model, gi, diffusion, schema, synth = mg.build(cfg)
    g = torch.Generator().manual_seed(1234)
    bx = torch.tensor(synth.C1_BOXES) if c.get("boxes") == "c1" else synth.random_boxes(c["n_boxes"], g)
    gb = synth.make_grounding_batch(c["batch"], bx, g, with_scribbles=c.get("with_scribbles", False),
                                    with_polygons=c.get("with_polygons", False), with_segs=c.get("with_segs", False))
    L = c["latent"]
    x = torch.randn(c["batch"], 4, L, L, generator=g)
    context = torch.randn(c["batch"], 77, 768, generator=g)
    t = torch.full((c["batch"],), 981, dtype=torch.long)
    inp = dict(x=x, timesteps=t, context=context, grounding_input=gi.prepare(gb))
    gold = torch.load(os.path.join(mg.GOLD, f"{tag}.pt"), weights_only=False)
    want = model(inp).float()
    assert torch.equal(want, gold["eps_cond"]), "must reproduce the committed golden bit for bit"
    out = {}
    for name, dt in (("bf16", torch.bfloat16), ("fp16", torch.float16)):
        try:
            with torch.autocast("cpu", dtype=dt):
                got = model(inp).float()
            out[name], out[name + "_maxabs_over_rms"] = rel(got, want)
        except Exception as e:                      # an op without a CPU fp16 kernel
            out[name] = None
            out[name + "_error"] = repr(e)[:200]
    print(tag, out, flush=True)
    return out


def main():
    mg.install_shims()
    torch.set_num_threads(os.cpu_count())
    tags = sys.argv[1:] or list(CASES)
    path = os.path.join(mg.GOLD, "noise_floor.json")
    res = json.load(open(path)) if os.path.exists(path) else {}
    for tag in tags:
        res[tag] = floor(tag, CASES[tag])
        json.dump(res, open(path, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
