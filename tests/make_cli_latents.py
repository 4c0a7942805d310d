"""Writes tests/golden/cli_latents.pt: the CPU oracle's final latent for each case of tests/test_inference_cli_gpu.py.

    python tests/make_cli_latents.py            # ~2.5 min on 8 cores (52 full-size fp32 forwards)

The fixture is ORACLE output, not reference output: ``inference.py --synthetic_weights`` has no counterpart a reference run could
produce offline (no trained weights, no CLIP); the oracle (oracle/ref_cpu.py) is what is pinned to the reference's goldens
(tests/test_oracle_golden.py), and this file only saves the GPU box from re-running it on every ``-m gpu`` pass.  Everything that
shapes the result is recorded next to it: steps, alpha, seed, negative prompt, the demo JSON and config of each case, torch version.
"""
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    from tests import test_inference_cli_gpu as t
    out = dict(steps=t.STEPS, alpha=t.ALPHA, seed=t.SEED, negative_prompt=t.DEFAULT_NEG, torch=str(torch.__version__), cases={})
    for name, cfg_name, input_json, mis in t.CASES:
        t0 = time.time()
        lat, n = t._oracle_latent(cfg_name, input_json, t.STEPS, mis, t.ALPHA, t.SEED, t.DEFAULT_NEG)
        out["cases"][cfg_name] = dict(latent=lat.float().contiguous(), n_forward=n, input_json=input_json, mis=mis)
        print(f"{name}: {n} oracle forwards in {time.time() - t0:.0f} s, latent rms {float(lat.float().pow(2).mean().sqrt()):.4f}", flush=True)
    torch.save(out, t.FIXTURE)
    print("wrote", t.FIXTURE)


if __name__ == "__main__":
    main()
