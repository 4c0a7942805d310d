#!/usr/bin/env python
"""Time AutoencoderKL.decode on one MI355X: full SD-1.5 KL-f8 decoder, 64x64 latents -> 512x512.

    python tools/vae_bench.py [batch] [iters]

Prints ONE JSON line: ms per image, images/s and achieved TFLOP/s against the 2514.5 GFLOP/image of the reference
decoder, plus per-op-family HIP-event times of one eager decode.
"""
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
GFLOP_PER_IMAGE = 2514.5


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    from instancediffusion_amd import synth     # seeded synthetic weights (no checkpoints offline)
    from instancediffusion_amd.host.config import instantiate_from_config, load_yaml
    cfg = load_yaml(os.path.join(REPO, "configs", "test_box.yaml"))
    with torch.device("meta"):
        ae = instantiate_from_config(cfg["autoencoder"])
    ae.load_state_dict(synth.synth_state_dict({k: tuple(v.shape) for k, v in ae.state_dict().items()}, 7), assign=True)
    ae.eval()
    ae.max_decode_batch = B
    z = torch.randn(B, 4, 64, 64, device="cuda") * 0.18215 * 4
    ae.decode(z)                                 # warm-up + graph capture
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        img = ae.decode(z)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    assert torch.isfinite(img).all()
    # per-family times of one eager decode (HIP events on the launch stream)
    eng = ae.engine
    real = eng.ops
    rec = []

    class Timer:
        def __getattr__(self, name):
            fn = getattr(real, name)
            if name in ("empty", "zeros", "device", "dtype"):
                return fn

            def timed(*a, **k):
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                r = fn(*a, **k)
                e.record()
                rec.append((name, s, e))
                return r
            return timed if callable(fn) else fn
    eng.ops = Timer()
    try:
        eng._decode_ops(eng.buf("io.z", z.shape, torch.float32), eng.buf("io.img", (B, 3, 512, 512), torch.float32))
    finally:
        eng.ops = real
    torch.cuda.synchronize()
    fam = {}
    for name, s, e in rec:
        fam[name] = fam.get(name, 0.0) + s.elapsed_time(e)
    print(json.dumps(dict(what="AutoencoderKL.decode, SD-1.5 KL-f8, 64x64 latent -> 512x512, bf16", batch=B,
                          ms_per_image=round(dt / B * 1e3, 3), images_per_s=round(B / dt, 2),
                          tflops=round(B * GFLOP_PER_IMAGE / dt / 1e3, 1),
                          eager_family_ms={k: round(v, 3) for k, v in sorted(fam.items(), key=lambda kv: -kv[1])})))


if __name__ == "__main__":
    main()
