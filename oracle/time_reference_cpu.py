#!/usr/bin/env python
"""TEST INFRASTRUCTURE (builder container only: needs /root/reference).  Times the UNMODIFIED reference UNet forward on this
container's CPU next to the oracle port (oracle/ref_cpu.py) on the same seeded weights and inputs, so that bench.py's
`cpu_baseline` (kind "port": /root/reference does not exist on the GPU box) can be related to the reference modules
themselves.  Full 1.228 B-parameter model, configs/test_box.yaml, B = 1, 64x64 latent, C1 boxes, fp32.

    python oracle/time_reference_cpu.py [threads] [iters]     ->  profiles/r02_cpu_reference_vs_port.json
"""
import json
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (installs nothing until install_shims() is called)


@torch.no_grad()
def main():
    threads = int(sys.argv[1]) if len(sys.argv) > 1 else (os.cpu_count() or 1)
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    torch.set_num_threads(threads)
    mg.install_shims()
    cfg = mg.load_cfg("test_box.yaml", "full")
    model, gi, diffusion, schema, synth = mg.build(cfg)
    g = torch.Generator().manual_seed(1234)
    gb = synth.make_grounding_batch(1, torch.tensor(synth.C1_BOXES), g)
    x = torch.randn(1, 4, 64, 64, generator=g)
    ctx = torch.randn(1, 77, 768, generator=g)
    t = torch.full((1,), 981, dtype=torch.long)
    grounding = gi.prepare(gb)

    def timed(fn):
        fn()                                   # warm-up
        ts = []
        for _ in range(iters):
            t0 = time.time()
            fn()
            ts.append(time.time() - t0)
        return sorted(ts)[len(ts) // 2], ts

    ref_med, ref_all = timed(lambda: model(dict(x=x, timesteps=t, context=ctx, grounding_input=grounding)))
    eps_ref = model(dict(x=x, timesteps=t, context=ctx, grounding_input=grounding))

    sys.path.insert(0, REPO)
    from oracle import ref_cpu
    sd = {k: v.detach() for k, v in model.state_dict().items()}
    pcfg = dict(cfg["model"]["params"])
    pcfg.update(cfg["model"]["params"]["grounding_tokenizer"]["params"])
    from instancediffusion_amd.host.config import SD15_BOX_CFG
    ocfg = dict(SD15_BOX_CFG)
    og = ref_cpu.prepare_grounding(gb)
    objs, _ = ref_cpu.unifusion(sd, ocfg, og)
    port_med, port_all = timed(lambda: ref_cpu.unet_forward(sd, ocfg, x, t, ctx, objs))
    eps_port = ref_cpu.unet_forward(sd, ocfg, x, t, ctx, objs)
    rel = float(((eps_port - eps_ref).pow(2).mean() / eps_ref.pow(2).mean()).sqrt())
    out = dict(what="UNet forward, full SD-1.5 InstanceDiffusion model (1.228 B parameters), B=1, 64x64 latent, C1 boxes, fp32, CPU",
               host="builder container (no GPU)", threads=threads, iters=iters,
               reference_module_s=round(ref_med, 3), reference_all_s=[round(v, 3) for v in ref_all],
               oracle_port_s=round(port_med, 3), oracle_port_all_s=[round(v, 3) for v in port_all],
               port_over_reference=round(port_med / ref_med, 3), port_vs_reference_rel_rms=rel,
               note="the port additionally hoists UniFusion out of the forward (56 GF of 1227 GF); the reference forward includes it")
    os.makedirs(os.path.join(REPO, "profiles"), exist_ok=True)
    json.dump(out, open(os.path.join(REPO, "profiles", "r02_cpu_reference_vs_port.json"), "w"), indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
