"""One-off functional + timing runs of the non-headline BASELINE.json configs on a single MI355X:
  C2  SD1.5 + 4 boxes (demo_cat_dog_robin layout), MIS off, alpha 0.8, bf16        (102 forwards / image)
  C4  test_mask.yaml, 768x768 (96x96 latent), N=12 instance masks, MIS 0.36, bf16  (558 forwards / image)
  C5  test_point.yaml and test_scribble.yaml, batch 4, N=8, MIS 0.36, fp16
Writes gpurun_out/configs.json.  (The headline C3 is bench.py.)  Usage: python tools/run_configs.py [c2 c4 c5p c5s]"""
import json
import os
import sys
import time
from functools import partial

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from instancediffusion_amd import synth  # noqa: E402
from instancediffusion_amd.host.alpha import alpha_generator, set_alpha_scale  # noqa: E402
from instancediffusion_amd.host.config import unet_kwargs_from_cfg  # noqa: E402
from instancediffusion_amd.host.diffusion import LatentDiffusion  # noqa: E402
from instancediffusion_amd.host.samplers import PLMSSampler, PLMSSamplerInst  # noqa: E402
from grounding_input.text_grounding_tokinzer_input import GroundingNetInput  # noqa: E402
from ldm.modules.diffusionmodules.openaimodel import UNetModel  # noqa: E402

BASE = dict(in_channels=4, out_channels=4, model_channels=320, attention_resolutions=(4, 2, 1), num_res_blocks=2,
            channel_mult=(1, 2, 4, 4), num_heads=8, context_dim=768, in_dim=768, out_dim=768, mid_dim=3072,
            test_drop_boxes=False, test_drop_points=False, test_drop_scribbles=True, test_drop_masks=True)
CASES = {
    "c2": dict(drops={}, latent=64, n=4, boxes="c1", batch=1, mis=0.0, dtype=torch.bfloat16),
    # the same at the reference CLI's own batch (inference.py --num_images 8): 16-row forwards instead of 2-row ones
    "c2x8": dict(drops={}, latent=64, n=4, boxes="c1", batch=8, mis=0.0, dtype=torch.bfloat16),
    "c4": dict(drops=dict(test_drop_masks=False), latent=96, n=12, batch=1, mis=0.36, dtype=torch.bfloat16, segs=True, poly=True),
    "c5p": dict(drops=dict(test_drop_boxes=True), latent=64, n=8, batch=4, mis=0.36, dtype=torch.float16),
    "c5s": dict(drops=dict(test_drop_scribbles=False, test_drop_masks=False), latent=64, n=8, batch=4, mis=0.36,
                dtype=torch.float16, segs=True, poly=True, scrib=True),
}


def run(tag):
    c = CASES[tag]
    cfg = dict(BASE)
    cfg.update(c["drops"])
    with torch.device("meta"):
        model = UNetModel(**unet_kwargs_from_cfg(cfg))
    model.load_state_dict(synth.synth_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}), assign=True)
    model.eval()
    model.first_conv_sd_override = synth.synth_first_conv_sd()
    model.compute_dtype = c["dtype"]
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(1234)
    boxes = torch.tensor(synth.C1_BOXES) if c.get("boxes") == "c1" else synth.random_boxes(c["n"], g)
    B, L = c["batch"], c["latent"]
    gb = synth.make_grounding_batch(B, boxes, g, with_scribbles=c.get("scrib", False), with_polygons=c.get("poly", False),
                                    with_segs=c.get("segs", False))
    x = torch.randn(B, 4, L, L, generator=g).to(dev)
    ctx, uc = torch.randn(B, 77, 768, generator=g).to(dev), torch.randn(B, 77, 768, generator=g).to(dev)
    gi = GroundingNetInput()
    model.grounding_tokenizer_input = gi

    def cu(d):
        return {k: v.to(dev) for k, v in d.items()}
    diffusion = LatentDiffusion(linear_start=0.00085, linear_end=0.012, timesteps=1000).to(dev)
    ag = partial(alpha_generator, type=[0.8, 0.0, 0.2])
    inp0 = dict(x=x, timesteps=None, context=ctx, grounding_input=gi.prepare(cu(gb)))
    if c["mis"] > 0:
        sampler = PLMSSamplerInst(diffusion, model, alpha_generator_func=ag, set_alpha_scale=set_alpha_scale, mis=c["mis"])
        inputs = [inp0]
        for i in range(boxes.shape[0]):
            inputs.append(dict(x=x, timesteps=None, context=torch.randn(B, 77, 768, generator=g).to(dev),
                               grounding_input=gi.prepare(cu(synth.instance_batch(gb, i)))))
        gi.prepare(cu(gb))
    else:
        sampler, inputs = PLMSSampler(diffusion, model, alpha_generator_func=ag, set_alpha_scale=set_alpha_scale), inp0
    shape = (B, 4, L, L)

    def once():
        ins = [dict(d) for d in inputs] if isinstance(inputs, list) else dict(inputs)
        out = sampler.sample(S=50, shape=shape, input=ins, uc=uc, guidance_scale=7.5)
        torch.cuda.synchronize()
        return out
    once()                                   # warm-up: packs weights, captures graphs
    t0 = time.perf_counter()
    out = once()
    dt = time.perf_counter() - t0
    ms = int(50 * c["mis"])
    nf = 2 * ((boxes.shape[0] + 1) * (ms + 1) + (50 - ms)) if c["mis"] > 0 else 2 * 51
    res = dict(config=tag, latent=L, n_instances=int(boxes.shape[0]), batch=B, mis=c["mis"],
               dtype=str(c["dtype"]).replace("torch.", ""), finite=bool(torch.isfinite(out).all()),
               seconds_per_batch=round(dt, 3), img_per_s=round(B / dt, 4), unet_forwards_per_image=nf,
               latent_rms=float(out.float().pow(2).mean().sqrt()))
    print(json.dumps(res), flush=True)
    del model
    torch.cuda.empty_cache()
    return res


if __name__ == "__main__":
    tags = sys.argv[1:] or ["c2", "c4", "c5p", "c5s"]
    out = [run(t) for t in tags]
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(out, open("gpurun_out/configs.json", "w"), indent=1)
