#!/usr/bin/env python
"""Entry-point mirror of the reference ``inference.py`` (demo CLI) for the MI355X sampling path.

Same flags and the same demo-JSON format (``demos/*.json``: caption, width/height, annos[{bbox, mask, point, scribble,
caption}]) as the reference (inference.py:165-297).  What is in scope here is the SAMPLING path: UNet denoise step +
PLMS / Multi-instance Sampler on the HIP engine.  The two neighbours of the path that need assets which do not exist
offline are handled explicitly, never silently:

  * text encoding (CLIP-L/14, `ldm/modules/encoders`, `utils/model.py:12-18`): ``--text_encoder synthetic`` (default)
    draws a deterministic embedding per string (seeded by its hash) -- layout-faithful, not semantically meaningful;
    ``--text_encoder clip --clip_path DIR`` uses a local HF CLIP checkpoint when one is available.
  * VAE decode (`ldm/models/autoencoder.py`, out of scope): the final LATENTS are saved as
    ``<output>/<name>/latents.pt``; decode them with the reference autoencoder.

Weights: ``--ckpt instancediffusion_sd15.pth`` loads the reference checkpoint (``['ema']`` else ``['model']``,
utils/checkpoint.py:238-244); without it ``--synthetic_weights`` must be given (seeded random weights).
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
from functools import partial

import torch

from instancediffusion_amd.host.alpha import alpha_generator, set_alpha_scale
from instancediffusion_amd.host.config import instantiate_from_config, load_yaml
from instancediffusion_amd.host.samplers import PLMSSampler, PLMSSamplerInst

MAX_OBJS = 30
N_SCRIBBLE, N_POLYGON = 20, 256


def text_embedding(s: str, dim: int = 768, rows: int = 1) -> torch.Tensor:
    g = torch.Generator().manual_seed(int.from_bytes(hashlib.sha256(s.encode()).digest()[:7], "little"))
    return torch.randn(rows, dim, generator=g)


class SyntheticTextEncoder:
    """Stand-in for FrozenCLIPEmbedder.encode (encoders/modules.py:144-172): [B, 77, 768] per prompt, deterministic."""

    def encode(self, prompts):
        return torch.stack([text_embedding("ctx:" + p, 768, 77) for p in prompts])

    def pooled(self, phrase):
        return text_embedding("pooled:" + phrase, 768, 1)[0]


def rescale_box(bbox, width, height):
    """xywh pixels -> normalised xyxy (inference.py:132-137)."""
    x0, y0 = bbox[0] / width, bbox[1] / height
    x1, y1 = (bbox[0] + bbox[2]) / width, (bbox[1] + bbox[3]) / height
    return [x0, y0, x1, y1]


def build_batch(data: dict, enc: SyntheticTextEncoder, batch: int) -> dict:
    """`prepare_batch` (utils/input.py:41-125) for the fields the tokenizer consumes; masks/polygons stay zero exactly
    as the reference demo script leaves them (inference.py:249 re-initialises the mask list)."""
    W, H = data.get("width", 512), data.get("height", 512)
    out = dict(boxes=torch.zeros(MAX_OBJS, 4), masks=torch.zeros(MAX_OBJS), text_embeddings=torch.zeros(MAX_OBJS, 768),
               points=torch.zeros(MAX_OBJS, 2), scribbles=torch.zeros(MAX_OBJS, N_SCRIBBLE * 2),
               polygons=torch.zeros(MAX_OBJS, N_POLYGON * 2), segs=torch.zeros(MAX_OBJS, 512, 512))
    phrases = []
    for i, a in enumerate(data["annos"][:MAX_OBJS]):
        box = rescale_box(a["bbox"], W, H)
        out["boxes"][i] = torch.tensor(box)
        out["masks"][i] = 1
        out["text_embeddings"][i] = enc.pooled(a["caption"])
        pt = a.get("point")
        out["points"][i] = torch.tensor([pt[0] / W, pt[1] / H]) if pt else torch.tensor(
            [(box[0] + box[2]) / 2, (box[1] + box[3]) / 2])
        sc = a.get("scribble")
        if sc:
            flat = torch.tensor([[p[0] / W, p[1] / H] for p in sc][:N_SCRIBBLE]).flatten()
            out["scribbles"][i, :flat.numel()] = flat
        phrases.append(a["caption"])
    return {k: v.unsqueeze(0).repeat(batch, *([1] * v.dim())) for k, v in out.items()}, phrases


def instance_batch(full: dict, i: int) -> dict:
    out = {k: torch.zeros_like(v) for k, v in full.items()}
    for k in full:
        out[k][:, 0] = full[k][:, i]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--output", type=str, default="OUTPUT")
    ap.add_argument("--num_images", type=int, default=8)
    ap.add_argument("--guidance_scale", type=float, default=7.5)
    ap.add_argument("--negative_prompt", type=str, default="longbody, lowres, bad anatomy, bad hands, missing fingers, "
                    "extra digit, fewer digits, cropped, worst quality, low quality")
    ap.add_argument("--input_json", type=str, required=True)
    ap.add_argument("--ckpt", type=str, default=None)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--alpha", type=float, default=0.8)
    ap.add_argument("--mis", type=float, default=0.4)
    ap.add_argument("--cascade_strength", type=float, default=0.0)
    ap.add_argument("--test_config", type=str, default="configs/test_box.yaml")
    ap.add_argument("--device", type=str, default="cuda")
    ap.add_argument("--text_encoder", choices=["synthetic", "clip"], default="synthetic")
    ap.add_argument("--clip_path", type=str, default=None)
    ap.add_argument("--synthetic_weights", action="store_true")
    ap.add_argument("--dtype", choices=["bf16", "fp16"], default="bf16")
    args = ap.parse_args()
    if args.cascade_strength > 0:
        raise SystemExit("the SDXL refiner cascade is outside the sampling path (needs diffusers + downloads)")
    if args.text_encoder == "clip":
        raise SystemExit("CLIP text encoding needs local HF weights; wire ldm.modules.encoders from the reference tree")

    cfg = load_yaml(args.test_config)
    with torch.device("meta"):
        model = instantiate_from_config(cfg["model"])
    if args.ckpt:
        saved = torch.load(args.ckpt, map_location="cpu")
        sd = saved["ema"] if "ema" in saved else saved["model"]
    elif args.synthetic_weights:
        from instancediffusion_amd import synth
        sd = synth.synth_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()})
        model.first_conv_sd_override = synth.synth_first_conv_sd()
    else:
        raise SystemExit("give --ckpt instancediffusion_sd15.pth, or --synthetic_weights for a dry run")
    model.load_state_dict(sd, assign=True)
    model.eval()
    model.compute_dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float16
    diffusion = instantiate_from_config(cfg["diffusion"]).to(args.device)
    gi = instantiate_from_config(cfg["grounding_tokenizer_input"])
    model.grounding_tokenizer_input = gi

    data = json.load(open(args.input_json))
    enc = SyntheticTextEncoder()
    dev = torch.device(args.device)
    batch, phrases = build_batch(data, enc, args.num_images)
    batch = {k: v.to(dev) for k, v in batch.items()}
    torch.manual_seed(args.seed)
    noise = torch.randn(args.num_images, 4, model.image_size, model.image_size).to(dev)
    context = enc.encode([data["caption"]] * args.num_images).to(dev)
    uc = enc.encode([args.negative_prompt] * args.num_images).to(dev)
    ag = partial(alpha_generator, type=[args.alpha, 0.0, 1 - args.alpha])
    inp = dict(x=noise, timesteps=None, context=context, grounding_input=gi.prepare(batch))
    shape = (args.num_images, model.in_channels, model.image_size, model.image_size)
    if args.mis > 0:
        sampler = PLMSSamplerInst(diffusion, model, alpha_generator_func=ag, set_alpha_scale=set_alpha_scale, mis=args.mis)
        inputs = [inp]
        for i, ph in enumerate(phrases):
            inputs.append(dict(x=noise, timesteps=None, context=enc.encode([ph] * args.num_images).to(dev),
                               grounding_input=gi.prepare(instance_batch(batch, i))))
        gi.prepare(batch)
        samples = sampler.sample(S=50, shape=shape, input=inputs, uc=uc, guidance_scale=args.guidance_scale)
    else:
        sampler = PLMSSampler(diffusion, model, alpha_generator_func=ag, set_alpha_scale=set_alpha_scale)
        samples = sampler.sample(S=50, shape=shape, input=inp, uc=uc, guidance_scale=args.guidance_scale)
    name = os.path.splitext(os.path.basename(args.input_json))[0]
    folder = os.path.join(args.output, name)
    os.makedirs(folder, exist_ok=True)
    torch.save(dict(latents=samples.cpu(), caption=data["caption"], phrases=phrases), os.path.join(folder, "latents.pt"))
    print(f"saved {tuple(samples.shape)} latents to {folder}/latents.pt (decode with the reference AutoencoderKL)")


if __name__ == "__main__":
    main()
