#!/usr/bin/env python
"""Entry-point mirror of the reference ``inference.py`` (demo CLI) on the MI355X path.

Same flags, same demo-JSON format (``demos/*.json``: caption, width/height, annos[{bbox, mask, point, scribble,
caption}]) and the same call sequence as the reference (inference.py:165-309): demo JSON -> ``meta`` ->
``utils.input.prepare_batch`` (+ ``prepare_instance_meta`` per instance when MIS is on) -> ``PLMSSampler`` /
``PLMSSamplerInst`` -> ``autoencoder.decode`` -> PNGs.  UNet, samplers and VAE decoder run on the HIP engine.

Two neighbours of the path need assets that do not exist offline; both are handled explicitly, never silently:
  * text encoding (CLIP-L/14: ``ldm/modules/encoders``, ``utils/model.py:12-18``): with ``--ckpt`` the checkpoint's
    text encoder is used (needs the BPE vocabulary locally, see host/text_encoder.py); ``--text_encoder synthetic``
    (default without a checkpoint) draws a deterministic embedding per string -- layout-faithful, not meaningful;
  * weights: ``--ckpt instancediffusion_sd15.pth`` goes through ``utils.checkpoint.load_model_ckpt`` (ema -> model
    fallback, autoencoder / text_encoder / diffusion sub-dicts); without it ``--synthetic_weights`` must be given.
The SDXL-refiner cascade (``--cascade_strength``, diffusers + downloads) is outside the path.
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
from instancediffusion_amd.host.input import meta_from_demo_json, prepare_batch, prepare_instance_meta
from instancediffusion_amd.host.samplers import PLMSSampler, PLMSSamplerInst

MAX_OBJS = 30


def text_embedding(s: str, dim: int = 768, rows: int = 1) -> torch.Tensor:
    g = torch.Generator().manual_seed(int.from_bytes(hashlib.sha256(s.encode()).digest()[:7], "little"))
    return torch.randn(rows, dim, generator=g)


class SyntheticTextEncoder:
    """Stand-in for FrozenCLIPEmbedder.encode (encoders/modules.py:144-172) and for the pooled CLIP phrase feature
    (utils/model.py:130-152): deterministic per string."""

    def encode(self, prompts):
        return torch.stack([text_embedding("ctx:" + p, 768, 77) for p in prompts])

    def pooled(self, phrase):
        return text_embedding("pooled:" + phrase, 768, 1)[0]


class ClipPhraseEncoder:
    """Pooled phrase features from the checkpoint's own CLIP text transformer (the reference loads a second copy of
    the same CLIP-L/14 text model through ``CLIPModel``; its pooler output is the same tensor)."""

    def __init__(self, text_encoder):
        self.text_encoder = text_encoder

    def pooled(self, phrase):
        return self.text_encoder.encode([phrase], return_pooler_output=True)[1][0]


def get_model_inputs(meta, gi, text_encoder, phrase_encoder, num_images, device, starting_noise, negative_prompt=None,
                     instance_input=False, use_masked_att=False):
    """inference.py:39-78.  ``use_masked_att``: also build the box-shaped visibility planes and hand them to the model
    (``eval_local.py --use_masked_att``; the reference's inference.py builds them but never passes them on)."""
    batch = prepare_batch(meta, batch=num_images, max_objs=MAX_OBJS, model=phrase_encoder, processor=None,
                          image_size=starting_noise.shape[-1], use_masked_att=use_masked_att, device=device)
    context = text_encoder.encode([meta["prompt"]] * num_images).to(device)
    uc = None
    if not instance_input:
        uc = text_encoder.encode(num_images * [negative_prompt if negative_prompt is not None else ""]).to(device)
    grounding_input = gi.prepare(batch, return_att_masks=use_masked_att)
    return dict(x=starting_noise, timesteps=None, context=context, grounding_input=grounding_input), uc


def save_images(images: torch.Tensor, folder: str) -> list:
    """inference.py:120-130: clamp to [-1, 1], map to uint8 RGB, one PNG per sample (numbered after existing files)."""
    import numpy as np
    from PIL import Image
    os.makedirs(folder, exist_ok=True)
    start = len(os.listdir(folder))
    names = []
    for i, sample in enumerate(images):
        sample = torch.clamp(sample, min=-1, max=1) * 0.5 + 0.5
        arr = (sample.float().cpu().numpy().transpose(1, 2, 0) * 255).astype(np.uint8)
        name = os.path.join(folder, f"{start + i}.png")
        Image.fromarray(arr).save(name)
        names.append(name)
    return names


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--output", type=str, default="OUTPUT")
    ap.add_argument("--num_images", type=int, default=8)
    ap.add_argument("--guidance_scale", type=float, default=7.5)
    ap.add_argument("--negative_prompt", type=str, default="longbody, lowres, bad anatomy, bad hands, missing fingers, "
                    "extra digit, fewer digits, cropped, worst quality, low quality")
    ap.add_argument("--input_json", type=str, default="demos/demo_four_boxes.json")
    ap.add_argument("--ckpt", type=str, default=None)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--alpha", type=float, default=0.75)
    ap.add_argument("--mis", type=float, default=0.36)
    ap.add_argument("--cascade_strength", type=float, default=0.0)
    ap.add_argument("--test_config", type=str, default="configs/test_mask.yaml")
    ap.add_argument("--device", type=str, default="cuda")
    ap.add_argument("--text_encoder", choices=["synthetic", "clip"], default=None)
    ap.add_argument("--synthetic_weights", action="store_true")
    ap.add_argument("--dtype", choices=["bf16", "fp16"], default="bf16")
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--save_latents", action="store_true")
    ap.add_argument("--use_masked_att", action="store_true",
                    help="masked gated self-attention (attention.py:187-255): instance patches only see their own box")
    args = ap.parse_args()
    if args.cascade_strength > 0:
        raise SystemExit("the SDXL refiner cascade is outside the sampling path (needs diffusers + downloads)")
    dev = torch.device(args.device)
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float16

    if args.ckpt:
        from utils.checkpoint import load_model_ckpt
        model, autoencoder, clip_text, diffusion, cfg = load_model_ckpt(args.ckpt, args, args.device)
        use_clip = (args.text_encoder or "clip") == "clip"
    elif args.synthetic_weights:
        from instancediffusion_amd import synth
        cfg = load_yaml(args.test_config)
        with torch.device("meta"):
            model = instantiate_from_config(cfg["model"])
            autoencoder = instantiate_from_config(cfg["autoencoder"])
        model.load_state_dict(synth.synth_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}),
                              assign=True)
        model.first_conv_sd_override = synth.synth_first_conv_sd()
        autoencoder.load_state_dict(
            synth.synth_state_dict({k: tuple(v.shape) for k, v in autoencoder.state_dict().items()}, 7), assign=True)
        model.eval(), autoencoder.eval()
        diffusion = instantiate_from_config(cfg["diffusion"]).to(args.device)
        clip_text, use_clip = None, False
        if args.text_encoder == "clip":
            raise SystemExit("--text_encoder clip needs --ckpt (the CLIP weights live in the checkpoint)")
    else:
        raise SystemExit("give --ckpt instancediffusion_sd15.pth, or --synthetic_weights for a dry run")
    model.compute_dtype = dtype
    autoencoder.compute_dtype = dtype
    if args.use_masked_att:
        model.efficient_attention = False      # the reference builds the mask only on its non-efficient path (:189)
        model.invalidate_engine()
    gi = instantiate_from_config(cfg["grounding_tokenizer_input"])
    model.grounding_tokenizer_input = gi
    text_encoder = clip_text if use_clip else SyntheticTextEncoder()
    phrase_encoder = ClipPhraseEncoder(clip_text) if use_clip else text_encoder

    data = json.load(open(args.input_json))
    save_folder_name = f"gc{args.guidance_scale}-seed{args.seed}-alpha{args.alpha}"
    meta = meta_from_demo_json(data, args.alpha, ckpt=args.ckpt, save_folder_name=save_folder_name)
    torch.manual_seed(args.seed)
    starting_noise = torch.randn(args.num_images, 4, model.image_size, model.image_size).to(dev)

    inp, uc = get_model_inputs(meta, gi, text_encoder, phrase_encoder, args.num_images, dev, starting_noise,
                               args.negative_prompt, use_masked_att=args.use_masked_att)
    ag = partial(alpha_generator, type=meta["alpha_type"])
    shape = (args.num_images, model.in_channels, model.image_size, model.image_size)
    if args.mis > 0:
        sampler = PLMSSamplerInst(diffusion, model, alpha_generator_func=ag, set_alpha_scale=set_alpha_scale, mis=args.mis)
        inputs = [inp]
        for i in range(len(meta["phrases"])):
            inst, _ = get_model_inputs(prepare_instance_meta(meta, i), gi, text_encoder, phrase_encoder, args.num_images,
                                       dev, starting_noise, instance_input=True, use_masked_att=args.use_masked_att)
            inputs.append(inst)
        samples = sampler.sample(S=args.steps, shape=shape, input=inputs, uc=uc, guidance_scale=args.guidance_scale)
    else:
        sampler = PLMSSampler(diffusion, model, alpha_generator_func=ag, set_alpha_scale=set_alpha_scale)
        samples = sampler.sample(S=args.steps, shape=shape, input=inp, uc=uc, guidance_scale=args.guidance_scale)
    images = autoencoder.decode(samples)                                   # inference.py:95
    folder = os.path.join(args.output, save_folder_name)
    names = save_images(images, folder)
    if args.save_latents:
        torch.save(dict(latents=samples.cpu(), caption=data["caption"], phrases=meta["phrases"]),
                   os.path.join(folder, "latents.pt"))
    print(f"saved {len(names)} images {tuple(images.shape[1:])} to {folder}")


if __name__ == "__main__":
    main()
