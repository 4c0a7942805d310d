"""Generate golden vectors from the UNMODIFIED reference (/root/reference) -- builder container only.

TEST INFRASTRUCTURE.  Run:  ``python oracle/make_golden.py [--only tiny|full|all|full_s50|full_c5|full_c2|...]``
Writes small fixtures into ``tests/golden/`` (committed), because /root/reference does not exist on the GPU
box.  The reference modules are imported as-is with three process-local shims:
  1. stub ``timm.models.layers`` / ``timm.models.registry`` (convnext.py:12-13 imports; timm not installed)
  2. ``torch.hub.load_state_dict_from_url`` -> ``{"model": {}}`` (convnext.py:154-157 downloads weights)
  3. PyYAML instead of OmegaConf for configs/*.yaml
``utils/model.py`` cannot be imported (needs CLIP weights), so its two tiny functions used by the samplers
(``set_alpha_scale`` :78-81, ``alpha_generator`` :83-117) are restated here.

Weights are synthetic and key-name-seeded (instancediffusion_amd/synth.py), loaded with load_state_dict, so
the oracle / HIP engine can regenerate identical parameters anywhere.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import types
from copy import deepcopy
from functools import partial

import numpy as np
import torch
import yaml

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
GOLD = os.path.join(REPO, "tests", "golden")


def install_shims():
    layers = types.ModuleType("timm.models.layers")
    layers.trunc_normal_ = lambda t, std=1.0, **k: torch.nn.init.trunc_normal_(t, std=std)

    class DropPath(torch.nn.Identity):
        def __init__(self, *a, **k):
            super().__init__()
    layers.DropPath = DropPath
    registry = types.ModuleType("timm.models.registry")
    registry.register_model = lambda f: f
    timm = types.ModuleType("timm")
    models = types.ModuleType("timm.models")
    import importlib.machinery
    for name, m in (("timm", timm), ("timm.models", models), ("timm.models.layers", layers), ("timm.models.registry", registry)):
        m.__spec__ = importlib.machinery.ModuleSpec(name, None)      # transformers probes find_spec("timm")
    sys.modules.update({"timm": timm, "timm.models": models, "timm.models.layers": layers,
                        "timm.models.registry": registry})
    torch.hub.load_state_dict_from_url = lambda *a, **k: {"model": {}}
    # the reference tree must win over this repo's same-named ``ldm`` mirror package
    sys.path = [p for p in sys.path if os.path.abspath(p or ".") != REPO]
    sys.path.insert(0, REF)
    os.chdir(REF)


def ref_set_alpha_scale(model, alpha_scale):            # utils/model.py:78-81
    from ldm.modules.attention import GatedSelfAttentionDense
    for m in model.modules():
        if type(m) == GatedSelfAttentionDense:
            m.scale = alpha_scale


def ref_alpha_generator(length, type=None):             # utils/model.py:83-117
    if type is None:
        type = [1, 0, 0]
    assert len(type) == 3 and type[0] + type[1] + type[2] == 1
    s0 = int(type[0] * length)
    s1 = int(type[1] * length)
    s2 = length - s0 - s1
    decay = list(np.arange(start=0, stop=1, step=1 / s1)[::-1]) if s1 != 0 else []
    al = [1] * s0 + decay + [0] * s2
    assert len(al) == length
    return al


# reduced architectures for fast tests.  "tiny": 64 base channels (head dims 8/16/32).  "mid": the real
# 320/640/1280 widths (head dims 40/80/160) but 3 levels x 1 ResBlock, run on 16x16 latents.
VARIANTS = {
    "full": {},
    "tiny": dict(model=dict(model_channels=64), tok=dict(mid_dim=256)),
    "mid": dict(model=dict(channel_mult=[1, 2, 4], num_res_blocks=1), tok=dict(mid_dim=512)),
}


def load_cfg(name: str, variant: str):
    cfg = yaml.safe_load(open(os.path.join(REF, "configs", name)))
    v = VARIANTS[variant]
    cfg["model"]["params"].update(v.get("model", {}))
    cfg["model"]["params"]["grounding_tokenizer"]["params"].update(v.get("tok", {}))
    cfg["model"]["params"]["use_checkpoint"] = False
    return cfg


def build(cfg, salt=0):
    from ldm.util import instantiate_from_config
    import importlib.util
    spec = importlib.util.spec_from_file_location("idf_synth", os.path.join(REPO, "instancediffusion_amd", "synth.py"))
    synth = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(synth)
    model = instantiate_from_config(cfg["model"]).eval()
    schema = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = synth.synth_state_dict(schema, salt)
    model.load_state_dict(sd, strict=True)
    gi = instantiate_from_config(cfg["grounding_tokenizer_input"])
    model.grounding_tokenizer_input = gi
    diffusion = instantiate_from_config(cfg["diffusion"])
    return model, gi, diffusion, schema, synth


def fp(t: torch.Tensor):
    t = t.detach().float()
    return dict(mean=float(t.mean()), std=float(t.std()), absmax=float(t.abs().max()),
                head=t.flatten()[:32].clone(), shape=list(t.shape))


def patch_first_conv(model, first_conv_sd):
    """restore_first_conv_from_SD (openaimodel.py:469-480) reads a cwd-relative .pth; for synthetic weights we
    feed it the synthetic file contents through th.load by writing a temp file in a temp cwd."""
    import tempfile
    d = tempfile.mkdtemp()
    os.makedirs(os.path.join(d, "pretrained"))
    torch.save(first_conv_sd, os.path.join(d, "pretrained", "SD_v1_5_input_conv_weight_bias.pth"))
    os.chdir(d)


@torch.no_grad()
def gen_case(tag: str, cfg_name: str, variant: str, latent: int, n_boxes: int, batch: int, *, boxes="rand",
             alpha_type=(0.6, 0.0, 0.4),
             with_scribbles=False, with_polygons=False, with_segs=False, samplers=True, S=5, mis=0.4, n_inst=2,
             seg_size=512):
    print(f"[golden] {tag}: {cfg_name} variant={variant} latent={latent}", flush=True)
    cfg = load_cfg(cfg_name, variant)
    model, gi, diffusion, schema, synth = build(cfg)
    from ldm.models.diffusion.plms import PLMSSampler
    from ldm.models.diffusion.plms_instance import PLMSSamplerInst
    g = torch.Generator().manual_seed(1234)
    bx = torch.tensor(synth.C1_BOXES) if boxes == "c1" else synth.random_boxes(n_boxes, g)
    gb = synth.make_grounding_batch(batch, bx, g, with_scribbles=with_scribbles, with_polygons=with_polygons,
                                    with_segs=with_segs, seg_size=seg_size)
    x = torch.randn(batch, 4, latent, latent, generator=g)
    context = torch.randn(batch, 77, 768, generator=g)
    uc = torch.randn(batch, 77, 768, generator=g)
    t = torch.full((batch,), 981, dtype=torch.long)
    grounding = gi.prepare(gb)
    out = {"meta": dict(tag=tag, cfg=cfg_name, variant=variant, alpha_type=list(alpha_type), latent=latent, n_boxes=int(bx.shape[0]), batch=batch,
                        boxes=boxes, with_scribbles=with_scribbles, with_polygons=with_polygons,
                        with_segs=with_segs, S=S, mis=mis, n_inst=n_inst, seg_size=seg_size,
                        x_fp=fp(x), ctx_fp=fp(context))}
    # --- single forwards, with probes via hooks
    probes = {}
    hooks = []
    for i, blk in enumerate(model.input_blocks):
        hooks.append(blk.register_forward_hook(lambda m, a, o, i=i: probes.__setitem__(f"input_blocks.{i}", fp(o))))
    hooks.append(model.middle_block.register_forward_hook(lambda m, a, o: probes.__setitem__("middle_block", fp(o))))
    for i, blk in enumerate(model.output_blocks):
        hooks.append(blk.register_forward_hook(lambda m, a, o, i=i: probes.__setitem__(f"output_blocks.{i}", fp(o))))
    objs_box = {}
    hooks.append(model.position_net.register_forward_hook(lambda m, a, o: objs_box.__setitem__("objs", o[0])))
    eps = model(dict(x=x, timesteps=t, context=context, grounding_input=grounding))
    out["eps_cond"] = eps.clone()
    out["probes_cond"] = dict(probes)
    objs = objs_box["objs"]
    out["objs_fp"] = fp(objs)
    out["objs_rows"] = objs[0, [0, 1, 29, 30, 59, 60, 90, 120, 183]].clone()   # one row per token family
    for h in hooks:
        h.remove()
    out["eps_uncond"] = model(dict(x=x, timesteps=t, context=uc)).clone()       # null grounding path
    ref_set_alpha_scale(model, 0.3)
    out["eps_scale03"] = model(dict(x=x, timesteps=t, context=context, grounding_input=grounding)).clone()
    ref_set_alpha_scale(model, 1)

    if samplers:
        alpha_type = list(alpha_type)
        first_sd = synth.synth_first_conv_sd()
        # PLMS + CFG
        m2 = deepcopy(model)
        m2.grounding_tokenizer_input = gi
        patch_first_conv(m2, first_sd)
        sampler = PLMSSampler(diffusion, m2, alpha_generator_func=partial(ref_alpha_generator, type=alpha_type),
                              set_alpha_scale=ref_set_alpha_scale)
        inp = dict(x=x.clone(), timesteps=None, context=context, grounding_input=grounding)
        out["plms"] = sampler.sample(S=S, shape=tuple(x.shape), input=inp, uc=uc, guidance_scale=7.5).clone()
        out["plms_timesteps"] = [int(v) for v in sampler.ddim_timesteps]
        # MIS
        m3 = deepcopy(model)
        m3.grounding_tokenizer_input = gi
        sampler = PLMSSamplerInst(diffusion, m3, alpha_generator_func=partial(ref_alpha_generator, type=alpha_type),
                                  set_alpha_scale=ref_set_alpha_scale, mis=mis)
        inputs = [dict(x=x.clone(), timesteps=None, context=context, grounding_input=grounding)]
        for i in range(n_inst):
            ctx_i = torch.randn(batch, 77, 768, generator=g)
            inputs.append(dict(x=x.clone(), timesteps=None, context=ctx_i,
                               grounding_input=gi.prepare(synth.instance_batch(gb, i))))
        out["mis"] = sampler.sample(S=S, shape=tuple(x.shape), input=inputs, uc=uc, guidance_scale=7.5).clone()
        os.chdir(REF)
    torch.save(out, os.path.join(GOLD, f"{tag}.pt"))
    return schema


@torch.no_grad()
def gen_plms_mask_case(tag="tiny_box_plms_mask", cfg_name="test_box.yaml", variant="tiny", latent=16, n_boxes=3, batch=2, S=5):
    """PLMSSampler.sample(mask=, x0=) -- the inpainting blend of plms.py:99-104 (``img = q_sample(x0, ts) * mask +
    (1 - mask) * img`` in front of every step) on the unmodified reference.  q_sample draws its noise from the global RNG
    (ldm.py:18); the draw is made here, recorded, and handed to the reference's own q_sample through its ``noise``
    argument, so a test can replay the same noise on any device.  Inputs are those of ``tiny_box`` (same generator)."""
    print(f"[golden] {tag}", flush=True)
    cfg = load_cfg(cfg_name, variant)
    model, gi, diffusion, schema, synth = build(cfg)
    from ldm.models.diffusion.plms import PLMSSampler
    g = torch.Generator().manual_seed(1234)
    bx = synth.random_boxes(n_boxes, g)
    gb = synth.make_grounding_batch(batch, bx, g, with_scribbles=False, with_polygons=False, with_segs=False, seg_size=512)
    x = torch.randn(batch, 4, latent, latent, generator=g)
    context = torch.randn(batch, 77, 768, generator=g)
    uc = torch.randn(batch, 77, 768, generator=g)
    g2 = torch.Generator().manual_seed(4321)
    mask = (torch.rand(batch, 1, latent, latent, generator=g2) > 0.5).float()
    x0 = torch.randn(batch, 4, latent, latent, generator=g2)
    noises = []
    real_q = diffusion.q_sample

    def q_sample(x_start, t, noise=None):
        n = torch.randn(x_start.shape, generator=g2)
        noises.append(n.clone())
        return real_q(x_start, t, noise=n)
    diffusion.q_sample = q_sample
    m2 = deepcopy(model)
    m2.grounding_tokenizer_input = gi
    patch_first_conv(m2, synth.synth_first_conv_sd())
    sampler = PLMSSampler(diffusion, m2, alpha_generator_func=partial(ref_alpha_generator, type=[1, 0, 0]),
                          set_alpha_scale=ref_set_alpha_scale)
    inp = dict(x=x.clone(), timesteps=None, context=context, grounding_input=gi.prepare(gb))
    out = sampler.sample(S=S, shape=tuple(x.shape), input=inp, uc=uc, guidance_scale=7.5, mask=mask, x0=x0).clone()
    os.chdir(REF)
    meta = dict(tag=tag, cfg=cfg_name, variant=variant, alpha_type=[1, 0, 0], latent=latent, n_boxes=int(bx.shape[0]), batch=batch,
                boxes="rand", with_scribbles=False, with_polygons=False, with_segs=False, S=S, mis=0.0, n_inst=0, seg_size=512,
                x_fp=fp(x), ctx_fp=fp(context))
    torch.save(dict(meta=meta, plms_masked=out, mask=mask, x0=x0, noises=torch.stack(noises)), os.path.join(GOLD, f"{tag}.pt"))


@torch.no_grad()
def gen_full_s50(tag="full_box_s50", S=50, mis=0.36, n_inst=8, alpha_type=(0.8, 0.0, 0.2)):
    """The BASELINE headline trajectory on the headline model: the UNMODIFIED reference ``PLMSSamplerInst``
    (plms_instance.py:59-158) driving the full 1.228 B-parameter UNet, B=1, 64x64 latent, S=50 (inference.py:64),
    N=8 boxes, mis 0.36 (mis_step 18), alpha_type [0.8, 0, 0.2] with the first-conv swap at step 40, guidance 7.5.
    406 UNet forwards on CPU fp32.  Besides the final latent, the latent entering selected ``p_sample_plms`` calls is
    kept (call 162 = the merged latent, call 162+7, call 162+22 = first step with alpha 0) so a GPU mismatch can be
    located in the trajectory."""
    print(f"[golden] {tag}: full model S={S} N={n_inst} mis={mis}", flush=True)
    import time
    cfg = load_cfg("test_box.yaml", "full")
    model, gi, diffusion, schema, synth = build(cfg)
    from ldm.models.diffusion.plms_instance import PLMSSamplerInst
    g = torch.Generator().manual_seed(1234)
    bx = synth.random_boxes(n_inst, g)
    gb = synth.make_grounding_batch(1, bx, g)
    x = torch.randn(1, 4, 64, 64, generator=g)
    context = torch.randn(1, 77, 768, generator=g)
    uc = torch.randn(1, 77, 768, generator=g)
    grounding = gi.prepare(gb)
    patch_first_conv(model, synth.synth_first_conv_sd())
    sampler = PLMSSamplerInst(diffusion, model, alpha_generator_func=partial(ref_alpha_generator, type=list(alpha_type)),
                              set_alpha_scale=ref_set_alpha_scale, mis=mis)
    inputs = [dict(x=x.clone(), timesteps=None, context=context, grounding_input=grounding)]
    for i in range(n_inst):
        ctx_i = torch.randn(1, 77, 768, generator=g)
        inputs.append(dict(x=x.clone(), timesteps=None, context=ctx_i, grounding_input=gi.prepare(synth.instance_batch(gb, i))))
    mis_step = int(S * mis)
    n_phase1 = (n_inst + 1) * mis_step
    keep = {n_phase1: "merged", n_phase1 + 7: "phase2_7", n_phase1 + (int(alpha_type[0] * S) - mis_step): "alpha0_first"}
    for j in range(n_inst + 1):
        keep[j * mis_step + mis_step - 1] = f"traj{j}_last_in"          # latent entering the last phase-1 step of input j
    marks = {}
    calls = [0]
    t0 = time.time()
    inner = sampler.p_sample_plms

    def traced(input, *a, **k):
        if calls[0] in keep:
            marks[keep[calls[0]]] = input["x"].clone()
        calls[0] += 1
        if calls[0] % 10 == 0:
            print(f"   call {calls[0]}  {time.time() - t0:.0f}s", flush=True)
        return inner(input, *a, **k)
    sampler.p_sample_plms = traced
    out = dict(meta=dict(tag=tag, cfg="test_box.yaml", variant="full", alpha_type=list(alpha_type), latent=64,
                         n_boxes=n_inst, batch=1, boxes="rand", with_scribbles=False, with_polygons=False, with_segs=False,
                         S=S, mis=mis, n_inst=n_inst, seg_size=512, x_fp=fp(x), ctx_fp=fp(context), calls=None))
    out["mis"] = sampler.sample(S=S, shape=tuple(x.shape), input=inputs, uc=uc, guidance_scale=7.5).clone()
    out["meta"]["calls"] = calls[0]
    out["marks"] = marks
    out["plms_timesteps"] = [int(v) for v in sampler.ddim_timesteps]
    os.chdir(REF)
    torch.save(out, os.path.join(GOLD, f"{tag}.pt"))
    print(f"[golden] {tag}: {calls[0]} sampler calls, {time.time() - t0:.0f}s, final std {float(out['mis'].std()):.4f}")


# reduced VAE for fast tests: 64 base channels, 3 levels (x4 upsampling), the same block structure (mid attention incl.)
VAE_VARIANTS = {
    "full": {},
    "tiny": dict(ch=64, ch_mult=[1, 2, 2], num_res_blocks=1),
}


@torch.no_grad()
def gen_vae_case(tag: str, variant: str, latent: int, batch: int):
    """AutoencoderKL.decode of the UNMODIFIED reference (ldm/models/autoencoder.py:32-36) on key-name-seeded weights."""
    print(f"[golden] {tag}: autoencoder variant={variant} latent={latent}", flush=True)
    cfg = yaml.safe_load(open(os.path.join(REF, "configs", "test_box.yaml")))["autoencoder"]
    cfg["params"]["ddconfig"].update(VAE_VARIANTS[variant])
    from ldm.util import instantiate_from_config
    import importlib.util
    spec = importlib.util.spec_from_file_location("idf_synth", os.path.join(REPO, "instancediffusion_amd", "synth.py"))
    synth = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(synth)
    ae = instantiate_from_config(cfg).eval()
    schema = {k: tuple(v.shape) for k, v in ae.state_dict().items()}
    ae.load_state_dict(synth.synth_state_dict(schema, salt=7), strict=True)
    g = torch.Generator().manual_seed(4321)
    z = torch.randn(batch, 4, latent, latent, generator=g) * 0.18215 * 4.0      # latents of a trained model: std ~ 0.18*O(1..5)
    probes = {}
    dec = ae.decoder
    hooks = [dec.conv_in.register_forward_hook(lambda m, a, o: probes.__setitem__("conv_in", fp(o))),
             dec.mid.attn_1.register_forward_hook(lambda m, a, o: probes.__setitem__("mid.attn_1", fp(o))),
             dec.mid.block_2.register_forward_hook(lambda m, a, o: probes.__setitem__("mid.block_2", fp(o)))]
    for i in range(dec.num_resolutions):
        hooks.append(dec.up[i].block[-1].register_forward_hook(lambda m, a, o, i=i: probes.__setitem__(f"up.{i}", fp(o))))
    img = ae.decode(z)
    for h in hooks:
        h.remove()
    out = dict(meta=dict(tag=tag, variant=variant, latent=latent, batch=batch, z_fp=fp(z), salt=7), probes=probes)
    if img.numel() > 400000:        # full-size image (3 MB): keep an 8x8 average-pooled digest + moments instead
        out["img_pool8"] = torch.nn.functional.avg_pool2d(img, 8).clone()
        out["img_fp"] = fp(img)
    else:
        out["img"] = img.clone()
    torch.save(out, os.path.join(GOLD, f"{tag}.pt"))
    return schema


def gen_input_case():
    """``prepare_batch`` / ``prepare_instance_meta`` of the UNMODIFIED reference ``utils/input.py`` (imported with stub
    modules for its un-installable imports: pycocotools, dataset.*, and ``utils.model.get_clip_feature`` replaced by a
    deterministic hash embedding -- the same stand-in the mirror's test uses).  Large tensors are stored as digests."""
    import hashlib
    import importlib

    def text_embedding(s, dim=768, rows=1):
        g = torch.Generator().manual_seed(int.from_bytes(hashlib.sha256(s.encode()).digest()[:7], "little"))
        return torch.randn(rows, dim, generator=g)

    def batch_to_device(batch, device):
        return batch
    stubs = {"pycocotools": types.ModuleType("pycocotools"), "pycocotools.mask": types.ModuleType("pycocotools.mask"),
             "dataset": types.ModuleType("dataset"), "dataset.jsondataset": types.ModuleType("dataset.jsondataset"),
             "dataset.decode_item": types.ModuleType("dataset.decode_item"), "utils": types.ModuleType("utils"),
             "utils.model": types.ModuleType("utils.model")}
    stubs["pycocotools"].mask = stubs["pycocotools.mask"]
    stubs["dataset.jsondataset"].batch_to_device = batch_to_device
    stubs["dataset.decode_item"].sample_random_points_from_mask = None
    stubs["dataset.decode_item"].sample_sparse_points_from_mask = None
    stubs["utils"].__path__ = [os.path.join(REF, "utils")]
    stubs["utils.model"].get_clip_feature = lambda model, processor, phrase, is_image=False: (
        None if phrase is None else text_embedding("pooled:" + phrase))
    saved = {k: sys.modules.get(k) for k in stubs}
    sys.modules.update(stubs)
    try:
        ref_input = importlib.import_module("utils.input")
        data = json.load(open(os.path.join(REF, "demos", "demo_cat_dog_robin.json")))
        W, H = data["width"], data["height"]
        locations = [[b[0] / W, b[1] / H, (b[0] + b[2]) / W, (b[1] + b[3]) / H] for b in (a["bbox"] for a in data["annos"])]
        n = len(locations)
        segs = np.zeros((n, 512, 512), dtype=np.float32)
        for i, l in enumerate(locations[:2]):                      # two box-filled instance masks, two empty
            segs[i, int(l[1] * 512):int(l[3] * 512), int(l[0] * 512):int(l[2] * 512)] = 1
        rng = np.random.RandomState(5)
        meta = dict(ckpt=None, prompt=data["caption"], phrases=[a["caption"] for a in data["annos"]],
                    polygons=[list(rng.rand(512).astype(np.float32)) for _ in range(n)],
                    scribbles=[list(rng.rand(40).astype(np.float32)) for _ in range(n)], segs=segs, locations=locations,
                    points=[[(l[0] + l[2]) / 2, (l[1] + l[3]) / 2] for l in locations], alpha_type=[0.8, 0.0, 0.2],
                    save_folder_name="x", text_mask=[1, 0, 1, 1])
        meta["instance_meta"] = [ref_input.prepare_instance_meta(meta, i) for i in range(n)]
        out = ref_input.prepare_batch(meta, batch=2, max_objs=30, model=None, processor=None, image_size=64,
                                      use_masked_att=True, device="cpu")

        def digest(d):
            r = {}
            for k, v in d.items():
                if not torch.is_tensor(v):
                    continue
                if v.numel() > 200000:
                    r[k] = dict(shape=list(v.shape), sums=v.reshape(v.shape[0], v.shape[1], -1).sum(-1).clone(),
                                pooled=torch.nn.functional.avg_pool2d(v[:1, :4], v.shape[-1] // 16).clone())
                else:                                              # batch rows are copies; slots >= 6 are zero padding
                    assert bool((v == v[:1]).all())
                    r[k] = dict(shape=list(v.shape), head=v[0, :6].clone(), rest_abs_sum=float(v[0, 6:].abs().sum()))
            return r
        gold = dict(meta=dict(n=n, text_mask=[1, 0, 1, 1], seed=5, phrases=list(meta["phrases"]), locations=locations),
                    main=digest(out),
                    inst=[digest(d) for d in out["instance_meta"]],
                    instance_meta_keys=sorted(meta["instance_meta"][0].keys()),
                    complete_mask=[ref_input.complete_mask(None, 5), ref_input.complete_mask(0.5, 5),
                                   ref_input.complete_mask([0, 1], 5)],
                    convert_points=ref_input.convert_points([100.0, 600.0, 800.0, 20.0], dict(width=768, height=512)))
        torch.save(gold, os.path.join(GOLD, "prepare_batch.pt"))
        print("[golden] prepare_batch:", {k: v["shape"] for k, v in gold["main"].items()})
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
        sys.modules.pop("utils.input", None)


@torch.no_grad()
def gen_masked_case(tag="tiny_masked_att"):
    """Masked gated self-attention (attention.py:187-255): the UNMODIFIED reference UNet built with
    ``efficient_attention=False`` and fed ``grounding_input['att_masks']`` (utils/input.py:34-37 layout), tiny width,
    64x64 latent (the only resolution at which the reference applies the mask)."""
    print(f"[golden] {tag}", flush=True)
    cfg = load_cfg("test_box.yaml", "tiny")
    cfg["model"]["params"]["efficient_attention"] = False
    model, gi, diffusion, schema, synth = build(cfg)
    g = torch.Generator().manual_seed(1234)
    bx = synth.random_boxes(3, g)
    gb = synth.make_grounding_batch(1, bx, g)
    att = torch.zeros(30, 64, 64)
    for i in range(3):                                                   # utils/input.py:34-37 (x on dim 0, y on dim 1)
        b = bx[i].tolist()
        x1, y1, x2, y2 = (int(np.round(b[0] * 64)), int(np.round(b[1] * 64)), int(np.round(b[2] * 64)), int(np.round(b[3] * 64)))
        att[i][x1:x2, y1:y2] = 1
    gb["att_masks"] = att.unsqueeze(0)
    x = torch.randn(1, 4, 64, 64, generator=g)
    context = torch.randn(1, 77, 768, generator=g)
    t = torch.full((1,), 981, dtype=torch.long)
    grounding = gi.prepare(gb, return_att_masks=True)
    eps_m = model(dict(x=x, timesteps=t, context=context, grounding_input=grounding)).clone()
    plain = {k: v for k, v in grounding.items() if k != "att_masks"}
    eps_u = model(dict(x=x, timesteps=t, context=context, grounding_input=plain)).clone()
    null = gi.get_null_input()
    assert "att_masks" in null and float(null["att_masks"].sum()) == 0
    eps_n = model(dict(x=x, timesteps=t, context=context, grounding_input=null)).clone()
    out = dict(meta=dict(tag=tag, cfg="test_box.yaml", variant="tiny", latent=64, n_boxes=3, batch=1, x_fp=fp(x)),
               eps_masked=eps_m, eps_unmasked=eps_u, eps_null=eps_n, att_fp=fp(att))
    print("   masked vs unmasked rel diff:", float((eps_m - eps_u).norm() / eps_u.norm()))
    torch.save(out, os.path.join(GOLD, f"{tag}.pt"))


# ---- f-3: checkpoint loader pin (utils/checkpoint.py:199-249) ---------------------------------------------------------
CKPT_TEXT_ENCODER = dict(target="torch.nn.Linear", params=dict(in_features=4, out_features=3))
CKPT_SALTS = dict(model=11, ema=12, autoencoder=7, text_encoder=13)


def ckpt_digest(sd):
    """{key: float64 sum} of a state dict -- small, order-free, sensitive to any swapped / missing tensor."""
    return {k: float(v.double().sum()) for k, v in sd.items()}


@torch.no_grad()
def gen_ckpt_case(tag="ckpt_tiny"):
    """``save_ckpt`` -> ``load_model_ckpt`` of the UNMODIFIED reference ``utils/checkpoint.py`` on a reduced-width model
    set (tiny UNet, tiny VAE, a 4->3 Linear standing in for the CLIP text encoder whose weights cannot be downloaded).
    Process-local stubs for its un-installable imports (torchvision, omegaconf, tensorboard, dataset.*); ``torch.load``
    is called the way torch < 2.6 did (weights_only=False), which is what the reference was written against.
    The golden keeps: the key set of the checkpoint dict the reference's ``save_ckpt`` wrote, per-tensor digests of the
    modules its ``load_model_ckpt`` returned (ema present / absent; config from the checkpoint / from --test_config),
    and the returned config."""
    import importlib
    import tempfile
    print(f"[golden] {tag}", flush=True)

    # Minimal stand-ins with OmegaConf's pickled layout: container nodes keep children in ``_content`` (dict / list of
    # nodes), leaves are value nodes with ``_val``; mapping access unwraps leaves (what the reference code relies on:
    # ``"target" in cfg``, ``cfg["target"]``, ``cfg.get("params", dict())``, ``**params``).
    class AnyNode:
        def __init__(self, val):
            self.__dict__.update(_metadata=None, _parent=None, _val=val)

    def wrap(v):
        if isinstance(v, dict):
            return DictConfig(v)
        if isinstance(v, (list, tuple)):
            return ListConfig(v)
        return AnyNode(v)

    def unwrap(n):
        return n._val if isinstance(n, AnyNode) else n

    class DictConfig:
        def __init__(self, content):
            self.__dict__.update(_metadata=None, _parent=None, _flags_cache=None,
                                 _content={k: wrap(v) for k, v in content.items()})

        def __contains__(self, k):
            return k in self._content

        def __getitem__(self, k):
            return unwrap(self._content[k])

        def get(self, k, default=None):
            return unwrap(self._content[k]) if k in self._content else default

        def keys(self):
            return self._content.keys()

        def items(self):
            return [(k, unwrap(v)) for k, v in self._content.items()]

        def __iter__(self):
            return iter(self._content)

        def __len__(self):
            return len(self._content)

    class ListConfig:
        def __init__(self, content):
            self.__dict__.update(_metadata=None, _parent=None, _flags_cache=None, _content=[wrap(v) for v in content])

        def __iter__(self):
            return iter(unwrap(v) for v in self._content)

        def __len__(self):
            return len(self._content)

        def __getitem__(self, i):
            return unwrap(self._content[i])

    def to_plain(n):
        if isinstance(n, DictConfig):
            return {k: to_plain(v) for k, v in n._content.items()}
        if isinstance(n, ListConfig):
            return [to_plain(v) for v in n._content]
        if isinstance(n, AnyNode):
            return n._val
        if isinstance(n, dict):
            return {k: to_plain(v) for k, v in n.items()}
        return n
    for cls, mod in ((DictConfig, "omegaconf.dictconfig"), (ListConfig, "omegaconf.listconfig"), (AnyNode, "omegaconf.nodes")):
        cls.__module__ = mod
        cls.__qualname__ = cls.__name__
    omegaconf = types.ModuleType("omegaconf")
    oc_dict = types.ModuleType("omegaconf.dictconfig")
    oc_list = types.ModuleType("omegaconf.listconfig")
    oc_nodes = types.ModuleType("omegaconf.nodes")
    oc_dict.DictConfig, oc_list.ListConfig, oc_nodes.AnyNode = DictConfig, ListConfig, AnyNode

    class OmegaConf:
        @staticmethod
        def load(path):
            return DictConfig(yaml.safe_load(open(path)))
    omegaconf.OmegaConf = OmegaConf
    omegaconf.DictConfig = DictConfig
    tb = types.ModuleType("torch.utils.tensorboard")
    tb.SummaryWriter = object
    jd = types.ModuleType("dataset.jsondataset")
    jd.sub_batch = jd.batch_to_device = None
    utils_pkg = types.ModuleType("utils")
    utils_pkg.__path__ = [os.path.join(REF, "utils")]
    stubs = {"torchvision": types.ModuleType("torchvision"), "omegaconf": omegaconf, "omegaconf.dictconfig": oc_dict,
             "omegaconf.listconfig": oc_list, "omegaconf.nodes": oc_nodes,
             "torch.utils.tensorboard": tb, "dataset": types.ModuleType("dataset"), "dataset.jsondataset": jd,
             "utils": utils_pkg}
    saved = {k: sys.modules.get(k) for k in stubs}
    sys.modules.update(stubs)
    real_load = torch.load
    torch.load = lambda *a, **k: real_load(*a, **{**k, "weights_only": k.get("weights_only", False) or False})
    try:
        ref_ckpt = importlib.import_module("utils.checkpoint")
        cfg = load_cfg("test_box.yaml", "tiny")
        cfg["autoencoder"]["params"]["ddconfig"].update(VAE_VARIANTS["tiny"])
        cfg["text_encoder"] = dict(CKPT_TEXT_ENCODER)
        model, gi, diffusion, schema, synth = build(cfg, salt=CKPT_SALTS["model"])
        from ldm.util import instantiate_from_config
        ae = instantiate_from_config(cfg["autoencoder"]).eval()
        ae_schema = {k: tuple(v.shape) for k, v in ae.state_dict().items()}
        ae.load_state_dict(synth.synth_state_dict(ae_schema, CKPT_SALTS["autoencoder"]))
        te = instantiate_from_config(cfg["text_encoder"])
        te.load_state_dict(synth.synth_state_dict({k: tuple(v.shape) for k, v in te.state_dict().items()},
                                                  CKPT_SALTS["text_encoder"]))
        ema = instantiate_from_config(cfg["model"]).eval()
        ema.load_state_dict(synth.synth_state_dict(schema, CKPT_SALTS["ema"]))
        opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=0.1)
        sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda i: 1.0)
        gold = dict(meta=dict(tag=tag, salts=dict(CKPT_SALTS), text_encoder=dict(CKPT_TEXT_ENCODER)), cases={})
        for with_ema in (True, False):
            for use_yaml in (False, True):
                d = tempfile.mkdtemp()
                run_cfg = types.SimpleNamespace(distributed=False, enable_ema=with_ema)
                ref_ckpt.save_ckpt(run_cfg, model, te, ae, opt, sched, vars(DictConfig(cfg)), diffusion, ema, 41, d)   # utils/misc.py:255
                path = os.path.join(d, "checkpoint_latest.pth")
                saved_keys = sorted(real_load(path, map_location="cpu", weights_only=False).keys())
                args = types.SimpleNamespace(test_config="")
                if use_yaml:
                    args.test_config = os.path.join(d, "cfg.yaml")
                    yaml.safe_dump(cfg, open(args.test_config, "w"))
                m, a, t, df, c = ref_ckpt.load_model_ckpt(path, args, "cpu")
                gold["cases"][(with_ema, use_yaml)] = dict(
                    saved_keys=saved_keys, model=ckpt_digest(m.state_dict()), autoencoder=ckpt_digest(a.state_dict()),
                    text_encoder=ckpt_digest(t.state_dict()), diffusion=ckpt_digest(df.state_dict()),
                    training=[m.training, a.training, t.training], config=json.loads(json.dumps(to_plain(c))),
                    config_type=type(c).__name__)
                os.remove(path)
        gold["cfg"] = json.loads(json.dumps(cfg))
        torch.save(gold, os.path.join(GOLD, f"{tag}.pt"))
        print("[golden] ckpt cases:", list(gold["cases"]))
    finally:
        torch.load = real_load
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
        sys.modules.pop("utils.checkpoint", None)
        sys.modules.pop("utils.dist", None)


# ---- f-3: CLIP text-encoder forward pin (ldm/modules/encoders/modules.py:144-172) --------------------------------------
# the hub's config.json of openai/clip-vit-large-patch14 (text part), which ``from_pretrained`` would have fetched
CLIP_HUB_TEXT_CONFIG = dict(vocab_size=49408, hidden_size=768, intermediate_size=3072, projection_dim=768,
                            num_hidden_layers=12, num_attention_heads=12, max_position_embeddings=77, hidden_act="quick_gelu",
                            layer_norm_eps=1e-5, pad_token_id=1, bos_token_id=0, eos_token_id=2)
CLIP_SALT = 21


def clip_input_ids():
    """Two fixed CLIP-BPE-shaped id rows (<|startoftext|> 49406, word ids, <|endoftext|> 49407, padded with 49407 as the
    openai tokenizer pads): the tokenizer itself cannot run offline (no vocabulary files), the transformer can."""
    g = torch.Generator().manual_seed(77)
    ids = torch.full((2, 77), 49407, dtype=torch.long)
    for r, n in enumerate((9, 40)):
        ids[r, 0] = 49406
        ids[r, 1:1 + n] = torch.randint(320, 49000, (n,), generator=g)
    return ids


@torch.no_grad()
def gen_clip_case(tag="clip_text"):
    """The UNMODIFIED reference ``FrozenCLIPEmbedder`` (its ``forward`` / ``encode`` / ``freeze`` code paths as written) on
    key-name-seeded weights and fixed ``input_ids``.  Process-local shims: ``clip`` / ``kornia`` stub modules (imported at the
    top of encoders/modules.py, unused by this class), ``CLIPTextModel.from_pretrained`` -> construction from the hub's
    config (no network), ``CLIPTokenizer.from_pretrained`` -> a callable returning the fixed ids."""
    import importlib
    import importlib.util
    import transformers
    print(f"[golden] {tag}: FrozenCLIPEmbedder forward (transformers {transformers.__version__})", flush=True)
    ids = clip_input_ids()

    class FixedTokenizer:
        def __call__(self, text, **kw):
            assert kw.get("max_length") == 77 and kw.get("padding") == "max_length" and kw.get("truncation") is True
            return {"input_ids": ids[:len(text)]}
    stubs = {"clip": types.ModuleType("clip"), "kornia": types.ModuleType("kornia")}
    saved = {k: sys.modules.get(k) for k in stubs}
    sys.modules.update(stubs)
    real_model_fp, real_tok_fp = transformers.CLIPTextModel.from_pretrained, transformers.CLIPTokenizer.from_pretrained
    transformers.CLIPTextModel.from_pretrained = classmethod(
        lambda cls, version, *a, **k: cls(transformers.CLIPTextConfig(**CLIP_HUB_TEXT_CONFIG)))
    transformers.CLIPTokenizer.from_pretrained = classmethod(lambda cls, version, *a, **k: FixedTokenizer())
    try:
        mod = importlib.import_module("ldm.modules.encoders.modules")
        enc = mod.FrozenCLIPEmbedder(device="cpu")
        spec = importlib.util.spec_from_file_location("idf_synth", os.path.join(REPO, "instancediffusion_amd", "synth.py"))
        synth = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(synth)
        schema = {k: tuple(v.shape) for k, v in enc.state_dict().items() if v.is_floating_point()}
        sd = synth.synth_state_dict(schema, CLIP_SALT)
        missing = enc.load_state_dict(sd, strict=False)
        assert not missing.unexpected_keys and all("position_ids" in k for k in missing.missing_keys), missing
        z, pooled = enc.encode(["a", "b"], return_pooler_output=True)
        z_only = enc(["a", "b"])
        assert torch.equal(z, z_only)
        out = dict(meta=dict(tag=tag, salt=CLIP_SALT, transformers=transformers.__version__, hub_config=dict(CLIP_HUB_TEXT_CONFIG),
                             key_layout=sorted(schema)[:3], n_keys=len(schema)),
                   input_ids=ids, last_hidden_state=z.clone(), pooler_output=pooled.clone(),
                   schema={k: list(v) for k, v in schema.items()})
        torch.save(out, os.path.join(GOLD, f"{tag}.pt"))
        print(f"[golden] {tag}: z {tuple(z.shape)} std {float(z.std()):.4f}, pooled {tuple(pooled.shape)}; "
              f"{sum(int(np.prod(v)) for v in schema.values())} parameters")
    finally:
        transformers.CLIPTextModel.from_pretrained, transformers.CLIPTokenizer.from_pretrained = real_model_fp, real_tok_fp
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
        sys.modules.pop("ldm.modules.encoders.modules", None)


@torch.no_grad()
def gen_full_c2(tag="full_box_c2_s50", S=50, alpha_type=(0.8, 0.0, 0.2)):
    """BASELINE config 2 at full size: the 1.228 B-parameter UNet, the C1 demo boxes, ONE image, Multi-instance Sampler OFF --
    the unmodified reference ``PLMSSampler`` (plms.py:72-113), S = 50, CFG 7.5, alpha [0.8, 0, 0.2] (the first-conv swap at
    step 40 included): 102 full-size CPU forwards (~5 min on 8 cores)."""
    print(f"[golden] {tag}", flush=True)
    cfg = load_cfg("test_box.yaml", "full")
    model, gi, diffusion, schema, synth = build(cfg)
    from ldm.models.diffusion.plms import PLMSSampler
    g = torch.Generator().manual_seed(1234)
    bx = torch.tensor(synth.C1_BOXES)
    gb = synth.make_grounding_batch(1, bx, g)
    x = torch.randn(1, 4, 64, 64, generator=g)
    context = torch.randn(1, 77, 768, generator=g)
    uc = torch.randn(1, 77, 768, generator=g)
    grounding = gi.prepare(gb)
    patch_first_conv(model, synth.synth_first_conv_sd())
    sampler = PLMSSampler(diffusion, model, alpha_generator_func=partial(ref_alpha_generator, type=list(alpha_type)),
                          set_alpha_scale=ref_set_alpha_scale)
    inp = dict(x=x.clone(), timesteps=None, context=context, grounding_input=grounding)
    out = {"meta": dict(tag=tag, cfg="test_box.yaml", variant="full", alpha_type=list(alpha_type), latent=64, n_boxes=int(bx.shape[0]),
                        batch=1, boxes="c1", with_scribbles=False, with_polygons=False, with_segs=False, S=S, mis=0.0, n_inst=0,
                        seg_size=512, x_fp=fp(x), ctx_fp=fp(context))}
    out["plms"] = sampler.sample(S=S, shape=tuple(x.shape), input=inp, uc=uc, guidance_scale=7.5).clone()
    out["plms_timesteps"] = [int(v) for v in sampler.ddim_timesteps]
    os.chdir(REF)
    torch.save(out, os.path.join(GOLD, f"{tag}.pt"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="all")
    args = ap.parse_args()
    install_shims()
    os.makedirs(GOLD, exist_ok=True)
    torch.set_num_threads(os.cpu_count())
    if args.only in ("all", "tiny"):
        # tiny: the first-conv swap hard-codes 320 channels (openaimodel.py:478) -> alpha never reaches 0 here
        gen_case("tiny_box", "test_box.yaml", "tiny", 16, 3, 2, alpha_type=(1, 0, 0))
        gen_case("tiny_mask", "test_mask.yaml", "tiny", 16, 3, 1, with_polygons=True, with_segs=True, S=4, mis=0.5,
                 n_inst=1, alpha_type=(1, 0, 0))
        gen_case("tiny_point", "test_point.yaml", "tiny", 16, 3, 1, samplers=False)
        gen_case("tiny_scribble", "test_scribble.yaml", "tiny", 16, 3, 1, with_scribbles=True, with_polygons=True,
                 with_segs=True, samplers=False)
    if args.only in ("all", "mid"):
        gen_case("mid_box", "test_box.yaml", "mid", 16, 3, 2)
    if args.only in ("all", "full"):
        schema = gen_case("full_box_c1", "test_box.yaml", "full", 64, 4, 1, boxes="c1", samplers=False)
        json.dump({k: list(v) for k, v in schema.items()}, open(os.path.join(GOLD, "unet_schema.json"), "w"))
    if args.only in ("all", "s50"):
        # BASELINE headline trajectory shape (inference.py:64 steps=50, N=8 instances, mis 0.36) on the reduced variants
        gen_case("tiny_box_s50", "test_box.yaml", "tiny", 16, 8, 1, alpha_type=(1, 0, 0), S=50, mis=0.36, n_inst=8)
        gen_case("mid_box_s50", "test_box.yaml", "mid", 16, 8, 1, alpha_type=(0.8, 0.0, 0.2), S=50, mis=0.36, n_inst=8)
    if args.only in ("full_s50",):          # 406 full-model forwards on CPU (~20 min on 8 cores): not part of "all"
        gen_full_s50()
    if args.only in ("all", "c5"):
        # C5 (point / scribble conditioning): forwards + S=5 PLMS / MIS trajectories
        gen_case("tiny_point_s5", "test_point.yaml", "tiny", 16, 3, 1, alpha_type=(1, 0, 0))
        gen_case("tiny_scribble_s5", "test_scribble.yaml", "tiny", 16, 3, 1, with_scribbles=True, with_polygons=True,
                 with_segs=True, alpha_type=(1, 0, 0))
    if args.only in ("full_c5",):
        # C5 at its stated size (round 6): the full model, 64x64 latent, batch 4, N = 8 -- test_point.yaml (only the point tokens
        # live) and test_scribble.yaml (nothing dropped: live scribbles through the 768 + 1280 -> 3072 MLP, polygons, ConvNeXt
        # mask tokens); forwards only (cond / uncond / gate scale 0.3)
        gen_case("full_point_c5", "test_point.yaml", "full", 64, 8, 4, samplers=False)
        gen_case("full_scribble_c5", "test_scribble.yaml", "full", 64, 8, 4, with_scribbles=True, with_polygons=True,
                 with_segs=True, samplers=False)
    if args.only in ("full_c2",):           # 102 full-model forwards on CPU (~5 min on 8 cores): not part of "all"
        gen_full_c2()
    if args.only in ("all", "c4"):
        # C4 at its stated size: test_mask.yaml, 96x96 latent (768x768), 12 instance masks with segs + polygons
        gen_case("full_mask_c4", "test_mask.yaml", "full", 96, 12, 1, with_polygons=True, with_segs=True, samplers=False)
    if args.only in ("all", "vae"):
        gen_vae_case("vae_tiny", "tiny", 8, 2)
        schema = gen_vae_case("vae_full_16", "full", 16, 1)
        gen_vae_case("vae_full_64", "full", 64, 1)              # the reference's real call: 64x64 latent -> 512x512
        json.dump({k: list(v) for k, v in schema.items()}, open(os.path.join(GOLD, "vae_schema.json"), "w"))
    if args.only in ("all", "masked"):
        gen_masked_case()
    if args.only in ("all", "input"):
        gen_input_case()
    if args.only in ("all", "ckpt"):
        gen_ckpt_case()
    if args.only in ("all", "clip"):
        gen_clip_case()
    if args.only in ("all", "plms_mask"):
        gen_plms_mask_case()
    print("done")


if __name__ == "__main__":
    main()
