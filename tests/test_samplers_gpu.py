"""Sampler parity on a real MI355X: PLMSSampler / PLMSSamplerInst (HIP engine, hipGraph replay, batched CFG and
batched MIS trajectories) vs the unmodified reference's trajectories (goldens).

Stated tolerance: bf16 storage, CFG 7.5 amplifies per-forward eps noise (~1.5e-2 rel-RMS) -> per-trajectory latent
rel-RMS <= 8e-2 on these 5-step / 4-step trajectories (CPU bf16 emulation of the same engine gives 2.4e-2..4.2e-2).
"""
from functools import partial

import pytest
import torch

pytestmark = pytest.mark.gpu
TRAJ_TOL = 8e-2


def _setup(tag):
    from instancediffusion_amd import synth
    from instancediffusion_amd.host.diffusion import LatentDiffusion
    from grounding_input.text_grounding_tokinzer_input import GroundingNetInput
    from tests import cases
    from tests.test_engine_emulated import build_model
    gold = cases.load_golden(tag)
    meta = gold["meta"]
    cfg = cases.cfg_for(meta["cfg"], meta["variant"])
    inp = cases.build_inputs(meta)
    model = build_model(cfg)
    model.first_conv_sd_override = synth.synth_first_conv_sd()
    gi = GroundingNetInput()
    model.grounding_tokenizer_input = gi
    diffusion = LatentDiffusion(linear_start=0.00085, linear_end=0.012, timesteps=1000).cuda()
    return gold, meta, inp, model, gi, diffusion


def _cuda(d):
    return {k: v.cuda() for k, v in d.items()}


@pytest.mark.parametrize("tag", ["tiny_box", "mid_box"])
def test_plms_and_mis_match_reference(tag):
    from instancediffusion_amd import synth
    from instancediffusion_amd.host.alpha import alpha_generator, set_alpha_scale
    from instancediffusion_amd.host.samplers import PLMSSampler, PLMSSamplerInst
    from tests import cases
    gold, meta, inp, model, gi, diffusion = _setup(tag)
    ag = partial(alpha_generator, type=meta["alpha_type"])
    shape = tuple(inp["x"].shape)
    sampler = PLMSSampler(diffusion, model, alpha_generator_func=ag, set_alpha_scale=set_alpha_scale)
    i0 = dict(x=inp["x"].cuda(), timesteps=None, context=inp["context"].cuda(), grounding_input=gi.prepare(_cuda(inp["gb"])))
    out = sampler.sample(S=meta["S"], shape=shape, input=i0, uc=inp["uc"].cuda(), guidance_scale=7.5)
    err = cases.rel_rms(out.cpu(), gold["plms"])
    print(f"[parity] {tag} PLMS S={meta['S']} CFG7.5 latent rel-rms {err:.3e}")
    assert torch.isfinite(out).all() and err < TRAJ_TOL

    gold, meta, inp, model, gi, diffusion = _setup(tag)
    sampler = PLMSSamplerInst(diffusion, model, alpha_generator_func=ag, set_alpha_scale=set_alpha_scale, mis=meta["mis"])
    inputs = [dict(x=inp["x"].cuda(), timesteps=None, context=inp["context"].cuda(), grounding_input=gi.prepare(_cuda(inp["gb"])))]
    for i in range(meta["n_inst"]):
        inputs.append(dict(x=inp["x"].cuda(), timesteps=None, context=inp["inst_ctx"][i].cuda(),
                           grounding_input=gi.prepare(_cuda(synth.instance_batch(inp["gb"], i)))))
    gi.prepare(_cuda(inp["gb"]))
    out = sampler.sample(S=meta["S"], shape=shape, input=inputs, uc=inp["uc"].cuda(), guidance_scale=7.5)
    err = cases.rel_rms(out.cpu(), gold["mis"])
    print(f"[parity] {tag} MIS S={meta['S']} mis={meta['mis']} latent rel-rms {err:.3e}")
    assert torch.isfinite(out).all() and err < TRAJ_TOL
    out2 = sampler.sample(S=meta["S"], shape=shape, input=[dict(d, x=inp["x"].cuda()) for d in inputs],
                          uc=inp["uc"].cuda(), guidance_scale=7.5)
    # second call: first conv stays swapped (reference quirk) -> not comparable to the golden, but must be finite
    assert torch.isfinite(out2).all()
