"""Pin the CPU oracle (oracle/ref_cpu.py) against golden vectors produced by the UNMODIFIED reference
(oracle/make_golden.py, run in the builder container).  fp32 vs fp32 on the same torch build: tolerance 2e-4
relative-RMS / 1e-3 of max for single forwards (summation-order differences only), looser for chained samplers.
"""
import pytest
import torch

from instancediffusion_amd import synth
from oracle import ref_cpu
from tests import cases

FWD_TOL = 2e-4
TRAJ_TOL = 5e-3


def _setup(tag):
    gold = cases.load_golden(tag)
    meta = gold["meta"]
    cfg = cases.cfg_for(meta["cfg"], meta["variant"])
    sd = synth.synth_state_dict(cases.unet_schema(cfg))
    inp = cases.build_inputs(meta)
    # the regenerated inputs must be the ones the reference saw
    assert torch.allclose(inp["x"].flatten()[:32], meta["x_fp"]["head"])
    assert torch.allclose(inp["context"].flatten()[:32], meta["ctx_fp"]["head"])
    return gold, meta, cfg, sd, inp


def _check(a, b, tol, what):
    err = cases.rel_rms(a, b)
    assert err < tol, f"{what}: rel-rms {err:.3e} >= {tol}"
    assert float((a - b).abs().max()) < 10 * tol * float(b.abs().max()) + 1e-6, what


@pytest.mark.parametrize("tag", ["tiny_box", "tiny_mask", "tiny_point", "tiny_scribble", "mid_box"])
def test_forward_matches_reference(tag):
    gold, meta, cfg, sd, inp = _setup(tag)
    with torch.no_grad():
        g = ref_cpu.prepare_grounding(inp["gb"])
        objs, _ = ref_cpu.unifusion(sd, cfg, g)
        _check(objs[0, [0, 1, 29, 30, 59, 60, 90, 120, 183]], gold["objs_rows"], FWD_TOL, "objs rows")
        probes = {}
        eps = ref_cpu.unet_forward(sd, cfg, inp["x"], inp["t"], inp["context"], objs, probes=probes)
        _check(eps, gold["eps_cond"], FWD_TOL, "eps_cond")
        for k, fpv in gold["probes_cond"].items():
            assert abs(float(probes[k].mean()) - fpv["mean"]) < 1e-3 * (abs(fpv["mean"]) + fpv["std"]), k
            assert torch.allclose(probes[k].flatten()[:32], fpv["head"], rtol=2e-3, atol=2e-3 * fpv["absmax"]), k
        objs0, _ = ref_cpu.unifusion(sd, cfg, ref_cpu.null_grounding(g))
        _check(ref_cpu.unet_forward(sd, cfg, inp["x"], inp["t"], inp["uc"], objs0), gold["eps_uncond"], FWD_TOL,
               "eps_uncond (null grounding)")
        _check(ref_cpu.unet_forward(sd, cfg, inp["x"], inp["t"], inp["context"], objs, fuser_scale=0.3),
               gold["eps_scale03"], FWD_TOL, "eps with fuser scale 0.3")


# heavy = minutes of CPU oracle time: opt-in (IDF_FULL_CPU_SUITE=1); the default suite still pins the S = 50 / N = 8 trajectory
# shape on the reduced-width model and the C5 (point / scribble) trajectories
import os
heavy = pytest.mark.skipif(os.environ.get("IDF_FULL_CPU_SUITE") != "1", reason="minutes of CPU oracle time: set IDF_FULL_CPU_SUITE=1")


@pytest.mark.parametrize("tag", ["tiny_box", "tiny_mask", "mid_box", "tiny_point_s5", "tiny_scribble_s5", "tiny_box_s50",
                                 pytest.param("mid_box_s50", marks=heavy)])
def test_samplers_match_reference(tag):
    gold, meta, cfg, sd, inp = _setup(tag)
    with torch.no_grad():
        g = ref_cpu.prepare_grounding(inp["gb"])
        model = ref_cpu.OracleModel(sd, cfg, synth.synth_first_conv_sd())
        i0 = dict(x=inp["x"].clone(), timesteps=None, context=inp["context"], grounding_input=g)
        out = ref_cpu.plms_sample(model, meta["S"], i0, inp["uc"], 7.5, alpha_type=meta["alpha_type"])
        _check(out, gold["plms"], TRAJ_TOL, "PLMS trajectory")
        assert list(ref_cpu._schedule(meta["S"])[0]) == gold["plms_timesteps"]

        model = ref_cpu.OracleModel(sd, cfg, synth.synth_first_conv_sd())
        inputs = [dict(x=inp["x"].clone(), timesteps=None, context=inp["context"], grounding_input=g)]
        for i in range(meta["n_inst"]):
            gi = ref_cpu.prepare_grounding(synth.instance_batch(inp["gb"], i))
            inputs.append(dict(x=inp["x"].clone(), timesteps=None, context=inp["inst_ctx"][i], grounding_input=gi))
        out = ref_cpu.plms_sample_mis(model, meta["S"], inputs, inp["uc"], 7.5, meta["mis"],
                                      alpha_type=meta["alpha_type"])
        _check(out, gold["mis"], TRAJ_TOL, "MIS trajectory")


def test_plms_mask_blend_matches_reference():
    """PLMSSampler.sample(mask=, x0=): the inpainting blend of plms.py:99-104, pinned to the unmodified reference
    (golden ``tiny_box_plms_mask``: its q_sample noise draws are stored with the golden and replayed here)."""
    gold, meta, cfg, sd, inp = _setup("tiny_box_plms_mask")
    with torch.no_grad():
        model = ref_cpu.OracleModel(sd, cfg, synth.synth_first_conv_sd())
        i0 = dict(x=inp["x"].clone(), timesteps=None, context=inp["context"], grounding_input=ref_cpu.prepare_grounding(inp["gb"]))
        out = ref_cpu.plms_sample(model, meta["S"], i0, inp["uc"], 7.5, alpha_type=meta["alpha_type"], mask=gold["mask"],
                                  x0=gold["x0"], noises=gold["noises"])
    _check(out, gold["plms_masked"], TRAJ_TOL, "PLMS trajectory with the mask / x0 blend")
    # the blend matters: without it the trajectory ends somewhere else
    assert cases.rel_rms(out, cases.load_golden("tiny_box")["plms"]) > 0.1


@heavy
def test_full_model_headline_trajectory_matches_reference():
    """The headline trajectory on the headline model (golden ``full_box_s50``: the unmodified reference PLMSSamplerInst, full
    1.228 B-parameter UNet, 64x64 latent, S = 50, N = 8, mis 0.36, alpha [0.8, 0, 0.2]): 406 full-size CPU forwards of the
    oracle (~20 min on 8 cores) -- opt-in; the recorded result of one run is profiles/r03_oracle_full_s50.log."""
    gold, meta, cfg, sd, inp = _setup("full_box_s50")
    with torch.no_grad():
        g = ref_cpu.prepare_grounding(inp["gb"])
        model = ref_cpu.OracleModel(sd, cfg, synth.synth_first_conv_sd())
        inputs = [dict(x=inp["x"].clone(), timesteps=None, context=inp["context"], grounding_input=g)]
        for i in range(meta["n_inst"]):
            gi = ref_cpu.prepare_grounding(synth.instance_batch(inp["gb"], i))
            inputs.append(dict(x=inp["x"].clone(), timesteps=None, context=inp["inst_ctx"][i], grounding_input=gi))
        out = ref_cpu.plms_sample_mis(model, meta["S"], inputs, inp["uc"], 7.5, meta["mis"], alpha_type=meta["alpha_type"])
    print(f"[oracle] full_box_s50 MIS trajectory vs the reference golden: rel-rms {cases.rel_rms(out, gold['mis']):.3e}")
    _check(out, gold["mis"], TRAJ_TOL, "full-model MIS trajectory")


@heavy
def test_full_size_c4_forward_matches_reference():
    """BASELINE config 4 at its stated size: test_mask.yaml, 96x96 latent, 12 masks with segs + polygons (ConvNeXt live)."""
    gold, meta, cfg, sd, inp = _setup("full_mask_c4")
    with torch.no_grad():
        g = ref_cpu.prepare_grounding(inp["gb"])
        objs, _ = ref_cpu.unifusion(sd, cfg, g)
        _check(ref_cpu.unet_forward(sd, cfg, inp["x"], inp["t"], inp["context"], objs), gold["eps_cond"], FWD_TOL, "C4 eps_cond")


@heavy
def test_full_size_forward_matches_reference():
    """Full SD-1.5 InstanceDiffusion UNet (1.228 B params), C1 inputs (demo_cat_dog_robin boxes), 64x64 latent.  Opt-in since
    round 3: its 12 GB of fp32 weights + synthesis left the test process under memory reclaim in this VM, and every test after
    it ran 10-50x slower (the whole CPU suite 8 min -> more than an hour); the full-size oracle is pinned on the GPU box by
    smoke() and by the live-oracle tests, and the recorded opt-in runs are profiles/r03_oracle_full_s50.log (trajectory, 1.4e-6)."""
    gold, meta, cfg, sd, inp = _setup("full_box_c1")
    assert sum(v.numel() for v in sd.values()) == 1228333437
    with torch.no_grad():
        g = ref_cpu.prepare_grounding(inp["gb"])
        objs, _ = ref_cpu.unifusion(sd, cfg, g)
        eps = ref_cpu.unet_forward(sd, cfg, inp["x"], inp["t"], inp["context"], objs)
    _check(eps, gold["eps_cond"], FWD_TOL, "full eps_cond")


def test_scaleu_4bin_identity():
    """Fourier_filter(x,1,s) == x + (s-1)*lowfreq_4bin(x)  (the identity the HIP ScaleU kernel relies on)."""
    g = torch.Generator().manual_seed(0)
    for (h, w) in [(8, 8), (16, 16), (12, 12), (24, 48), (64, 64), (96, 96)]:
        x = torch.randn(2, 5, h, w, generator=g)
        s = torch.tensor([1.37])
        ref = ref_cpu.fourier_filter(x, 1, s)
        mine = x + (s - 1) * ref_cpu.lowfreq_4bin(x)
        assert float((ref - mine).abs().max()) < 2e-5, (h, w)


def test_alpha_generator_and_schedule():
    assert ref_cpu.alpha_generator(50, [0.8, 0.0, 0.2]) == [1] * 40 + [0] * 10
    steps, a, a_prev = ref_cpu._schedule(5)
    assert list(steps) == [1, 201, 401, 601, 801]
    assert a_prev[0] == pytest.approx(float(torch.tensor(ref_cpu.alphas_cumprod(), dtype=torch.float32)[0]))
