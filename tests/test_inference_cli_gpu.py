"""``inference.py`` end to end on a real MI355X (the entry point was covered by a round-1 log only).

For each conditioning modality the demo-JSON format carries (reference inference.py:189-297) -- boxes, points, scribbles --
the CLI mirror runs demo JSON -> ``meta`` -> ``prepare_batch`` (+ ``prepare_instance_meta`` per instance under the Multi-instance
Sampler) -> sampler -> VAE decode -> PNGs with ``--synthetic_weights`` (the full 1.228 B-parameter UNet, key-seeded weights; no
trained weights exist offline), and the final latent is compared with the CPU oracle (``oracle/ref_cpu.py``: the restatement of
the reference's samplers / UNet, pinned to the reference's goldens) run on the same meta, the same starting noise and the same
weights.

The oracle's side of that comparison is a committed fixture, ``tests/golden/cli_latents.pt`` (``python tests/make_cli_latents.py``:
this module's ``_oracle_latent`` run once per case -- 12 to 28 full-size CPU forwards each, which on the GPU box's host cores was
a third of the whole ``-m gpu`` suite's wall time, round 6).  ``IDF_CLI_LIVE_ORACLE=1`` runs the oracle live instead, as rounds
5-6 did.
"""
import json
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _oracle_latent(cfg_name, input_json, steps, mis, alpha, seed, negative_prompt):
    """The same pipeline as inference.main() on the CPU oracle; returns (latent, n_forward)."""
    import inference
    from instancediffusion_amd import synth
    from instancediffusion_amd.host.input import meta_from_demo_json, prepare_batch, prepare_instance_meta
    from oracle import ref_cpu
    from tests import cases
    cfg = cases.cfg_for(cfg_name, "full")
    schema = cases.unet_schema(cfg)
    sd = synth.synth_state_dict(schema)
    om = ref_cpu.OracleModel(sd, cfg, synth.synth_first_conv_sd())
    data = json.load(open(os.path.join(REPO, input_json)))
    meta = meta_from_demo_json(data, alpha)
    enc = inference.SyntheticTextEncoder()
    torch.manual_seed(seed)
    x0 = torch.randn(1, 4, 64, 64)

    def grounding(m):
        batch = prepare_batch(m, batch=1, max_objs=inference.MAX_OBJS, model=enc, processor=None, image_size=64, device="cpu")
        return ref_cpu.prepare_grounding(batch)

    inp = dict(x=x0.clone(), timesteps=None, context=enc.encode([meta["prompt"]]), grounding_input=grounding(meta))
    uc = enc.encode([negative_prompt])
    with torch.no_grad():
        if mis > 0:
            inputs = [inp]
            for i in range(len(meta["phrases"])):
                mi = prepare_instance_meta(meta, i)
                inputs.append(dict(x=x0.clone(), timesteps=None, context=enc.encode([mi["prompt"]]), grounding_input=grounding(mi)))
            lat = ref_cpu.plms_sample_mis(om, steps, inputs, uc, 7.5, mis, alpha_type=meta["alpha_type"])
        else:
            lat = ref_cpu.plms_sample(om, steps, inp, uc, 7.5, alpha_type=meta["alpha_type"])
    return lat, om.n_forward


CASES = [
    ("boxes + Multi-instance Sampler", "test_box.yaml", "demos/demo_four_boxes.json", 0.36),
    ("points", "test_point.yaml", "demos/demo_points.json", 0.0),
    ("scribbles", "test_scribble.yaml", "demos/demo_scribbles.json", 0.0),
]
STEPS, ALPHA, SEED = 5, 0.8, 3
DEFAULT_NEG = ("longbody, lowres, bad anatomy, bad hands, missing fingers, extra digit, fewer digits, cropped, worst quality, "
               "low quality")                      # inference.py's --negative_prompt default (reference inference.py:171)
FIXTURE = os.path.join(REPO, "tests", "golden", "cli_latents.pt")


def _want(cfg_name, input_json, mis):
    """(oracle latent, oracle forwards, source)"""
    if os.environ.get("IDF_CLI_LIVE_ORACLE") == "1":
        return _oracle_latent(cfg_name, input_json, STEPS, mis, ALPHA, SEED, DEFAULT_NEG) + ("run live",)
    fx = torch.load(FIXTURE)
    assert (fx["steps"], fx["alpha"], fx["seed"], fx["negative_prompt"]) == (STEPS, ALPHA, SEED, DEFAULT_NEG), \
        "tests/golden/cli_latents.pt was made for other settings: re-run tests/make_cli_latents.py"
    case = fx["cases"][cfg_name]
    assert case["input_json"] == input_json and case["mis"] == mis
    return case["latent"], case["n_forward"], "fixture"


@pytest.mark.parametrize("name,cfg_name,input_json,mis", CASES)
def test_inference_cli_end_to_end_matches_oracle(tmp_path, monkeypatch, capsys, name, cfg_name, input_json, mis):
    import inference
    from PIL import Image
    from tests import cases
    steps, alpha, seed = STEPS, ALPHA, SEED
    out_dir = tmp_path / "OUT"
    argv = ["inference.py", "--synthetic_weights", "--num_images", "1", "--steps", str(steps), "--mis", str(mis), "--alpha", str(alpha),
            "--seed", str(seed), "--input_json", os.path.join(REPO, input_json), "--test_config", os.path.join(REPO, "configs", cfg_name),
            "--output", str(out_dir), "--save_latents", "--dtype", "bf16"]
    monkeypatch.setattr(sys, "argv", argv)
    monkeypatch.chdir(REPO)
    inference.main()
    folder = out_dir / f"gc7.5-seed{seed}-alpha{alpha}"
    pngs = sorted(p for p in os.listdir(folder) if p.endswith(".png"))
    assert len(pngs) == 1
    img = Image.open(folder / pngs[0])
    assert img.size == (512, 512) and img.mode == "RGB"
    import numpy as np
    px = torch.from_numpy(np.asarray(img, dtype=np.float32))
    assert float(px.std()) > 1.0, "a constant image means the decode path did nothing"
    saved = torch.load(folder / "latents.pt")
    lat = saved["latents"].float()
    assert tuple(lat.shape) == (1, 4, 64, 64) and torch.isfinite(lat).all()
    want, n_fwd, src = _want(cfg_name, input_json, mis)
    err = cases.rel_rms(lat, want)
    print(f"[parity] inference.py end to end, {name} ({cfg_name}, S={steps}, mis={mis}): latent rel-rms {err:.3e} vs the CPU oracle "
          f"({n_fwd} oracle forwards, {src}; tol 5e-2), image {img.size}")
    assert err < 5e-2
