"""SURVEY.md §8 f-1 (no GPU): the ``utils.input`` mirror against goldens of the UNMODIFIED reference ``prepare_batch`` /
``prepare_instance_meta`` / ``complete_mask`` / ``convert_points`` (oracle/make_golden.py:gen_input_case), the demo-JSON
parser against the reference demo's boxes, and the broadcast (stride-0) layout of the large tensors."""
import hashlib
import json
import os

import numpy as np
import torch

from tests import cases

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class HashEncoder:
    """The deterministic stand-in make_golden.py gave the reference as ``get_clip_feature``."""

    def pooled(self, s):
        g = torch.Generator().manual_seed(int.from_bytes(hashlib.sha256(("pooled:" + s).encode()).digest()[:7], "little"))
        return torch.randn(1, 768, generator=g)[0]


def _meta():
    from utils.input import prepare_instance_meta
    data = json.load(open(os.path.join(REPO, "demos", "demo_four_boxes.json")))
    ref_caps = None
    W, H = data["width"], data["height"]
    locations = [[b[0] / W, b[1] / H, (b[0] + b[2]) / W, (b[1] + b[3]) / H] for b in (a["bbox"] for a in data["annos"])]
    n = len(locations)
    segs = np.zeros((n, 512, 512), dtype=np.float32)
    for i, l in enumerate(locations[:2]):
        segs[i, int(l[1] * 512):int(l[3] * 512), int(l[0] * 512):int(l[2] * 512)] = 1
    rng = np.random.RandomState(5)
    meta = dict(ckpt=None, prompt="p", phrases=None, polygons=[list(rng.rand(512).astype(np.float32)) for _ in range(n)],
                scribbles=[list(rng.rand(40).astype(np.float32)) for _ in range(n)], segs=segs, locations=locations,
                points=[[(l[0] + l[2]) / 2, (l[1] + l[3]) / 2] for l in locations], alpha_type=[0.8, 0.0, 0.2],
                save_folder_name="x", text_mask=[1, 0, 1, 1])
    return meta, prepare_instance_meta, ref_caps


def _check(mine: dict, gold: dict, what: str):
    for k, g in gold.items():
        v = mine[k].cpu()
        assert list(v.shape) == g["shape"], (what, k)
        assert bool((v == v[:1]).all()), (what, k, "batch rows must be copies")
        if "sums" in g:
            assert torch.equal(v.reshape(v.shape[0], v.shape[1], -1).sum(-1), g["sums"]), (what, k)
            assert torch.equal(torch.nn.functional.avg_pool2d(v[:1, :4].contiguous(), v.shape[-1] // 16), g["pooled"]), (what, k)
        else:
            assert torch.equal(v[0, :6], g["head"]), (what, k)
            assert float(v[0, 6:].abs().sum()) == g["rest_abs_sum"], (what, k)


def test_prepare_batch_matches_reference_golden():
    from utils.input import prepare_batch
    gold = cases.load_golden("prepare_batch")
    meta, prepare_instance_meta, _ = _meta()
    # the demo JSON in this repo has the reference demo's boxes but its own captions: take the phrases the golden used
    meta["phrases"] = gold["meta"]["phrases"]
    assert meta["locations"] == gold["meta"]["locations"]
    meta["instance_meta"] = [prepare_instance_meta(meta, i) for i in range(len(meta["locations"]))]
    assert sorted(meta["instance_meta"][0].keys()) == gold["instance_meta_keys"]
    out = prepare_batch(meta, batch=2, max_objs=30, model=HashEncoder(), processor=None, image_size=64,
                        use_masked_att=True, device="cpu")
    _check(out, gold["main"], "main")
    assert len(out["instance_meta"]) == len(gold["inst"])
    for i, (m, g) in enumerate(zip(out["instance_meta"], gold["inst"])):
        _check(m, g, f"instance {i}")
    # MI355X-first layout: the large tensors are broadcast views, not per-sample copies
    assert out["segs"].stride(0) == 0 and out["att_masks"].stride(0) == 0 and out["boxes"].stride(0) != 0


def test_small_helpers_match_reference_golden():
    from utils.input import complete_mask, convert_points
    gold = cases.load_golden("prepare_batch")
    got = [complete_mask(None, 5), complete_mask(0.5, 5), complete_mask([0, 1], 5)]
    assert all(torch.equal(a, b) for a, b in zip(got, gold["complete_mask"]))
    assert convert_points([100.0, 600.0, 800.0, 20.0], dict(width=768, height=512)) == gold["convert_points"]


def test_meta_from_demo_json():
    from instancediffusion_amd import synth
    from instancediffusion_amd.host.input import meta_from_demo_json
    data = json.load(open(os.path.join(REPO, "demos", "demo_four_boxes.json")))
    meta = meta_from_demo_json(data, alpha=0.8)
    assert np.allclose(np.array(meta["locations"]), np.array(synth.C1_BOXES), atol=2e-3)
    assert meta["alpha_type"][0] == 0.8 and abs(meta["alpha_type"][2] - 0.2) < 1e-12 and len(meta["phrases"]) == 4
    assert meta["points"][0] == [(meta["locations"][0][0] + meta["locations"][0][2]) / 2,
                                 (meta["locations"][0][1] + meta["locations"][0][3]) / 2]
    assert meta["segs"].shape == (4, 512, 512) and not meta["segs"].any() and len(meta["polygons"][0]) == 512
    pt = meta_from_demo_json(dict(caption="c", width=768, height=512,
                                  annos=[dict(caption="a", point=[384, 128]), dict(caption="b", point=[0, 512])]), 0.75)
    assert pt["locations"] == [[0, 0, 0, 0], [0, 0, 0, 0]] and pt["points"] == [[0.5, 0.25], [0.0, 1.0]]
