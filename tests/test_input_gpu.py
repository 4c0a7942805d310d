"""SURVEY.md §8 f-1 on the GPU box.  The grounding-input builder stays HOST-side by design (it runs once per
prompt, its work is ~100 scalar slot fills and one 30 x 64 x 64 box rasterisation; what mattered for the device was the
layout of its outputs -- batch-broadcast views instead of 31 MB per sample -- not where the loops run).  What has to hold
on the device is that the tensors it hands to the MI355X path, built directly with ``device="cuda"``, are bit-identical to
the CPU-built ones (which tests/test_input_host.py pins to the reference goldens), and that the conditioning the engine
derives from either (UniFusion tokens, per-layer K / V^T caches, visibility words) is bit-identical too."""
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _flat(d, prefix=""):
    for k, v in d.items():
        if torch.is_tensor(v):
            yield prefix + k, v
        elif isinstance(v, list):
            for i, e in enumerate(v):
                yield from _flat(e, f"{prefix}{k}[{i}].")


def test_prepare_batch_on_device_equals_host_built_and_same_conditioning():
    from grounding_input.text_grounding_tokinzer_input import GroundingNetInput
    from tests import cases
    from tests.test_engine_emulated import build_model
    from tests.test_input_host import HashEncoder, _meta
    from utils.input import prepare_batch
    gold = cases.load_golden("prepare_batch")
    meta, prepare_instance_meta, _ = _meta()
    meta["phrases"] = gold["meta"]["phrases"]
    meta["instance_meta"] = [prepare_instance_meta(meta, i) for i in range(len(meta["locations"]))]
    kw = dict(batch=2, max_objs=30, model=HashEncoder(), processor=None, image_size=64, use_masked_att=True)
    host = prepare_batch(meta, device="cpu", **kw)
    dev = prepare_batch(meta, device="cuda", **kw)
    names = []
    for (k, a), (k2, b) in zip(_flat(host), _flat(dev)):
        assert k == k2 and b.is_cuda and a.shape == b.shape and a.dtype == b.dtype, k
        assert torch.equal(a, b.cpu()), k
        names.append(k)
    assert dev["segs"].stride(0) == 0 and dev["att_masks"].stride(0) == 0          # still broadcast views on the device
    assert {"boxes", "segs", "att_masks", "instance_meta[0].boxes"} <= set(names)
    # the conditioning derived from both: masked gated self-attention model (visibility words are part of the Cond)
    cfg = cases.cfg_for("test_box.yaml", "tiny")
    model = build_model(cfg, efficient_attention=False)
    gi = GroundingNetInput()
    ctx = torch.randn(2, 77, 768, generator=torch.Generator().manual_seed(5))
    eng = model.engine
    with torch.no_grad():
        c_host = eng.prepare_cond(ctx.cuda(), {k: v.cuda() for k, v in gi.prepare(host, return_att_masks=True).items()})
        c_dev = eng.prepare_cond(ctx.cuda(), gi.prepare(dev, return_att_masks=True))
    ta, tb = c_host._tensors(), c_dev._tensors()
    assert len(ta) == len(tb) and len(ta) > 4
    for a, b in zip(ta, tb):
        assert torch.equal(a, b)
