"""bench.py pieces that can be checked without a GPU: the forward count of the headline config, and that the
memory-lean input construction (one sample built, mask stack batch-broadcast) yields exactly the tensors the
materialised per-sample construction would."""
import torch

import bench
from instancediffusion_amd import synth
from instancediffusion_amd.host.config import SD15_BOX_CFG


def test_forward_count_matches_survey():
    # n_fwd = 2 [(N+1)(floor(S*mis)+1) + (S - floor(S*mis))]; C3 -> 406, no MIS -> 2 (S + 1) - 2... = 2*S
    assert bench.n_forwards(8, 50, 0.36) == 406
    assert bench.n_forwards(4, 50, 0.36) == 2 * (5 * 19 + 32)
    assert abs(406 * bench.GFLOP_PER_FWD / 1e3 - 498.3) < 0.1            # TFLOP per image, SURVEY §8d


def test_make_inputs_equals_materialised_construction():
    n = 3
    inputs, uc, gi, host = bench.make_inputs(dict(SD15_BOX_CFG), n, "cpu")
    g = torch.Generator().manual_seed(1234)
    boxes = synth.random_boxes(bench.N_INST, g)
    gb = synth.make_grounding_batch(n, boxes, g)
    x = torch.randn(n, 4, bench.LATENT, bench.LATENT, generator=g)
    ctx = torch.randn(n, 77, 768, generator=g)
    uc2 = torch.randn(1, 77, 768, generator=g)         # ONE negative-prompt context, broadcast (as inference.py encodes it)
    ic = [torch.randn(n, 77, 768, generator=g) for _ in range(bench.N_INST)]
    assert len(inputs) == bench.N_INST + 1
    assert torch.equal(inputs[0]["x"], x) and torch.equal(inputs[0]["context"], ctx)
    assert uc.shape == (n, 77, 768) and uc.stride(0) == 0 and torch.equal(uc[1], uc2[0])
    ref0 = gi.prepare(gb)
    assert all(torch.equal(v, ref0[k]) for k, v in inputs[0]["grounding_input"].items())
    for i in range(bench.N_INST):
        assert torch.equal(inputs[i + 1]["context"], ic[i])
        r = gi.prepare(synth.instance_batch(gb, i))
        assert all(torch.equal(v, r[k]) for k, v in inputs[i + 1]["grounding_input"].items())
        assert inputs[i + 1]["grounding_input"]["segs"].stride(0) == 0          # 31 MB/sample stays ONE plane set
    assert host["gb"]["boxes"].shape[0] == 1
