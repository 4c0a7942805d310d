"""VAE decoder, no GPU: the oracle restatement against goldens of the unmodified reference
``AutoencoderKL.decode``, the host mirror's state-dict schema, and the engine's host logic (weight packing, op order,
buffer ping-pong, the GEMM -> softmax -> GEMM attention with the v-bias folded into the P.V epilogue) through the CPU
op emulation."""
import json
import os

import pytest
import torch

from oracle import ref_cpu
from tests import cases
from tests.emul_ops import EmulOps


@pytest.mark.parametrize("tag", ["vae_tiny", "vae_full_16"])
def test_oracle_vae_decode_vs_reference_golden(tag):
    gold = cases.load_golden(tag)
    meta = gold["meta"]
    cfg = cases.vae_cfg_for(meta["variant"])
    ae = cases.build_vae(cfg, meta["salt"])
    sd = {k: v.detach() for k, v in ae.state_dict().items()}
    z = cases.vae_latent(meta)
    assert abs(float(z.std()) - meta["z_fp"]["std"]) < 1e-6
    with torch.no_grad():
        img = ref_cpu.vae_decode(sd, cfg, z)
    err = cases.rel_rms(img, gold["img"])
    print(f"[parity] oracle vae_decode {tag}: rel-rms {err:.3e}")
    assert img.shape == gold["img"].shape and err < 2e-4


def test_vae_schema_matches_reference():
    ref = json.load(open(os.path.join(cases.GOLD, "vae_schema.json")))
    ae = cases.build_vae(cases.vae_cfg_for("full"))
    mine = {k: tuple(v.shape) for k, v in ae.state_dict().items()}
    assert set(ref) == set(mine) and len(ref) == 248
    assert all(tuple(ref[k]) == mine[k] for k in ref)


@pytest.mark.parametrize("tag,dtype,tol", [("vae_tiny", torch.float32, 3e-4), ("vae_tiny", torch.bfloat16, 4e-2),
                                           ("vae_full_16", torch.float32, 3e-4)])
def test_vae_engine_vs_reference(tag, dtype, tol):
    from instancediffusion_amd.vae_engine import VAEDecoderEngine
    gold = cases.load_golden(tag)
    meta = gold["meta"]
    ae = cases.build_vae(cases.vae_cfg_for(meta["variant"]), meta["salt"])
    eng = VAEDecoderEngine(ae, ops=EmulOps(dtype), use_graphs=False)
    with torch.no_grad():
        img = eng.decode(cases.vae_latent(meta))
    err = cases.rel_rms(img, gold["img"])
    print(f"[parity] emulated VAE engine {tag} {dtype}: rel-rms {err:.3e}")
    assert img.shape == gold["img"].shape and err < tol
    assert eng.ops.calls["softmax_rows"] == 1 and eng.ops.calls["pointwise_nchw"] == 1


def test_vae_decode_refuses_without_hip():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    ae = cases.build_vae(cases.vae_cfg_for("tiny"))
    with pytest.raises(RuntimeError):
        ae.decode(torch.zeros(1, 4, 8, 8))
    with pytest.raises(NotImplementedError):
        ae.encode(torch.zeros(1, 3, 64, 64))


def test_oracle_vae_decode_full_size_vs_reference_digest():
    """The reference's real call (inference.py:95): 64x64 latent -> 512x512 image; golden kept as an 8x8 average-pooled
    digest + moments (the full image would be 3 MB)."""
    gold = cases.load_golden("vae_full_64")
    meta = gold["meta"]
    cfg = cases.vae_cfg_for("full")
    ae = cases.build_vae(cfg, meta["salt"])
    with torch.no_grad():
        img = ref_cpu.vae_decode({k: v.detach() for k, v in ae.state_dict().items()}, cfg, cases.vae_latent(meta))
    assert list(img.shape) == gold["img_fp"]["shape"] == [1, 3, 512, 512]
    assert cases.rel_rms(torch.nn.functional.avg_pool2d(img, 8), gold["img_pool8"]) < 2e-4
    assert abs(float(img.std()) - gold["img_fp"]["std"]) < 1e-4 * gold["img_fp"]["std"] + 1e-6
    assert torch.allclose(img.flatten()[:32], gold["img_fp"]["head"], atol=1e-4, rtol=1e-4)
