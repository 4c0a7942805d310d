"""the masked gated self-attention (reference attention.py:187-255; reached with
``efficient_attention=False`` + ``grounding_input['att_masks']``).  CPU part: oracle restatement vs a golden of the
unmodified reference; the bit-word formulation the HIP kernel uses vs the reference's dense [N, N] mask; the engine's
host logic through the op emulation.  GPU part: kernel vs the dense-mask fp32 reference, engine vs the golden."""
import pytest
import torch

from oracle import ref_cpu
from tests import cases


def _setup():
    gold = cases.load_golden("tiny_masked_att")
    cfg = cases.cfg_for("test_box.yaml", "tiny")
    return gold, cfg, cases.build_masked_inputs()


def test_oracle_masked_forward_vs_reference_golden():
    from instancediffusion_amd import synth
    gold, cfg, inp = _setup()
    sd = synth.synth_state_dict(cases.unet_schema(cfg))
    assert abs(float(inp["x"].std()) - gold["meta"]["x_fp"]["std"]) < 1e-6
    with torch.no_grad():
        g = ref_cpu.prepare_grounding(inp["gb"])
        objs, drop_box = ref_cpu.unifusion(sd, cfg, g)
        assert not drop_box
        eps = ref_cpu.unet_forward(sd, cfg, inp["x"], inp["t"], inp["context"], objs, att_masks=inp["gb"]["att_masks"])
        eps_u = ref_cpu.unet_forward(sd, cfg, inp["x"], inp["t"], inp["context"], objs)
    e_m, e_u = cases.rel_rms(eps, gold["eps_masked"]), cases.rel_rms(eps_u, gold["eps_unmasked"])
    print(f"[parity] oracle masked forward: rel-rms {e_m:.3e} (unmasked {e_u:.3e}); mask effect "
          f"{cases.rel_rms(gold['eps_masked'], gold['eps_unmasked']):.2f}")
    assert e_m < 2e-4 and e_u < 2e-4


def test_bit_words_reproduce_the_dense_mask():
    """visibility(q, k) = (qbits[q] & kbits[k]) != 0 or k is q's own token  ==  reference mask > 0, visual query rows."""
    from instancediffusion_amd.host.attention import visibility_words
    _, _, inp = _setup()
    att = inp["gb"]["att_masks"]
    dense = ref_cpu.fuser_attention_mask(att, 4096 + 184)[:, 0, :4096] > 0          # [B, 4096, 4280]
    qb, kb0, kb1 = visibility_words(att)
    assert qb.dtype == torch.int32 and qb.shape == (1, 4096) and kb0.shape == (1, 4096) and kb1.shape[1] >= 184
    vis0 = (qb[:, :, None] & kb0[:, None, :]) != 0
    vis0 |= torch.eye(4096, dtype=torch.bool)[None]
    vis1 = (qb[:, :, None] & kb1[:, None, :184]) != 0
    assert torch.equal(torch.cat([vis0, vis1], -1), dense)
    # an all-zero mask TENSOR means "no mask" in the reference (attention.py:200): every word pair must intersect
    qz, kz0, kz1 = visibility_words(torch.zeros_like(att))
    assert bool(((qz[:, :, None] & kz0[:, None, :64]) != 0).all()) and bool(((qz[:, :1, None] & kz1[:, None, :184]) != 0).all())


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 3e-4), (torch.bfloat16, 4e-2)])
def test_engine_masked_forward_emulated(dtype, tol):
    from grounding_input.text_grounding_tokinzer_input import GroundingNetInput
    from instancediffusion_amd.engine import UNetEngine
    from tests.emul_ops import EmulOps
    from tests.test_engine_emulated import build_model
    gold, cfg, inp = _setup()
    model = build_model(cfg, efficient_attention=False)
    eng = UNetEngine(model, ops=EmulOps(dtype), use_graphs=False)
    gi = GroundingNetInput()
    with torch.no_grad():
        cond = eng.prepare_cond(inp["context"], gi.prepare(inp["gb"], return_att_masks=True))
        if dtype != torch.float32:                                             # storage-rounding check: one forward
            assert cases.rel_rms(eng.forward_cond(inp["x"], inp["t"], cond), gold["eps_masked"]) < tol
            return
        # masked sample and null-grounding sample (zero att_masks -> unmasked, attention.py:200) in ONE batched forward
        null = eng.prepare_cond(inp["context"], gi.get_null_input())
        both = eng.forward_cond(torch.cat([inp["x"]] * 2), torch.cat([inp["t"]] * 2), type(cond).cat([cond, null]))
        assert cases.rel_rms(both[:1], gold["eps_masked"]) < tol and cases.rel_rms(both[1:], gold["eps_null"]) < tol
        assert eng.ops.calls["attention_masked"] == sum(1 for p in eng._st_layers() if p["c"] == cfg["model_channels"])
        # a grounding input WITHOUT att_masks on the same model is the unmasked path
        plain = eng.prepare_cond(inp["context"], gi.prepare({k: v for k, v in inp["gb"].items() if k != "att_masks"}))
        assert cases.rel_rms(eng.forward_cond(inp["x"], inp["t"], plain), gold["eps_unmasked"]) < tol


# ---- GPU ------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("d,heads,nq,n1", [(40, 2, 4096, 184), (8, 8, 4096, 184), (40, 1, 300, 0), (80, 2, 256, 184)])
def test_masked_attention_kernel(d, heads, nq, n1):
    from instancediffusion_amd.ops import HipOps
    from tests.emul_ops import EmulOps
    ops, ref = HipOps(torch.bfloat16), EmulOps(torch.float32)
    g = torch.Generator().manual_seed(5)
    B, C = 2, d * heads
    q, k0 = torch.randn(B, nq, C, generator=g), torch.randn(B, nq, C, generator=g)
    ld0, ld1 = (nq + 63) // 64 * 64, 192
    vt0 = torch.zeros(B, C, ld0)
    vt0[:, :, :nq] = torch.randn(B, C, nq, generator=g)
    k1 = torch.randn(B, 184, C, generator=g)
    vt1 = torch.zeros(B, C, ld1)
    vt1[:, :, :184] = torch.randn(B, C, 184, generator=g)
    # random instance memberships: ~40 % of the tokens in no instance (they only see themselves + unconditional keys)
    obj = (torch.rand(B, nq, 5, generator=g) < 0.2)
    words = (obj.int() * (1 << torch.arange(5))).sum(-1).int()
    qb = (words | torch.tensor(-2 ** 31, dtype=torch.int32)).contiguous()
    kb0 = words.contiguous()
    kb1 = torch.full((B, 192), -1, dtype=torch.int32)
    kb1[:, :5] = (1 << torch.arange(5)).int()
    kb1[1, 7] = 0                                                       # a key nobody sees
    b16 = [t.to(torch.bfloat16) for t in (q, k0, vt0, k1, vt1)]
    kw = dict(qbits=qb, kbits0=kb0) if not n1 else dict(k1=b16[3].float(), vt1=b16[4].float(), n1=n1, qbits=qb, kbits0=kb0, kbits1=kb1)
    want = ref.attention(b16[0].float(), b16[1].float(), b16[2].float(), nq, torch.empty(B, nq, C), heads, **kw)
    kwd = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in kw.items()}
    if n1:
        kwd.update(k1=b16[3].cuda(), vt1=b16[4].cuda())
    out = ops.attention(b16[0].cuda(), b16[1].cuda(), b16[2].cuda(), nq, ops.empty((B, nq, C)), heads, **kwd)
    torch.cuda.synchronize()
    err = float((out.float().cpu() - want).abs().max() / want.abs().max())
    assert torch.isfinite(out).all() and err < 2.0 ** -6, err
    # the mask matters: the unmasked result differs
    plain = ref.attention(b16[0].float(), b16[1].float(), b16[2].float(), nq, torch.empty(B, nq, C), heads,
                          **{k: v for k, v in kw.items() if "bits" not in k})
    assert float((plain - want).abs().max()) > 0.05


@pytest.mark.gpu
def test_engine_masked_forward_gpu():
    from grounding_input.text_grounding_tokinzer_input import GroundingNetInput
    from tests.test_engine_emulated import build_model
    gold, cfg, inp = _setup()
    model = build_model(cfg, efficient_attention=False)
    gi = GroundingNetInput()
    model.grounding_tokenizer_input = gi
    g = gi.prepare({k: v.cuda() for k, v in inp["gb"].items()}, return_att_masks=True)
    with torch.no_grad():
        eps = model(dict(x=inp["x"].cuda(), timesteps=inp["t"].cuda(), context=inp["context"].cuda(), grounding_input=g))
        eps2 = model(dict(x=inp["x"].cuda(), timesteps=inp["t"].cuda(), context=inp["context"].cuda(), grounding_input=g))
        eps_n = model(dict(x=inp["x"].cuda(), timesteps=inp["t"].cuda(), context=inp["context"].cuda()))
    err, err_n = cases.rel_rms(eps.float().cpu(), gold["eps_masked"]), cases.rel_rms(eps_n.float().cpu(), gold["eps_null"])
    print(f"[parity] masked gated self-attention forward bf16: rel-rms {err:.3e}; null grounding {err_n:.3e}")
    assert torch.equal(eps, eps2) and err < 3e-2 and err_n < 3e-2


def test_bit_words_random_masks_property():
    """Random (overlapping, partly empty) instance masks on a small grid: words == dense reference mask."""
    from instancediffusion_amd.host.attention import visibility_words
    g = torch.Generator().manual_seed(11)
    for n_objs in (1, 7, 30):
        att = (torch.rand(2, n_objs, 64, 64, generator=g) < 0.08).float()
        att[1, n_objs // 2] = 0       # an instance with an empty mask (n_objs == 1: a sample with NO instance pixel)
        dense = ref_cpu.fuser_attention_mask(att, 4096 + 4 * n_objs + 64)[:, 0, :4096] > 0
        qb, kb0, kb1 = visibility_words(att)
        n1 = 4 * n_objs + 64
        vis0 = ((qb[:, :, None] & kb0[:, None, :]) != 0) | torch.eye(4096, dtype=torch.bool)[None]
        vis1 = (qb[:, :, None] & kb1[:, None, :n1]) != 0
        assert torch.equal(torch.cat([vis0, vis1], -1), dense), n_objs


def test_visibility_words_travel_with_the_conditioning_bank():
    """The samplers assemble [cond | uncond] / per-unit batches by row-gathering a bank of conditionings into the
    engine's static slots: the visibility words must be gathered like every other per-sample tensor (first gather
    allocates the slot, later gathers write it in place -- the hipGraph-visible path)."""
    from grounding_input.text_grounding_tokinzer_input import GroundingNetInput
    from instancediffusion_amd.engine import UNetEngine
    from tests.emul_ops import EmulOps
    from tests.test_engine_emulated import build_model
    _, cfg, inp = _setup()
    model = build_model(cfg, efficient_attention=False)
    eng = UNetEngine(model, ops=EmulOps(torch.float32), use_graphs=False)
    gi = GroundingNetInput()
    with torch.no_grad():
        cond = eng.prepare_cond(inp["context"], gi.prepare(inp["gb"], return_att_masks=True))
        null = eng.prepare_cond(inp["context"], gi.get_null_input())
    assert len(cond.vis) == 3 and cond.vis[0].dtype == torch.int32 and cond.vis[0].shape == (1, 4096)
    assert bool((null.vis[0] == -1).all()) and not bool((cond.vis[0] == -1).all())
    bank = type(cond).cat([cond, null])
    slot = eng.gather_cond(bank, torch.tensor([1, 0, 0]))
    for k in range(3):
        assert torch.equal(slot.vis[k][0], null.vis[k][0]) and torch.equal(slot.vis[k][2], cond.vis[k][0])
    slot2 = eng.gather_cond(bank, torch.tensor([0, 1, 1]))                    # same batch size: in-place refill
    assert slot2 is slot and torch.equal(slot.vis[0][0], cond.vis[0][0]) and torch.equal(slot.vis[2][1], null.vis[2][0])
    assert torch.equal(slot.objs[1], null.objs[0])
