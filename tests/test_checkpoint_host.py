"""SURVEY.md §8 f-3 (host side, no GPU): ``utils.checkpoint.load_model_ckpt`` / ``read_official_ckpt`` mirrors on a
synthetic checkpoint with the reference's layout (``ema`` / ``model`` / ``autoencoder`` / ``text_encoder`` /
``diffusion`` / ``config_dict`` with pickled OmegaConf-like nodes), and the CLIP text-encoder wrapper."""
import os
import sys
import types

import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

TINY = dict(
    model=dict(target="ldm.modules.diffusionmodules.openaimodel.UNetModel", params=dict(
        image_size=16, in_channels=4, out_channels=4, model_channels=64, attention_resolutions=[4, 2, 1],
        num_res_blocks=1, channel_mult=[1, 2], num_heads=8, context_dim=768, fuser_type="gatedSA", use_checkpoint=False,
        sd_v1_5=True, efficient_attention=True,
        grounding_tokenizer=dict(target="ldm.modules.diffusionmodules.text_grounding_net.UniFusion",
                                 params=dict(in_dim=768, out_dim=768, mid_dim=128)))),
    autoencoder=dict(target="ldm.models.autoencoder.AutoencoderKL", params=dict(
        scale_factor=0.18215, embed_dim=4, ddconfig=dict(double_z=True, z_channels=4, resolution=64, in_channels=3,
                                                        out_ch=3, ch=64, ch_mult=[1, 2], num_res_blocks=1,
                                                        attn_resolutions=[], dropout=0.0))),
    text_encoder=dict(target="tests.test_checkpoint_host.StubTextEncoder"),
    diffusion=dict(target="ldm.models.diffusion.ldm.LatentDiffusion",
                   params=dict(linear_start=0.00085, linear_end=0.012, timesteps=1000)),
    grounding_tokenizer_input=dict(target="grounding_input.text_grounding_tokinzer_input.GroundingNetInput"),
)


class StubTextEncoder(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.w = torch.nn.Parameter(torch.zeros(3))


def _build_all():
    from instancediffusion_amd.host.config import instantiate_from_config
    return {k: instantiate_from_config(TINY[k]) for k in ("model", "autoencoder", "text_encoder", "diffusion")}


def _fake_omegaconf_nodes(cfg):
    """Pickle-compatible stand-ins laid out like OmegaConf's DictConfig/ListConfig/AnyNode (``_content`` / ``_val``),
    registered under the module names a real checkpoint references, removed again before loading."""
    mod = types.ModuleType("omegaconf")
    sub = types.ModuleType("omegaconf.dictconfig")
    nodes = types.ModuleType("omegaconf.nodes")

    class DictConfig:
        pass

    class ListConfig:
        pass

    class AnyNode:
        pass
    for cls, m in ((DictConfig, sub), (ListConfig, sub), (AnyNode, nodes)):
        cls.__module__ = m.__name__
        cls.__qualname__ = cls.__name__
        setattr(m, cls.__name__, cls)
    sys.modules.update({"omegaconf": mod, "omegaconf.dictconfig": sub, "omegaconf.nodes": nodes})

    def wrap(v):
        if isinstance(v, dict):
            n = DictConfig()
            n.__dict__.update(_metadata=None, _parent=None, _content={k: wrap(x) for k, x in v.items()})
            return n
        if isinstance(v, (list, tuple)):
            n = ListConfig()
            n.__dict__.update(_metadata=None, _parent=None, _content=[wrap(x) for x in v])
            return n
        n = AnyNode()
        n.__dict__.update(_metadata=None, _parent=None, _val=v)
        return n
    return wrap(cfg)


def _drop_fake_omegaconf():
    for k in ("omegaconf", "omegaconf.dictconfig", "omegaconf.nodes"):
        sys.modules.pop(k, None)


def _digest(sd):
    return {k: float(v.double().sum()) for k, v in sd.items()}


def _same_digest(got, want, tol=1e-9):
    assert sorted(got) == sorted(want)
    for k, v in want.items():
        assert abs(got[k] - v) <= tol * max(1.0, abs(v)), (k, got[k], v)


@pytest.fixture(scope="module")
def ref_layout_ckpt(tmp_path_factory):
    """The reduced checkpoint of the golden case, written ONCE (with and without ``ema``): same key-seeded weights the
    reference was given, config pickled as OmegaConf-layout nodes in ``vars(cfg)`` form (utils/misc.py:255)."""
    from instancediffusion_amd import synth
    from instancediffusion_amd.host.config import instantiate_from_config
    gold = torch.load(os.path.join(REPO, "tests", "golden", "ckpt_tiny.pt"), weights_only=False)
    cfg, salts = gold["cfg"], gold["meta"]["salts"]
    mods = {k: instantiate_from_config(cfg[k]) for k in ("model", "autoencoder", "text_encoder", "diffusion")}

    def synth_sd(mod, salt):
        return synth.synth_state_dict({k: tuple(v.shape) for k, v in mod.state_dict().items()}, salt)
    ckpt = dict(model=synth_sd(mods["model"], salts["model"]), text_encoder=synth_sd(mods["text_encoder"], salts["text_encoder"]),
                autoencoder=synth_sd(mods["autoencoder"], salts["autoencoder"]), opt={}, scheduler={}, iters=42,
                config_dict=dict(_metadata=None, _parent=None, _flags_cache=None,
                                 _content=_fake_omegaconf_nodes(cfg).__dict__["_content"]),
                diffusion=mods["diffusion"].state_dict(), ema=synth_sd(mods["model"], salts["ema"]))
    d = tmp_path_factory.mktemp("ckpt")
    paths = {True: str(d / "with_ema.pth"), False: str(d / "no_ema.pth")}
    torch.save(ckpt, paths[True])
    torch.save({k: v for k, v in ckpt.items() if k != "ema"}, paths[False])
    _drop_fake_omegaconf()
    return gold, ckpt, paths


@pytest.mark.parametrize("with_ema,use_yaml", [(True, False), (False, True)])     # the golden also holds the other two
def test_load_model_ckpt_matches_reference_golden(tmp_path, monkeypatch, ref_layout_ckpt, with_ema, use_yaml):
    """PIN (tests/golden/ckpt_tiny.pt, oracle/make_golden.py:gen_ckpt_case): the unmodified reference's ``save_ckpt`` wrote
    a reduced-size checkpoint and its ``load_model_ckpt`` (utils/checkpoint.py:224-249) loaded it; the golden holds the
    checkpoint's key set, per-tensor digests of the four returned modules and the returned config.  Here the same
    checkpoint (same key-seeded weights, same key set) goes through the mirror loader, which must return the same
    5-tuple contents."""
    import yaml
    from instancediffusion_amd.host import checkpoint as ck
    from utils.checkpoint import load_model_ckpt
    gold, ckpt, paths = ref_layout_ckpt
    want = gold["cases"][(with_ema, use_yaml)]
    monkeypatch.setattr(ck, "ALLOWED_TARGET_ROOTS", ck.ALLOWED_TARGET_ROOTS + ("torch",))   # the 4->3 Linear stand-in
    assert sorted(k for k in ckpt if with_ema or k != "ema") == want["saved_keys"]   # layout the reference's save_ckpt wrote
    args = types.SimpleNamespace(test_config="")
    if use_yaml:
        args.test_config = str(tmp_path / "cfg.yaml")
        yaml.safe_dump(gold["cfg"], open(args.test_config, "w"))
    model, ae, te, diffusion, config = load_model_ckpt(paths[with_ema], args, "cpu")
    _same_digest(_digest(model.state_dict()), want["model"])
    _same_digest(_digest(ae.state_dict()), want["autoencoder"])
    _same_digest(_digest(te.state_dict()), want["text_encoder"])
    _same_digest(_digest(diffusion.state_dict()), want["diffusion"], tol=1e-6)
    assert [model.training, ae.training, te.training] == want["training"]
    assert config == want["config"]
    # ema present -> ema weights; absent -> model weights (the bare-except fallback of the reference)
    src = ckpt["ema"] if with_ema else ckpt["model"]
    assert all(torch.equal(model.state_dict()[k], v) for k, v in src.items())
    assert type(model).__module__.startswith("instancediffusion_amd") and type(ae).__module__.startswith("instancediffusion_amd")


def test_checkpoint_loading_refuses_foreign_globals_and_targets(tmp_path):
    """A checkpoint is data: pickled globals outside torch / containers / numpy / (bagged) omegaconf are refused, and a
    checkpoint-embedded config may only name classes of this code base."""
    import pickle
    from instancediffusion_amd.host import checkpoint as ck

    class Evil:
        def __reduce__(self):
            return (os.system, ("true",))
    path = str(tmp_path / "evil.pth")
    torch.save(dict(model={}, payload=Evil()), path)
    with pytest.raises(pickle.UnpicklingError):
        ck.tolerant_torch_load(path)
    # an allow-listed module ROOT is not enough: callables inside torch / numpy that execute their argument
    # must be refused, and refused BEFORE anything runs
    import numpy.testing._private.utils as npu
    import torch.utils.collect_env as ce
    marker = tmp_path / "ran"

    class RunString:
        def __reduce__(self):
            return (npu.runstring, (f"open({str(marker)!r}, 'w').write('x')", {}))

    class CollectEnv:
        def __reduce__(self):
            return (ce.run, (f"touch {marker}",))

    class HubLoad:
        def __reduce__(self):
            return (torch.hub.load, ("x/y", "z"))
    for k, payload in enumerate((RunString(), CollectEnv(), HubLoad())):
        path_k = str(tmp_path / f"evil{k}.pth")
        torch.save(dict(model={}, payload=payload), path_k)
        with pytest.raises(pickle.UnpicklingError, match="refused"):
            ck.tolerant_torch_load(path_k)
        with open(path_k, "wb") as f:                       # a bare pickle (no zip container), as in the advisor's PoC
            pickle.dump(dict(payload=payload), f)
        with pytest.raises(Exception):
            ck.tolerant_torch_load(path_k)
        assert not marker.exists(), "the payload must not have run"
    # what a real checkpoint holds still loads: tensors of several dtypes, parameters, numpy arrays / scalars, an
    # argparse.Namespace, an OrderedDict, and pytorch_lightning callback objects as inert bags
    import argparse
    import collections
    import sys
    import types
    import numpy as np
    pl = types.ModuleType("pytorch_lightning")
    plc = types.ModuleType("pytorch_lightning.callbacks")

    class ModelCheckpoint:
        def __init__(self):
            self.best = 0.5
    ModelCheckpoint.__module__, ModelCheckpoint.__qualname__ = "pytorch_lightning.callbacks", "ModelCheckpoint"
    plc.ModelCheckpoint = ModelCheckpoint
    sys.modules.update({"pytorch_lightning": pl, "pytorch_lightning.callbacks": plc})
    try:
        good = dict(state_dict=collections.OrderedDict(a=torch.ones(2, dtype=torch.float16), b=torch.arange(3)),
                    p=torch.nn.Parameter(torch.zeros(1)), arr=np.arange(4.0), sc=np.float32(2.5),
                    args=argparse.Namespace(lr=0.1), callbacks={"ckpt": ModelCheckpoint()}, dt=torch.bfloat16)
        path_g = str(tmp_path / "good.pth")
        torch.save(good, path_g)
    finally:
        sys.modules.pop("pytorch_lightning"), sys.modules.pop("pytorch_lightning.callbacks")
    back = ck.tolerant_torch_load(path_g)
    assert torch.equal(back["state_dict"]["a"], good["state_dict"]["a"]) and back["args"].lr == 0.1
    assert float(back["sc"]) == 2.5 and back["arr"].tolist() == [0.0, 1.0, 2.0, 3.0] and back["dt"] == torch.bfloat16
    assert type(back["callbacks"]["ckpt"]).__name__ == "ModelCheckpoint" and back["callbacks"]["ckpt"].best == 0.5
    assert isinstance(back["callbacks"]["ckpt"], ck._Bag)
    # a missing file is an I/O error, not something to retry with the weaker unpickler
    with pytest.raises(FileNotFoundError):
        ck.tolerant_torch_load(str(tmp_path / "nope.pth"))
    with pytest.raises(ValueError):
        ck.check_target_namespace(dict(model=dict(target="os.system", params={})))
    ck.check_target_namespace(dict(model=dict(target="ldm.modules.diffusionmodules.openaimodel.UNetModel")))


def test_read_official_ckpt_split(tmp_path):
    from utils.checkpoint import read_official_ckpt
    sd = {"model.diffusion_model.input_blocks.0.0.weight": torch.ones(1), "cond_stage_model.transformer.x": torch.ones(2),
          "first_stage_model.decoder.conv_in.weight": torch.ones(3), "model_ema.decay": torch.ones(1), "betas": torch.ones(4)}
    path = str(tmp_path / "sd.ckpt")
    torch.save({"state_dict": sd}, path)
    out = read_official_ckpt(path)
    assert list(out["model"]) == ["input_blocks.0.0.weight"] and list(out["text_encoder"]) == ["transformer.x"]
    assert list(out["autoencoder"]) == ["decoder.conv_in.weight"] and list(out["diffusion"]) == ["betas"]
    assert list(out["unexpected"]) == ["model_ema.decay"]


def test_clip_text_encoder_wrapper_offline():
    """Built from the fixed CLIP-L/14 text config (no download); loads the reference's 4.x key layout; refuses to
    tokenize with the empty tokenizer transformers >= 5 would otherwise fabricate."""
    pytest.importorskip("transformers")
    from ldm.modules.encoders.modules import FrozenCLIPEmbedder
    enc = FrozenCLIPEmbedder(device="cpu")
    assert sum(p.numel() for p in enc.parameters()) == 123060480 and not enc.transformer.training
    mine = enc.state_dict()
    old_layout = {}
    for k, v in mine.items():
        k4 = k if k.startswith("transformer.text_model.") else "transformer.text_model." + k[len("transformer."):]
        old_layout[k4] = torch.full_like(v, 0.25)
    old_layout["transformer.text_model.embeddings.position_ids"] = torch.arange(77)[None]       # 4.x buffer
    enc.load_state_dict(old_layout, strict=False)
    assert all(bool((v == 0.25).all()) for v in enc.state_dict().values() if v.is_floating_point())
    with pytest.raises(RuntimeError):
        enc.load_state_dict({"something.else": torch.zeros(1)}, strict=False)
    if not os.environ.get("IDF_CLIP_PATH"):
        with pytest.raises(RuntimeError, match="vocabulary"):
            enc.encode(["a cat"])


def _clip_vs_golden(device):
    """The wrapper (built from ITS OWN fixed config, weights through ITS load_state_dict key remap) against the
    last_hidden_state / pooler_output the unmodified reference FrozenCLIPEmbedder produced on the same key-seeded weights
    and input_ids (tests/golden/clip_text.pt, oracle/make_golden.py --only clip).  The tokenizer is bypassed on both
    sides: its vocabulary files do not exist offline."""
    from instancediffusion_amd import synth
    from ldm.modules.encoders.modules import FrozenCLIPEmbedder
    from tests import cases
    gold = cases.load_golden("clip_text")
    enc = FrozenCLIPEmbedder(device=device)
    sd = synth.synth_state_dict({k: tuple(v) for k, v in gold["schema"].items()}, gold["meta"]["salt"])
    # hand the weights over in the OTHER key layout than the golden's, so the remap is what is being tested too
    nested = any(k.startswith("transformer.text_model.") for k in sd)
    if nested:
        other = {"transformer." + k[len("transformer.text_model."):]: v for k, v in sd.items()}
    else:
        other = {"transformer.text_model." + k[len("transformer."):]: v for k, v in sd.items()}
    res = enc.load_state_dict(other, strict=False)
    assert not res.unexpected_keys and all("position_ids" in k for k in res.missing_keys), res
    enc = enc.to(device)
    ids = gold["input_ids"]

    class FixedTokenizer:
        def __call__(self, text, **kw):
            return {"input_ids": ids[:len(text)]}
    enc._tokenizer = FixedTokenizer()
    z, pooled = enc.encode(["a", "b"], return_pooler_output=True)
    ez = cases.rel_rms(z.float().cpu(), gold["last_hidden_state"])
    ep = cases.rel_rms(pooled.float().cpu(), gold["pooler_output"])
    print(f"[parity] CLIP text encoder wrapper on {device} fp32 vs the reference class: last_hidden_state rel-rms {ez:.2e}, "
          f"pooler_output {ep:.2e} (tol 1e-5)")
    assert tuple(z.shape) == (2, 77, 768) and tuple(pooled.shape) == (2, 768)
    assert float(gold["last_hidden_state"][0].std(0).mean()) > 1e-3, "degenerate golden"
    assert ez < 1e-5 and ep < 1e-5
    # pooler_output = the hidden state at the FIRST <|endoftext|> (row 0: position 10, row 1: position 41)
    assert torch.allclose(pooled[0].cpu(), z[0, 10].cpu(), atol=1e-6) and torch.allclose(pooled[1].cpu(), z[1, 41].cpu(), atol=1e-6)


def test_clip_text_encoder_forward_matches_reference_golden():
    pytest.importorskip("transformers")
    _clip_vs_golden("cpu")


@pytest.mark.gpu
def test_clip_text_encoder_forward_matches_reference_golden_gpu():
    pytest.importorskip("transformers")
    _clip_vs_golden("cuda")


def test_first_conv_file_is_found_next_to_the_checkpoint_or_fails_before_sampling(tmp_path, monkeypatch):
    """``restore_first_conv_from_SD`` (openaimodel.py:469-480) reads ``pretrained/SD_v1_5_input_conv_weight_bias.pth``
    relative to the cwd; a run from this tree used to die at the first alpha == 0 step (40 of 50).  Now the file is searched in
    the cwd, ``$IDF_PRETRAINED_DIR``, next to the checkpoint and in the repository, and a sampler whose alpha schedule reaches 0
    asks for it BEFORE the first step."""
    from functools import partial
    import torch
    from ldm.modules.diffusionmodules.openaimodel import UNetModel
    from instancediffusion_amd.host.alpha import alpha_generator, set_alpha_scale
    from instancediffusion_amd.host.config import unet_kwargs_from_cfg
    from instancediffusion_amd.host.samplers import PLMSSampler
    from tests import cases
    with torch.device("meta"):
        m = UNetModel(**unet_kwargs_from_cfg(cases.cfg_for("test_box.yaml", "full")))
    monkeypatch.chdir(tmp_path)
    monkeypatch.delenv("IDF_PRETRAINED_DIR", raising=False)
    import instancediffusion_amd.host.unet as unet_mod
    monkeypatch.setattr(unet_mod.os.path if hasattr(unet_mod, "os") else __import__("os").path, "isfile",
                        lambda p, _real=__import__("os").path.isfile: _real(p) and str(tmp_path) in os.path.abspath(p))
    with pytest.raises(FileNotFoundError) as e:
        m.first_conv_file()
    assert "IDF_PRETRAINED_DIR" in str(e.value) and "pretrained" in str(e.value)

    class _Diff:
        num_timesteps = 1000
        betas = torch.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000) ** 2
        alphas_cumprod = torch.cumprod(1 - betas, 0)
        alphas_cumprod_prev = torch.cat([torch.ones(1), alphas_cumprod[:-1]])
    s = PLMSSampler(_Diff(), m, alpha_generator_func=partial(alpha_generator, type=[0.8, 0.0, 0.2]), set_alpha_scale=set_alpha_scale)
    with pytest.raises(FileNotFoundError):
        s._check_first_conv(alpha_generator(50, type=[0.8, 0.0, 0.2]))           # reaches 0 at step 40: asks now
    s._check_first_conv(alpha_generator(50, type=[1.0, 0.0, 0.0]))               # never 0: nothing needed
    m.first_conv_sd_override = dict(weight=torch.zeros(320, 4, 3, 3), bias=torch.zeros(320))
    s._check_first_conv(alpha_generator(50, type=[0.8, 0.0, 0.2]))               # an in-memory override satisfies it
    m.first_conv_sd_override = None
    # next to the checkpoint
    ck = tmp_path / "weights" / "instancediffusion_sd15.pth"
    (tmp_path / "weights" / "pretrained").mkdir(parents=True)
    f = tmp_path / "weights" / "pretrained" / "SD_v1_5_input_conv_weight_bias.pth"
    torch.save(dict(weight=torch.zeros(320, 4, 3, 3), bias=torch.zeros(320)), f)
    m.ckpt_path = str(ck)
    assert os.path.samefile(m.first_conv_file(), f)
    s._check_first_conv(alpha_generator(50, type=[0.8, 0.0, 0.2]))
    # the environment variable
    m.ckpt_path = None
    with pytest.raises(FileNotFoundError):
        m.first_conv_file()
    monkeypatch.setenv("IDF_PRETRAINED_DIR", str(tmp_path / "weights" / "pretrained"))
    assert os.path.samefile(m.first_conv_file(), f)
