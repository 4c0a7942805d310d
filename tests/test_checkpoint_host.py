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


@pytest.mark.parametrize("with_ema,use_yaml", [(True, False), (False, False), (True, True)])
def test_load_model_ckpt_reference_layout(tmp_path, with_ema, use_yaml):
    import yaml
    from utils.checkpoint import load_model_ckpt
    objs = _build_all()
    torch.manual_seed(3)
    sds = {k: {n: torch.randn_like(v) if v.is_floating_point() else v.clone() for n, v in o.state_dict().items()}
           for k, o in objs.items()}
    ckpt = dict(model={k: v + 1 for k, v in sds["model"].items()}, autoencoder=sds["autoencoder"],
                text_encoder=dict(sds["text_encoder"], extra_unexpected=torch.zeros(1)), diffusion=sds["diffusion"],
                config_dict={"_content": _fake_omegaconf_nodes(TINY)}, iters=123)
    if with_ema:
        ckpt["ema"] = sds["model"]
    path = str(tmp_path / "ckpt.pth")
    torch.save(ckpt, path)
    _drop_fake_omegaconf()
    args = types.SimpleNamespace(test_config="")
    if use_yaml:
        ypath = str(tmp_path / "cfg.yaml")
        yaml.safe_dump(TINY, open(ypath, "w"))
        args.test_config = ypath
    model, ae, te, diffusion, config = load_model_ckpt(path, args, "cpu")
    want = sds["model"] if with_ema else ckpt["model"]
    got = model.state_dict()
    assert all(torch.equal(got[k], want[k]) for k in want)
    assert all(torch.equal(ae.state_dict()[k], v) for k, v in sds["autoencoder"].items())
    assert torch.equal(te.w, sds["text_encoder"]["w"]) and not model.training and not ae.training
    assert torch.equal(diffusion.betas, sds["diffusion"]["betas"])
    assert config["model"]["params"]["model_channels"] == 64 and isinstance(config["model"], dict)
    assert type(model).__module__.startswith("instancediffusion_amd") and type(ae).__module__.startswith("instancediffusion_amd")


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
