"""Config / plugin glue: YAML ``target:`` dotted path -> class(**params).

Mirrors the reference plugin mechanism (``ldm/util.py:71-85``) so unchanged ``configs/*.yaml`` drive this
implementation.  OmegaConf is not required: plain dicts from PyYAML (or any Mapping with ``get``) work.
"""
from __future__ import annotations

import importlib
from typing import Any, Dict, Mapping


def get_obj_from_str(string: str, reload: bool = False):
    module, cls = string.rsplit(".", 1)
    mod = importlib.import_module(module, package=None)
    if reload:
        mod = importlib.reload(mod)
    return getattr(mod, cls)


def instantiate_from_config(config: Mapping[str, Any]):
    """Same contract as the reference: needs key ``target``; two sentinel strings return None."""
    if "target" not in config:
        if config == "__is_first_stage__" or config == "__is_unconditional__":
            return None
        raise KeyError("Expected key `target` to instantiate.")
    params = config.get("params", dict())
    return get_obj_from_str(config["target"])(**(dict(params) if params is not None else {}))


def load_yaml(path: str) -> Dict[str, Any]:
    import yaml
    with open(path) as f:
        return yaml.safe_load(f)


# SD-1.5 InstanceDiffusion UNet + UniFusion hyper-parameters (reference configs/test_box.yaml:9-40), flat form
SD15_BOX_CFG = dict(
    in_channels=4, out_channels=4, model_channels=320, attention_resolutions=(4, 2, 1), num_res_blocks=2,
    channel_mult=(1, 2, 4, 4), num_heads=8, context_dim=768, in_dim=768, out_dim=768, mid_dim=3072,
    test_drop_boxes=False, test_drop_points=False, test_drop_scribbles=True, test_drop_masks=True)


def unet_kwargs_from_cfg(cfg: Mapping[str, Any]) -> Dict[str, Any]:
    """Flat test/bench config (oracle-style keys) -> ``UNetModel`` constructor kwargs."""
    return dict(
        image_size=cfg.get("image_size", 64), in_channels=cfg["in_channels"], model_channels=cfg["model_channels"],
        out_channels=cfg["out_channels"], num_res_blocks=cfg["num_res_blocks"],
        attention_resolutions=list(cfg["attention_resolutions"]), channel_mult=list(cfg["channel_mult"]),
        num_heads=cfg["num_heads"], context_dim=cfg["context_dim"], fuser_type="gatedSA", use_checkpoint=False,
        sd_v1_5=True, efficient_attention=True,
        grounding_tokenizer=dict(
            target="ldm.modules.diffusionmodules.text_grounding_net.UniFusion",
            params=dict(in_dim=cfg["in_dim"], out_dim=cfg["out_dim"], mid_dim=cfg["mid_dim"],
                        test_drop_boxes=cfg["test_drop_boxes"], test_drop_points=cfg["test_drop_points"],
                        test_drop_scribbles=cfg["test_drop_scribbles"], test_drop_masks=cfg["test_drop_masks"])),
    )
