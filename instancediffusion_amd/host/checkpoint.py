"""Host mirror of ``utils/checkpoint.py`` for the sampling path: ``load_model_ckpt`` and
``read_official_ckpt``, so a trained ``instancediffusion_sd15.pth`` drives the MI355X path unchanged.

Same behaviour as the reference (utils/checkpoint.py:224-248): config from ``args.test_config`` (YAML) when given, else
from the checkpoint's pickled ``config_dict``; UNet weights from ``['ema']`` with fallback to ``['model']``;
``['autoencoder']``, ``['text_encoder']`` (strict=False) and ``['diffusion']`` sub-dicts loaded into the objects the
config's ``target:`` paths resolve to (here: the MI355X mirrors).  Two robustness additions, both host-side only:
  * the checkpoint pickles OmegaConf objects (``config_dict``); OmegaConf is not required here -- unknown classes are
    unpickled as attribute bags and converted to plain dicts (``plain_config``);
  * training / saving helpers of the reference file (tensorboard writer, auto-resume, ``save_ckpt``) are training-only
    and are not mirrored.
"""
from __future__ import annotations

import pickle
from typing import Any, Dict

import torch

from .config import instantiate_from_config, load_yaml


# ---- tolerant unpickling -----------------------------------------------------------------------------------------
class _Bag:
    """Stand-in for a class whose module is not importable (omegaconf.*): keeps whatever state pickle hands over."""

    def __init__(self, *a, **k):
        self._args = a

    def __setstate__(self, state):
        if isinstance(state, dict):
            self.__dict__.update(state)
        elif isinstance(state, tuple) and len(state) == 2 and isinstance(state[0], (dict, type(None))):
            for part in state:
                if isinstance(part, dict):
                    self.__dict__.update(part)
        else:
            self._state = state

    def __reduce_ex__(self, protocol):           # pragma: no cover - bags are never re-pickled
        raise pickle.PicklingError("placeholder object")


# Globals a reference checkpoint legitimately pickles: tensors / storages (torch), containers (collections, builtins
# data types), numpy scalars, and the OmegaConf node classes of ``config_dict`` (turned into attribute bags, never
# imported).  Everything else is refused: a checkpoint is data, it must not be able to name arbitrary callables.
_SAFE_BUILTINS = {"dict", "list", "tuple", "set", "frozenset", "int", "float", "bool", "str", "bytes", "bytearray",
                  "complex", "slice", "range", "object"}
_SAFE_MODULE_ROOTS = ("torch", "collections", "numpy", "typing", "pathlib", "argparse", "enum")
_BAG_MODULE_ROOTS = ("omegaconf",)


class _TolerantUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        root = module.split(".", 1)[0]
        if root in _BAG_MODULE_ROOTS:
            return type(name, (_Bag,), {"__module__": module})
        if module == "builtins":
            if name in _SAFE_BUILTINS:
                return super().find_class(module, name)
            raise pickle.UnpicklingError(f"checkpoint names builtins.{name}: refused")
        if root in _SAFE_MODULE_ROOTS:
            try:
                return super().find_class(module, name)
            except (ImportError, AttributeError):
                return type(name, (_Bag,), {"__module__": module})
        raise pickle.UnpicklingError(f"checkpoint names {module}.{name}: refused (not a tensor / container / config class)")


class _tolerant_pickle:
    """``pickle_module`` for torch.load: the stdlib pickle with a class-tolerant Unpickler."""
    __name__ = "pickle"
    Unpickler = _TolerantUnpickler
    load = staticmethod(lambda f, **k: _TolerantUnpickler(f, **k).load())
    loads = staticmethod(pickle.loads)
    dump = staticmethod(pickle.dump)
    dumps = staticmethod(pickle.dumps)
    HIGHEST_PROTOCOL = pickle.HIGHEST_PROTOCOL
    PicklingError = pickle.PicklingError
    UnpicklingError = pickle.UnpicklingError


def tolerant_torch_load(path: str) -> Dict[str, Any]:
    """``torch.load(path, map_location='cpu')`` for the reference's checkpoints without executing what they name:
    first torch's own ``weights_only`` loader (pure tensor files: the SD-1.5 first-conv file, official SD checkpoints);
    files that also pickle OmegaConf nodes (``config_dict`` of instancediffusion_sd15.pth) fall back to an allow-listing
    unpickler (``_TolerantUnpickler``) that turns the OmegaConf classes into attribute bags and refuses every global
    outside torch / containers / numpy."""
    try:
        return torch.load(path, map_location="cpu", weights_only=True)
    except Exception:
        return torch.load(path, map_location="cpu", weights_only=False, pickle_module=_tolerant_pickle)


def plain_config(obj: Any) -> Any:
    """OmegaConf DictConfig / ListConfig / value nodes (real or unpickled as bags) -> plain dict / list / scalars."""
    try:                                          # real OmegaConf objects, when the package is present
        from omegaconf import OmegaConf           # type: ignore
        if OmegaConf.is_config(obj):
            return OmegaConf.to_container(obj, resolve=True)
    except ImportError:
        pass
    if isinstance(obj, dict):
        return {plain_config(k): plain_config(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [plain_config(v) for v in obj]
    d = getattr(obj, "__dict__", None)
    if d is not None and "_content" in d:         # DictConfig / ListConfig container node
        return plain_config(d["_content"])
    if d is not None and "_val" in d:             # AnyNode / StringNode / IntegerNode ... value node
        return plain_config(d["_val"])
    return obj


ALLOWED_TARGET_ROOTS = ("ldm", "grounding_input", "instancediffusion_amd")


def check_target_namespace(cfg: Any):
    """Every ``target:`` of a config that came out of a checkpoint must live in this code base's namespaces."""
    if isinstance(cfg, dict):
        t = cfg.get("target")
        if isinstance(t, str) and t.split(".", 1)[0] not in ALLOWED_TARGET_ROOTS:
            raise ValueError(f"checkpoint config names target {t!r} outside {ALLOWED_TARGET_ROOTS}: refused")
        for v in cfg.values():
            check_target_namespace(v)
    elif isinstance(cfg, (list, tuple)):
        for v in cfg:
            check_target_namespace(v)


# ---- the reference functions -------------------------------------------------------------------------------------
def read_official_ckpt(ckpt_path: str) -> Dict[str, Dict[str, torch.Tensor]]:
    """utils/checkpoint.py:13-34: split an official SD checkpoint's flat ``state_dict`` into our sub-dicts."""
    state_dict = tolerant_torch_load(ckpt_path)["state_dict"]
    out: Dict[str, Dict[str, torch.Tensor]] = dict(model={}, text_encoder={}, autoencoder={}, unexpected={}, diffusion={})
    for k, v in state_dict.items():
        if k.startswith("model.diffusion_model"):
            out["model"][k.replace("model.diffusion_model.", "")] = v
        elif k.startswith("cond_stage_model"):
            out["text_encoder"][k.replace("cond_stage_model.", "")] = v
        elif k.startswith("first_stage_model"):
            out["autoencoder"][k.replace("first_stage_model.", "")] = v
        elif k in ["model_ema.decay", "model_ema.num_updates"]:
            out["unexpected"][k] = v
        else:
            out["diffusion"][k] = v
    return out


def load_model_ckpt(ckpt_path: str, args, device):
    """utils/checkpoint.py:224-248.  Returns (model, autoencoder, text_encoder, diffusion, config)."""
    saved_ckpt = tolerant_torch_load(ckpt_path)
    if hasattr(args, "test_config") and args.test_config != "":
        config = load_yaml(args.test_config)
        print("config for evaluation: ", config)
    else:
        config = plain_config(saved_ckpt["config_dict"])
        if isinstance(config, dict) and "_content" in config:
            config = config["_content"]
        check_target_namespace(config)              # a checkpoint's own config may only name classes of this code base

    model = instantiate_from_config(config["model"]).to(device).eval()
    autoencoder = instantiate_from_config(config["autoencoder"]).to(device).eval()
    text_encoder = instantiate_from_config(config["text_encoder"]).to(device).eval()
    diffusion = instantiate_from_config(config["diffusion"]).to(device)

    try:
        print("Loading ema")
        model.load_state_dict(saved_ckpt["ema"])
    except Exception:                               # the reference uses a bare except (missing key or mismatch)
        print("Loading non-ema model")
        model.load_state_dict(saved_ckpt["model"])
    autoencoder.load_state_dict(saved_ckpt["autoencoder"])
    text_encoder.load_state_dict(saved_ckpt["text_encoder"], strict=False)
    diffusion.load_state_dict(saved_ckpt["diffusion"])
    return model, autoencoder, text_encoder, diffusion, config
