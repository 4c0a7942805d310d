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


# Globals a reference checkpoint legitimately pickles, as EXACT (module, name) pairs: the tensor / parameter rebuild
# functions and storage classes torch.save emits, torch dtypes, containers, numpy array / scalar reconstruction, and
# the two plain-data classes training scripts leave in ``config_dict`` / ``args``.  A module ROOT is never enough:
# torch.utils.collect_env.run, torch.hub.load or numpy.testing._private.utils.runstring are all "inside torch / numpy"
# and execute what a pickle hands them.  The OmegaConf node classes of ``config_dict`` and the
# pytorch_lightning callback objects of official SD checkpoints become inert attribute bags (never imported, never
# called).  Everything else is refused.
_SAFE_BUILTINS = {"dict", "list", "tuple", "set", "frozenset", "int", "float", "bool", "str", "bytes", "bytearray",
                  "complex", "slice", "range", "object"}
_TORCH_STORAGES = {"DoubleStorage", "FloatStorage", "HalfStorage", "BFloat16Storage", "LongStorage", "IntStorage",
                   "ShortStorage", "CharStorage", "ByteStorage", "BoolStorage", "ComplexFloatStorage",
                   "ComplexDoubleStorage", "UntypedStorage"}
_SAFE_GLOBALS = {
    ("torch._utils", "_rebuild_tensor_v2"), ("torch._utils", "_rebuild_tensor"), ("torch._utils", "_rebuild_parameter"),
    ("torch._utils", "_rebuild_parameter_with_state"), ("torch._tensor", "_rebuild_from_type_v2"),
    ("torch", "Tensor"), ("torch", "Size"), ("torch", "device"), ("torch.nn.parameter", "Parameter"),
    ("torch.storage", "UntypedStorage"), ("torch.storage", "TypedStorage"),
    ("collections", "OrderedDict"), ("collections", "defaultdict"),
    ("numpy", "ndarray"), ("numpy", "dtype"),
    ("numpy.core.multiarray", "_reconstruct"), ("numpy.core.multiarray", "scalar"),
    ("numpy._core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "scalar"),
    ("_codecs", "encode"),                          # numpy arrays pickled with protocol 2 carry their bytes through it
    ("typing", "Any"), ("argparse", "Namespace"), ("pathlib", "PosixPath"), ("pathlib", "PurePosixPath"),
}
# (torch.storage._load_from_bytes is deliberately absent: it unpickles a nested stream with the stock pickle.)
_BAG_MODULE_ROOTS = ("omegaconf", "pytorch_lightning", "lightning", "lightning_fabric")


class _TolerantUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        root = module.split(".", 1)[0]
        if root in _BAG_MODULE_ROOTS:
            return type(name, (_Bag,), {"__module__": module})
        if module == "builtins":
            if name in _SAFE_BUILTINS:
                return super().find_class(module, name)
            raise pickle.UnpicklingError(f"checkpoint names builtins.{name}: refused")
        if (module, name) in _SAFE_GLOBALS or (module == "torch" and name in _TORCH_STORAGES):
            return super().find_class(module, name)
        if module == "torch" and isinstance(getattr(torch, name, None), torch.dtype):
            return getattr(torch, name)                  # torch.float32 ... pickle as the global torch.<name>
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
    a file that loader REFUSES because it also pickles non-tensor classes (the OmegaConf ``config_dict`` of
    instancediffusion_sd15.pth, pytorch_lightning callbacks) goes through ``_TolerantUnpickler``, which resolves an exact
    allow-list of (module, name) globals, turns the config / callback classes into inert attribute bags and refuses every
    other global.  I/O errors and corrupt files are NOT retried: only the strict loader's refusal of a global is."""
    try:
        return torch.load(path, map_location="cpu", weights_only=True)
    except pickle.UnpicklingError as strict_refusal:
        try:
            return torch.load(path, map_location="cpu", weights_only=False, pickle_module=_tolerant_pickle)
        except pickle.UnpicklingError as e:
            raise pickle.UnpicklingError(f"{e} (torch's weights_only loader had refused the file: "
                                         f"{str(strict_refusal).splitlines()[0]})") from strict_refusal


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
    model.ckpt_path = ckpt_path          # UNetModel.first_conv_file looks next to the checkpoint for the SD first-conv file
    return model, autoencoder, text_encoder, diffusion, config
