"""Deterministic synthetic weights and inputs for the InstanceDiffusion sampling path.

There is no network in the build/bench environment, so neither the trained
``instancediffusion_sd15.pth`` nor SD-1.5/CLIP/ConvNeXt weights exist.  Parity and
throughput are therefore measured on *seeded random* weights.

The weights are a pure function of ``(key name, shape)`` -- NOT of module construction
order -- so the unmodified reference model (in the builder container), the CPU oracle
and the HIP engine can all be given bit-identical parameters via ``load_state_dict``.

Every parameter that the reference zero-initialises (``zero_module`` convs
``openaimodel.py:210-212,463``, ``attention.py:360``; ``alpha_attn/alpha_dense``
``attention.py:297-298``; ``scaleu_*`` ``openaimodel.py:442-443``; ``null_*_feature``
``text_grounding_net.py:92-102``) gets a non-zero value here, otherwise eps == 0 and
nothing is tested.
"""
from __future__ import annotations

import hashlib
import math
from typing import Dict, Mapping, Sequence, Tuple

import torch


def _seed_for(key: str, salt: int) -> int:
    h = hashlib.sha256(f"{salt}:{key}".encode()).digest()
    return int.from_bytes(h[:7], "little")


def _randn(key: str, shape: Sequence[int], salt: int) -> torch.Tensor:
    g = torch.Generator(device="cpu")
    g.manual_seed(_seed_for(key, salt))
    return torch.randn(tuple(shape), generator=g, dtype=torch.float32)


def synth_param(key: str, shape: Sequence[int], salt: int = 0) -> torch.Tensor:
    """One synthetic parameter.  Scales keep activations O(1) through the UNet."""
    shape = tuple(int(s) for s in shape)
    r = _randn(key, shape, salt)
    leaf = key.rsplit(".", 1)[-1]
    # --- special, reference-zero-initialised or learned-scalar parameters
    if leaf in ("alpha_attn", "alpha_dense"):
        return r * 0.5 + 0.6            # tanh(.) comfortably non-zero
    if key.startswith("scaleu_b_"):
        return r * 0.3
    if key.startswith("scaleu_s_"):
        return r * 0.5 - 0.3
    if leaf == "pos_embedding":
        return r * 0.02
    if leaf.startswith("null_") and leaf.endswith("_feature"):
        return r * 0.5
    if leaf == "gamma":                 # ConvNeXt layer-scale
        return r * 0.05 + 0.3
    # --- norm layers (GroupNorm / LayerNorm): 1-D weight near 1, bias near 0
    is_norm = any(t in key for t in (".norm", "in_layers.0.", "out_layers.0.", "out.0.", "downsample_layers.0.1.",
                                     "downsample_layers.1.0.", "downsample_layers.2.0.", "downsample_layers.3.0."))
    if len(shape) == 1 and is_norm:
        if leaf == "weight":
            return 1.0 + 0.1 * r
        return 0.05 * r
    if leaf == "bias":
        return 0.05 * r
    if leaf == "weight" and len(shape) >= 2:
        fan_in = 1
        for s in shape[1:]:
            fan_in *= s
        return r * (1.0 / math.sqrt(fan_in))
    return r * 0.1


def synth_state_dict(schema: Mapping[str, Sequence[int]], salt: int = 0) -> Dict[str, torch.Tensor]:
    """schema: ``{key: shape}`` (e.g. from ``model.state_dict()``)."""
    return {k: synth_param(k, tuple(v), salt) for k, v in schema.items()}


def schema_of(module: torch.nn.Module) -> Dict[str, Tuple[int, ...]]:
    return {k: tuple(v.shape) for k, v in module.state_dict().items()}


def synth_first_conv_sd(salt: int = 0) -> Dict[str, torch.Tensor]:
    """Stand-in for ``pretrained/SD_v1_5_input_conv_weight_bias.pth`` (``openaimodel.py:469-480``)."""
    return {
        "weight": synth_param("sd_first_conv.weight", (320, 4, 3, 3), salt + 17),
        "bias": synth_param("sd_first_conv.bias", (320,), salt + 17),
    }


# ------------------------------------------------------------------------------------------------
# Synthetic sampler inputs
# ------------------------------------------------------------------------------------------------

C1_BOXES = [  # demos/demo_cat_dog_robin.json, xywh/512 -> xyxy
    [0.0, 0.0996, 0.3496, 0.5488],
    [0.3496, 0.1992, 0.6484, 0.4980],
    [0.6484, 0.1992, 0.9980, 0.6973],
    [0.0, 0.6992, 1.0, 0.9980],
]


def random_boxes(n: int, g: torch.Generator) -> torch.Tensor:
    xy0 = torch.rand(n, 2, generator=g) * 0.7
    wh = torch.rand(n, 2, generator=g) * 0.15 + 0.15
    xy1 = (xy0 + wh).clamp(max=1.0)
    return torch.cat([xy0, xy1], dim=1)


def make_grounding_batch(batch: int, boxes: torch.Tensor, g: torch.Generator, *, max_objs: int = 30,
                         with_scribbles: bool = False, with_polygons: bool = False, with_segs: bool = False,
                         seg_size: int = 512, text_dim: int = 768) -> Dict[str, torch.Tensor]:
    """A ``prepare_batch``-shaped dict (``utils/input.py:41-125``): [B, 30, ...] padded tensors."""
    n = boxes.shape[0]
    out = {
        "boxes": torch.zeros(batch, max_objs, 4),
        "masks": torch.zeros(batch, max_objs),
        "text_embeddings": torch.zeros(batch, max_objs, text_dim),
        "points": torch.zeros(batch, max_objs, 2),
        "scribbles": torch.zeros(batch, max_objs, 40),
        "polygons": torch.zeros(batch, max_objs, 512),
        "segs": torch.zeros(batch, max_objs, seg_size, seg_size),
    }
    emb = torch.randn(1, n, text_dim, generator=g)
    out["text_embeddings"][:, :n] = emb
    out["boxes"][:, :n] = boxes
    out["masks"][:, :n] = 1.0
    out["points"][:, :n] = (boxes[:, :2] + boxes[:, 2:]) / 2.0

    def pts_in_box(npts):
        u = torch.rand(n, npts, 2, generator=g)
        p = boxes[:, None, :2] + u * (boxes[:, None, 2:] - boxes[:, None, :2])
        order = (p ** 2).sum(-1).argsort(dim=1)
        return torch.gather(p, 1, order[..., None].expand(-1, -1, 2)).reshape(n, -1)

    if with_scribbles:
        out["scribbles"][:, :n] = pts_in_box(20)
    if with_polygons:
        out["polygons"][:, :n] = pts_in_box(256)
    if with_segs:
        for i in range(n):
            x0, y0, x1, y1 = (boxes[i] * seg_size).round().int().tolist()
            out["segs"][:, i, y0:max(y1, y0 + 1), x0:max(x1, x0 + 1)] = 1.0
    return out


def instance_batch(full: Mapping[str, torch.Tensor], i: int) -> Dict[str, torch.Tensor]:
    """Single-instance grounding batch for MIS (``utils/input.py:130-144``): slot 0 = instance i."""
    out = {k: torch.zeros_like(v) for k, v in full.items()}
    for k in full:
        out[k][:, 0] = full[k][:, i]
    return out
