"""Host mirror of ``utils/input.py``: the grounding-input builder that feeds the sampling path.

Same functions, argument meaning and output dict as the reference (``prepare_batch`` utils/input.py:41-125,
``prepare_instance_meta`` :128-144, ``get_attmask_w_box`` :34-37, ``complete_mask`` :22-31, ``convert_points`` :152-159,
``create_zero_input_tensors`` :9-20).  MI355X-first differences, none of which changes a value:
  * every tensor is assembled ONCE on the host and crosses PCIe once; the batch dimension of the two large tensors
    (``segs`` 30 x 512 x 512 fp32 = 31 MB per sample, ``att_masks``) is a stride-0 ``expand`` view on the device instead
    of the reference's ``.repeat(batch, ...)`` copies (31 MB x batch x (N+1) instance inputs in HBM and on the bus) --
    ``.shape`` / indexing / ``.sum()`` behave identically, and the engine's tokenizer detects the broadcast and runs the
    ConvNeXt mask backbone once per distinct mask stack;
  * text features come from ``get_clip_feature`` (utils/model.py:130-152 mirror below), which accepts either the HF
    CLIP model + processor the reference passes, or any object with ``.pooled(phrase) -> [768]`` (offline stand-in).
COCO-polygon helpers of the reference file (``annToMask``, ``prepare_scribble_and_instmask``) are dataset-evaluation
utilities that need pycocotools and are not on the inference path; they are not mirrored.
"""
from __future__ import annotations

from typing import Any, Dict, List, Optional

import numpy as np
import torch


def create_zero_input_tensors(max_objs, n_polygon_points, n_scribble_points):
    """utils/input.py:9-20."""
    masks = torch.zeros(max_objs)
    text_masks = torch.zeros(max_objs)
    text_embeddings = torch.zeros(max_objs, 768)
    boxes_embeddings = torch.zeros(max_objs, 4)
    polygons_embeddings = torch.zeros(max_objs, n_polygon_points * 2)
    scribbles_embeddings = torch.zeros(max_objs, n_scribble_points * 2)
    segs_embeddings = torch.zeros(max_objs, 512, 512)
    points_embeddings = torch.zeros(max_objs, 2)
    return (boxes_embeddings, masks, text_masks, text_embeddings, polygons_embeddings, scribbles_embeddings,
            segs_embeddings, points_embeddings)


def complete_mask(has_mask, max_objs):
    """utils/input.py:22-31 (also inference.py:26-36)."""
    mask = torch.ones(1, max_objs)
    if has_mask is None:
        return mask
    if type(has_mask) == int or type(has_mask) == float:
        return mask * has_mask
    for idx, value in enumerate(has_mask):
        mask[0, idx] = value
    return mask


def get_attmask_w_box(att_masks, idx, box, image_size):
    """utils/input.py:34-37.  NOTE the reference's index order, kept: dim-0 of the mask is sliced with the box's
    x-range and dim-1 with its y-range."""
    x1, y1, x2, y2 = (int(np.round(box[0] * image_size)), int(np.round(box[1] * image_size)),
                      int(np.round(box[2] * image_size)), int(np.round(box[3] * image_size)))
    att_masks[idx][x1:x2, y1:y2] = 1
    return att_masks


def get_clip_feature(model, processor, input, is_image=False):
    """utils/model.py:130-152: pooled CLIP text feature [1, 768] of a phrase (None for a None phrase)."""
    if input is None:
        return None
    if hasattr(model, "pooled"):                                   # offline stand-in encoder
        return model.pooled(input).reshape(1, -1)
    inputs = processor(text=input, return_tensors="pt", padding=True)
    dev = next(model.parameters()).device
    inputs["input_ids"] = inputs["input_ids"].to(dev)
    inputs["pixel_values"] = torch.ones(1, 3, 224, 224, device=dev)          # placeholder, as in the reference
    inputs["attention_mask"] = inputs["attention_mask"].to(dev)
    outputs = model(**inputs)
    return outputs.text_model_output.pooler_output


def batch_to_device(batch, device):
    """dataset/jsondataset.py:59-69."""
    for k in batch:
        if isinstance(batch[k], torch.Tensor):
            batch[k] = batch[k].to(device)
        if isinstance(batch[k], list):
            for i in range(len(batch[k])):
                if isinstance(batch[k][i], dict):
                    for j in batch[k][i]:
                        if isinstance(batch[k][i][j], torch.Tensor):
                            batch[k][i][j] = batch[k][i][j].to(device)
    return batch


_BIG = ("segs", "att_masks")


def _batched(d: Dict[str, torch.Tensor], batch: int, device) -> Dict[str, torch.Tensor]:
    """[..] -> [batch, ..] on ``device``: small tensors are real copies (the reference's ``repeat``); the large ones are
    moved once and broadcast over the batch with a stride-0 view."""
    out = {}
    for k, v in d.items():
        v = v.to(device)
        if k in _BIG:
            out[k] = v.unsqueeze(0).expand(batch, *v.shape)
        else:
            out[k] = v.unsqueeze(0).repeat(batch, *([1] * v.dim()))
    return out


def _t(v) -> torch.Tensor:
    return torch.as_tensor(np.asarray(v), dtype=torch.float32)


def _padded_slots(entries, max_objs: int, text_mask_spec, att_rows=None, image_size: int = 64) -> Dict[str, torch.Tensor]:
    """One sample's [max_objs, ...] grounding tensors.  ``entries``: per live instance a tuple
    (box, text_feature | None, polygon | None, scribble | None, seg | None, point | None); slot k takes entry k, the
    remaining slots stay zero (``create_zero_input_tensors``).  ``att_rows``: optional per-entry [S, S] visibility
    masks for the masked gated self-attention."""
    (boxes, masks, text_masks, text_emb, polygons, scribbles, segs, points) = create_zero_input_tensors(
        max_objs, N_POLYGON_POINTS, N_SCRIBBLE_POINTS)
    fields = (None, None, polygons, scribbles, segs, points)
    for k, entry in enumerate(entries):
        boxes[k] = _t(entry[0])
        masks[k] = 1
        if entry[1] is not None:
            text_emb[k] = entry[1].detach().float().cpu().reshape(-1)
            text_masks[k] = 1
        for dst, val in zip(fields[2:], entry[2:]):
            if val is not None:
                dst[k] = _t(val).reshape(dst[k].shape)
    out = dict(boxes=boxes, masks=masks, text_masks=text_masks * complete_mask(text_mask_spec, max_objs)[0],
               text_embeddings=text_emb, polygons=polygons, scribbles=scribbles, segs=segs, points=points)
    if att_rows is not None:
        att = torch.zeros(max_objs, image_size, image_size)
        for k, row in enumerate(att_rows):
            att[k] = row
        out["att_masks"] = att
    return out


N_SCRIBBLE_POINTS, N_POLYGON_POINTS = 20, 256


@torch.no_grad()
def prepare_batch(meta, batch=1, max_objs=30, model=None, processor=None, image_size=64, use_masked_att=False,
                  device="cuda"):
    """utils/input.py:39-125: meta (phrases, locations, polygons, scribbles, segs, points[, text_mask,
    instance_meta]) -> ``{boxes, masks, text_masks, text_embeddings, polygons, scribbles, segs, points[, att_masks,
    instance_meta: [same dict per instance]]}``, every tensor [batch, max_objs, ...] on ``device``."""
    phrases = meta.get("phrases")
    phrases = [None] * len(meta["locations"]) if phrases is None else phrases
    text_features = [get_clip_feature(model, processor, phrase, is_image=False) for phrase in phrases]
    entries = list(zip(meta["locations"], text_features, meta.get("polygons"), meta.get("scribbles"), meta.get("segs"),
                       meta.get("points")))
    att_rows = None
    if use_masked_att:                                   # one box-shaped visibility plane per instance (:34-37)
        planes = torch.zeros(len(entries), image_size, image_size)
        for k, e in enumerate(entries):
            get_attmask_w_box(planes, k, e[0], image_size)
        att_rows = list(planes)
    out: Dict[str, Any] = _batched(_padded_slots(entries, max_objs, meta.get("text_mask"), att_rows, image_size),
                                   batch, device)
    if "instance_meta" in meta:
        # the Multi-instance Sampler's per-instance inputs: instance i alone, in slot 0 (:92-120)
        out["instance_meta"] = []
        for i, im in enumerate(meta["instance_meta"]):
            entry = (im["locations"][0], text_features[i], im["polygons"][0], im["scribbles"][0], im["segs"][0],
                     im["points"][0])
            one = _padded_slots([entry], max_objs, im.get("text_mask"), None if att_rows is None else [att_rows[i]],
                                image_size)
            out["instance_meta"].append(_batched(one, batch, device))
    return out


@torch.no_grad()
def prepare_instance_meta(test_info, i, file_name=None, save_folder_name=None, ckpt=None):
    """utils/input.py:127-144: the single-instance meta the Multi-instance Sampler's input i+1 is built from."""
    return {
        "ckpt": test_info.get("ckpt", None),
        "phrases": [test_info["phrases"][i]],
        "locations": [test_info["locations"][i]],
        "polygons": [test_info["polygons"][i]],
        "segs": [test_info["segs"][i]],
        "scribbles": [test_info["scribbles"][i]],
        "points": [test_info["points"][i]],
        "alpha_type": test_info["alpha_type"],
        "prompt": test_info["phrases"][i],
        "file_name": file_name,
        "save_folder_name": save_folder_name,
    }


def convert_points(points: List[float], img_info: Dict[str, int]) -> List[float]:
    """utils/input.py:152-159: absolute [x0,y0,x1,y1,...] -> relative, clipped at 1."""
    for i in range(len(points)):
        if i % 2 == 0:
            points[i] = min(points[i] / img_info["width"], 1.0)
        else:
            points[i] = min(points[i] / img_info["height"], 1.0)
    return points


# ---- demo-JSON parsing (inference.py:189-297) --------------------------------------------------------------------
def equally_spaced_sampling_with_replacement(points_list, sample_size):
    """dataset/decode_item.py:79-100."""
    if sample_size <= len(points_list):
        gap_size = len(points_list) // sample_size
        return [points_list[i * gap_size] for i in range(sample_size)]
    return [points_list[(i * len(points_list)) // sample_size % len(points_list)] for i in range(sample_size)]


def reorder_scribbles(scribbles):
    """dataset/decode_item.py:102-108 (called by inference.py:258 on the LIST of per-instance scribbles, as there)."""
    center = np.array([0, 0])
    scribbles = sorted(scribbles, key=lambda x: np.linalg.norm(np.array(x) - center))
    scribbles = equally_spaced_sampling_with_replacement(scribbles, 20)
    return sorted(scribbles, key=lambda x: np.linalg.norm(np.array(x) - center))


def rescale_box(bbox, width, height):
    """inference.py:132-137: xywh pixels -> normalised xyxy."""
    return [bbox[0] / width, bbox[1] / height, (bbox[0] + bbox[2]) / width, (bbox[1] + bbox[3]) / height]


def meta_from_demo_json(data: dict, alpha: float, ckpt: Optional[str] = None, save_folder_name: Optional[str] = None) -> dict:
    """inference.py:189-297: demo JSON (caption, width, height, annos[{bbox, mask, point, scribble, caption}]) -> the
    ``meta`` dict ``prepare_batch`` consumes.  As in the reference script, instance masks are NOT used (it re-initialises
    its mask list to zeros at :249), so ``segs`` and ``polygons`` are zero and scribbles default to zeros."""
    W, H = data["width"], data["height"]
    annos = data["annos"]
    boxes = [a["bbox"] if "bbox" in a else [0, 0, 0, 0] for a in annos]
    points_list = [a["point"] for a in annos if "point" in a]
    scribbles_list = [a["scribble"] for a in annos if "scribble" in a]
    phrases = [a["caption"] for a in annos]
    locations = [rescale_box(b, W, H) for b in boxes]
    if len(points_list) == 0:
        points = [[(b[0] + b[2]) / 2.0, (b[1] + b[3]) / 2.0] for b in locations]
    else:
        points = [[p[0] / float(W), p[1] / float(H)] for p in points_list]
    n = len(locations)
    if len(scribbles_list) == 0:
        scribbles = [[0.0] * 40 for _ in range(n)]                 # sample_random_points_from_mask(zeros) -> zeros
    else:
        scribbles = [[[s[0] / float(W), s[1] / float(H)] for s in sc] for sc in scribbles_list]
        scribbles = reorder_scribbles(scribbles)
    polygons = [[0.0] * 512 for _ in range(n)]
    segs = np.zeros((n, 512, 512), dtype=np.float32)
    return dict(ckpt=ckpt, prompt=data["caption"], phrases=phrases, polygons=polygons, scribbles=scribbles, segs=segs,
                locations=locations, points=points, alpha_type=[alpha, 0.0, 1 - alpha],
                save_folder_name=save_folder_name)
