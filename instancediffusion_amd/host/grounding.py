"""Host mirror of ``grounding_input/text_grounding_tokinzer_input.py``: the boundary object that turns a
``prepare_batch`` dict into the UNet's ``grounding_input`` and manufactures the all-zero null input used for the
classifier-free-guidance (unconditional) pass."""
from __future__ import annotations

import torch as th


class GroundingNetInput:
    def __init__(self):
        self.set = False
        self.return_att_masks = False
        self.image_size = 64
        self.return_att_masks32 = False

    def prepare(self, batch, image_size=64, device=None, dtype=None, return_att_masks=False):
        """Pass the batch tensors through (renaming text_embeddings -> positive_embeddings) and remember the
        shapes/device/dtype so ``get_null_input`` can be called later (text_grounding_tokinzer_input.py:13-56)."""
        self.set = True
        self.return_att_masks = return_att_masks
        emb = batch["text_embeddings"]
        self.batch, self.max_box, self.in_dim = emb.shape
        self.device, self.dtype = emb.device, emb.dtype
        self.dim_scribbles = batch["scribbles"].shape[-1]
        self.dim_polygons = batch["polygons"].shape[-1]
        self.dim_segs = batch["segs"].shape[-1]
        out = {
            "boxes": batch["boxes"], "masks": batch["masks"], "positive_embeddings": emb,
            "scribbles": batch["scribbles"], "polygons": batch["polygons"], "segs": batch["segs"],
            "points": batch["points"],
        }
        if return_att_masks:
            assert "att_masks" in batch
            out["att_masks"] = batch["att_masks"]
        return out

    def get_null_input(self, batch=None, device=None, dtype=None):
        """All-zero grounding of the remembered shapes (:59-94).  Requires a prior ``prepare``."""
        assert self.set, "not set yet, cannot call this funcion"
        b = self.batch if batch is None else batch
        device = self.device if device is None else device
        dtype = self.dtype if dtype is None else dtype
        key = (b, str(device), dtype, self.max_box, self.dim_segs, self.dim_scribbles, self.dim_polygons,
               self.return_att_masks)
        cached = getattr(self, "_null_cache", None)
        if cached is not None and cached[0] == key:
            return cached[1]        # identical object every call -> the engine's token cache hits (exact hoist)

        def z(*shape):
            return th.zeros(*shape, dtype=dtype, device=device)
        out = {
            "boxes": z(b, self.max_box, 4), "masks": z(b, self.max_box),
            "positive_embeddings": z(b, self.max_box, self.in_dim),
            "scribbles": z(b, self.max_box, self.dim_scribbles), "polygons": z(b, self.max_box, self.dim_polygons),
            # 31 MB per sample at 512^2: one zero plane set, broadcast over the batch (stride-0 view, read-only use)
            "segs": z(1, self.max_box, self.dim_segs, self.dim_segs).expand(b, -1, -1, -1), "points": z(b, self.max_box, 2),
        }
        if self.return_att_masks:
            out["att_masks"] = z(b, self.max_box, self.image_size, self.image_size)
        self._null_cache = (key, out)
        return out
