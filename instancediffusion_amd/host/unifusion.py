"""Host mirror of ``ldm/modules/diffusionmodules/text_grounding_net.py`` (UniFusion instance tokenizer) and the
ConvNeXt-T parameter tree of ``convnext.py`` -- parameter containers + eval-mode drop flags.

The tokens are computed by the HIP engine (``engine.Tokenizer``) ONCE per distinct grounding input: UniFusion
depends on neither ``x`` nor ``t`` and is deterministic in eval mode (text_grounding_net.py:206-207).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .params import Affine, Conv, Dense, Slots, mlp3

CONVNEXT_DEPTHS = (3, 3, 9, 3)
CONVNEXT_DIMS = (96, 192, 384, 768)


class _CNBlock(nn.Module):
    """convnext.py:15-50: dwconv7x7 -> LN -> Linear(4x) -> GELU -> Linear -> gamma * x -> + input."""

    def __init__(self, dim: int):
        super().__init__()
        self.dwconv = Conv(dim, dim, 7, groups=dim)
        self.norm = Affine(dim)
        self.pwconv1 = Dense(dim, 4 * dim)
        self.pwconv2 = Dense(4 * dim, dim)
        self.gamma = nn.Parameter(1e-6 * torch.ones(dim))


class ConvNeXtTiny(nn.Module):
    """convnext.py:52-110 (forward_features only; no head)."""

    def __init__(self, in_chans: int = 3):
        super().__init__()
        dims = CONVNEXT_DIMS
        self.downsample_layers = nn.ModuleList()
        self.downsample_layers.append(Slots({0: Conv(in_chans, dims[0], 4), 1: Affine(dims[0])}))
        for i in range(3):
            self.downsample_layers.append(Slots({0: Affine(dims[i]), 1: Conv(dims[i], dims[i + 1], 2)}))
        self.stages = nn.ModuleList(
            [Slots({j: _CNBlock(dims[i]) for j in range(CONVNEXT_DEPTHS[i])}) for i in range(4)])


class UniFusion(nn.Module):
    """text_grounding_net.py:7-102.  Token order in ``objs``: 30 box, 30 point, 30 scribble, 30 polygon, 64 seg."""

    N_SCRIBBLE_POINTS = 20
    N_POLYGON_POINTS = 256
    FOURIER_FREQS = 16
    NUM_SEG_TOKENS = 64
    CONVNEXT_FEATURE_DIM = 3072

    def __init__(self, in_dim, out_dim, mid_dim=3072, fourier_freqs=8,
                 train_add_boxes=True, train_add_points=True, train_add_scribbles=True, train_add_masks=True,
                 test_drop_boxes=False, test_drop_points=False, test_drop_scribbles=True, test_drop_masks=False,
                 use_seperate_tokenizer=True):
        super().__init__()
        if not (train_add_boxes and train_add_points and train_add_scribbles and train_add_masks
                and use_seperate_tokenizer):
            raise NotImplementedError("only the configuration used by every reference YAML is built "
                                      "(all modalities on, separate tokenizers)")
        self.in_dim, self.out_dim, self.mid_dim = in_dim, out_dim, mid_dim
        f = self.FOURIER_FREQS
        self.position_dim = f * 2 * 4
        self.point_dim = f * 2 * 2
        self.scribble_dim = f * 2 * self.N_SCRIBBLE_POINTS * 2
        self.polygon_dim = f * 2 * self.N_POLYGON_POINTS * 2
        self.resize_input = 512
        self.num_tokens = self.NUM_SEG_TOKENS
        self.convnext_feature_dim = self.CONVNEXT_FEATURE_DIM
        self.test_drop_boxes = test_drop_boxes
        self.test_drop_points = test_drop_points
        self.test_drop_scribbles = test_drop_scribbles
        self.test_drop_masks = test_drop_masks
        self.test_drop_segs = test_drop_masks

        self.in_conv = Conv(30, 3, 3)
        self.convnext_tiny_backbone = ConvNeXtTiny()
        self.pos_embedding = nn.Parameter(torch.empty(1, self.num_tokens, self.convnext_feature_dim))
        if self.pos_embedding.device.type != "meta":
            with torch.no_grad():
                self.pos_embedding.normal_(std=0.02)
        dims = [in_dim + self.position_dim, in_dim + self.point_dim, in_dim + self.scribble_dim,
                in_dim + self.polygon_dim, self.convnext_feature_dim]
        self.linears_list = nn.ModuleList([mlp3(d, mid_dim, out_dim) for d in dims])
        self.null_positive_feature = nn.Parameter(torch.zeros(in_dim))
        self.null_position_feature = nn.Parameter(torch.zeros(self.position_dim))
        self.null_point_feature = nn.Parameter(torch.zeros(self.point_dim))
        self.null_scribble_feature = nn.Parameter(torch.zeros(self.scribble_dim))
        self.null_polygon_feature = nn.Parameter(torch.zeros(self.polygon_dim))
        self.null_seg_feature = nn.Parameter(torch.zeros(self.convnext_feature_dim))

    def eval_drops(self):
        """(drop_point, drop_box, drop_scribble, drop_polygons, drop_segs) in eval mode
        (reset_dropout_test :104-116 + the all-dropped rule :211-213)."""
        dp, db = self.test_drop_points, self.test_drop_boxes
        ds, dg, dsg = self.test_drop_scribbles, self.test_drop_masks, self.test_drop_masks
        if dp and db and ds and dg and dsg:
            db = False
        return dp, db, ds, dg, dsg

    @property
    def drop_box_mask(self) -> bool:
        _, db, _, dg, _ = self.eval_drops()
        return bool(db and dg)
