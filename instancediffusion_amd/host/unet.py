"""Host mirror of ``ldm/modules/diffusionmodules/openaimodel.py``: the InstanceDiffusion UNet denoiser.

Same constructor, attributes, state-dict keys and ``forward(input: dict) -> eps`` contract as the reference
(SURVEY.md §8b); the arithmetic runs in the HIP engine (``instancediffusion_amd.engine.UNetEngine``) on an
MI355X -- there is no CPU / eager fallback: calling ``forward`` without a GPU or without the built
``libidf_gfx950.so`` raises.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
import torch.nn as nn

from .attention import SpatialTransformer
from .config import instantiate_from_config
from .params import Affine, Conv, Dense, Slots


class TimestepBlock(nn.Module):
    """Marker base class (openaimodel.py:50-59)."""


class TimestepEmbedSequential(Slots, TimestepBlock):
    """openaimodel.py:62-79 -- ordered children 0..n-1."""

    def __init__(self, *layers: nn.Module):
        super().__init__({i: m for i, m in enumerate(layers)})


class Upsample(nn.Module):
    """nearest x2 then 3x3 conv (openaimodel.py:82-110)."""

    def __init__(self, channels, use_conv, dims=2, out_channels=None, padding=1):
        super().__init__()
        assert use_conv and dims == 2
        self.channels = channels
        self.out_channels = out_channels or channels
        self.conv = Conv(channels, self.out_channels, 3)


class Downsample(nn.Module):
    """3x3 stride-2 pad-1 conv (openaimodel.py:115-141)."""

    def __init__(self, channels, use_conv, dims=2, out_channels=None, padding=1):
        super().__init__()
        assert use_conv and dims == 2
        self.channels = channels
        self.out_channels = out_channels or channels
        self.op = Conv(channels, self.out_channels, 3)


class ResBlock(TimestepBlock):
    """openaimodel.py:144-257 (no up/down, use_scale_shift_norm=False)."""

    def __init__(self, channels, emb_channels, dropout, out_channels=None, use_conv=False,
                 use_scale_shift_norm=False, dims=2, use_checkpoint=False, up=False, down=False):
        super().__init__()
        assert not (use_conv or use_scale_shift_norm or up or down) and dims == 2
        self.channels = channels
        self.emb_channels = emb_channels
        self.out_channels = out_channels or channels
        self.in_layers = Slots({0: Affine(channels), 2: Conv(channels, self.out_channels, 3)})
        self.emb_layers = Slots({1: Dense(emb_channels, self.out_channels)})
        self.out_layers = Slots({0: Affine(self.out_channels), 3: Conv(self.out_channels, self.out_channels, 3, zero=True)})
        self.skip_connection = nn.Identity() if self.out_channels == channels else Conv(channels, self.out_channels, 1)


class UNetModel(nn.Module):
    """openaimodel.py:308-566."""

    def __init__(self, image_size, in_channels, model_channels, out_channels, num_res_blocks, attention_resolutions,
                 dropout=0, channel_mult=(1, 2, 4, 8), conv_resample=True, dims=2, use_checkpoint=False, num_heads=8,
                 use_scale_shift_norm=False, transformer_depth=1, context_dim=None, fuser_type=None,
                 inpaint_mode=False, grounding_downsampler=None, grounding_tokenizer=None, sd_v1_5=False,
                 efficient_attention=False):
        super().__init__()
        # openaimodel.py:349 accepts all three names; BasicTransformerBlock builds a GatedSelfAttentionDense whatever the
        # value (attention.py:325), so all three name the same network here too
        assert fuser_type in ["gatedSA", "gatedSA2", "gatedCA"]
        assert dims == 2 and conv_resample and not use_scale_shift_norm and transformer_depth == 1
        self.image_size = image_size
        self.in_channels = in_channels
        self.model_channels = model_channels
        self.out_channels = out_channels
        self.num_res_blocks = num_res_blocks
        self.attention_resolutions = list(attention_resolutions)
        self.dropout = dropout
        self.channel_mult = list(channel_mult)
        self.conv_resample = conv_resample
        self.use_checkpoint = use_checkpoint
        self.num_heads = num_heads
        self.context_dim = context_dim
        self.fuser_type = fuser_type
        self.inpaint_mode = inpaint_mode
        self.sd_v1_5 = sd_v1_5
        self.efficient_attention = efficient_attention
        self.grounding_tokenizer_input = None      # set externally (inference.py:307)
        self.enable_freeu = False
        self.enable_scaleu = True
        self.enable_se_scaleu = False
        self.first_conv_restorable = True
        self.first_conv_sd_override: Optional[Dict[str, torch.Tensor]] = None   # tests/bench: synthetic SD conv
        self.compute_dtype = torch.bfloat16        # 16-bit storage / MFMA input type of the HIP engine (or float16)

        ted = model_channels * 4
        self.time_embed = Slots({0: Dense(model_channels, ted), 2: Dense(ted, ted)})

        def st(ch):
            return SpatialTransformer(ch, key_dim=context_dim, value_dim=context_dim, n_heads=num_heads,
                                      d_head=ch // num_heads, depth=transformer_depth, fuser_type=fuser_type,
                                      use_checkpoint=use_checkpoint, efficient_attention=efficient_attention)

        self.input_blocks = nn.ModuleList([TimestepEmbedSequential(Conv(in_channels, model_channels, 3))])
        chans = [model_channels]
        ch, ds = model_channels, 1
        for level, mult in enumerate(self.channel_mult):
            for _ in range(num_res_blocks):
                layers: List[nn.Module] = [ResBlock(ch, ted, dropout, out_channels=mult * model_channels)]
                ch = mult * model_channels
                if ds in self.attention_resolutions:
                    layers.append(st(ch))
                self.input_blocks.append(TimestepEmbedSequential(*layers))
                chans.append(ch)
            if level != len(self.channel_mult) - 1:
                self.input_blocks.append(TimestepEmbedSequential(Downsample(ch, True, out_channels=ch)))
                chans.append(ch)
                ds *= 2

        self.middle_block = TimestepEmbedSequential(ResBlock(ch, ted, dropout), st(ch), ResBlock(ch, ted, dropout))

        self.output_blocks = nn.ModuleList([])
        idx = 0
        for level, mult in list(enumerate(self.channel_mult))[::-1]:
            for i in range(num_res_blocks + 1):
                ich = chans.pop()
                layers = [ResBlock(ch + ich, ted, dropout, out_channels=model_channels * mult)]
                self.register_parameter(f"scaleu_b_{idx}", nn.Parameter(torch.zeros(ch)))
                self.register_parameter(f"scaleu_s_{idx}", nn.Parameter(torch.zeros(1)))
                idx += 1
                ch = model_channels * mult
                if ds in self.attention_resolutions:
                    layers.append(st(ch))
                if level and i == num_res_blocks:
                    layers.append(Upsample(ch, True, out_channels=ch))
                    ds //= 2
                self.output_blocks.append(TimestepEmbedSequential(*layers))

        self.out = Slots({0: Affine(ch), 2: Conv(model_channels, out_channels, 3, zero=True)})
        self.position_net = instantiate_from_config(grounding_tokenizer)

        self._engine = None
        self._engine_version = 0

    # ---- state management -------------------------------------------------------------------------
    def load_state_dict(self, *a, **k):
        r = super().load_state_dict(*a, **k)
        self.invalidate_engine()
        return r

    def invalidate_engine(self):
        """Call after mutating parameters in place: the engine re-packs its bf16 weight images lazily."""
        self._engine = None

    def first_conv_file(self):
        """Where the SD first-conv weights (openaimodel.py:473-476: a 48-KB ``{weight, bias}`` file the reference repository ships
        under ``pretrained/``) are looked for: the reference's cwd-relative path first, then ``$IDF_PRETRAINED_DIR``, then
        ``pretrained/`` next to (and the directory of) the checkpoint this model was loaded from (``self.ckpt_path``, set by
        ``utils.checkpoint.load_model_ckpt``), then this repository's own ``pretrained/``.  Returns the first existing path,
        else raises FileNotFoundError naming every place tried."""
        import os
        name = "SD_v1_5_input_conv_weight_bias.pth" if self.sd_v1_5 else "SD_input_conv_weight_bias.pth"
        tried = [os.path.join("pretrained", name)]
        if os.environ.get("IDF_PRETRAINED_DIR"):
            tried.append(os.path.join(os.environ["IDF_PRETRAINED_DIR"], name))
        ck = getattr(self, "ckpt_path", None)
        if ck:
            d = os.path.dirname(os.path.abspath(ck))
            tried += [os.path.join(d, "pretrained", name), os.path.join(d, name)]
        tried.append(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "pretrained", name))
        for t in tried:
            if os.path.isfile(t):
                return t
        raise FileNotFoundError(
            f"{name} not found (tried: {', '.join(tried)}).  The samplers swap the UNet's first conv for Stable Diffusion's own "
            "as soon as the gated-self-attention gate alpha reaches 0 (openaimodel.py:469-480); the file ships with the reference "
            "repository under pretrained/ -- copy it there, next to the checkpoint, or point IDF_PRETRAINED_DIR at its directory "
            "(or set model.first_conv_sd_override = {'weight': ..., 'bias': ...}).")

    def check_first_conv_available(self):
        """Fail BEFORE a sampling run instead of at the step where alpha first reaches 0 (step 40 of 50 with the default
        alpha_type [0.8, 0, 0.2]): called by the samplers when their alpha schedule contains a 0."""
        if self.first_conv_restorable and self.first_conv_sd_override is None and not getattr(self, "_first_conv_swapped", False):
            self.first_conv_file()

    def restore_first_conv_from_SD(self):
        """openaimodel.py:469-480: swap input_blocks[0][0] for the SD-1.5 first conv (never undone).
        The reference re-reads the 48 KB file on EVERY alpha==0 step; here the swap is idempotent."""
        if not self.first_conv_restorable:
            return
        if getattr(self, "_first_conv_swapped", False):
            return
        if self.first_conv_sd_override is not None:
            sdw = self.first_conv_sd_override
        else:
            sdw = torch.load(self.first_conv_file(), map_location="cpu", weights_only=True)
        old = self.input_blocks[0][0]
        self.first_conv_state_dict = {k: v.detach().clone() for k, v in old.state_dict().items()}
        new = Conv(self.in_channels, self.model_channels, 3)
        new.load_state_dict({"weight": sdw["weight"], "bias": sdw["bias"]})
        new.to(old.weight.device)
        self.input_blocks[0][0] = new
        self._first_conv_swapped = True
        if self._engine is not None:
            self._engine.repack_first_conv(new)

    # ---- execution --------------------------------------------------------------------------------
    @property
    def engine(self):
        if self._engine is None:
            from ..engine import UNetEngine      # imports the C-ABI loader; raises if the .so is missing
            self._engine = UNetEngine(self, dtype=self.compute_dtype)
        return self._engine

    def forward_single_input(self, input):
        """openaimodel.py:482-563 contract: dict{x, timesteps, context[, grounding_input]} -> eps [B,4,H,W] fp32."""
        if "grounding_input" in input:
            g = input["grounding_input"]
        else:
            g = self.grounding_tokenizer_input.get_null_input()
        return self.engine.forward(input["x"], input["timesteps"], input["context"], g)

    def forward(self, input):
        return self.forward_single_input(input)
