"""Host mirror of ``ldm/models/autoencoder.py`` + the VAE blocks of ``ldm/modules/diffusionmodules/model.py``.

SURVEY.md §8 row f-2: ``AutoencoderKL.decode`` is the step right after the sampling path (reference
``inference.py:95``: ``samples_fake = autoencoder.decode(samples_fake)``).  Same constructor, same state-dict keys
(``encoder.*``, ``decoder.*``, ``quant_conv.*``, ``post_quant_conv.*`` -- 248 tensors for the SD-1.5 KL-f8 config) and
the same ``decode(z) -> image [B, 3, 8H, 8W]`` contract as the reference; the arithmetic runs in
``instancediffusion_amd.vae_engine.VAEDecoderEngine`` on an MI355X.  There is no CPU / eager fallback.

``encode`` is NOT on the inference path (it is used by training and inpainting only, ``trainer.py``) and is not
built: the encoder's parameters exist here so that the reference checkpoint's ``['autoencoder']`` sub-dict loads with
``strict=True`` (``utils/checkpoint.py:245``), and ``encode`` raises.
"""
from __future__ import annotations

from typing import Sequence

import torch
import torch.nn as nn

from .params import Affine, Conv


class ResnetBlock(nn.Module):
    """model.py:82-143 with temb_channels == 0 (no temb_proj), dropout 0, nin (1x1) shortcut."""

    def __init__(self, in_channels: int, out_channels: int):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.norm1 = Affine(in_channels)
        self.conv1 = Conv(in_channels, out_channels, 3)
        self.norm2 = Affine(out_channels)
        self.conv2 = Conv(out_channels, out_channels, 3)
        if in_channels != out_channels:
            self.nin_shortcut = Conv(in_channels, out_channels, 1)


class AttnBlock(nn.Module):
    """model.py:152-202: single-head attention over the H*W positions, head dim = channels."""

    def __init__(self, in_channels: int):
        super().__init__()
        self.in_channels = in_channels
        self.norm = Affine(in_channels)
        self.q = Conv(in_channels, in_channels, 1)
        self.k = Conv(in_channels, in_channels, 1)
        self.v = Conv(in_channels, in_channels, 1)
        self.proj_out = Conv(in_channels, in_channels, 1)


class Upsample(nn.Module):
    """model.py:42-56: nearest x2 then 3x3 conv."""

    def __init__(self, in_channels: int, with_conv: bool = True):
        super().__init__()
        assert with_conv
        self.with_conv = with_conv
        self.conv = Conv(in_channels, in_channels, 3)


class Downsample(nn.Module):
    """model.py:59-79 (parameter container only: the encoder is not on the inference path)."""

    def __init__(self, in_channels: int, with_conv: bool = True):
        super().__init__()
        assert with_conv
        self.with_conv = with_conv
        self.conv = Conv(in_channels, in_channels, 3)


def _level(blocks: Sequence[nn.Module], attns: Sequence[nn.Module]) -> nn.Module:
    m = nn.Module()
    m.block = nn.ModuleList(blocks)
    m.attn = nn.ModuleList(attns)
    return m


def _mid(ch: int) -> nn.Module:
    m = nn.Module()
    m.block_1 = ResnetBlock(ch, ch)
    m.attn_1 = AttnBlock(ch)
    m.block_2 = ResnetBlock(ch, ch)
    return m


class Encoder(nn.Module):
    """model.py:368-460 -- parameter container (state-dict compatibility); no forward."""

    def __init__(self, *, ch, out_ch, ch_mult=(1, 2, 4, 8), num_res_blocks, attn_resolutions, dropout=0.0,
                 resamp_with_conv=True, in_channels, resolution, z_channels, double_z=True, use_linear_attn=False,
                 attn_type="vanilla", **ignore_kwargs):
        super().__init__()
        assert attn_type == "vanilla" and not use_linear_attn and resamp_with_conv
        self.ch, self.num_resolutions, self.num_res_blocks = ch, len(ch_mult), num_res_blocks
        self.resolution, self.in_channels = resolution, in_channels
        self.conv_in = Conv(in_channels, ch, 3)
        curr_res = resolution
        in_ch_mult = (1,) + tuple(ch_mult)
        self.down = nn.ModuleList()
        block_in = ch
        for i_level in range(self.num_resolutions):
            blocks, attns = [], []
            block_in = ch * in_ch_mult[i_level]
            block_out = ch * ch_mult[i_level]
            for _ in range(num_res_blocks):
                blocks.append(ResnetBlock(block_in, block_out))
                block_in = block_out
                if curr_res in attn_resolutions:
                    attns.append(AttnBlock(block_in))
            down = _level(blocks, attns)
            if i_level != self.num_resolutions - 1:
                down.downsample = Downsample(block_in, resamp_with_conv)
                curr_res //= 2
            self.down.append(down)
        self.mid = _mid(block_in)
        self.norm_out = Affine(block_in)
        self.conv_out = Conv(block_in, 2 * z_channels if double_z else z_channels, 3)


class Decoder(nn.Module):
    """model.py:462-568 -- same module tree / key names; executed by ``VAEDecoderEngine``."""

    def __init__(self, *, ch, out_ch, ch_mult=(1, 2, 4, 8), num_res_blocks, attn_resolutions, dropout=0.0,
                 resamp_with_conv=True, in_channels, resolution, z_channels, give_pre_end=False, tanh_out=False,
                 use_linear_attn=False, attn_type="vanilla", **ignorekwargs):
        super().__init__()
        assert attn_type == "vanilla" and not use_linear_attn and resamp_with_conv and not give_pre_end
        self.ch, self.num_resolutions, self.num_res_blocks = ch, len(ch_mult), num_res_blocks
        self.resolution, self.in_channels, self.out_ch = resolution, in_channels, out_ch
        self.tanh_out = tanh_out
        self.ch_mult = tuple(ch_mult)
        block_in = ch * ch_mult[self.num_resolutions - 1]
        curr_res = resolution // 2 ** (self.num_resolutions - 1)
        self.z_shape = (1, z_channels, curr_res, curr_res)
        self.conv_in = Conv(z_channels, block_in, 3)
        self.mid = _mid(block_in)
        self.up = nn.ModuleList()
        for i_level in reversed(range(self.num_resolutions)):
            blocks, attns = [], []
            block_out = ch * ch_mult[i_level]
            for _ in range(num_res_blocks + 1):
                blocks.append(ResnetBlock(block_in, block_out))
                block_in = block_out
                if curr_res in attn_resolutions:
                    attns.append(AttnBlock(block_in))
            up = _level(blocks, attns)
            if i_level != 0:
                up.upsample = Upsample(block_in, resamp_with_conv)
                curr_res *= 2
            self.up.insert(0, up)                           # prepend: up[0] is the highest resolution
        self.norm_out = Affine(block_in)
        self.conv_out = Conv(block_in, out_ch, 3)


class AutoencoderKL(nn.Module):
    """ldm/models/autoencoder.py:12-37."""

    def __init__(self, ddconfig, embed_dim, scale_factor=1):
        super().__init__()
        ddconfig = dict(ddconfig)
        assert ddconfig["double_z"]
        self.encoder = Encoder(**ddconfig)
        self.decoder = Decoder(**ddconfig)
        self.quant_conv = Conv(2 * ddconfig["z_channels"], 2 * embed_dim, 1)
        self.post_quant_conv = Conv(embed_dim, ddconfig["z_channels"], 1)
        self.embed_dim = embed_dim
        self.scale_factor = scale_factor
        self.compute_dtype = torch.bfloat16     # 16-bit storage / MFMA input type of the HIP engine (or float16)
        self.max_decode_batch = 4               # images per decoder pass (activations at 512^2 x 128 ch: 67 MB each)
        self._engine = None

    def load_state_dict(self, *a, **k):
        r = super().load_state_dict(*a, **k)
        self._engine = None
        return r

    @property
    def engine(self):
        if self._engine is None:
            from ..vae_engine import VAEDecoderEngine      # imports the C-ABI loader; raises if the .so is missing
            self._engine = VAEDecoderEngine(self, dtype=self.compute_dtype)
        return self._engine

    def encode(self, x):
        raise NotImplementedError(
            "AutoencoderKL.encode is outside the sampling path (training / inpainting only, SURVEY.md §8 f-2); "
            "use the reference encoder")

    @torch.no_grad()
    def decode(self, z: torch.Tensor) -> torch.Tensor:
        """autoencoder.py:32-36: z [B, 4, H, W] latent -> image [B, 3, 8H, 8W] fp32 (unclamped, roughly [-1, 1])."""
        outs = [self.engine.decode(z[i:i + self.max_decode_batch]) for i in range(0, z.shape[0], self.max_decode_batch)]
        return outs[0] if len(outs) == 1 else torch.cat(outs, 0)
