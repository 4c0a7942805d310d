"""Shared test-case construction: regenerates exactly the inputs ``oracle/make_golden.py`` fed the reference."""
from __future__ import annotations

import os
from typing import Dict

import torch

from instancediffusion_amd import synth
from oracle import ref_cpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(REPO, "tests", "golden")

# mirrors oracle/make_golden.py:VARIANTS
VARIANTS = {
    "full": dict(),
    "tiny": dict(model_channels=64, mid_dim=256),
    "mid": dict(channel_mult=(1, 2, 4), num_res_blocks=1, mid_dim=512),
}
DROPS = {
    "test_box.yaml": dict(test_drop_boxes=False, test_drop_points=False, test_drop_scribbles=True, test_drop_masks=True),
    "test_mask.yaml": dict(test_drop_boxes=False, test_drop_points=False, test_drop_scribbles=True, test_drop_masks=False),
    "test_point.yaml": dict(test_drop_boxes=True, test_drop_points=False, test_drop_scribbles=True, test_drop_masks=True),
    "test_scribble.yaml": dict(test_drop_boxes=False, test_drop_points=False, test_drop_scribbles=False, test_drop_masks=False),
}


def cfg_for(cfg_name: str, variant: str) -> dict:
    cfg = dict(ref_cpu.DEFAULT_CFG)
    cfg.update(DROPS[cfg_name])
    cfg.update(VARIANTS[variant])
    return cfg


def load_golden(tag: str) -> dict:
    return torch.load(os.path.join(GOLD, f"{tag}.pt"), weights_only=False)


def build_inputs(meta: dict) -> Dict[str, torch.Tensor]:
    """Re-create (x, context, uc, grounding batch, per-instance contexts) exactly as make_golden.gen_case did."""
    g = torch.Generator().manual_seed(1234)
    bx = torch.tensor(synth.C1_BOXES) if meta["boxes"] == "c1" else synth.random_boxes(meta["n_boxes"], g)
    gb = synth.make_grounding_batch(meta["batch"], bx, g, with_scribbles=meta["with_scribbles"],
                                    with_polygons=meta["with_polygons"], with_segs=meta["with_segs"],
                                    seg_size=meta["seg_size"])
    L = meta["latent"]
    x = torch.randn(meta["batch"], 4, L, L, generator=g)
    context = torch.randn(meta["batch"], 77, 768, generator=g)
    uc = torch.randn(meta["batch"], 77, 768, generator=g)
    inst_ctx = [torch.randn(meta["batch"], 77, 768, generator=g) for _ in range(meta["n_inst"])]
    return dict(x=x, context=context, uc=uc, gb=gb, inst_ctx=inst_ctx,
                t=torch.full((meta["batch"],), 981, dtype=torch.long))


def unet_schema(cfg: dict) -> Dict[str, tuple]:
    """State-dict schema {key: shape} of the reference UNet for ``cfg`` -- derived from the HOST mirror classes
    (whose key set is itself pinned against the reference's in tests/test_schema.py)."""
    from ldm.modules.diffusionmodules.openaimodel import UNetModel
    from instancediffusion_amd.host.config import unet_kwargs_from_cfg
    with torch.device("meta"):
        m = UNetModel(**unet_kwargs_from_cfg(cfg))
    return {k: tuple(v.shape) for k, v in m.state_dict().items()}


def rel_rms(a: torch.Tensor, b: torch.Tensor) -> float:
    a = a.double(); b = b.double()
    return float(((a - b).pow(2).mean() / b.pow(2).mean().clamp_min(1e-30)).sqrt())


# ---- VAE decoder cases (oracle/make_golden.py:gen_vae_case) -------------------------------------------------------
VAE_VARIANTS = {"full": dict(), "tiny": dict(ch=64, ch_mult=(1, 2, 2), num_res_blocks=1)}


def vae_cfg_for(variant: str) -> dict:
    cfg = dict(ref_cpu.DEFAULT_VAE_CFG)
    cfg.update(VAE_VARIANTS[variant])
    return cfg


def build_vae(cfg: dict, salt: int = 7):
    """Host AutoencoderKL mirror with key-name-seeded weights (the parameters make_golden gave the reference)."""
    from ldm.models.autoencoder import AutoencoderKL
    dd = dict(double_z=True, z_channels=cfg["z_channels"], resolution=cfg["resolution"], in_channels=3,
              out_ch=cfg["out_ch"], ch=cfg["ch"], ch_mult=list(cfg["ch_mult"]), num_res_blocks=cfg["num_res_blocks"],
              attn_resolutions=list(cfg["attn_resolutions"]), dropout=0.0)
    with torch.device("meta"):
        ae = AutoencoderKL(ddconfig=dd, embed_dim=cfg["embed_dim"], scale_factor=cfg["scale_factor"])
    ae = ae.to_empty(device="cpu")
    ae.load_state_dict(synth.synth_state_dict({k: tuple(v.shape) for k, v in ae.state_dict().items()}, salt))
    return ae.eval()


def vae_latent(meta: dict) -> torch.Tensor:
    g = torch.Generator().manual_seed(4321)
    return torch.randn(meta["batch"], 4, meta["latent"], meta["latent"], generator=g) * 0.18215 * 4.0


# ---- masked gated self-attention case (oracle/make_golden.py:gen_masked_case) ------------------------------------
def build_masked_inputs():
    import numpy as np
    g = torch.Generator().manual_seed(1234)
    bx = synth.random_boxes(3, g)
    gb = synth.make_grounding_batch(1, bx, g)
    att = torch.zeros(30, 64, 64)
    for i in range(3):                                   # utils/input.py:34-37 index order (x on dim 0)
        b = bx[i].tolist()
        x1, y1, x2, y2 = (int(np.round(b[0] * 64)), int(np.round(b[1] * 64)), int(np.round(b[2] * 64)),
                          int(np.round(b[3] * 64)))
        att[i][x1:x2, y1:y2] = 1
    gb["att_masks"] = att.unsqueeze(0)
    x = torch.randn(1, 4, 64, 64, generator=g)
    context = torch.randn(1, 77, 768, generator=g)
    return dict(gb=gb, x=x, context=context, t=torch.full((1,), 981, dtype=torch.long))
