from instancediffusion_amd.host.unet import (  # noqa: F401
    Downsample, ResBlock, TimestepBlock, TimestepEmbedSequential, UNetModel, Upsample)
