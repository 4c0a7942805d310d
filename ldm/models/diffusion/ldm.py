from instancediffusion_amd.host.diffusion import LatentDiffusion  # noqa: F401
