from instancediffusion_amd.host.diffusion import DDPM  # noqa: F401
