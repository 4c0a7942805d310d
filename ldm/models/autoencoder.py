from instancediffusion_amd.host.autoencoder import AutoencoderKL  # noqa: F401
