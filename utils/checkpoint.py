from instancediffusion_amd.host.checkpoint import load_model_ckpt, read_official_ckpt  # noqa: F401
