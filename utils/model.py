from instancediffusion_amd.host.alpha import alpha_generator, set_alpha_scale  # noqa: F401
