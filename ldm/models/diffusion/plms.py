from instancediffusion_amd.host.samplers import PLMSSampler  # noqa: F401
