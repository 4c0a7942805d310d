from instancediffusion_amd.host.samplers import PLMSSamplerInst  # noqa: F401
