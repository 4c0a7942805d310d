from instancediffusion_amd.host.unifusion import UniFusion  # noqa: F401
