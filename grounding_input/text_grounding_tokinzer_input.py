from instancediffusion_amd.host.grounding import GroundingNetInput  # noqa: F401
