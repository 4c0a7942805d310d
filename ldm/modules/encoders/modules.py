from instancediffusion_amd.host.text_encoder import AbstractEncoder, FrozenCLIPEmbedder  # noqa: F401
