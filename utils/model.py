from instancediffusion_amd.host.alpha import alpha_generator, set_alpha_scale  # noqa: F401
from instancediffusion_amd.host.input import get_clip_feature  # noqa: F401,E402
