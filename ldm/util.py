from instancediffusion_amd.host.config import get_obj_from_str, instantiate_from_config  # noqa: F401
