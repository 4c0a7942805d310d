from instancediffusion_amd.host.input import (batch_to_device, complete_mask, convert_points,  # noqa: F401
                                              create_zero_input_tensors, get_attmask_w_box, prepare_batch,
                                              prepare_instance_meta)
