"""instancediffusion_amd -- MI355X-native (gfx950) InstanceDiffusion sampling hot path.

Layout:
  csrc/        hand-written HIP kernels + the C-ABI (``libidf_gfx950.so``, declared in ``include/idf.h``)
  _lib.py      ctypes loader for the C-ABI (fails loudly if the library is missing)
  ops.py       thin tensor -> pointer marshalling over the C-ABI
  engine.py    UNet denoise-step executor (weight packing, workspace, op sequencing, hipGraph replay)
  host/        host-side mirror of the reference's Python interface for this path
               (``ldm.*`` / ``grounding_input.*`` dotted names resolve here via the top-level shim packages)
  synth.py     seeded synthetic weights / inputs (no checkpoints exist offline)

Nothing in this package imports ``oracle/``.
"""
__version__ = "0.1.0"
