"""lidarcrafter_amd -- MI355X (gfx950) native hot path for LiDARCrafter's range-image diffusion denoiser.

Layout:
  csrc/      hand-written HIP kernels + the C-ABI (include/lidarcrafter_hip.h)
  _lib.py    ctypes binding of that C-ABI (fails loudly when the .so is missing)
  ops.py     thin tensor-level wrappers (device pointers + sizes + stream -> C-ABI)
  lidargen/  host-side mirror of the reference `lidargen` API surface for this path
"""
__version__ = "0.1.0"
