"""Host-side mirror of the reference `lidargen` package for the denoising hot path.

Importable as `lidarcrafter_amd.lidargen` and -- through the alias package at the repository
root -- as plain `lidargen`, so the reference's tools/generate/*.py and
tools/evaluation/sample_and_save_*.py resolve every `lidargen.*` name they touch.
Unlike the reference's __init__ (lidargen/__init__.py:4) no generated version.py is required.
"""
__version__ = "0.1.0+mi355x"

from lidarcrafter_amd import torch_ops as _torch_ops  # noqa: E402,F401  registers torch.ops.lidarcrafter.*
