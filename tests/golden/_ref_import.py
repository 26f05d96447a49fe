"""Import the reference `lidargen` read-only from /root/reference (this container only).

Used ONLY by tests/golden/make_fixtures.py to generate golden vectors; nothing under tests/
that runs on the GPU box (or in the CPU suite) imports this module.  The reference package
cannot be imported normally: `lidargen/__init__.py:4` needs a generated version.py,
`lidargen/models/unets/__init__.py:8` pulls `timm`, `lidargen/dataset/__init__.py` pulls
loguru/clip/compiled CUDA extensions.  We therefore register empty namespace stubs for the
package nodes and import only the leaf modules on the hot path ("Shim 1"),
and neutralise the hard-coded `.cuda()` in layout_encoder.py:217 ("Shim 2").
"""
import importlib
import sys
import types

REF = "/root/reference"

_PKGS = [
    "lidargen",
    "lidargen.models",
    "lidargen.models.unets",
    "lidargen.dataset",
    "lidargen.dataset.transforms_3d",
    "lidargen.utils",
]


def install():
    import torch

    for name in list(sys.modules):
        if name == "lidargen" or name.startswith("lidargen."):
            del sys.modules[name]
    for name in _PKGS:
        m = types.ModuleType(name)
        m.__path__ = [REF + "/" + name.replace(".", "/")]
        m.__package__ = name
        sys.modules[name] = m
    # Shim 2: CPU-only container
    torch.Tensor.cuda = lambda self, *a, **k: self


def ref(name: str):
    """ref('models.unets.efficient_unet') -> reference module object."""
    return importlib.import_module("lidargen." + name)
