"""CPU-side checks of the C-ABI boundary: the library builds/loads and exports every symbol that
include/lidarcrafter_hip.h declares (no compute calls -- there is no GPU in the CPU suite)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "lidarcrafter_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(lc_[a-z0-9_]+)\s*\(", src)))


def test_header_matches_binding():
    from lidarcrafter_amd import _lib

    assert _declared() == sorted(_lib.SIGNATURES)


def test_library_exports_every_symbol():
    from lidarcrafter_amd import _lib, build

    build.build(verbose=False)
    handle = _lib.lib()
    for name in _declared():
        assert hasattr(handle, name), name
    assert handle.lc_abi_version() == 1
    # pure host-side helpers are callable without a GPU
    assert handle.lc_packed_conv_weight_elems(2, 64, 3) == 9 * 64 * 64
    assert handle.lc_packed_conv_weight_elems(100, 42, 1) == 48 * 128
    # statistics blocks: 64 KiB chunks, 16 KiB when that would leave fewer than 512 blocks
    assert handle.lc_groupnorm_partials_elems(32, 64, 32, 1024, 8) == 32 * 8 * 16 * 2
    assert handle.lc_groupnorm_partials_elems(2, 64, 32, 1024, 8) == 2 * 8 * 64 * 2


def test_no_cpu_fallback():
    """The product path must fail loudly on CPU tensors instead of silently computing."""
    import torch

    from lidarcrafter_amd import ops

    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.groupnorm(torch.zeros(1, 8, 2, 2), 8, 1e-6)
    from lidargen.models.unets import EfficientUNet

    m = EfficientUNet(2, (8, 64), base_channels=16, coords_encoding="fourier_features")
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 2, 8, 64), torch.zeros(1))


def test_product_never_imports_oracle():
    bad = []
    for base in ("lidarcrafter_amd", "lidargen"):
        for dp, _, fs in os.walk(os.path.join(ROOT, base)):
            for f in fs:
                if f.endswith(".py"):
                    txt = open(os.path.join(dp, f)).read()
                    if re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M):
                        bad.append(os.path.join(dp, f))
    assert not bad, bad
