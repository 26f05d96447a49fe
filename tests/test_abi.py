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
    assert handle.lc_abi_version() == 5
    # pure host-side helpers are callable without a GPU
    assert handle.lc_packed_conv_weight_elems(2, 64, 3) == 9 * 64 * 64
    assert handle.lc_packed_conv_weight_elems(100, 42, 1) == 48 * 128
    # statistics blocks: 64 KiB chunks, 16 KiB when that would leave fewer than 512 blocks
    assert handle.lc_groupnorm_partials_elems(32, 64, 32, 1024, 8) == 32 * 8 * 16 * 2
    assert handle.lc_groupnorm_partials_elems(2, 64, 32, 1024, 8) == 2 * 8 * 64 * 2


def test_argument_checks_refuse_before_any_launch():
    """Shapes a kernel does not take are refused by the host entry with LC_EUNSUP / LC_EINVAL before anything is
    dereferenced or launched (callable without a GPU): the ping-pong tile configuration (33) outside its shapes, a sample
    too large for the kernels' 32-bit byte offsets, attention heads wider than 64 channels."""
    import ctypes

    from lidarcrafter_amd import _lib

    h = _lib.lib()
    buf = (ctypes.c_float * 64)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    LC_EINVAL, LC_EUNSUP = -1, -2
    src = open(os.path.join(ROOT, "include", "lidarcrafter_hip.h")).read()
    m = dict(re.findall(r"#define\s+(LC_E[A-Z]+)\s+\(?(-?\d+)\)?", src))
    if m:
        LC_EINVAL, LC_EUNSUP = int(m.get("LC_EINVAL", LC_EINVAL)), int(m.get("LC_EUNSUP", LC_EUNSUP))

    def conv(Ci, Co, H, W, cfg, B=1):
        return h.lc_conv2d_ring_f16x2_fwd(p, Ci * H * W, p, p, None, None, 0, p, Co * H * W, B, Ci, Co, H, W, 3, 1.0, cfg,
                                          None, 0, 0, None, None, 8, p, p, None)

    assert conv(48, 64, 8, 64, 33) == LC_EUNSUP        # Ci % 16 == 0 but < 64
    assert conv(64, 96, 8, 64, 33) == LC_EUNSUP        # Co % 64
    assert conv(64, 64, 6, 64, 33) == LC_EUNSUP        # H % 4
    assert conv(64, 64, 8, 96, 33) == LC_EUNSUP        # W % 64
    assert conv(1024, 64, 8, 64, 33) == LC_EUNSUP      # Ci > 512
    assert conv(64, 64, 4096, 4095, 0) == LC_EUNSUP    # H * W >= 2^24 / 32-bit offsets
    assert conv(64, 4096, 1024, 1024, 0) == LC_EUNSUP  # Co * H * W * 4 >= 2^31
    assert h.lc_attention_bwd(p, p, p, p, p, p, p, p, p, p, 1, 32, 32, 96, 32, 1.0, None) == LC_EUNSUP
    assert h.lc_attention_bwd(p, p, p, p, p, p, p, p, p, p, 0, 32, 32, 32, 32, 1.0, None) == LC_EINVAL
    assert h.lc_project_workspace_init(None, 16, None) == LC_EINVAL


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
    # the folded down-sampling stage (round 6): eligibility is a pure predicate, the op itself refuses CPU tensors
    assert ops.can_fold_down(64, 128, 32, 1024) and ops.can_fold_down(16, 32, 4, 128)
    assert not ops.can_fold_down(64, 128, 32, 1000) and not ops.can_fold_down(24, 48, 32, 1024)   # W % 128, Ci % 16
    assert not ops.can_fold_down(64, 128, 3, 1024) and not ops.can_fold_down(64, 128, 2, 1024)     # H odd / H < 4
    pk = ops.PackedConv("down")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.conv_down2(torch.zeros(1, 16, 4, 128), pk, torch.zeros(32, 16, 3, 3), None)


def test_fold_down_symbols_reject_bad_shapes():
    """The C entry points of the folded stage validate their arguments before any launch (no GPU needed)."""
    from lidarcrafter_amd import _lib

    LC_EINVAL, LC_EUNSUP = -1, -2
    h = _lib.lib()
    assert h.lc_fir_down2_split_units(8, 64, 32, 1024) == 8 * 2 * 8 * 35 * 1024
    assert h.lc_fir_down2_split_units(8, 60, 32, 1024) == 0                       # C % 8
    assert h.lc_conv2d_ring_s2_stats_slots(16, 512) == 16 * 8 * 2 and h.lc_conv2d_ring_s2_stats_slots(16, 500) == 0
    p = 4096
    assert h.lc_fir_down2_prefilter_split(None, 0, p, 1, 16, 4, 128, p, None) == LC_EINVAL
    assert h.lc_fir_down2_prefilter_split(p, 16 * 4 * 128, p, 1, 24, 4, 128, p, None) == LC_EUNSUP     # C % 16
    assert h.lc_fir_down2_prefilter_split(p, 16 * 3 * 128, p, 1, 16, 3, 128, p, None) == LC_EUNSUP     # H odd
    assert h.lc_fir_down2_prefilter_split(p, 16 * 4 * 64, p, 1, 16, 4, 64, p, None) == LC_EUNSUP       # W % 128
    assert h.lc_conv2d_ring_s2_f16x2_ps_fwd(p, p, p, None, None, 0, 1, 16, 32, 2, 64, 1.0, None, 8, p, p, None) == LC_EINVAL
    assert h.lc_conv2d_ring_s2_f16x2_ps_fwd(p, p, p, None, p, 0, 1, 24, 32, 2, 64, 1.0, None, 8, p, p, None) == LC_EUNSUP
    assert h.lc_conv2d_ring_s2_f16x2_ps_fwd(p, p, p, None, p, 0, 1, 16, 32, 2, 48, 1.0, None, 8, p, p, None) == LC_EUNSUP


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


def test_struct_layouts_match_header(tmp_path):
    """The ctypes mirrors of the header's structs have the C compiler's size and field offsets."""
    import ctypes as C
    import subprocess

    from lidarcrafter_amd import _lib

    structs = {"lc_cm_operand": _lib.CmOperand, "lc_oct_stats": _lib.OctStats,
               "lc_gn_stats_input": _lib.GnStatsInput}
    lines = []
    for cname, cls in structs.items():
        lines.append(f'printf("{cname} %zu", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'printf(" %zu", offsetof({cname}, {fname}));')
        lines.append('printf("\\n");')
    src = tmp_path / "layout.c"
    src.write_text('#include <stddef.h>\n#include <stdio.h>\n#include "lidarcrafter_hip.h"\n'
                   "int main(void) {\n" + "\n".join(lines) + "\nreturn 0; }\n")
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split("\n")
    for line, (cname, cls) in zip(out, structs.items()):
        got = line.split()
        assert got[0] == cname
        want = [C.sizeof(cls)] + [getattr(cls, f).offset for f, _ in cls._fields_]
        assert [int(v) for v in got[1:]] == want, (cname, got, want)


def test_stats_slots_helper():
    """lc_conv2d_ring_f16x2_stats_slots is host-only: entries per (sample, octet) for the tile the
    heuristic picks, 0 where the chosen kernel emits none."""
    from lidarcrafter_amd import _lib

    h = _lib.lib()
    assert h.lc_conv2d_ring_f16x2_stats_slots(8, 64, 64, 32, 1024, 3, 0) == 128 * 4   # 4x64 tiles, 4 px waves
    # 1x1: the 2-blocks/CU kernel, plane folded to 512 x 64, 2 x 64 tiles of 4 pixel waves (it writes entries since round 5)
    assert h.lc_conv2d_ring_f16x2_stats_slots(8, 64, 64, 32, 1024, 1, 0) == 256 * 4
    assert h.lc_conv2d_ring_f16x2_stats_slots(8, 64, 64, 32, 1024, 3, 5) == 0         # ... its 3x3 launches write none
    assert h.lc_conv2d_ring_f16x2_stats_slots(8, 2, 64, 32, 1024, 3, 0) == 0          # Ci < 24: likewise
    assert h.lc_conv2d_ring_f16x2_stats_slots(8, 64, 62, 32, 1024, 3, 0) == 0         # Co % 8
    # the x2 down-sampler: one entry per channel, output row and 256-column input segment where its vector kernel runs
    assert h.lc_resample2x_stats_slots(32, 1024, -1) == 16 * 4
    assert h.lc_resample2x_stats_slots(32, 1000, -1) == 0 and h.lc_resample2x_stats_slots(31, 1024, -1) == 0
    assert h.lc_resample2x_stats_slots(32, 1024, 1) == 0                              # up-sampling leaves none


def test_producer_stats_bookkeeping():
    """Host logic of the producer-side GroupNorm statistics (ops._attach_stats / _find_stats /
    _drop_stats) on CPU tensors: channel slices of a concat buffer, two-segment lookup,
    invalidation by overlap, refusal of shapes / group sizes the kernels cannot fold."""
    import torch

    from lidarcrafter_amd import ops as K

    B, H, W = 2, 4, 8
    buf = torch.zeros(B, 128, H, W)
    lo, hi = buf[:, :64], buf[:, 64:]
    h0 = K._OctStatsHandle(torch.zeros(1), 64, 3, (B, H * W))
    h1 = K._OctStatsHandle(torch.zeros(1), 64, 5, (B, H * W))
    K._attach_stats(lo, h0)
    assert K._find_stats(buf, 8) is None                       # second half unknown
    K._attach_stats(hi, h1)
    assert K._find_stats(buf, 8) == (h0, h1)
    assert K._find_stats(buf[:, :64], 8) == (h0,)              # a fresh view of the same range
    assert K._find_stats(buf[:, 64:], 4) == (h1,)
    assert K._find_stats(buf, 32) is None                      # 4 channels per group: not octets
    assert K._find_stats(buf, 1) is None                       # group would straddle the segments
    assert K._find_stats(buf[:, 32:96], 8) is None             # not a recorded range
    assert K._find_stats(buf[:, :, :2], 8) is None             # not a plain channel slice
    K._drop_stats(buf[:, 60:70])                               # overlaps both ranges
    assert K._find_stats(lo, 8) is None and K._find_stats(hi, 8) is None
    # entry units (round 5): a consumer folds entries of any unit that divides its group; the pre-split apply pass
    # (octet_groups) additionally wants groups of whole octets or of 2 / 4 channels, and the segment boundary on an octet
    q0 = K._OctStatsHandle(torch.zeros(1), 64, 3, (B, H * W), 4)    # quad entries
    p1 = K._OctStatsHandle(torch.zeros(1), 64, 5, (B, H * W), 2)    # pair entries
    K._attach_stats(lo, q0)
    K._attach_stats(hi, p1)
    assert K._find_stats(buf, 32) == (q0, p1)                  # 4 per group: quads and pairs both divide it
    assert K._find_stats(buf, 32, octet_groups=True) == (q0, p1)
    assert K._find_stats(buf, 64) is None                      # 2 per group: the quad entries do not
    assert K._find_stats(hi, 32) == (p1,) and K._find_stats(hi, 32, octet_groups=True) == (p1,)   # 2 per group from pairs
    assert K._find_stats(buf, 8) == (q0, p1) and K._find_stats(buf, 8, octet_groups=True) == (q0, p1)
    assert K._find_stats(lo, 16) == (q0,) and K._find_stats(lo, 64) is None          # 4 per group / 1 per group
    c1 = K._OctStatsHandle(torch.zeros(1), 64, 7, (B, H * W), 1)    # per-channel entries (the x2 down-sampler)
    K._attach_stats(lo, c1)
    assert K._find_stats(lo, 64) == (c1,) and K._find_stats(lo, 64, octet_groups=True) is None    # 1 per group: not that pass
    tri = torch.zeros(B, 96, H, W)                              # 36 + 60 channels: boundary off the octet grid
    K._attach_stats(tri[:, :36], K._OctStatsHandle(torch.zeros(1), 36, 3, (B, H * W), 2))
    K._attach_stats(tri[:, 36:], K._OctStatsHandle(torch.zeros(1), 60, 3, (B, H * W), 2))
    assert K._find_stats(tri, 24) is not None and K._find_stats(tri, 24, octet_groups=True) is None
    K._drop_stats(buf)
    K._attach_stats(lo, h0)
    K._attach_stats(hi, h1)
    K._drop_stats(hi)
    assert K._find_stats(lo, 8) == (h0,) and K._find_stats(buf, 8) is None
    other = torch.zeros(B, 64, H, 2 * W)
    K._attach_stats(other, h0)                                 # handle of a different plane shape
    assert K._find_stats(other, 8) is None
    own = torch.zeros(B, 64, H, W)
    K._attach_stats(own, h0)
    assert K._find_stats(own, 8) == (h0,)
    K._drop_stats(own)
    assert K._find_stats(own, 8) is None
    # the plane may be re-indexed (token views of the attention blocks): same sums
    K._attach_stats(own, h0)
    assert K._find_stats(own.view(B, 64, 1, H * W), 8) == (h0,)
    # torch.inference_mode (the samplers) records no `_base`: slices / token views made through
    # ops.chan_slice / ops.alias keep the bookkeeping on the buffer they belong to
    with torch.inference_mode():
        cat = torch.zeros(B, 128, H, W)
        assert cat[:, 64:]._base is None
        K._attach_stats(K.chan_slice(cat, 0, 64), h0)
        hi2 = K.chan_slice(cat, 64, 128)
        K._attach_stats(hi2, h1)
        assert K._find_stats(cat, 8) == (h0, h1)
        tok = K.alias(torch.as_strided(hi2, (B, 64, H * W), (hi2.stride(0), H * W, 1)), hi2)
        tok4 = K.alias(torch.as_strided(tok, (B, 64, 1, H * W), (tok.stride(0), H * W, H * W, 1)), tok)
        assert K._find_stats(tok4, 8) == (h1,)
        K._drop_stats(K.chan_slice(cat, 64, 128))               # a writer into that range
        assert K._find_stats(tok4, 8) is None and K._find_stats(cat, 8) is None
        assert K._find_stats(K.chan_slice(cat, 0, 64), 8) == (h0,)


def test_bench_line_contract():
    """The committed bench line (profiles/, produced by `python bench.py` on an MI355X) carries every
    field of the driver's contract, the roofline and the CPU baseline objects."""
    import glob
    import json

    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_n1.json")))
    assert paths
    d = json.load(open(paths[-1]))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
              "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    # (lines recorded before the metric string was made verbatim say "32x1024" without the GPU list)
    assert base["metric"].startswith(d["metric"].replace("32x1024", "32\u00d71024"))
    assert json.dumps(base["metric"]) in open(os.path.join(ROOT, "bench.py")).read()
    assert d["n_gpus"] == 1 and d["scaling"] == "weak"
    assert d["higher_is_better"] is True and d["vs_baseline"] is None and "workload" in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["traffic"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    assert abs(d["value"] - 1000.0 / d["ms_per_step"]) / d["value"] < 1e-3   # steps/s of the job
    if "r06" in os.path.basename(paths[-1]) or "rows" in d:
        # round 6: per-rank verification, same-process rows behind their fixture checks, the box's calibration
        v = d["verify"]
        assert v["ok"] is True and v["ranks_verified"] == 1 and v["per_rank"][0]["samples"] == [0, 7]
        for k in ("uncond_32x1024_batch1", "uncond_32x1024_batch32", "cond_layout_v6_32x1024_batch8"):
            row = d["rows"][k]
            assert row["ms_per_step"] > 0 and row["check"]["forward_max_rel_l2_vs_reference"] < row["check"]["tolerance"]
        cal = d["box_calibration"]
        assert cal["mfma_f16_tflops_random_operands"] > 100 and 1.0 < cal["stream_copy_tb_s"] < 8.0
        assert abs(r["frac_of_box_ceiling"] - r["achieved"] / cal["three_product_ceiling_tflops"]) < 1e-3
        assert r["achieved_executed"] <= r["achieved"]


def test_single_product_library_loads():
    """liblidarcrafter_hip_p1.so (the convolution kernels with one fp16 product per multiply, for fp16-autocast
    callers) is built next to the product library and exports the two entry points ops routes to it."""
    from lidarcrafter_amd import _lib, ops

    h = _lib.lib_p1()
    for name in ("lc_conv2d_ring_f16x2_fwd", "lc_conv2d_ring_f16x2_ps_fwd"):
        assert getattr(h, name).argtypes == _lib.SIGNATURES[name][1]
    assert ops.conv_products() == 3            # never the default


def test_route_signature_moves_with_switches_and_epoch(monkeypatch):
    """ops.route_signature() is part of the key under which a sampler replays a graph captured by an earlier run
    (continuous_time.py::_graph_key): every routing switch of ops, the effective product count and the epoch of the
    packed weights / range slots must move it."""
    from lidarcrafter_amd import ops as K

    base = K.route_signature()
    assert K.route_signature() == base
    for name in ("PRODUCER_GN_STATS", "RESAMPLE_STATS", "FUSE_GN", "PRESPLIT", "FOLD_DOWN", "SPLITK"):
        monkeypatch.setattr(K, name, not getattr(K, name))
        assert K.route_signature() != base, name
        monkeypatch.undo()
    for name, val in (("CONV_PRECISION", "f32"), ("ATTN_PRECISION", "f32"), ("FUSE_GN_MAX_CO", 128), ("PS1X1_MIN_CO", 0)):
        monkeypatch.setattr(K, name, val)
        assert K.route_signature() != base, name
        monkeypatch.undo()
    old = K.set_conv_products(1)
    try:
        assert K.route_signature() != base
    finally:
        K.set_conv_products(old)
    assert K.route_signature() == base
    K.bump_epoch()
    assert K.route_signature() != base
