"""ctypes binding of include/lidarcrafter_hip.h.  There is NO fallback: if the shared library is
missing the import of any compute op raises, so a GPU box can never silently run a CPU path."""
from __future__ import annotations

import ctypes as C
import os

from .build import LIB

i32, i64, f32, f64, vp = C.c_int, C.c_int64, C.c_float, C.c_double, C.c_void_p


class CmOperand(C.Structure):
    """struct lc_cm_operand: channel-major attention operand (pointer + batch/head/channel strides)."""
    _fields_ = [("p", vp), ("bs", i64), ("hs", i64), ("cs", i64)]


_op = C.POINTER(CmOperand)


class OctStats(C.Structure):
    """struct lc_oct_stats: producer-side GroupNorm statistics of one channel segment."""
    _fields_ = [("p", vp), ("channels", i32), ("slots", i32), ("unit", i32)]


_os = C.POINTER(OctStats)


class GnStatsInput(C.Structure):
    """struct lc_gn_stats_input: GroupNorm statistics + parameters for the conv's fused input norm."""
    _fields_ = [("partials", vp), ("G", i32), ("nch", i32), ("eps", f32), ("gamma", vp), ("beta", vp),
                ("scale", vp), ("shift", vp), ("ss_bs", i64), ("os0", _os), ("os1", _os)]


_gs = C.POINTER(GnStatsInput)


class ConvRange(C.Structure):
    """struct lc_conv_range: device-resident pre-scale + running amax of one conv layer's input."""
    _fields_ = [("x_scale", f32), ("x_unscale", f32), ("amax_scaled", f32), ("reserved", f32)]


# name -> (restype, argtypes); mirrors include/lidarcrafter_hip.h one to one
SIGNATURES = {
    "lc_abi_version": (i32, []),
    "lc_device_arch": (i32, [C.c_char_p, i32]),
    "lc_load_code_objects": (i32, []),
    "lc_packed_conv_weight_elems": (i64, [i32, i32, i32]),
    "lc_pack_conv_weight": (i32, [vp, vp, i32, i32, i32, vp]),
    "lc_conv2d_ring_fwd": (i32, [vp, i64, vp, vp, vp, i64, vp, i64, i32, i32, i32, i32, i32, i32,
                                 f32, i32, vp]),
    "lc_packed_conv_weight_f16x2_elems": (i64, [i32, i32, i32]),
    "lc_pack_conv_weight_f16x2": (i32, [vp, vp, vp, i32, i32, i32, vp, vp]),
    "lc_pack_conv_weight_f16x2_dx": (i32, [vp, vp, vp, i32, i32, i32, vp, vp, vp]),
    "lc_pack_conv_weights_f16x2_multi": (i32, [vp, i32, i32, vp]),
    "lc_range_from_amax": (i32, [vp, i64, f32, vp, vp]),
    "lc_groupnorm_amax_partials": (i64, [i32, i32, i32, i32, i32, i32]),
    "lc_conv2d_ring_f16x2_fwd": (i32, [vp, i64, vp, vp, vp, vp, i64, vp, i64, i32, i32, i32, i32,
                                       i32, i32, f32, i32, vp, i32, i32, _gs, vp, i32, vp, vp, vp]),
    "lc_split_act_units": (i64, [i32, i32, i32, i32]),
    "lc_groupnorm_apply_split": (i32, [vp, i64, vp, vp, vp, vp, vp, i64, vp, i32, i32, i32, i32, i32,
                                       f32, i32, vp, vp]),
    "lc_groupnorm_apply_os_split": (i32, [vp, i64, _os, _os, vp, vp, vp, vp, i64, vp, i32, i32, i32,
                                          i32, i32, f32, i32, vp, vp]),
    "lc_conv1x1_f16x2_ps_fwd": (i32, [vp, vp, vp, vp, vp, i64, vp, i64, i32, i32, i32, i32, i32, f32, vp, vp, vp]),
    "lc_conv2d_ring_f16x2_ps_fwd": (i32, [vp, vp, vp, vp, vp, i64, vp, i64, i32, i32, i32, i32, i32,
                                          f32, i32, vp, i32, vp, i32, vp, vp, vp]),
    "lc_range_from_tensor": (i32, [vp, i64, i32, i64, vp, vp]),
    "lc_splitk_stats_slots": (i64, [i32, i32]),
    "lc_splitk_reduce": (i32, [vp, i32, vp, vp, i64, vp, i64, i32, i32, i32, i32, f32, vp, vp]),
    "lc_conv2d_ring_f16x2_stats_slots": (i64, [i32, i32, i32, i32, i32, i32, i32]),
    "lc_groupnorm_coeffs": (i32, [vp, i64, vp, vp, vp, vp, vp, i64, vp, i32, i32, i32, i32, i32,
                                  i32, f32, vp]),
    "lc_groupnorm_partials_elems": (i64, [i32, i32, i32, i32, i32]),
    "lc_groupnorm_stats": (i32, [vp, i64, vp, i32, i32, i32, i32, i32, vp]),
    "lc_groupnorm_apply": (i32, [vp, i64, vp, vp, vp, vp, vp, i64, vp, i64, i32, i32, i32, i32,
                                 i32, f32, i32, vp]),
    "lc_groupnorm_apply_train": (i32, [vp, i64, vp, vp, vp, vp, vp, i64, vp, i64, i32, i32, i32, i32,
                                       i32, f32, i32, vp, vp, vp]),
    "lc_groupnorm_apply_os": (i32, [vp, i64, _os, _os, vp, vp, vp, vp, i64, vp, i64, i32, i32, i32,
                                    i32, i32, f32, i32, vp]),
    "lc_conv2d_ring_wgrad_scratch_elems": (i64, [i32, i32, i32, i32, i32, i32]),
    "lc_conv2d_ring_wgrad": (i32, [vp, i64, vp, i64, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]),
    "lc_conv2d_ring_wgrad_f16x2": (i32, [vp, i64, vp, i64, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32,
                                         i32, i32, vp]),
    "lc_groupnorm_meanrstd": (i32, [vp, i64, vp, vp, i32, i32, i32, i32, i32, f32, vp]),
    "lc_groupnorm_bwd": (i32, [vp, i64, vp, i64, vp, vp, vp, vp, vp, i64, vp, vp, i64, i32, i32, i32,
                               i32, i32, i32, vp]),
    "lc_groupnorm_bwd_train": (i32, [vp, i64, vp, i64, vp, vp, vp, vp, vp, i64, vp, vp, i64, vp, vp, vp, vp,
                                     i32, i32, i32, i32, i32, i32, vp, vp]),
    "lc_resample2x_fwd": (i32, [vp, i64, vp, i64, i32, i32, i32, i32, i32, vp]),
    "lc_resample2x_stats_slots": (i64, [i32, i32, i32]),
    "lc_resample2x_stats_fwd": (i32, [vp, i64, vp, i64, i32, i32, i32, i32, i32, vp, vp]),
    "lc_linear_fwd": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "lc_sinusoid_fwd": (i32, [vp, vp, i32, i32, f32, vp]),
    "lc_attention_fwd": (i32, [_op, _op, _op, _op, _op, _op, _op, _op, vp, i64, i64, i64,
                               i32, i32, i32, i32, i32, i32, i32, i32, f32, vp]),
    "lc_attention_f16x2_fwd": (i32, [_op, _op, _op, _op, _op, _op, _op, _op, vp, i64, i64, i64,
                                     i32, i32, i32, i32, i32, i32, i32, i32, f32, vp]),
    "lc_conv1x1_f16x2_ps_qkv_fwd": (i32, [vp, vp, vp, vp, vp, i64, vp, i32, i32, i32, i32, i32, i32, vp, vp, vp]),
    "lc_groupnorm_coeffs_os": (i32, [_os, _os, vp, vp, vp, vp, i64, vp, i32, i32, i32, i32, f32, vp]),
    "lc_resample2x_pair_fwd": (i32, [vp, i64, vp, i32, vp, i64, vp, i64, i32, i32, i32, i32, i32, vp]),
    "lc_split_act_fwd": (i32, [vp, i64, vp, i32, i32, i32, i32, vp, vp]),
    "lc_up2_combine9_stats_slots": (i64, [i32, i32]),
    "lc_up2_combine9_fwd": (i32, [vp, i64, vp, vp, i64, i32, i32, i32, i32, vp, vp]),
    "lc_up2_combine9_xup_fwd": (i32, [vp, i64, vp, vp, i64, vp, i64, vp, i64, i32, i32, i32, i32, vp, vp]),
    "lc_attention_units_elems": (i64, [i32, i32, i32, i32]),
    "lc_attention_pack_units": (i32, [_op, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
    "lc_attention_units_fwd": (i32, [_op, _op, vp, vp, i64, i64, i64, i32, i32, i32, i32, i32, i32, i32, i32, f32, vp]),
    "lc_attention_train_fwd": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, f32, i32, vp, vp]),
    "lc_attention_bwd": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, f32, vp]),
    "lc_attention_bwd_f16x2": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, f32, vp, vp]),
    "lc_pstep_fwd": (i32, [vp, i64, vp, i64, vp, i64, vp, vp, i64, i32, i64, i32, i32, vp]),
    "lc_gate_bias_act": (i32, [vp, i64, vp, vp, i64, vp, i64, vp, i64, i32, i32, i32, i32, vp]),
    "lc_fir_down2_split_units": (i64, [i32, i32, i32, i32]),
    "lc_fir_down2_prefilter_split": (i32, [vp, i64, vp, i32, i32, i32, i32, vp, vp]),
    "lc_conv2d_ring_s2_stats_slots": (i64, [i32, i32]),
    "lc_conv2d_ring_s2_f16x2_ps_fwd": (i32, [vp, vp, vp, vp, vp, i64, i32, i32, i32, i32, i32, f32, vp, i32, vp, vp, vp]),
    "lc_calibrate_mfma_f16": (i64, [vp, i32, i32, vp, vp]),
    "lc_calibrate_stream_copy": (i32, [vp, vp, i64, vp]),
    "lc_copy_strided": (i32, [vp, i64, vp, i64, i32, i64, vp]),
    "lc_add_scale": (i32, [vp, i64, vp, i64, vp, i64, i32, i64, f32, vp]),
    "lc_project_points": (i32, [vp, i32, i32, i32, f64, f64, f32, f32, vp, vp, vp, vp, i32, vp]),
    "lc_project_workspace_init": (i32, [vp, i32, vp]),
    "lc_project_points_ws": (i32, [vp, i32, i32, i32, f64, f64, f32, f32, vp, vp, vp, vp, i32, vp]),
    "lc_project_points_f64": (i32, [vp, i32, i32, i32, f64, f64, f64, f64, vp, vp, vp, vp]),
    "lc_range_postprocess": (i32, [vp, i64, vp, vp, i32, i32, i32, i32, f32, f32, vp]),
    "lc_condition_preprocess": (i32, [vp, i64, vp, i64, i32, i32, i32, i32, i32, f32, f32, vp]),
    "lc_layout_scratch_bytes": (i64, [i32, i32]),
    "lc_layout_condition": (i32, [vp, i32, i32, vp, i32, i32, i32, i32, f64, f64, vp, vp, vp, vp, vp]),
    "lc_roiaware_pool3d_fwd": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp, vp, vp,
                                     vp, vp]),
    "lc_roiaware_pool3d_bwd": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]),
    "lc_points_in_boxes_mask": (i32, [vp, i32, vp, i32, f32, vp, vp]),
    "lc_points_in_boxes_mask4": (i32, [vp, i32, vp, i32, f32, vp, vp, vp]),
    "lc_transform_points": (i32, [vp, i32, C.POINTER(C.c_double), vp, vp]),
    "lc_transform_points_f64": (i32, [vp, i32, C.POINTER(C.c_double), i32, vp, vp]),
    "lc_image_to_points": (i32, [vp, i64, vp, vp, i32, i32, f32, f32, f32, vp, vp, vp]),
    "lc_bev_occupancy_accumulate": (i32, [vp, i32, i32, f32, f32, f32, f32, f32, i32, i32, i32, i32,
                                          i32, vp, vp, vp]),
    "lc_sparse_quantize_scratch_bytes": (i64, [i32, i32]),
    "lc_sparse_quantize": (i32, [vp, i32, i32, f32, f32, f32, vp, vp, vp, vp, vp, vp]),
    "lc_bev_histogram": (i32, [vp, i32, i32, vp, i32, f32, f32, vp, vp, vp]),
    "lc_rbf_partials_elems": (i64, [i32, i32]),
    "lc_rbf_kernel_sum": (i32, [vp, vp, i32, i32, i32, f32, vp, vp]),
    "lc_chamfer3d_fwd": (i32, [vp, vp, i32, i32, i32, vp, vp, vp, vp, vp]),
    "lc_compact_scratch_elems": (i64, [i32]),
    "lc_compact_points": (i32, [vp, vp, i32, i32, vp, vp, vp, vp, vp]),
    "lc_points_in_boxes_index": (i32, [vp, vp, i32, i32, i32, f32, vp, vp]),
}

_lib = None
ABI_VERSION = 5   # include/lidarcrafter_hip.h lc_abi_version: bumped with every change of an exported signature


class HipLibraryMissing(RuntimeError):
    pass


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        path = os.environ.get("LC_HIP_LIB", LIB)   # developer override: A/B a differently built library
        if not os.path.exists(path):
            raise HipLibraryMissing(
                f"{path} not found: run `python -m lidarcrafter_amd.build` "
                "(or __graft_entry__.build()). There is no CPU fallback for the hot path.")
        handle = C.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the .so lacks a declared symbol
            fn.restype, fn.argtypes = res, args
        if handle.lc_abi_version() != ABI_VERSION:
            raise HipLibraryMissing("ABI version mismatch, rebuild the library")
        _lib = handle
    return _lib


_lib_p1 = None


def lib_p1() -> C.CDLL:
    """The single-product build of the convolution kernels (liblidarcrafter_hip_p1.so, -DLC_F16X2_TERMS=1):
    same entry points and arguments as the product library, used only through ops.conv_products() == 1."""
    global _lib_p1
    if _lib_p1 is None:
        path = os.path.join(os.path.dirname(LIB), "liblidarcrafter_hip_p1.so")
        if not os.path.exists(path):
            raise HipLibraryMissing(f"{path} not found: run `python -m lidarcrafter_amd.build`")
        handle = C.CDLL(path)
        for name in ("lc_conv2d_ring_f16x2_fwd", "lc_conv2d_ring_f16x2_ps_fwd", "lc_conv1x1_f16x2_ps_fwd",
                     "lc_conv1x1_f16x2_ps_qkv_fwd"):
            fn = getattr(handle, name)
            fn.restype, fn.argtypes = SIGNATURES[name]
        _lib_p1 = handle
    return _lib_p1


class HipError(RuntimeError):
    pass


def check(code: int, what: str) -> None:
    if code != 0:
        kind = {-1: "invalid argument", -2: "unsupported shape"}.get(code, f"hipError {code}")
        raise HipError(f"{what}: {kind}")
