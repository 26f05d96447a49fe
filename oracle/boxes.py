"""numpy front-end of oracle/points_in_boxes.c (TEST INFRASTRUCTURE, see oracle/__init__.py)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import build_c

_lib = None


def _c():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build_c.build_c())
        fp, ip = C.POINTER(C.c_float), C.POINTER(C.c_int)
        _lib.oracle_points_in_boxes_mask.argtypes = [fp, C.c_int, fp, C.c_int, C.c_float, ip]
        _lib.oracle_points_in_boxes_index.argtypes = [fp, fp, C.c_int, C.c_int, C.c_int, C.c_float, ip]
    return _lib


def _f(a):
    a = np.ascontiguousarray(a, np.float32)
    return a, a.ctypes.data_as(C.POINTER(C.c_float))


def points_in_boxes_mask(points, boxes, margin):
    """points [M,3], boxes [N,7] -> int32 [N,M] (roiaware_pool3d.cpp:143-168 semantics)."""
    p, pp = _f(points)
    b, bp = _f(boxes)
    out = np.zeros((b.shape[0], p.shape[0]), np.int32)
    _c().oracle_points_in_boxes_mask(bp, b.shape[0], pp, p.shape[0], margin,
                                     out.ctypes.data_as(C.POINTER(C.c_int)))
    return out


def points_in_boxes_index(points, boxes, margin):
    """points [B,M,3], boxes [B,T,7] -> int32 [B,M] (roiaware_pool3d_kernel.cu:313-336)."""
    p, pp = _f(points)
    b, bp = _f(boxes)
    out = np.zeros(p.shape[:2], np.int32)
    _c().oracle_points_in_boxes_index(bp, pp, p.shape[0], b.shape[1], p.shape[1], margin,
                                      out.ctypes.data_as(C.POINTER(C.c_int)))
    return out


def points_in_boxes_cpu(points, boxes):
    """The Python wrapper of the reference (roiaware_pool3d_utils.py:9-25): inflates
    boxes[:, 3:6] by 0.2 IN PLACE, MARGIN 1e-2."""
    boxes[:, 3:6] += 0.2
    return points_in_boxes_mask(points, boxes, 1e-2)
