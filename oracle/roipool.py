"""numpy restatement of the RoI-aware voxel pooling kernels (TEST INFRASTRUCTURE).

Follows the CUDA kernel text of the reference,
/root/reference/lidargen/ops/roiaware_pool3d/src/roiaware_pool3d_kernel.cu:
generate_pts_mask_for_box3d :39-75, collect_inside_pts_for_box3d :78-108, roiaware_maxpool3d
:111-157, roiaware_avgpool3d :160-190, *_backward :236-284.
PARITY PARTLY PINNED: the CUDA kernels cannot run in the build container and the reference ships no
vectors for them.  Pinned: the inside test `_local` (z slab, rotation into the box frame, the two
half-extent comparisons) -- its arithmetic is the reference's check_pt_in_box3d, which the CUDA file
(:23-36, MARGIN 1e-5) and the compiled C++ file (roiaware_pool3d.cpp:128-140, MARGIN 1e-2) share word
for word except for the constant; tests/test_oracle_vs_golden.py::test_roipool_inside_test_vs_reference_cpp
runs `_local(margin=1e-2)` against the compiled reference (oracle/_ref) on boundary-heavy point sets.
UNPINNED (a restatement of CUDA text only): the voxel index arithmetic, the per-voxel slot order /
overflow rule, max / avg pooling and their backward."""
from __future__ import annotations

import numpy as np

f32 = np.float32


def _local(pt, bx, margin=1e-5):
    x, y, z = (f32(v) for v in pt)
    cx, cy, cz, dx, dy, dz, rz = (f32(v) for v in bx)
    if float(abs(f32(z - cz))) > float(dz) / 2.0:
        return None
    cosa, sina = f32(np.cos(np.float64(-rz))), f32(np.sin(np.float64(-rz)))
    sx, sy = f32(x - cx), f32(y - cy)
    lx = f32(f32(sx * cosa) + f32(sy * f32(-sina)))
    ly = f32(f32(sx * sina) + f32(sy * cosa))
    if float(abs(lx)) < float(dx) / 2.0 + float(f32(margin)) and \
            float(abs(ly)) < float(dy) / 2.0 + float(f32(margin)):
        return lx, ly
    return None


def _vidx(val, n):
    i = int(val)                       # C truncation toward zero
    u = i & 0xFFFFFFFF                 # stored into unsigned int
    return min(u, n - 1)               # min(max(u, 0), n-1)


def forward(rois, pts, feat, out_size, max_pts, method):
    ox, oy, oz = out_size
    N, P, C = len(rois), len(pts), feat.shape[1]
    vox = np.zeros((N, ox, oy, oz, max_pts), np.int32)
    pooled = np.zeros((N, ox, oy, oz, C), np.float32)
    argmax = np.zeros((N, ox, oy, oz, C), np.int32)
    for b in range(N):
        dx, dy, dz = (f32(v) for v in rois[b, 3:6])
        xr, yr, zr = f32(dx / f32(ox)), f32(dy / f32(oy)), f32(dz / f32(oz))
        for k in range(P):
            loc = _local(pts[k], rois[b])
            if loc is None:
                continue
            lx, ly = loc
            lz = f32(f32(pts[k, 2]) - f32(rois[b, 2]))
            xi = _vidx(f32(f32(lx + f32(dx / f32(2))) / xr), ox)
            yi = _vidx(f32(f32(ly + f32(dy / f32(2))) / yr), oy)
            zi = _vidx(f32(f32(lz + f32(dz / f32(2))) / zr), oz)
            c = vox[b, xi, yi, zi, 0]
            if c < max_pts - 1:
                vox[b, xi, yi, zi, c + 1] = k
                vox[b, xi, yi, zi, 0] = c + 1
        for v in np.ndindex(ox, oy, oz):
            lst = vox[(b, *v)]
            idx = lst[1:1 + lst[0]]
            for c in range(C):
                if method == 0:
                    am, mx = -1, f32(-np.inf)
                    for k in idx:
                        if feat[k, c] > mx:
                            mx, am = feat[k, c], k
                    if am != -1:
                        pooled[(b, *v, c)] = mx
                    argmax[(b, *v, c)] = am
                elif len(idx):
                    s = f32(0)
                    for k in idx:
                        s = f32(s + feat[k, c])
                    pooled[(b, *v, c)] = f32(s / f32(len(idx)))
    return pooled, vox, argmax


def backward(vox, argmax, grad_out, num_pts, method):
    N, ox, oy, oz, C = grad_out.shape
    gin = np.zeros((num_pts, C), np.float64)
    for b in range(N):
        for v in np.ndindex(ox, oy, oz):
            lst = vox[(b, *v)]
            for c in range(C):
                if method == 0:
                    am = argmax[(b, *v, c)]
                    if am != -1:
                        gin[am, c] += grad_out[(b, *v, c)]
                else:
                    g = f32(grad_out[(b, *v, c)] * f32(1.0 / max(float(lst[0]), 1.0)))
                    for k in lst[1:1 + lst[0]]:
                        gin[k, c] += g
    return gin.astype(np.float32)
