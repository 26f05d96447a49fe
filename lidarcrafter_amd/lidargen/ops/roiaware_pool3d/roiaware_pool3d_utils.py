"""Points-in-boxes -- API mirror of the reference's
lidargen/ops/roiaware_pool3d/roiaware_pool3d_utils.py:9-45 on the HIP kernels
(lc_points_in_boxes_mask / _index).  The reference's `points_in_boxes_cpu` runs a C++ double loop
on the host (called >= 2x per frame in the temporal loop); here both entry points
run on the GPU.  RoIAwarePool3d (voxel pooling, :48-107; not called by any generation script) runs
on lc_roiaware_pool3d_fwd/_bwd."""
from __future__ import annotations

import numpy as np
import torch

from lidarcrafter_amd import ops as K


def _device():
    if not torch.cuda.is_available():
        raise RuntimeError("points_in_boxes needs the MI355X: the hot path has no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def _to_torch(x):
    if isinstance(x, np.ndarray):
        return torch.from_numpy(x).float(), True
    return x, False


def points_in_boxes_cpu(points, boxes):
    """points (M,3), boxes (N,7) [x,y,z,dx,dy,dz,heading] -> int (N,M) 0/1.
    Same contract as the reference, INCLUDING its in-place `boxes[:, 3:6] += 0.2` on the caller's
    tensor (a numpy input is inflated on its float32 copy, exactly like the reference) and the
    host-side MARGIN of 1e-2; numpy in -> numpy out."""
    assert boxes.shape[1] == 7 and points.shape[1] == 3
    points, is_numpy = _to_torch(points)
    boxes, _ = _to_torch(boxes)
    boxes[:, 3:6] += boxes.new_tensor([0.2, 0.2, 0.2])[None, :]
    dev = points.device if points.is_cuda else _device()
    out = K.points_in_boxes_mask(points.float().to(dev), boxes.float().to(dev), 1e-2)
    out = out if points.is_cuda else out.cpu()
    return out.numpy() if is_numpy else out


def points_in_boxes_gpu(points, boxes):
    """points (B,M,3), boxes (B,T,7) -> int32 (B,M): index of the first containing box, -1 bg."""
    assert boxes.shape[0] == points.shape[0]
    assert boxes.shape[2] == 7 and points.shape[2] == 3
    return K.points_in_boxes_index(points.float(), boxes.float(), 1e-5)


class RoIAwarePool3d(torch.nn.Module):
    def __init__(self, out_size, max_pts_each_voxel=128):
        super().__init__()
        self.out_size = out_size
        self.max_pts_each_voxel = max_pts_each_voxel

    def forward(self, rois, pts, pts_feature, pool_method="max"):
        assert pool_method in ["max", "avg"]
        return RoIAwarePool3dFunction.apply(rois, pts, pts_feature, self.out_size,
                                            self.max_pts_each_voxel, pool_method)


class RoIAwarePool3dFunction(torch.autograd.Function):
    """rois (N,7), pts (P,3), pts_feature (P,C) -> pooled (N, X, Y, Z, C); reference
    roiaware_pool3d_utils.py:55-107 (forward + atomicAdd backward) on lc_roiaware_pool3d_*."""

    @staticmethod
    def forward(ctx, rois, pts, pts_feature, out_size, max_pts_each_voxel, pool_method):
        assert rois.shape[1] == 7 and pts.shape[1] == 3
        if isinstance(out_size, int):
            out_size = (out_size,) * 3
        else:
            assert len(out_size) == 3 and all(isinstance(k, int) for k in out_size)
        method = {"max": 0, "avg": 1}[pool_method]
        pooled, vox, argmax = K.roiaware_pool3d_forward(rois.float(), pts.float(),
                                                        pts_feature.float(), tuple(out_size),
                                                        max_pts_each_voxel, method)
        ctx.roiaware_pool3d_for_backward = (vox, argmax, method, pts.shape[0])
        return pooled

    @staticmethod
    def backward(ctx, grad_out):
        vox, argmax, method, num_pts = ctx.roiaware_pool3d_for_backward
        grad_in = K.roiaware_pool3d_backward(vox, argmax, grad_out.float(), num_pts, method)
        return None, None, grad_in, None, None, None
