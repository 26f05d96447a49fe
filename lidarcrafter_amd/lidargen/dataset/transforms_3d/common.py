"""Point cloud -> range image -- API mirror of the reference's
lidargen/dataset/transforms_3d/common.py (mask_points_with_distance :16-24,
load_points_as_images :26-91 with scan_unfolding=False) on the HIP z-buffer kernel
(lc_project_points): one atomic-min pass + one gather pass instead of numpy argsort + scatter."""
from __future__ import annotations

import numpy as np
import torch

from lidarcrafter_amd import ops as K


def mask_points_with_distance(points: np.ndarray, min_depth: float = 1.45, max_depth: float = 80.0):
    x = points[:, :3].astype(np.float32)
    depth = np.sqrt((x[:, 0] * x[:, 0] + x[:, 1] * x[:, 1]) + x[:, 2] * x[:, 2])
    return (depth >= min_depth) & (depth <= max_depth)


def load_points_as_images(point_path: str = None, points=None, scan_unfolding: bool = True,
                          H: int = 64, W: int = 2048, min_depth: float = 1.45,
                          max_depth: float = 80.0, fov_up: float = 10.0, fov_down: float = -30.0,
                          custom_feat_dim: int = 0):
    """-> float32 [H, W, 6] = (x, y, z, intensity, depth, mask).  numpy in -> numpy out, CUDA
    tensor in -> CUDA tensor out (no host round trip).
    Elevation -> row arithmetic in float64 like the reference under numpy >= 2 (ops.PROJECTION_DTYPE
    = "native"; "f32" = its pinned numpy 1.23.5); equal-depth ties go to the lowest
    point index."""
    assert point_path is not None or points is not None, "Either point_path or points must be provided."
    if scan_unfolding:
        raise NotImplementedError("scan_unfolding=True (KITTI ring unfolding) is not used by any "
                                  "nuScenes config; only the spherical projection is on the path")
    if custom_feat_dim:
        raise NotImplementedError("custom_feat_dim > 0 is not on the path")
    if point_path is not None:
        points = np.fromfile(point_path, dtype=np.float32).reshape(-1, 5)[:, :4]
    is_numpy = isinstance(points, np.ndarray)
    if not torch.cuda.is_available():
        raise RuntimeError("load_points_as_images needs the MI355X: no CPU fallback on the hot path")
    # float64 points (the temporal glue's promoted sets) are projected in float64 like the reference
    f64 = (points.dtype == np.float64) if is_numpy else (points.dtype == torch.float64)
    p = torch.from_numpy(np.ascontiguousarray(points[:, :4], np.float64 if f64 else np.float32)) \
        if is_numpy else points
    p = p[:, :4].to(torch.float64 if f64 else torch.float32).contiguous().cuda()
    img, _ = K.project_points(p, H, W, fov_up, fov_down, min_depth, max_depth)
    return img.cpu().numpy() if is_numpy else img


def convert_boxes_to_2d(boxes_3d, H: int = 64, W: int = 2048, min_depth: float = 1.45,
                        max_depth: float = 80.0, fov_up: float = 10.0, fov_down: float = -30.0):
    """boxes_3d [n, >=8] (x,y,z,l,w,h,yaw,class) -> (corners_2d [n,4], condition_mask [2,H,W],
    scene_loss_weight_map [H,W]) -- reference :99-181, on the device (lc_layout_condition): the
    Python loop over boxes becomes one rectangle kernel + one paint kernel.
    numpy in -> numpy out; CUDA tensor in -> CUDA tensors out."""
    is_numpy = isinstance(boxes_3d, np.ndarray)
    if not torch.cuda.is_available():
        raise RuntimeError("convert_boxes_to_2d needs the MI355X: no CPU fallback on the hot path")
    # float64 boxes (NuscDataset.pre_process hands them over, nuscenes_dataset.py:384-397) keep the
    # reference's all-float64 flow; anything else follows its float32 flow
    f64 = (boxes_3d.dtype == np.float64) if is_numpy else (boxes_3d.dtype == torch.float64)
    b = torch.from_numpy(np.ascontiguousarray(boxes_3d, np.float64 if f64 else np.float32)) \
        if is_numpy else boxes_3d
    b = b.to(torch.float64 if f64 else torch.float32).contiguous().cuda()[None]
    n = torch.tensor([b.shape[1]], dtype=torch.int32, device=b.device)
    c2d, mask, wmap = K.layout_condition(b, n, H, W, fov_up, fov_down, with_weight_map=True)
    c2d, mask, wmap = c2d[0], mask[0], wmap[0]
    if is_numpy:
        return c2d.cpu().numpy(), mask.cpu().numpy(), wmap.cpu().numpy()
    return c2d, mask, wmap
