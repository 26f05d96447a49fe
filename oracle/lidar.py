"""CPU restatement of range-image utilities and the point-cloud -> range-image projection.
TEST INFRASTRUCTURE -- see oracle/__init__.py.

Follows /root/reference/lidargen/utils/lidar.py:34-132 (LiDARUtility) and
/root/reference/lidargen/dataset/transforms_3d/common.py:26-91 (load_points_as_images,
scan_unfolding=False) + :11-14 (scatter).
"""
from __future__ import annotations

import numpy as np
import torch


def convert_depth(metric, min_depth, max_depth, depth_format="log_depth", mask=None):
    if mask is None:
        mask = ((metric > min_depth) & (metric < max_depth)).float()
    if depth_format == "log_depth":
        n = torch.log2(metric + 1) / np.log2(max_depth + 1)
    elif depth_format == "inverse_depth":
        n = min_depth / metric.add(1e-8)
    elif depth_format == "depth":
        n = metric.div(max_depth)
    else:
        raise ValueError(depth_format)
    return n.clamp(0, 1) * mask


def revert_depth(normalized, min_depth, max_depth, depth_format="log_depth"):
    if depth_format == "log_depth":
        m = torch.exp2(normalized * np.log2(max_depth + 1)) - 1
    elif depth_format == "inverse_depth":
        m = min_depth / normalized.add(1e-8)
    elif depth_format == "depth":
        m = normalized.mul(max_depth)
    else:
        raise ValueError(depth_format)
    return m * ((m > min_depth) & (m < max_depth)).float()


def to_xyz(metric, ray_angles, min_depth, max_depth):
    mask = ((metric > min_depth) & (metric < max_depth)).float()
    phi, theta = ray_angles[:, [0]], ray_angles[:, [1]]
    xyz = torch.cat([metric * phi.cos() * theta.cos(), metric * phi.cos() * theta.sin(),
                     metric * phi.sin()], dim=1)
    return xyz * mask


def project_cells(points: np.ndarray, H, W, fov_up, fov_down, mode="f32"):
    """Per-point (grid_h, grid_w, depth) -- common.py:44-45, 72-81.

    mode="f32"   : THE CONTRACT of the HIP kernel.  Every operation is a correctly-rounded
                   float32 operation (the reference's pinned numpy 1.23.5 keeps float32
                   throughout, environment.yml:233); asin/atan2 are evaluated in float64 and
                   rounded once to float32, i.e. correctly-rounded float32 libm.
    mode="native": what the reference computes under numpy>=2 (NEP 50): np.deg2rad() of a
                   python float is a float64 *scalar*, which promotes the elevation maths to
                   float64.  Used only to compare against fixtures generated
                   by running the reference in this container.
    """
    pts = points.astype(np.float32)
    x, y, z = pts[:, 0], pts[:, 1], pts[:, 2]
    depth = np.sqrt((x * x + y * y) + z * z)  # == np.linalg.norm(xyz, axis=1) in float32
    f32 = np.float32
    if mode == "f32":
        h_up, h_down = f32(np.deg2rad(fov_up)), f32(np.deg2rad(fov_down))
        t = z / (depth + f32(1e-6))
        elev = np.arcsin(t.astype(np.float64)).astype(f32) + abs(h_down)
        gh = f32(1) - elev / (h_up - h_down)
        gh = np.floor(gh * f32(H)).clip(0, H - 1).astype(np.int32)
        az = -(np.arctan2(y.astype(np.float64), x.astype(np.float64)).astype(f32))
        gw = np.mod((az / f32(np.pi) + f32(1)) / f32(2), f32(1))
        gw = np.floor(gw * f32(W)).clip(0, W - 1).astype(np.int32)
    elif mode == "native":
        h_up, h_down = np.deg2rad(fov_up), np.deg2rad(fov_down)
        elev = np.arcsin(z / (depth + 1e-6)) + abs(h_down)
        gh = 1 - elev / (h_up - h_down)
        gh = np.floor(gh * H).clip(0, H - 1).astype(np.int32)
        az = -np.arctan2(y, x)
        gw = (az / np.pi + 1) / 2 % 1
        gw = np.floor(gw * W).clip(0, W - 1).astype(np.int32)
    else:
        raise ValueError(mode)
    return gh, gw, depth


def load_points_as_images(points: np.ndarray, H, W, min_depth=1.45, max_depth=80.0,
                          fov_up=10.0, fov_down=-30.0, mode="f32"):
    """[N,4] (x,y,z,intensity) -> ([H,W,6] (x,y,z,i,depth,mask), winner[H,W] point index or -1).
    Far-to-near last-write-wins == nearest point per cell wins; equal-depth ties are
    implementation-defined in the reference (np.argsort quicksort) -- here and in the HIP
    kernel the LOWEST point index wins."""
    gh, gw, depth = project_cells(points, H, W, fov_up, fov_down, mode)
    mask = ((depth >= min_depth) & (depth <= max_depth)).astype(np.float32)
    feats = np.concatenate([points.astype(np.float32), depth[:, None], mask[:, None]], axis=1)
    order = np.lexsort((-np.arange(len(depth)), -depth))  # far first; among ties high idx first
    img = np.zeros((H, W, 6), np.float32)
    win = np.full((H, W), -1, np.int32)
    img[gh[order], gw[order]] = feats[order]
    win[gh[order], gw[order]] = order.astype(np.int32)
    return img, win
