"""numpy restatement of the layout-condition rasteriser (TEST INFRASTRUCTURE).

Follows /root/reference/lidargen/dataset/transforms_3d/common.py: rotz :93-97,
convert_boxes_to_2d :99-181, convert_points_to_2d :184-215 with the dtypes the reference ends up
with: for float32 boxes cos/sin of the yaw in float32, corner geometry in float64, centre depth in
float32 (pinned by tests/golden/layout_cond.npz, the reference's own output); for float64 boxes --
what `NuscDataset.pre_process` hands over (nuscenes_dataset.py:384-386: float64 `gt_boxes` with the
class column appended) -- everything in float64, the centre depth rounded to float32 only when it is
painted (pinned by tests/golden/pipe_next.npz, the reference's own CustomDataset item)."""
from __future__ import annotations

import numpy as np


def box_rectangles(boxes: np.ndarray, H: int, W: int, fov_up=10.0, fov_down=-30.0):
    """boxes float32 / float64 [n, >=7] -> (corners_2d float64 [n,4] (x1,y1,x2,y2 normalised),
    rect int [n,4] (x1,y1,x2,y2 pixels), wrap bool [n], depth float32 [n])."""
    f64 = np.asarray(boxes).dtype == np.float64
    b = np.asarray(boxes, np.float64 if f64 else np.float32)
    n = len(b)
    l, w, h = b[:, 3].astype(np.float64), b[:, 4].astype(np.float64), b[:, 5].astype(np.float64)
    sx = np.array([1, 1, -1, -1, 1, 1, -1, -1]) * 0.5
    sy = np.array([1, -1, -1, 1, 1, -1, -1, 1]) * 0.5
    sz = np.array([1, 1, 1, 1, -1, -1, -1, -1]) * 0.5
    X, Y, Z = l[:, None] * sx, w[:, None] * sy, h[:, None] * sz
    c = np.cos(b[:, 6]).astype(b.dtype).astype(np.float64)[:, None]   # float32 cos for float32 boxes
    s = np.sin(b[:, 6]).astype(b.dtype).astype(np.float64)[:, None]
    cx, cy, cz = (b[:, k].astype(np.float64)[:, None] for k in range(3))
    px, py, pz = c * X - s * Y + cx, s * X + c * Y + cy, Z + cz
    depth = np.sqrt(px * px + py * py + pz * pz) + 1e-6
    h_up, h_down = np.deg2rad(fov_up), np.deg2rad(fov_down)
    gh = 1 - (np.arcsin(pz / depth) + abs(h_down)) / (h_up - h_down)
    gh = np.floor(gh * H).clip(0, H - 1) / H
    gw = (-np.arctan2(py, px) / np.pi + 1) / 2 % 1
    gw = np.floor(gw * W).clip(0, W - 1) / W
    c2d = np.stack([gw.min(1), gh.min(1), gw.max(1), gh.max(1)], 1)
    rect = np.stack([(c2d[:, 0] * W).astype(int), (c2d[:, 1] * H).astype(int),
                     (c2d[:, 2] * W).astype(int), (c2d[:, 3] * H).astype(int)], 1)
    wrap = (rect[:, 2] - rect[:, 0]) / W > 0.6
    xyz = b[:, :3]
    cdep = np.sqrt((xyz[:, 0] * xyz[:, 0] + xyz[:, 1] * xyz[:, 1]) + xyz[:, 2] * xyz[:, 2]) + \
        b.dtype.type(1e-6)
    return c2d, rect, wrap, cdep.astype(np.float32)


def convert_boxes_to_2d(boxes: np.ndarray, H: int, W: int, fov_up=10.0, fov_down=-30.0):
    """-> (corners_2d [n,4], condition_mask float32 [2,H,W], scene_loss_weight_map float32 [H,W]):
    painter's order -- later boxes overwrite earlier ones."""
    c2d, rect, wrap, cdep = box_rectangles(boxes, H, W, fov_up, fov_down)
    n = len(boxes)
    mask = np.zeros((2, H, W), np.float32)
    cover = np.zeros((H, W, n), np.float32)
    areas = np.zeros(n, np.float32)
    for k in range(n):
        x1, y1, x2, y2 = rect[k]
        cols = np.zeros(W, bool)
        if wrap[k]:
            cols[:x1] = True
            cols[x2:] = True
            areas[k] = (W - x2 + x1) * (y2 - y1)
        else:
            cols[x1:x2] = True
            areas[k] = (x2 - x1) * (y2 - y1)
        rows = np.zeros(H, bool)
        rows[y1:y2] = True
        sel = rows[:, None] & cols[None, :]
        mask[0][sel] = boxes[k, 7]
        mask[1][sel] = cdep[k]
        cover[..., k][sel] = 1.0
    wts = (3 - areas / np.max(areas))[None, None, :]
    wmap = np.exp(np.sum(cover * wts, axis=-1))
    return c2d, mask, wmap.astype(np.float32)
