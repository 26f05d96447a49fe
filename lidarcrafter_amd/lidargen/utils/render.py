"""Visualisation helpers the reference's CLIs and training loops import from `lidargen.utils.render`
(tools/generate/generate.py:56-66, generate_cond.py:113-123, train_lidm_cond.py:224-242):
`colorize`, `make_Rt`, `render_point_clouds`, `bilinear_rasterizer`, `estimate_surface_normal`.

API mirror of the reference's lidargen/utils/render.py (same names, arguments, shapes, dtypes),
written on plain torch tensor ops -- this is logging / image export AFTER the sampling loop, not the
hot path, and it runs on whatever device its inputs live on (so the frames never have to leave the
GPU before they are rasterised).  The reference builds its rotations and its pinhole projection
with `kornia` (absent here); both are closed forms restated below:
  axis_angle_to_rotation_matrix(0,0,a) etc. = the elementary rotations Rz / Ry / Rx (Rodrigues),
  project_points(p, K)                      = (fx x/z + cx, fy y/z + cy).
Parity: `bilinear_rasterizer`, `estimate_surface_normal` and `colorize` are pinned on outputs of the
reference functions (tests/golden/render.npz); `make_Rt` / `render_point_clouds` depend on kornia
in the reference and are checked against the closed forms only ("parity unpinned")."""
from __future__ import annotations

import math

import numpy as np
import torch

try:                                            # the reference's default colour map (cm.turbo)
    import matplotlib.cm as cm

    _DEFAULT_CMAP = cm.turbo
except Exception:                               # pragma: no cover - matplotlib is optional
    cm = None

    def _DEFAULT_CMAP(x):
        """Polynomial approximation of the `turbo` colour map (A. Mikhailov, Google, 2019)."""
        x = np.asarray(x, np.float64)
        v = np.stack([np.ones_like(x), x, x ** 2, x ** 3, x ** 4, x ** 5], -1)
        r = v @ [0.13572138, 4.61539260, -42.66032258, 132.13108234, -152.94239396, 59.28637943]
        g = v @ [0.09140261, 2.19418839, 4.84296658, -14.18503333, 4.27729857, 2.82956604]
        b = v @ [0.10667330, 12.64194608, -60.58204836, 110.36276771, -89.90310912, 27.34824973]
        return np.clip(np.stack([r, g, b, np.ones_like(x)], -1), 0.0, 1.0)


def _rot(axis: int, angle: float, device) -> torch.Tensor:
    c, s = math.cos(angle), math.sin(angle)
    i, j = [(1, 2), (2, 0), (0, 1)][axis]       # the plane the rotation acts in (right-handed)
    R = torch.eye(3, device=device)
    R[i, i], R[i, j], R[j, i], R[j, j] = c, -s, s, c
    return R


def make_Rt(roll: float = 0.0, pitch: float = 0.0, yaw: float = 0.0, x: float = 0.0, y: float = 0.0,
            z: float = 0.0, device="cpu"):
    """Extrinsics of the logging camera: R [1,3,3] = Rz(yaw) Ry(pitch) Rx(roll), t [1,3]
    (reference render.py:9-29)."""
    R = _rot(2, float(yaw), device) @ _rot(1, float(pitch), device) @ _rot(0, float(roll), device)
    return R[None], torch.tensor([[x, y, z]], device=device, dtype=torch.float32)


def bilinear_rasterizer(coords: torch.Tensor, values: torch.Tensor, out_shape):
    """Splat values [B,N,C] at sub-pixel positions coords [B,N,2] = (row, column) into [B,C,H,W] with
    bilinear weights; corners outside the image get weight 0, weights below 1e-3 are dropped
    (reference render.py:83-142)."""
    B, N, C = values.shape
    H, W = out_shape
    r, c = coords[..., 0], coords[..., 1]
    r0, c0 = torch.floor(r), torch.floor(c)
    out = torch.zeros(B, H * W, C, device=coords.device, dtype=values.dtype)
    for dr in (0, 1):
        rr = r0 + dr
        wr = ((r0 + 1) - r) if dr == 0 else (r - r0)
        rs = rr.clamp(0, H - 1)
        wr = wr * (rr == rs)
        for dc in (0, 1):
            cc = c0 + dc
            wc = ((c0 + 1) - c) if dc == 0 else (c - c0)
            cs = cc.clamp(0, W - 1)
            w = wr * (wc * (cc == cs))
            w = w * (w >= 1e-3)
            idx = (cs + W * rs).long()
            out.scatter_add_(1, idx[..., None].expand(-1, -1, C), values * w[..., None])
    return out.reshape(B, H, W, C).permute(0, 3, 1, 2)


def render_point_clouds(points: torch.Tensor, colors: torch.Tensor | None = None, size: int = 800,
                        R: torch.Tensor | None = None, t: torch.Tensor | None = None,
                        focal_length=1.0) -> torch.Tensor:
    """Soft z-buffered pinhole rendering of [B,N,3] points (+ colours [B,N,3]) into [B,3,size,size]
    (reference render.py:32-80): flip z, apply `points @ R + t`, project with fx = fy = focal_length,
    cx = cy = 0.5, weight every splat by exp(-3 * distance)."""
    p = points.clone()
    p[..., 2] = -p[..., 2]
    if colors is None:
        colors = torch.ones(p.shape[0], p.shape[1], 3).to(p)
    if R is not None:
        assert R.shape[-2:] == (3, 3)
        p = p @ R
    if t is not None:
        assert t.shape[-1:] == (3,)
        p = p + t
    zc = p[..., 2:3]
    inv_z = torch.where(zc.abs() > 1e-8, 1.0 / (zc + 1e-8), torch.ones_like(zc))   # kornia's rule
    uv = (p[..., :2] * inv_z * focal_length + 0.5) * size
    inside = ((0 < uv) & (uv < size - 1)).all(dim=-1, keepdim=True)
    colors = colors * inside
    uv = size - uv
    dist = p.norm(p=2, dim=-1, keepdim=True)
    weight = torch.exp(-3.0 * dist) * (dist > 1e-8)
    img = bilinear_rasterizer(uv, weight * colors, (size, size))
    return img / (bilinear_rasterizer(uv, weight, (size, size)) + 1e-8)


def estimate_surface_normal(points: torch.Tensor, d: int = 2, mode: str = "closest") -> torch.Tensor:
    """Surface normals of an organised point image [B,3,H,W] from the 8 neighbours at distance d
    (rows replicate at the border, columns wrap): cross product of the two neighbour vectors of the
    CLOSEST consecutive pair ("closest") or the mean over all 8 pairs ("mean"), normalised
    (reference render.py:145-236)."""
    assert points.dim() == 4, f"expected (B,3,H,W), but got {points.shape}"
    B, C, H, W = points.shape
    assert C == 3, f"expected C==3, but got {C}"
    rows = torch.arange(H, device=points.device)
    cols = torch.arange(W, device=points.device)
    # neighbour k of the reference's table, as (row offset, column offset)
    offs = [(-d, 0), (-d, d), (0, d), (d, d), (d, 0), (d, -d), (0, -d), (-d, -d)]

    def shifted(dh, dw):
        return points[:, :, (rows + dh).clamp(0, H - 1)][:, :, :, (cols + dw) % W]

    nb = torch.stack([shifted(*o) for o in offs], 1) - points[:, None]        # [B,8,3,H,W]
    nb2 = torch.roll(nb, shifts=-2, dims=1)                                    # neighbour (k+2) % 8
    if mode == "closest":
        i = (nb.norm(dim=2) + nb2.norm(dim=2)).argmin(dim=1)                   # [B,H,W]
        gi = i[:, None, None].expand(-1, 1, 3, -1, -1)
        n = torch.cross(nb.gather(1, gi)[:, 0], nb2.gather(1, gi)[:, 0], dim=1)
    elif mode == "mean":
        n = torch.cross(nb, nb2, dim=2).mean(dim=1)
    else:
        raise NotImplementedError(mode)
    return n / (n.norm(dim=1, keepdim=True) + 1e-8)


@torch.no_grad()
def colorize(tensor: torch.Tensor, cmap_fn=_DEFAULT_CMAP) -> torch.Tensor:
    """[B,1,H,W] or [B,H,W] values in [0,1] -> uint8 RGB [B,3,H,W] through a 256-entry table of
    `cmap_fn` (reference render.py:239-246)."""
    table = torch.from_numpy(np.asarray(cmap_fn(np.linspace(0, 1, 256)))[:, :3]).to(tensor)
    t = tensor.squeeze(1) if tensor.ndim == 4 else tensor
    ids = (t * 256).clamp(0, 255).long()
    return table[ids].permute(0, 3, 1, 2).mul(255).clamp(0, 255).byte()
