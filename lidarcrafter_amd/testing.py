"""Deterministic synthetic weights / inputs shared by tests, bench and the fixture generator.

There are no pretrained checkpoints in this environment, and a freshly
constructed denoiser has 82 zero-initialised tensors (reference `ops.zero_out`
lidargen/models/unets/ops.py:9-11, `zero_module` nn.py:79-85) so it would output exactly 0.
`seeded_fill` therefore overwrites EVERY parameter with values that depend only on the
parameter's state_dict key and shape -- not on construction order or torch's global RNG --
so the reference modules (fixture generator) and this repo's modules get bit-identical
weights as long as their state_dict keys/shapes agree (which is itself the checkpoint contract).
"""
from __future__ import annotations

import zlib

import torch


def _gen_for(key: str, salt: int) -> torch.Generator:
    g = torch.Generator(device="cpu")
    g.manual_seed((zlib.crc32(key.encode()) ^ (salt * 2654435761)) & 0x7FFFFFFF)
    return g


@torch.no_grad()
def seeded_fill(module: torch.nn.Module, salt: int = 0) -> torch.nn.Module:
    """Fill all parameters of `module` in place, keyed by their state_dict name."""
    for key, p in module.named_parameters():
        g = _gen_for(key, salt)
        r = torch.randn(p.shape, generator=g, dtype=torch.float32)
        if p.ndim >= 2:
            fan_in = max(1, p.numel() // p.shape[0])
            v = r / fan_in ** 0.5
        elif key.endswith("weight"):  # norm gains
            v = 1.0 + 0.1 * r
        else:  # biases
            v = 0.1 * r
        p.copy_(v.to(device=p.device, dtype=p.dtype))
    return module


def seeded_randn(*shape: int, seed: int) -> torch.Tensor:
    g = torch.Generator(device="cpu").manual_seed(seed)
    return torch.randn(*shape, generator=g, dtype=torch.float32)


def rel_l2(a: torch.Tensor, b: torch.Tensor) -> float:
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def synth_points(N: int, seed: int):
    """Synthetic LiDAR sweep: azimuth U(-pi,pi), elevation U(-30.5,10.5) deg,
    range log-U(0.8,95) m, intensity U(0,255) -> float32 [N,4]."""
    import numpy as np

    g = np.random.default_rng(seed)
    az = g.uniform(-np.pi, np.pi, N)
    el = np.deg2rad(g.uniform(-30.5, 10.5, N))
    r = np.exp(g.uniform(np.log(0.8), np.log(95.0), N))
    return np.stack([r * np.cos(el) * np.cos(az), r * np.cos(el) * np.sin(az), r * np.sin(el),
                     g.uniform(0, 255, N)], axis=1).astype(np.float32)


def synth_layout_batch(B: int, H: int, W: int, seed: int, n_extra: int = 0) -> dict:
    """Synthetic layout-condition batch: 13 object slots,
    n_valid ~ U{1..12}; scaled 3-D boxes U(-1,1) with class id U{1..8} in the last column (0 for
    padding); sorted 2-D corners U(0,1); `concat_cond` [B,10,H,W] = one-hot(9) of a rasterised
    class map + one log-depth channel (the output of `preprocess_condition_mask`,
    tools/evaluation/sample_and_save_cond.py:106-117).  n_extra>0 adds `autoregressive_cond`."""
    import numpy as np

    g = np.random.default_rng(seed)
    boxes = np.zeros((B, 13, 9), np.float32)
    b2d = np.zeros((B, 13, 4), np.float32)
    valid = np.zeros((B, 13), np.float32)
    cls_map = np.zeros((B, H, W), np.int64)
    dep_map = np.zeros((B, H, W), np.float32)
    for b in range(B):
        n = int(g.integers(1, 13))
        valid[b, :n] = 1
        boxes[b, :n, :8] = g.uniform(-1, 1, (n, 8))
        boxes[b, :n, 8] = g.integers(1, 9, n)
        xs = np.sort(g.uniform(0, 1, (n, 2)), axis=1)
        ys = np.sort(g.uniform(0, 1, (n, 2)), axis=1)
        b2d[b, :n] = np.stack([xs[:, 0], ys[:, 0], xs[:, 1], ys[:, 1]], 1)
        for k in range(n):
            x0, x1 = int(xs[k, 0] * W), max(int(xs[k, 1] * W), int(xs[k, 0] * W) + 1)
            y0, y1 = int(ys[k, 0] * H), max(int(ys[k, 1] * H), int(ys[k, 0] * H) + 1)
            cls_map[b, y0:y1, x0:x1] = int(boxes[b, k, 8])
            dep_map[b, y0:y1, x0:x1] = g.uniform(2, 60)
    onehot = np.eye(9, dtype=np.float32)[cls_map].transpose(0, 3, 1, 2)
    logd = np.clip(np.log2(dep_map + 1) / np.log2(81.0), 0, 1) * ((dep_map > 1.45) & (dep_map < 80))
    out = {
        "scaled_gt_boxes": torch.from_numpy(boxes),
        "gt_boxes_2d": torch.from_numpy(b2d),
        "is_valid_obj": torch.from_numpy(valid),
        "concat_cond": torch.from_numpy(np.ascontiguousarray(
            np.concatenate([onehot, logd[:, None].astype(np.float32)], 1))),
    }
    if n_extra:
        out["autoregressive_cond"] = torch.from_numpy(
            g.uniform(0, 1, (B, n_extra, H, W)).astype(np.float32))
    return out


def synth_boxes(n: int, pts, seed: int):
    """n rotated boxes [x,y,z,dx,dy,dz,heading] centred on random points of `pts` (float32)."""
    import numpy as np

    g = np.random.default_rng(seed)
    bx = np.stack([np.zeros(n), np.zeros(n), np.zeros(n), g.uniform(1.5, 10, n),
                   g.uniform(1.5, 5, n), g.uniform(1.5, 4, n), g.uniform(-3.2, 3.2, n)],
                  1).astype(np.float32)
    bx[:, :3] = pts[g.integers(0, len(pts), n), :3]
    return bx


def synth_scene_boxes(n: int, seed: int):
    """n scene boxes float32 [n, 8] = (x, y, z, l, w, h, yaw, class 1..8) at 5-60 m range, one of
    them straddling the +-pi azimuth seam (exercises the wrap-around case of convert_boxes_to_2d)."""
    import numpy as np

    g = np.random.default_rng(seed)
    r = g.uniform(5, 60, n)
    az = g.uniform(-np.pi, np.pi, n)
    az[0] = np.pi - 0.01
    b = np.stack([r * np.cos(az), r * np.sin(az), g.uniform(-2.0, 0.5, n), g.uniform(1.5, 9, n),
                  g.uniform(1.2, 3, n), g.uniform(1.2, 3.5, n), g.uniform(-np.pi, np.pi, n),
                  g.integers(1, 9, n).astype(np.float64)], 1)
    return b.astype(np.float32)


def synth_temporal_inputs(seed=0, K=5, T=6):
    """Seeded inputs of the temporal glue: ego + K object per-step offsets, K boxes, a point set."""
    import numpy as np

    g = np.random.default_rng(seed)
    ego = np.stack([g.normal(0.05, 0.02, T), g.uniform(0.3, 1.2, T)], 1)      # mostly forward (+y)
    ego[2] = [0.01, 0.02]                                                      # a < 0.1 m step
    obj = g.normal(0.0, 0.6, (K, T, 2))
    obj[1] = 0.0                                                               # a standing object
    obj[2, 3] = 0.0                                                            # stops for one step
    trajs = np.concatenate([ego[None], obj], 0)                                # [1+K, T, 2] offsets
    r, az = g.uniform(6, 40, K), g.uniform(-np.pi, np.pi, K)
    boxes = np.stack([r * np.cos(az), r * np.sin(az), g.uniform(-1.5, 0.0, K), g.uniform(1.5, 6, K),
                      g.uniform(1.2, 2.5, K), g.uniform(1.2, 2.5, K), g.uniform(-np.pi, np.pi, K)], 1)
    return trajs, boxes


def synth_object_batch(B: int, seed: int) -> dict:
    """Synthetic foreground-object condition batch: box codes `fg_encoding_box`
    [B, 6] = (x, y, z, l, w, unique yaw) in the scaled ranges of nuscenes_dataset.encoding_boxes_3d
    and class ids `fg_class` [B] in 0..7."""
    g = torch.Generator().manual_seed(seed)
    box = torch.rand(B, 6, generator=g) * 2 - 1
    cls = torch.randint(0, 8, (B,), generator=g)
    return {"fg_encoding_box": box, "fg_class": cls}


def synth_text_features(seed: int = 77) -> dict:
    """Stand-in for the CLIP class-name features of obj_text_feat.pkl (8 x 512, unit norm)."""
    names = ["car", "truck", "construction_vehicle", "bus", "trailer", "motorcycle", "bicycle",
             "pedestrian"]
    g = torch.Generator().manual_seed(seed)
    f = torch.randn(8, 512, generator=g)
    f = f / f.norm(dim=1, keepdim=True)
    return {n: f[i] for i, n in enumerate(names)}
