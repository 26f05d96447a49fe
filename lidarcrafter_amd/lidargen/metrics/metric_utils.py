"""Voxel scatter of the weight-free metrics -- mirror of the reference's
lidargen/metrics/metric_utils.py: `ravel_hash` :28-40, `sparse_quantize` :43-66, `pcd2bev_sum`
:233-258 (the BEV occupancy volume the JSD of eval_utils.compute_jsd :84-95 is computed from).
The point sets stay on the device: sweeps are scattered with atomics (lc_bev_occupancy_accumulate),
unique voxels come from a device radix sort (lc_sparse_quantize).  numpy in -> numpy out, CUDA
tensors in -> CUDA tensors out.  The feature-extractor front-ends of that module (pcd2range,
pcd2voxel, compute_logits: RangeNet++ / MinkowskiNet / SPVCNN inputs) are out of scope
(SURVEY.md section 2 row 17)."""
from __future__ import annotations

import math

import numpy as np
import torch

from lidarcrafter_amd import ops as K

# lidargen/metrics/__init__.py:28-33
VOXEL_SIZE = 0.05
DATA_CONFIG = {"64": {"x": [-50, 50], "y": [-50, 50], "z": [-3, 1]},
               "32": {"x": [-30, 30], "y": [-30, 30], "z": [-3, 6]}}


def _dev(a):
    if isinstance(a, np.ndarray):
        if not torch.cuda.is_available():
            raise RuntimeError("metric_utils needs the MI355X: no CPU fallback on the hot path")
        return torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda(), True
    return a.float().contiguous(), False


def ravel_hash(x):
    """[N, D] integer coordinates -> uint64 hash, row-major over (x - min) with extents max + 1."""
    t = torch.as_tensor(x)
    t = (t - t.min(dim=0).values).to(torch.int64)
    ext = t.max(dim=0).values + 1
    h = torch.zeros(t.shape[0], dtype=torch.int64, device=t.device)
    for k in range(t.shape[1] - 1):
        h = (h + t[:, k]) * ext[k + 1]
    h = h + t[:, -1]
    return h.cpu().numpy().astype(np.uint64) if isinstance(x, np.ndarray) else h


def sparse_quantize(coords, voxel_size=1, *, return_index: bool = False, return_inverse: bool = False):
    c, is_np = _dev(coords)
    out = K.sparse_quantize(c, voxel_size, return_index=return_index, return_inverse=return_inverse)
    if not is_np:
        return out
    if isinstance(out, (list, tuple)):
        return [o.cpu().numpy() for o in out]
    return out.cpu().numpy()


def pcd2bev_sum(data_type, *args, voxel_size=VOXEL_SIZE):
    """For every set of sweeps in `args`: float32 [nx, ny] volume whose cell (i, j) counts the
    sweeps that have at least one point in voxel (i, j) of the BEV range of `data_type`."""
    cfg = DATA_CONFIG[data_type]
    output = tuple()
    for data in args:
        acc, is_np = None, False
        for pcd in data:
            p, is_np = _dev(pcd)
            if acc is None:
                acc = K.BevOccupancy(cfg["x"], cfg["y"], voxel_size, p.device)
            acc.add(p)
        if acc is None:
            vol = torch.zeros((math.ceil((cfg["x"][1] - cfg["x"][0]) / voxel_size),
                               math.ceil((cfg["y"][1] - cfg["y"][0]) / voxel_size)))
            output += (vol.numpy(),)
        else:
            output += (acc.grid.cpu().numpy() if is_np else acc.grid,)
    return output


def compute_jsd(reference, samples, data):
    """eval_utils.compute_jsd :84-95 (value returned instead of printed)."""
    from scipy.spatial.distance import jensenshannon

    r, s = pcd2bev_sum(data, reference, samples)
    r = r.cpu().numpy() if isinstance(r, torch.Tensor) else r
    s = s.cpu().numpy() if isinstance(s, torch.Tensor) else s
    return float(jensenshannon((r / np.sum(r)).flatten(), (s / np.sum(s)).flatten()))
