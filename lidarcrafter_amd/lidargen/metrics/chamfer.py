"""Chamfer distance -- mirror of the reference's `lidargen/metrics/modules/chamfer3D/
dist_chamfer_3D.py` (`chamfer_3DDist`) and of `compute_pairwise_cd` / `compute_pairwise_cd_batch`
(`lidargen/metrics/metric_utils.py:415-444`).  Forward only (evaluation)."""
from __future__ import annotations

import numpy as np
import torch
from torch import nn

from lidarcrafter_amd import ops as K


class chamfer_3DDist(nn.Module):
    def forward(self, input1, input2):
        return K.chamfer3d(input1.float(), input2.float())


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda() if isinstance(a, np.ndarray) else a


def compute_pairwise_cd(x, y, module=None):
    module = chamfer_3DDist() if module is None else module
    x, y = _dev(x), _dev(y)
    if x.ndim == 2 and y.ndim == 2:
        x, y = x[None], y[None]
    dist1, dist2, _, _ = module(x, y)
    return ((dist1.mean() + dist2.mean()) / 2).item()


def compute_pairwise_cd_batch(reference, samples):
    """One reference cloud against a list of clouds: shorter clouds are padded with points at 1e6
    (as the reference does) and the padded tail is excluded from the means."""
    assert reference.ndim == 2 and reference.shape[1] == 3, "3-D clouds (the 2-D variant is out of scope)"
    len_r, len_s = reference.shape[0], [s.shape[0] for s in samples]
    max_len = max([len_r] + len_s)
    padv = lambda a: np.vstack([a, np.ones((max_len - a.shape[0], 3), dtype=np.float32) * 1e6])
    ref = _dev(padv(np.asarray(reference, np.float32)))
    smp = _dev(np.stack([padv(np.asarray(s, np.float32)) for s in samples]))
    dist_r, dist_s, _, _ = chamfer_3DDist()(ref[None].expand_as(smp).contiguous(), smp)
    return [((dist_r[i, :len_r].mean() + dist_s[i, :len_s[i]].mean()) / 2.).item()
            for i in range(smp.shape[0])]
