"""BEV histogram metrics -- mirror of the reference's `lidargen/metrics/bev.py` (same names and
arguments): `point_cloud_to_histogram` :5-24 (the reference bins on the CPU with
torch.histogramdd; here one HIP pass with integer atomics, identical counts), `cdist_rbf` :27-34,
`compute_jsd_2d` :37-45, `compute_mmd_2d` :47-55."""
from __future__ import annotations

import torch

from lidarcrafter_amd import ops as K


def point_cloud_to_histogram(point_cloud: torch.Tensor, field_size: float = 160.0, bins: int = 100,
                             min_depth: float = 3.0, max_depth: float = 70.0) -> torch.Tensor:
    assert point_cloud.ndim == 2, "must be (N, 3)"
    assert bins % 2 == 0
    return K.bev_histogram(point_cloud.float().contiguous(), field_size, bins, min_depth, max_depth)


def cdist_rbf_mean(p: torch.Tensor, q: torch.Tensor, sigma: float = 0.5) -> torch.Tensor:
    """`cdist_rbf(p, q, sigma).mean()` without materialising the [M, Mq] matrix."""
    return K.rbf_kernel_mean(p.float().contiguous(), q.float().contiguous(), sigma)


@torch.no_grad()
def compute_jsd_2d(hist1: torch.Tensor, hist2: torch.Tensor) -> float:
    """BEV Jensen-Shannon distance of the pooled histograms (scipy on 10^4 numbers, as the reference)."""
    from scipy.spatial.distance import jensenshannon

    hist1, hist2 = hist1.flatten(1), hist2.flatten(1)
    p = hist1.sum(dim=0) / hist1.sum()
    q = hist2.sum(dim=0) / hist2.sum()
    return jensenshannon(p.cpu().numpy(), q.cpu().numpy())


@torch.no_grad()
def compute_mmd_2d(hist1: torch.Tensor, hist2: torch.Tensor) -> float:
    """BEV maximum mean discrepancy with the RBF kernel (sigma 0.5) on per-sample histograms."""
    hist1, hist2 = hist1.flatten(1), hist2.flatten(1)
    p = (hist1 / hist1.sum(dim=1, keepdim=True)).contiguous()
    q = (hist2 / hist2.sum(dim=1, keepdim=True)).contiguous()
    mmd = cdist_rbf_mean(p, p) + cdist_rbf_mean(q, q) - 2 * cdist_rbf_mean(p, q)
    return mmd.item()
