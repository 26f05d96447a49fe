"""Voxel scatter of the metrics front-end (SURVEY.md §8f-3 ii): `pcd2bev_sum`, `sparse_quantize`,
`ravel_hash` of lidargen/metrics/metric_utils.py on the device vs outputs of the reference's own
functions (tests/golden/voxel.npz) -- integer / index work: BIT-EXACT.  `pytest -m gpu`
(`test_ravel_hash` runs on the CPU)."""
import numpy as np
import pytest
import torch

from lidarcrafter_amd.testing import synth_points

T = torch.from_numpy


def test_ravel_hash(golden):
    from lidargen.metrics import metric_utils as M

    g = golden("voxel")
    q = np.floor(synth_points(20000, seed=11)[:500, :3] / 0.7).astype(np.int32)
    assert np.array_equal(M.ravel_hash(q), g["hash"])


@pytest.mark.gpu
@pytest.mark.parametrize("tag,cols,vs", [("2d", 2, 0.5), ("3d", 3, (0.4, 0.4, 0.2))])
def test_sparse_quantize_bit_exact(golden, tag, cols, vs):
    from lidargen.metrics import metric_utils as M

    g = golden("voxel")
    pts = synth_points(20000, seed=11)[:, :cols].copy()
    c, idx, inv = M.sparse_quantize(pts, vs, return_index=True, return_inverse=True)     # numpy path
    assert np.array_equal(c, g[f"sq_{tag}_coords"]) and c.dtype == np.int32
    assert np.array_equal(idx, g[f"sq_{tag}_index"])
    assert np.array_equal(inv, g[f"sq_{tag}_inverse"])
    d = T(pts).cuda()
    cd, id_, iv = M.sparse_quantize(d, vs, return_index=True, return_inverse=True)         # device path
    assert cd.is_cuda and torch.equal(cd.cpu(), T(g[f"sq_{tag}_coords"]))
    assert torch.equal(id_.cpu(), T(g[f"sq_{tag}_index"])) and torch.equal(iv.cpu(), T(g[f"sq_{tag}_inverse"]))
    only = M.sparse_quantize(d, vs)
    assert torch.equal(only, cd)
    # property at a size the CPU reference would take long for: unique rows reproduce the input
    big = T(synth_points(1 << 20, seed=12)[:, :3].copy()).cuda()
    cb, ib, vb = M.sparse_quantize(big, 0.25, return_index=True, return_inverse=True)
    q = torch.floor(big.double() / 0.25).to(torch.int32)
    assert torch.equal(cb[vb], q) and torch.equal(q[ib], cb)
    assert torch.equal(cb, torch.unique(q, dim=0))              # lexicographic order, no duplicates


@pytest.mark.gpu
def test_pcd2bev_sum_bit_exact(golden):
    from lidargen.metrics import metric_utils as M

    g = golden("voxel")
    sets = [[synth_points(30000, seed=20 + 10 * k + i) for i in range(3)] for k in range(2)]
    sets[0][0][:6, :2] = [[30.0, 0.0], [-30.0, 0.0], [29.999998, 1.0], [-29.999998, 1.0],
                          [0.0, 29.999998], [0.025, 0.05]]
    vols = M.pcd2bev_sum("32", sets[0], sets[1])
    dev_sets = [[T(p).cuda() for p in s] for s in sets]
    dvols = M.pcd2bev_sum("32", dev_sets[0], dev_sets[1])
    for k in range(2):
        ref = np.zeros(tuple(g[f"bev{k}_shape"]), np.float32)
        ref.flat[g[f"bev{k}_idx"]] = g[f"bev{k}_cnt"]
        assert vols[k].dtype == np.float32 and np.array_equal(vols[k], ref)
        assert dvols[k].is_cuda and np.array_equal(dvols[k].cpu().numpy(), ref)
    # a sweep counts once per voxel however many points it has there; JSD of a set with itself is 0
    assert float(vols[0].max()) <= 3.0
    assert M.compute_jsd(dev_sets[0], dev_sets[0], "32") < 1e-12
    assert 0.0 < M.compute_jsd(dev_sets[0], dev_sets[1], "32") < 1.0
