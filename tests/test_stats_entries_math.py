"""The arithmetic of the producer-side GroupNorm statistics entries (include/lidarcrafter_hip.h lc_oct_stats),
restated in numpy: a producer leaves {pivot, n, S = sum(v - pivot), Q = sum((v - pivot)^2)} per (channel unit, slot) in float32;
a consumer re-centres every entry of a group on ONE pivot P0 -- S' = S + n d, Q' = Q + d (2 S + n d), d = pivot - P0 -- and
sums N, S', Q' in float64: mean = P0 + S'/N, var = Q'/N - (S'/N)^2.  Units of 8 / 4 / 2 / 1 channels per entry, several slots
per plane, groups of any whole number of entries, two segments with different units (a concat of two producers).  CPU suite."""
import numpy as np
import pytest


def producer_entries(y, unit, slot_px):
    """y [C, P] float32 -> entries [C / unit, slots, 4] float32, pivot = first value of the (unit, slot) block."""
    C, P = y.shape
    slots = (P + slot_px - 1) // slot_px
    e = np.zeros((C // unit, slots, 4), np.float32)
    for u in range(C // unit):
        for s in range(slots):
            blk = y[u * unit:(u + 1) * unit, s * slot_px:(s + 1) * slot_px].astype(np.float32)
            p = np.float32(blk[0, 0])
            d = (blk - p).astype(np.float32)
            e[u, s] = (p, np.float32(blk.size), np.float32(d.sum(dtype=np.float32)), np.float32((d * d).sum(dtype=np.float32)))
    return e


def fold(entries):
    """entries [k, 4] of ONE group -> (mean, var) in float64, the consumer's order of operations."""
    e = entries.astype(np.float64)
    P0 = e[0, 0]
    d = e[:, 0] - P0
    n, S, Q = e[:, 1], e[:, 2], e[:, 3]
    N = n.sum()
    Sp = (S + n * d).sum()
    Qp = (Q + d * (2.0 * S + n * d)).sum()
    m = Sp / N
    return P0 + m, max(Qp / N - m * m, 0.0)


@pytest.mark.parametrize("C,G,units,split", [(64, 8, (8,), None), (128, 32, (4,), None), (128, 32, (4, 2), 64), (64, 32, (2,), None),
                                             (64, 64, (1,), None), (256, 32, (8, 4), 128), (96, 24, (2, 1), 36)])
@pytest.mark.parametrize("offset", [0.0, 37.5])
def test_fold_matches_direct_statistics(C, G, units, split, offset):
    g = np.random.default_rng(C * 131 + G)
    P, slot_px = 1000, 96                                   # ragged last slot
    y = (g.standard_normal((C, P)) * (1.0 + g.random((C, 1))) + offset + g.standard_normal((C, 1))).astype(np.float32)
    cpg = C // G
    if split is None:
        segs = [(0, C, units[0])]
    else:
        segs = [(0, split, units[0]), (split, C, units[1])]
    ent = {c0: producer_entries(y[c0:c1], u, slot_px) for c0, c1, u in segs}
    for grp in range(G):
        lo = grp * cpg
        c0, c1, u = next(sg for sg in segs if sg[0] <= lo < sg[1])
        assert cpg % u == 0 and (lo - c0) % u == 0 and lo + cpg <= c1        # the kernels' preconditions (os_from_segments)
        e = ent[c0][(lo - c0) // u:(lo - c0 + cpg) // u].reshape(-1, 4)
        mean, var = fold(e)
        ref = y[lo:lo + cpg].astype(np.float64)
        assert abs(mean - ref.mean()) < 2e-6 * max(1.0, abs(ref.mean()))
        assert abs(var - ref.var()) < 3e-6 * ref.var()
