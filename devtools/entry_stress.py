"""Stress of the deferred epilogue's GroupNorm statistics entries (conv_f16x2.hip, DefEpi::finalize_with).

Launches the level-0 emitting convolution of the headline config (8 x 64 x 32 x 1024, fused input GroupNorm,
residual) `burst` times back to back -- so that the tail of one launch overlaps the head of the next, as in
the model -- and recomputes EVERY statistics entry of every launch from the output that launch stored.
Prints one line per entry unit: entries checked, entries off.

    LC_HIP_LIB=<library> LC_GN_PRODUCER_STATS=1 python devtools/entry_stress.py --entries 1e8
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from lidarcrafter_amd import ops as K  # noqa: E402
from lidarcrafter_amd.testing import seeded_randn  # noqa: E402


def check(y, unit, B, C, H, W):
    h = y._lc_gnstats[(0, C)]
    assert h.unit == unit and h.slots == (H // 4) * (W // 64) * 4
    e = h.buf.double()
    yv = y.double().view(B, C // unit, unit, H // 4, 4, W // 64, 64).permute(0, 1, 3, 5, 4, 2, 6)
    ref = yv.reshape(B, C // unit, h.slots, unit * 64)
    rs, rq = ref.sum(-1), (ref * ref).sum(-1)
    p, n, s_, q = e[..., 0], e[..., 1], e[..., 2], e[..., 3]
    es, eq = p * n + s_, q + 2 * p * s_ + p * p * n
    bad = (n != unit * 64.0) | ((es - rs).abs() > 2e-3) | (((eq - rq).abs() / rq.clamp(min=1.0)) > 1e-4)
    bad |= ~torch.isfinite(es) | ~torch.isfinite(eq)
    return bad.sum(), e.shape[0] * e.shape[1] * e.shape[2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--entries", type=float, default=1e7, help="entries to check per unit")
    ap.add_argument("--burst", type=int, default=8)
    ap.add_argument("--units", default="8,2")
    a = ap.parse_args()
    K.PRODUCER_GN_STATS = True
    dev = torch.device("cuda:0")
    B, C, H, W = 8, 64, 32, 1024
    x = seeded_randn(B, C, H, W, seed=301).to(dev)
    w = (seeded_randn(C, C, 3, 3, seed=302) / 17.0).to(dev)
    res = seeded_randn(B, C, H, W, seed=303).to(dev)
    pc = K.PackedConv()
    for unit in [int(u) for u in a.units.split(",")]:
        gn = K.groupnorm_stats(x, 32 if unit == 2 else 8, 1e-6)
        total, nbad, launches = 0, torch.zeros((), dtype=torch.long, device=dev), 0
        t0 = time.time()
        while total < a.entries:
            ys = [K.conv2d_ring(x, pc, w, None, tile_cfg=23, emit_stats=True if unit == 8 else 2, gn_coeffs=gn, res=res)
                  for _ in range(a.burst)]
            for y in ys:
                b, n = check(y, unit, B, C, H, W)
                nbad += b
                total += n
            launches += a.burst
        torch.cuda.synchronize()
        print("lib=%s unit=%d launches=%d entries=%d bad=%d (%.1f s)" % (
            os.path.basename(os.environ.get("LC_HIP_LIB", "default")), unit, launches, total, int(nbad),
            time.time() - t0), flush=True)


if __name__ == "__main__":
    main()
