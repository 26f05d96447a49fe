"""Do a memory-bound 1x1 skip convolution and the MFMA-bound 3x3 conv1 of the same ResidualBlock overlap when they are issued
on two branches of a HIP graph?  (u_block1.RB0 of the C2 denoiser: skip 128 -> 64 @ 32x1024, conv1 128 -> 64 with the fused
GroupNorm, batch B.)  python devtools/overlap_probe.py [B]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lidarcrafter_amd import ops as K  # noqa: E402
from lidarcrafter_amd.testing import seeded_fill, seeded_randn  # noqa: E402
from lidargen.models.unets import ops  # noqa: E402


def timed_graph(fn, reps=20):
    for _ in range(3):
        fn()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        g.replay()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / reps)
    return min(ts) * 1e6


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    dev = torch.device("cuda:0")
    for Ci, Co, H, W in ((128, 64, 32, 1024), (256, 64, 16, 512), (512, 128, 8, 256)):
        x = seeded_randn(B, Ci, H, W, seed=Ci).to(dev)
        gn = seeded_fill(ops.GroupNorm(8, Ci, 1e-6), salt=1).to(dev)
        conv1 = seeded_fill(ops.Conv2d(Ci, Co, 3, 1, 1, ring=True), salt=2).to(dev)
        skip = seeded_fill(ops.Conv2d(Ci, Co, 1, 1, 0), salt=3).to(dev)
        h = torch.empty((B, Co, H, W), device=dev)
        sk = torch.empty((B, Co, H, W), device=dev)
        with torch.no_grad():
            st = gn.coeffs(x) if K.fuse_gn(Co) else None
            a = None if st is not None else gn(x, act_silu=True, split_for=conv1._packed)

            def f_conv():
                if st is not None:
                    conv1(x, gn_coeffs=st, out=h)
                else:
                    conv1(a, out=h)

            def f_skip():
                skip(x, out=sk)

            side = torch.cuda.Stream()

            def both_serial():
                f_conv(); f_skip()

            def both_forked():
                main_s = torch.cuda.current_stream()
                side.wait_stream(main_s)
                with torch.cuda.stream(side):
                    f_skip()
                f_conv()
                main_s.wait_stream(side)

            t1, t2 = timed_graph(f_conv), timed_graph(f_skip)
            ts, tf = timed_graph(both_serial), timed_graph(both_forked)
        print(f"B={B} {Ci}->{Co} @ {H}x{W}: conv1 {t1:.1f} us, skip {t2:.1f} us, serial {ts:.1f} us, forked {tf:.1f} us", flush=True)


if __name__ == "__main__":
    main()
