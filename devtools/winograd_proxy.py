"""Proxy measurement for Winograd F(2x2, 3x3) on the pre-split 256 / 512-channel layers, from kernels that exist:
the 16 element-wise GEMMs of F(2x2, 3x3) over H*W/4 tiles cost what ONE 1x1 projection Ci -> Co over 4*H*W pixels costs (same MACs, same
operand bytes: the transformed input is 4x the input), so the pre-split 1x1 kernel on a [B, Ci, H, 4W] tensor is the GEMM core -- generous
to Winograd in that its epilogue is a plain store (no output transform), harsh in that it writes 4x the output bytes.  Next to it: the
3x3 pre-split conv it would replace, and the GroupNorm apply + split pass at 1x and 4x the pixels (what writing the 4x transformed
operand costs).  Every figure = 20 launches replayed as one HIP graph.  python devtools/winograd_proxy.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lidarcrafter_amd import ops as K  # noqa: E402
from lidarcrafter_amd.testing import seeded_fill, seeded_randn  # noqa: E402
from lidargen.models.unets import ops as O  # noqa: E402
from lidargen.models.unets.nn import PointwiseConv1d  # noqa: E402


def graph_time(fn, n=20, reps=5):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best


def main():
    dev = torch.device("cuda:0")
    K.PS1X1_MIN_CO = 128
    B = 8
    from lidargen.models.unets.nn import GroupNorm32
    for C, H, W in ((256, 8, 256), (512, 4, 128)):
        norm = seeded_fill(O.GroupNorm(8, C, 1e-6), salt=1).to(dev)
        conv3 = seeded_fill(O.Conv2d(C, C, 3, 1, 1, ring=True), salt=2).to(dev)
        conv1 = seeded_fill(PointwiseConv1d(C, C), salt=3).to(dev)
        n32 = seeded_fill(GroupNorm32(32, C), salt=6).to(dev)
        with torch.inference_mode():
            x = seeded_randn(B, C, H, W, seed=4).to(dev)
            x4 = seeded_randn(B, C, H, 4 * W, seed=5).to(dev)
            xs3 = norm(x, act_silu=True, split_for=conv3._packed)
            t3 = graph_time(lambda: conv3(xs3, emit_stats=True))
            xs1 = n32(x4.view(B, C, H * 4 * W), split_for=conv1._packed)
            t1 = graph_time(lambda: conv1(xs1))
            tp1 = graph_time(lambda: norm(x, act_silu=True, split_for=conv3._packed))
            tp4 = graph_time(lambda: norm(x4, act_silu=True, split_for=conv3._packed))
            print(f"{C} -> {C} @ {H} x {W}, batch {B}: 3x3 pre-split conv {t3:6.1f} us (+ apply/split pass {tp1:5.1f}) | Winograd proxy: "
                  f"GEMM core = 1x1 over 4x the pixels {t1:6.1f} us, apply/split pass over 4x the pixels {tp4:5.1f} us "
                  f"(reads 4x too: the transform pass would read 1x, write 4x)", flush=True)


if __name__ == "__main__":
    main()
