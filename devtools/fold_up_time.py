"""Timing of `Resample(up=2) -> Conv2d(3x3, ring)` at the up-path shapes of C2 (EfficientUNet Block.upsample: plain input) and
C3 (LayoutUnetV1 up-sampling ResBlock: GroupNorm + SiLU in front, x up-sampled beside it), batch B: the reference's order on
the HIP kernels against the fold (split / apply+split pass at the low resolution, 1x1 projection to nine tap planes, combine
pass).  Every figure = 20 launches of the sequence replayed as one HIP graph, best of 5.
    python devtools/fold_up_time.py [B]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("LC_FOLD_UP_MIN_CI", "32")
from lidarcrafter_amd import ops as K  # noqa: E402
from lidarcrafter_amd.testing import rel_l2, seeded_fill, seeded_randn  # noqa: E402
from lidargen.models.unets import ops  # noqa: E402


def graph_us(fn, n=20, reps=5):
    fn()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n):
                fn()
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / n)
    return best


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    dev = torch.device("cuda:0")
    shapes = [("C2 u4", 256, 4, 128, False), ("C2 u3", 128, 8, 256, False), ("C2 u2", 64, 16, 512, False),
              ("C3 up1", 512, 4, 128, True), ("C3 up2", 256, 8, 256, True), ("C3 up3", 128, 16, 512, True)]
    for tag, C, H, W, gn in shapes:
        conv = seeded_fill(ops.Conv2d(C, C, 3, 1, 1, ring=True), salt=C).to(dev)
        norm = seeded_fill(ops.GroupNorm(32, C), salt=C + 1).to(dev) if gn else None
        x = (seeded_randn(B, C, H, W, seed=C) + 0.2).to(dev)
        pk = K.PackedConv("time.up9")
        with torch.no_grad():
            w9 = K.up9_weight(conv.weight)
            if gn:
                xn = x.clone()          # (statistics pass inside both routes: no producer here)

                def old():
                    a, xu = K.groupnorm_resample_pair(xn, 32, norm.eps, norm.weight, norm.bias, up=True)
                    return conv(a, emit_stats=True), xu

                def new():
                    a = norm(xn, act_silu=True, split_for=pk)
                    return K.conv_up2(a, pk, w9, conv.bias, emit_stats=True), K.resample2x(xn, up=True)

                def new_front():
                    return norm(xn, act_silu=True, split_for=pk), K.resample2x(xn, up=True)
            else:
                def old():
                    return conv(K.resample2x(x, up=True), emit_stats=True), None

                def new():
                    return K.conv_up2(K.split_act(x, pk), pk, w9, conv.bias, emit_stats=True), None

                def new_front():
                    return K.split_act(x, pk), None
            r = rel_l2(new()[0], old()[0])
            t_old, t_new, t_front = graph_us(old), graph_us(new), graph_us(new_front)
            xs = new_front()[0]
            p9 = torch.empty((B, 9 * C, H, W), device=dev)
            wh, wl = pk.get_f16x2(w9)

            def proj():
                K.check(K._conv_lib().lc_conv1x1_f16x2_ps_fwd(xs.buf.data_ptr(), wh.data_ptr(), wl.data_ptr(), None, None, 0,
                                                              p9.data_ptr(), 9 * C * H * W, B, C, 9 * C, H, W, 1.0,
                                                              pk.wmeta.data_ptr(), pk.range_ptr(dev),
                                                              torch.cuda.current_stream().cuda_stream), "proj")
            t_proj = graph_us(proj)
        print(f"B={B} {tag} {C}->{C} @ {H}x{W} -> {2 * H}x{2 * W}: reference order {t_old:.1f} us | folded {t_new:.1f} us "
              f"(front passes {t_front:.1f}, 1x1 -> 9 planes {t_proj:.1f} = {2.0 * B * H * W * 9 * C * C / t_proj / 1e6:.0f} TFLOP/s executed, "
              f"combine ~{t_new - t_front - t_proj:.1f}) | rel-L2 {r:.2e}", flush=True)


if __name__ == "__main__":
    main()
