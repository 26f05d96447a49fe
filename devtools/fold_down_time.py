"""Timing of Block.downsample at the three C2 shapes (batch B): the reference's order on the HIP kernels (3x3 conv at full
resolution + FIR x2 down pass) against the folded form (FIR pre-filter + stride-2 conv).  python devtools/fold_down_time.py [B]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lidarcrafter_amd import ops as K  # noqa: E402
from lidarcrafter_amd.testing import rel_l2, seeded_fill, seeded_randn  # noqa: E402
from lidargen.models.unets import ops  # noqa: E402


def t_us(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    dev = torch.device("cuda:0")
    for Ci, Co, H, W in ((64, 128, 32, 1024), (128, 256, 16, 512), (256, 512, 8, 256)):
        conv = seeded_fill(ops.Conv2d(Ci, Co, 3, 1, 1, ring=True), salt=Ci).to(dev)
        x = seeded_randn(B, Ci, H, W, seed=Ci).to(dev)
        with torch.no_grad():
            full = torch.empty((B, Co, H, W), device=dev)
            a = K.resample2x(conv(x, out=full), up=False)
            b = K.conv_down2(x, conv._packed, conv.weight, conv.bias, emit_stats=8)
            r = rel_l2(b, a)
            t_conv = t_us(lambda: conv(x, out=full))
            t_down = t_us(lambda: K.resample2x(full, up=False))
            t_old = t_us(lambda: K.resample2x(conv(x, out=full), up=False))
            t_new = t_us(lambda: K.conv_down2(x, conv._packed, conv.weight, conv.bias, emit_stats=8))
            K.PROFILE = []
            for _ in range(20):
                K.conv_down2(x, conv._packed, conv.weight, conv.bias, emit_stats=8)
            torch.cuda.synchronize()
            rec, K.PROFILE = K.PROFILE, None
            tp = sum(e0.elapsed_time(e1) for n_, _, e0, e1, *_ in rec if n_ == "resample") * 1e3 / 20
            tc = sum(e0.elapsed_time(e1) for n_, _, e0, e1, *_ in rec if n_ == "conv3x3") * 1e3 / 20
        import hashlib
        digest = hashlib.sha1(b.cpu().numpy().tobytes()).hexdigest()[:12]
        gf = 2.0 * B * (H // 2) * (W // 2) * Co * Ci * 9 / 1e9
        print(f"B={B} {Ci}->{Co} @ {H}x{W}: conv {t_conv:.1f} + down {t_down:.1f} = {t_old:.1f} us | folded {t_new:.1f} us "
              f"(pre-filter {tp:.1f}, stride-2 conv {tc:.1f} = {gf / tc * 1e3:.0f} TFLOP/s executed) | rel-L2 {r:.2e} sha1 {digest}", flush=True)


if __name__ == "__main__":
    main()
