"""The qkv projection of ObjectAwareCrossAttention (GroupNorm split + lc_conv1x1_f16x2_ps_qkv_fwd) at the layout model's shapes,
and the plain pre-split 1x1 projection; LC_P1_BP=256 / 128 forces the pixel tile.  python devtools/qkv_time.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lidarcrafter_amd import ops as K  # noqa: E402
from lidarcrafter_amd.testing import seeded_fill, seeded_randn  # noqa: E402
from lidargen.models.unets.nn import GroupNorm32, PointwiseConv1d  # noqa: E402


def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    dev = torch.device("cuda:0")
    K.PS1X1_MIN_CO = 128
    for B, C, L in ((8, 256, 2048), (8, 512, 512), (1, 256, 2048), (1, 512, 512), (4, 256, 8192), (4, 512, 2048)):
        heads = C // 32
        norm = seeded_fill(GroupNorm32(32, C), salt=41).to(dev)
        proj = seeded_fill(PointwiseConv1d(C, 3 * C), salt=42).to(dev)
        x = seeded_randn(B, C, L, seed=43).to(dev)
        with torch.no_grad():
            xs = norm(x, split_for=proj._packed)
            u = K.AttnUnits(B, heads, L, 13, 32, 32, 32, dev)
            t_plain = timed(lambda: proj(xs))
            t_qkv = timed(lambda: K.qkv_project_units(xs, proj._packed, proj.weight, proj.bias, u))
        print(f"B={B} {C}->{3 * C} @ {L} px, LC_P1_BP={os.environ.get('LC_P1_BP', 'auto')}: plain {t_plain:6.1f} us, "
              f"units epilogue {t_qkv:6.1f} us", flush=True)


if __name__ == "__main__":
    main()
