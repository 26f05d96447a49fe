"""ObjectAwareCrossAttention's two shapes at batch 8 (ds 4: 8 heads x 2048 + 13 keys; ds 8: 16 heads x 512 + 13): lc_attention_f16x2_fwd
on fp32 operands against the unit form (pack K + pack V per step, lc_attention_units_fwd).  python devtools/attn_units_time.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lidarcrafter_amd import ops as K  # noqa: E402
from lidarcrafter_amd.testing import seeded_randn  # noqa: E402


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
    for B, heads, L in ((8, 8, 2048), (8, 16, 512), (1, 8, 2048), (4, 8, 8192)):
        d, L2 = 32, 13
        mk = lambda c, n, s: seeded_randn(B, heads * c, n, seed=s).to(dev)
        qkv = seeded_randn(B, 3 * heads * d, L, seed=1).to(dev)
        q, k, v = qkv[:, :heads * d], qkv[:, heads * d:2 * heads * d], qkv[:, 2 * heads * d:]
        pos = seeded_randn(1, heads * d, L, seed=2).to(dev).expand(B, -1, -1)
        k2, v2, k2p = mk(d, L2, 3), mk(d, L2, 4), mk(d, L2, 5)
        scale = (2 * d) ** -0.5
        out = torch.empty((B, heads * d, L), device=dev)
        old = lambda: K.attention_cm(q, k, v, heads, scale, k2=k2, v2=v2, q_pos=pos, k_pos=pos, k2_pos=k2p, out=out)
        u = K.AttnUnits(B, heads, L, L2, d, d, d, dev)
        K.attention_pack_units(u, pos, "k_pos"), K.attention_pack_units(u, k2, "k", 1)
        K.attention_pack_units(u, k2p, "k_pos", 1), K.attention_pack_units(u, v2, "v", 1)
        out2 = torch.empty_like(out)
        pk = lambda: (K.attention_pack_units(u, k, "k"), K.attention_pack_units(u, v, "v"))
        at = lambda: K.attention_units(q, u, heads, scale, q_pos=pos, out=out2)
        old(), pk(), at()
        same = torch.equal(out, out2)
        t_old, t_pk, t_at = timed(old), timed(pk), timed(at)
        print(f"B={B} heads={heads} L={L}+{L2}: f16x2 kernel {t_old:7.1f} us | units: pack k+v {t_pk:6.1f} + attention {t_at:7.1f} "
              f"= {t_pk + t_at:7.1f} us ({(t_pk + t_at) / t_old:.2f}x)  bit-equal {same}", flush=True)


if __name__ == "__main__":
    main()
