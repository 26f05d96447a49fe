"""Isolated cost of the fused-GroupNorm / statistics options of one conv shape (developer tool):
    python devtools/gnconv_time.py B Ci Co H W [cfg]
plain | emit | gn from partials (stats launch timed separately) | gn from producer stats | both."""
import sys

import torch

sys.path.insert(0, ".")
from lidarcrafter_amd import ops as K

B, Ci, Co, H, W = [int(v) for v in sys.argv[1:6]]
cfg = int(sys.argv[6]) if len(sys.argv) > 6 else 0
dev = torch.device("cuda:0")
G = 8
xsrc = torch.randn(B, 32, H, W, device=dev)
wp = torch.randn(Ci, 32, 3, 3, device=dev) / 17.0
x = K.conv2d_ring(xsrc, K.PackedConv(), wp, None, emit_stats=True)     # carries producer stats
xc = x.clone()                                                         # carries none
w = torch.randn(Co, Ci, 3, 3, device=dev) / (Ci * 9) ** 0.5
b = torch.randn(Co, device=dev)
pk = K.PackedConv()
out = torch.empty(B, Co, H, W, device=dev)
ga, be = torch.ones(Ci, device=dev), torch.zeros(Ci, device=dev)


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


st_p = K.groupnorm_stats(xc, G, 1e-6, ga, be)
st_o = K.groupnorm_stats(x, G, 1e-6, ga, be)
assert st_p._struct.partials and not st_o._struct.partials
rows = {
    "plain": lambda: K.conv2d_ring(xc, pk, w, b, out=out, tile_cfg=cfg),
    "emit": lambda: K.conv2d_ring(xc, pk, w, b, out=out, tile_cfg=cfg, emit_stats=True),
    "gn_partials": lambda: K.conv2d_ring(xc, pk, w, b, out=out, tile_cfg=cfg, gn_coeffs=st_p),
    "gn_ostats": lambda: K.conv2d_ring(x, pk, w, b, out=out, tile_cfg=cfg, gn_coeffs=st_o),
    "gn_ostats+emit": lambda: K.conv2d_ring(x, pk, w, b, out=out, tile_cfg=cfg, gn_coeffs=st_o,
                                            emit_stats=True),
    "stats_launch": lambda: K.groupnorm_stats(xc, G, 1e-6, ga, be),
}
print(f"B{B} Ci{Ci} Co{Co} {H}x{W} cfg{cfg}: " + "  ".join(f"{k} {timed(f):.1f}us" for k, f in rows.items()))
