"""Time one conv shape (graph of 20 launches) with the library named by LC_HIP_LIB.
    python devtools/conv_time.py B:Ci:Co:H:W[:ks] ... [--gn] [--res] [--emit] [--cfg N]
    python devtools/conv_time.py B Ci Co H W ks cfg [--gn] [--res] [--emit]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from lidarcrafter_amd import ops as K  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
cfg = int(sys.argv[sys.argv.index("--cfg") + 1]) if "--cfg" in sys.argv else 0
if "--cfg" in sys.argv:
    args = [a for a in args if a != str(cfg)]
if len(args) >= 7 and ":" not in args[0]:        # older form (devtools/pmc_conv.sh): B Ci Co H W ks cfg
    cfg = int(args[6])
    args = [":".join(args[:6])]
dev = torch.device("cuda:0")
for shape in args:
    v = [int(t) for t in shape.split(":")]
    B, Ci, Co, H, W = v[:5]
    ks = v[5] if len(v) > 5 else 3
    x = torch.randn(B, Ci, H, W, device=dev)
    w = torch.randn(Co, Ci, ks, ks, device=dev) / (Ci * ks * ks) ** 0.5
    if "--zeros" in sys.argv:      # same instruction stream on zero operands: what the power limit costs (DVFS)
        x, w = x * 0 + 1e-30, w * 0
    b = torch.randn(Co, device=dev)
    pk = K.PackedConv()
    out = torch.empty(B, Co, H, W, device=dev)
    kw = {}
    if "--gn" in sys.argv:
        kw["gn_coeffs"] = K.groupnorm_stats(x, 8, 1e-6)
    if "--res" in sys.argv:
        kw["res"] = torch.randn(B, Co, H, W, device=dev)
    f = lambda: K.conv2d_ring(x, pk, w, b, out=out, precision="f16x2", emit_stats="--emit" in sys.argv,
                              tile_cfg=cfg, **kw)
    for _ in range(3):
        f()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20):
            f()
    g.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        g.replay()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / 20)
    print(f"{shape} cfg {cfg} {' '.join(a for a in sys.argv if a.startswith('--') and a != '--cfg')}: {min(ts) * 1e6:.1f} us")
