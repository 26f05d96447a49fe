"""Time the pre-split 3x3 conv (GroupNorm-apply output as hi / lo planes -> LDS-DMA kernel) on the C2 layer shapes.
    LC_HIP_LIB=<variant> python devtools/ps_time.py [B] [--cfg N] [--emit 0|8|4] [--shape 0..3]   (statistics entries: none / octets / quads)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from lidarcrafter_amd import ops as K  # noqa: E402

cfg = int(sys.argv[sys.argv.index("--cfg") + 1]) if "--cfg" in sys.argv else 0
emit = int(sys.argv[sys.argv.index("--emit") + 1]) if "--emit" in sys.argv else 8
only = int(sys.argv[sys.argv.index("--shape") + 1]) if "--shape" in sys.argv else -1      # one of the four shapes (PMC runs)
pos = [a for i, a in enumerate(sys.argv[1:], 1) if not a.startswith("--") and sys.argv[i - 1] not in ("--cfg", "--emit", "--shape")]
B = int(pos[0]) if pos else 8
dev = torch.device("cuda:0")
for si, (Ci, Co, H, W) in enumerate(((128, 128, 16, 512), (256, 256, 8, 256), (512, 512, 4, 128), (256, 128, 16, 512))):
    if only >= 0 and si != only:
        continue
    x = torch.randn(B, Ci, H, W, device=dev)
    w = torch.randn(Co, Ci, 3, 3, device=dev) / (Ci * 9) ** 0.5
    b = torch.randn(Co, device=dev)
    res = torch.randn(B, Co, H, W, device=dev)
    pk = K.PackedConv()
    out = torch.empty(B, Co, H, W, device=dev)
    sa = K.groupnorm(x, 8, 1e-6, act_silu=True, split_for=pk)
    assert isinstance(sa, K.SplitAct)
    f = lambda: K.conv2d_ring(sa, pk, w, b, out=out, res=res, out_scale=0.7071, emit_stats={0: False, 8: True}.get(emit, emit), tile_cfg=cfg)
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
    fl = 2.0 * B * H * W * Ci * Co * 9
    print(f"ps {B}:{Ci}:{Co}:{H}:{W} cfg {cfg} emit {emit}: {min(ts) * 1e6:.1f} us  {fl / min(ts) / 1e12:.0f} TF")
