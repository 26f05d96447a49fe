"""Time the pre-split GroupNorm apply pass (producer entries -> hi / lo planes) on the C2 layer shapes, graph of 20 launches.
    LC_HIP_LIB=<variant> python devtools/gns_time.py [B]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from lidarcrafter_amd import ops as K  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0")
for (C, H, W) in ((128, 16, 512), (256, 8, 256), (512, 4, 128), (128, 32, 1024)):
    x0 = torch.randn(B, C, H, W, device=dev)
    w = torch.randn(C, C, 3, 3, device=dev) / (C * 9) ** 0.5
    pk0, pk = K.PackedConv(), K.PackedConv()
    x = K.conv2d_ring(K.groupnorm(x0, 8, 1e-6, act_silu=True, split_for=pk0), pk0, w, None, emit_stats=True)   # x carries entries
    ss = torch.randn(B, 2 * C, device=dev) * 0.1
    f = lambda: K.groupnorm(x, 8, 1e-6, None, None, ss[:, :C], ss[:, C:], act_silu=True, split_for=pk)
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
    nb = 8.0 * B * C * H * W
    print(f"gns {B}:{C}:{H}:{W}: {min(ts) * 1e6:.1f} us  {nb / min(ts) / 1e12:.2f} TB/s")
