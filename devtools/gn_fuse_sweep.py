"""Developer tool: per-layer cost of fusing GN+SiLU into the conv staging vs the separate apply."""
import sys
import torch
sys.path.insert(0, ".")
from lidarcrafter_amd import ops as K

B = 8
dev = torch.device("cuda:0")
shapes = [(64, 64, 32, 1024, 8), (128, 64, 32, 1024, 8), (128, 128, 16, 512, 8), (256, 64, 16, 512, 8),
          (256, 256, 8, 256, 8), (512, 128, 8, 256, 8), (512, 512, 4, 128, 8), (512, 256, 4, 128, 8),
          (1024, 512, 4, 128, 32), (256, 128, 16, 512, 32)]

def timeit(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for (Ci, Co, H, W, G) in shapes:
    x = torch.randn(B, Ci, H, W, device=dev)
    w = torch.randn(Co, Ci, 3, 3, device=dev) / (Ci * 9) ** 0.5
    b = torch.randn(Co, device=dev)
    ga, be = torch.randn(Ci, device=dev), torch.randn(Ci, device=dev)
    pk = K.PackedConv()
    out = torch.empty(B, Co, H, W, device=dev)
    tmp = torch.empty_like(x)
    def unfused():
        K.groupnorm(x, G, 1e-6, ga, be, act_silu=True, out=tmp)
        K.conv2d_ring(tmp, pk, w, b, out=out)
    def fused():
        co = K.groupnorm_coeffs(x, G, 1e-6, ga, be)
        K.conv2d_ring(x, pk, w, b, out=out, gn_coeffs=co)
    def conv_only():
        K.conv2d_ring(tmp, pk, w, b, out=out)
    print(f"Ci{Ci:5d} Co{Co:4d} {H:2d}x{W:4d}: gn+conv {timeit(unfused):7.1f}us  fused {timeit(fused):7.1f}us  conv-only {timeit(conv_only):7.1f}us", flush=True)
