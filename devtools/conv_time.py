import sys
import torch
sys.path.insert(0, ".")
from lidarcrafter_amd import ops as K
B, Ci, Co, H, W, ks, cfg = [int(v) for v in sys.argv[1:8]]
dev = torch.device("cuda:0")
x = torch.randn(B, Ci, H, W, device=dev); w = torch.randn(Co, Ci, ks, ks, device=dev) / (Ci * ks * ks) ** 0.5
b = torch.randn(Co, device=dev); pk = K.PackedConv(); out = torch.empty(B, Co, H, W, device=dev)
for _ in range(5): K.conv2d_ring(x, pk, w, b, out=out, tile_cfg=cfg, precision="f16x2")
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): K.conv2d_ring(x, pk, w, b, out=out, tile_cfg=cfg, precision="f16x2")
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print(f"  Ci{Ci} Co{Co} {H}x{W} cfg{cfg}: {ms*1e3:.1f} us  {2.0*B*H*W*Co*Ci*ks*ks/ms/1e9:.1f} TF")
