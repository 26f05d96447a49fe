"""Developer tool: run ONE conv shape/config repeatedly (for rocprofv3 --pmc passes)."""
import sys
import torch
sys.path.insert(0, ".")
from lidarcrafter_amd import ops as K

B, Ci, Co, H, W, ks, cfg = [int(v) for v in sys.argv[1:8]]
prec = sys.argv[8] if len(sys.argv) > 8 else "f16x2"
n = int(sys.argv[9]) if len(sys.argv) > 9 else 20
dev = torch.device("cuda:0")
x = torch.randn(B, Ci, H, W, device=dev)
w = torch.randn(Co, Ci, ks, ks, device=dev) / (Ci * ks * ks) ** 0.5
b = torch.randn(Co, device=dev)
pk = K.PackedConv()
out = torch.empty(B, Co, H, W, device=dev)
for _ in range(n):
    K.conv2d_ring(x, pk, w, b, out=out, tile_cfg=cfg, precision=prec)
torch.cuda.synchronize()
