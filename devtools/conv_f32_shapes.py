"""fp32 conv smoke over the shapes of the layout-conditioned training graph (forward and dX)."""
import sys, torch
sys.path.insert(0, ".")
from lidarcrafter_amd import ops as K
dev = torch.device("cuda:0")
shapes = [(2, 2, 256, 32, 1024, 3), (2, 256, 2, 32, 1024, 3), (2, 42, 256, 32, 1024, 3), (2, 256, 42, 32, 1024, 3),
          (2, 256, 256, 32, 1024, 3), (2, 512, 256, 32, 1024, 3), (2, 256, 512, 32, 1024, 3),
          (2, 512, 256, 32, 1024, 1), (2, 256, 512, 32, 1024, 1), (2, 768, 512, 8, 256, 3), (2, 512, 768, 8, 256, 3),
          (2, 1024, 512, 8, 256, 3), (2, 512, 1024, 8, 256, 3), (2, 512, 1536, 1, 2048, 1), (2, 1536, 512, 1, 2048, 1),
          (2, 512, 512, 1, 2048, 1), (2, 256, 512, 1, 2048, 1), (2, 512, 256, 1, 2048, 1),
          (2, 64, 192, 32, 1024, 1), (2, 64, 192, 32, 1024, 3), (1, 8, 320, 4, 64, 3)]
for (B, Ci, Co, H, W, ks) in shapes:
    x = torch.randn(B, Ci, H, W, device=dev)
    w = torch.randn(Co, Ci, ks, ks, device=dev) / (Ci * ks * ks) ** 0.5
    print((B, Ci, Co, H, W, ks), end=" ", flush=True)
    y = K.conv2d_ring(x, K.PackedConv(), w, None, precision="f32")
    torch.cuda.synchronize()
    if ks == 1:
        ref = torch.einsum("oc,bchw->bohw", w[:, :, 0, 0].double(), x.double())
        print("rel", float((y.double() - ref).norm() / ref.norm()), flush=True)
    else:
        print("ok", float(y.abs().mean()), flush=True)
