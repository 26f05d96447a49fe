"""Developer micro-benchmark: time lc_conv2d_ring_fwd tile configurations on the layer shapes of
EfficientUNet at a given batch (not part of the product; run via gpurun)."""
import sys
import torch
sys.path.insert(0, ".")
from lidarcrafter_amd import ops as K

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
PREC = sys.argv[2] if len(sys.argv) > 2 else "f32"
CFGS = (1, 2, 3, 4, 5) if PREC == "f32" else (0, 2, 5, 13, 23, 25, 223, 225)
dev = torch.device("cuda:0")
shapes = [  # Ci, Co, H, W, ks  (EfficientUNet nuscenes-unet-uncond layer shapes)
    (32, 64, 32, 1024, 3), (64, 64, 32, 1024, 3), (64, 128, 32, 1024, 3), (128, 64, 32, 1024, 3),
    (128, 128, 16, 512, 3), (256, 128, 16, 512, 3), (128, 256, 16, 512, 3),
    (256, 256, 8, 256, 3), (512, 256, 8, 256, 3), (256, 512, 8, 256, 3),
    (512, 512, 4, 128, 3), (512, 256, 4, 128, 3), (256, 256, 4, 128, 3), (64, 2, 32, 1024, 3),
]
for (Ci, Co, H, W, ks) in shapes:
    x = torch.randn(B, Ci, H, W, device=dev)
    w = torch.randn(Co, Ci, ks, ks, device=dev) / (Ci * ks * ks) ** 0.5
    b = torch.randn(Co, device=dev)
    pk = K.PackedConv()
    out = torch.empty(B, Co, H, W, device=dev)
    fl = 2.0 * B * H * W * Co * Ci * ks * ks
    line = f"Ci{Ci:4d} Co{Co:4d} {H:2d}x{W:4d} k{ks}: "
    for cfg in CFGS:
        try:
            for _ in range(3):
                K.conv2d_ring(x, pk, w, b, out=out, tile_cfg=cfg, precision=PREC)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                K.conv2d_ring(x, pk, w, b, out=out, tile_cfg=cfg, precision=PREC)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            line += f" c{cfg}:{ms*1e3:7.0f}us {fl/ms/1e9:5.1f}TF |"
        except Exception as ex:
            line += f" c{cfg}: ERR |"
    print(line, flush=True)
