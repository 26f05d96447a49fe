"""Time the weight-gradient kernels (exact fp32 vs f16x2 split) on the C2 layer shapes at batch 8."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from lidarcrafter_amd import ops as K  # noqa: E402
from lidarcrafter_amd._lib import check, lib  # noqa: E402

dev = torch.device("cuda:0")
for (B, Ci, Co, H, W, ks) in ((8, 64, 64, 32, 1024, 3), (8, 128, 128, 16, 512, 3), (8, 256, 256, 8, 256, 3),
                              (8, 512, 512, 4, 128, 3), (8, 128, 64, 32, 1024, 3), (8, 512, 512, 4, 128, 1)):
    x = torch.randn(B, Ci, H, W, device=dev)
    dy = torch.randn(B, Co, H, W, device=dev) * 1e-3
    rx, rdy = K.PackedConv("t.x"), K.PackedConv("t.dy")
    K.range_from_tensor(x, rx)
    K.range_from_tensor(dy, rdy)
    n = int(lib().lc_conv2d_ring_wgrad_scratch_elems(B, Ci, Co, H, W, ks))
    scratch = torch.empty(n, device=dev)
    dw, db = torch.empty(Co, Ci, ks, ks, device=dev), torch.empty(Co, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    fl = 2.0 * B * H * W * Ci * Co * ks * ks
    res = []
    for split in (False, True):
        def f():
            if split:
                check(lib().lc_conv2d_ring_wgrad_f16x2(x.data_ptr(), Ci * H * W, dy.data_ptr(), Co * H * W,
                                                       rx.range_ptr(dev), rdy.range_ptr(dev), scratch.data_ptr(),
                                                       dw.data_ptr(), None, B, Ci, Co, H, W, ks, 0, st), "h")
            else:
                check(lib().lc_conv2d_ring_wgrad(x.data_ptr(), Ci * H * W, dy.data_ptr(), Co * H * W,
                                                 scratch.data_ptr(), dw.data_ptr(), None, B, Ci, Co, H, W, ks, 0,
                                                 st), "f")
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            f()
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t0) / 20)
    print(f"{B}:{Ci}:{Co}:{H}:{W}:{ks}  fp32 {res[0] * 1e6:7.1f} us ({fl / res[0] / 1e12:5.1f} TF)   "
          f"f16x2 {res[1] * 1e6:7.1f} us ({fl / res[1] / 1e12:5.1f} TF)   (kernel + fold)")
