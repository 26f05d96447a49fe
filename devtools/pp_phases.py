"""Phase totals of the ping-pong conv (cfg 33) from a -DLC_TIMING=1 build (devtools/variants/liblc_timing.so): per wave
group, s_memtime ticks a wave spends computing, waiting behind its compute phase (vmcnt + barrier), in its stage slot,
waiting behind the stage slot, in the prologue and in the final drain.
    python devtools/pp_phases.py B:Ci:Co:H:W [--gn] [--res] [--emit] [--cfg 33]"""
import ctypes as C
import os
import sys

os.environ["LC_HIP_LIB"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "variants", "liblc_timing.so")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from lidarcrafter_amd import ops as K  # noqa: E402
from lidarcrafter_amd._lib import lib  # noqa: E402

cfg = int(sys.argv[sys.argv.index("--cfg") + 1]) if "--cfg" in sys.argv else 33
for shape in [a for a in sys.argv[1:] if ":" in a]:
    B, Ci, Co, H, W = (int(v) for v in shape.split(":"))
    dev = torch.device("cuda:0")
    x = torch.randn(B, Ci, H, W, device=dev)
    w = torch.randn(Co, Ci, 3, 3, device=dev) / (Ci * 9) ** 0.5
    b = torch.randn(Co, device=dev)
    pk = K.PackedConv()
    out = torch.empty(B, Co, H, W, device=dev)
    kw = {}
    if "--gn" in sys.argv:
        kw["gn_coeffs"] = K.groupnorm_stats(x, 8, 1e-6)
    if "--res" in sys.argv:
        kw["res"] = torch.randn(B, Co, H, W, device=dev)
    run = lambda: K.conv2d_ring(x, pk, w, b, out=out, precision="f16x2", emit_stats="--emit" in sys.argv, tile_cfg=cfg, **kw)
    for _ in range(3):
        run()
    h = lib()
    h.lc_debug_read.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
    buf = (C.c_ulonglong * 16)()
    h.lc_debug_read(buf, 1)
    N = 10
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(N):
        run()
    e1.record()
    torch.cuda.synchronize()
    h.lc_debug_read(buf, 0)
    v = list(buf)
    us = e0.elapsed_time(e1) / N * 1e3
    print(f"{shape} {' '.join(a for a in sys.argv if a.startswith('--'))}: {us:.1f} us per launch (instrumented)")
    for g in (0, 1):
        d = v[8 * g:8 * g + 8]
        n = max(d[6], 1)
        print("  group %d per wave: lifetime %7.0f | prologue %6.0f  compute %6.0f  wait behind compute %6.0f  stage %6.0f  "
              "wait behind stage %6.0f  drain %6.0f   (ticks; lifetime = %.2f ticks/ns of the launch)" % (
                  g, d[0] / n, d[1] / n, d[2] / n, d[3] / n, d[4] / n, d[5] / n, d[7] / n, d[0] / n / (us * 1e3)))
