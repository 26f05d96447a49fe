"""Phase totals of the tall level-0 conv (tile cfg 27) from a -DLC_TALL_TIMING=1 build (devtools/variants/liblc_ttim.so):
per-wave s_memtime ticks in prologue / taps / wait in front of the chunk barrier / barrier / park / drain, for the older
(waves 0-3) and younger (4-7) half of the block.
    python devtools/tall_phases.py B:Ci:Co:H:W [--gn] [--res] [--emit]"""
import ctypes as C
import os
import sys

os.environ["LC_HIP_LIB"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "variants", os.environ.get("LC_TIMING_LIB", "liblc_ttim.so"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from lidarcrafter_amd import ops as K  # noqa: E402
from lidarcrafter_amd._lib import lib  # noqa: E402

shape = [a for a in sys.argv[1:] if not a.startswith("--")][0]
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
run = lambda: K.conv2d_ring(x, pk, w, b, out=out, precision="f16x2", emit_stats="--emit" in sys.argv, tile_cfg=27, **kw)
for _ in range(3):
    run()
h = lib()
h.lc_debug_read_tall.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
buf = (C.c_ulonglong * 32)()
h.lc_debug_read_tall(buf, 1)
N = 1 if '--one' in sys.argv else 10
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(N):
    run()
e1.record()
torch.cuda.synchronize()
h.lc_debug_read_tall(buf, 0)
v = list(buf)
print(f"{shape} {' '.join(a for a in sys.argv if a.startswith('--'))}: {e0.elapsed_time(e1) / N * 1e3:.1f} us per launch (instrumented, eager); s_memtime ticks per wave:")
names = ("lifetime", "prologue", "taps (16 chunks)", "wait before barrier", "barrier", "park", "drain")
for half in (0, 1):
    d = v[8 * half: 8 * half + 8]
    n = max(d[7], 1)
    print(f"  waves {4 * half}-{4 * half + 3}: " + ", ".join(f"{nm} {d[i] / n:.0f}" for i, nm in enumerate(names)))
print(f"  lifetime max {v[16]} min {v[17]}; loop (taps + wait + barrier) max {v[18]} min {v[19]}; first start -> last end {v[20] - v[21]} ticks, "
      f"start spread {v[22] - v[21]} (meaningful with --one: a single launch)")
