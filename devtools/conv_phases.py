"""Phase totals of the fp32-input pipelined conv from a -DLC_TIMING=1 build (devtools/variants/liblc_timing.so):
wave-summed s_memtime cycles in chunk compute / chunk barrier / tile epilogue.
    python devtools/conv_phases.py B:Ci:Co:H:W [--gn] [--res] [--emit]"""
import ctypes as C
import os
import sys

os.environ["LC_HIP_LIB"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "variants", os.environ.get("LC_TIMING_LIB", "liblc_timing.so"))
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
run = lambda: K.conv2d_ring(x, pk, w, b, out=out, precision="f16x2", emit_stats="--emit" in sys.argv, **kw)
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
waves = v[6]
print(f"{shape}: {e0.elapsed_time(e1) / N * 1e3:.1f} us per launch (instrumented); per wave (s_memtime ticks, 100 MHz?):")
for name, i in (("lifetime", 0), ("prologue", 1), ("chunk compute", 2), ("chunk barrier", 3), ("epilogue", 4),
                ("  epi: MFMA drain", 7), ("  epi: residual wait", 8), ("  epi: math + store issue", 9),
                ("  epi: store ack (instr.)", 10)):
    print(f"  {name:28s} {v[i] / waves:10.0f}")
print(f"  chunks/wave {v[5] / waves:.1f}; compute per chunk {v[2] / max(v[5], 1):.0f}, barrier per chunk {v[3] / max(v[5], 1):.0f}")
