"""Per-layer timing of the conv kernels under HIP-graph replay (no host launch overhead):
    python devtools/conv_bench.py [--lib path.so] [--cfg N] [--emit] B:Ci:Co:H:W:ks ...
Prints one line per shape: microseconds per launch (median of 5 replays of a 20-launch graph) and
algorithmic TFLOP/s.  `uncond8` / `uncond1` expand to the distinct conv shapes of the C2 denoiser."""
import os
import sys

args = sys.argv[1:]
lib = cfg = None
emit = False
ps = False
gn = res = False
shapes = []
i = 0
while i < len(args):
    if args[i] == "--lib":
        lib = args[i + 1]; i += 2
    elif args[i] == "--cfg":
        cfg = int(args[i + 1]); i += 2
    elif args[i] == "--emit":
        emit = True; i += 1
    elif args[i] == "--ps":
        ps = True; i += 1
    elif args[i] == "--gn":          # GroupNorm + SiLU fused into the conv's staging
        gn = True; i += 1
    elif args[i] == "--res":         # residual operand in the epilogue
        res = True; i += 1
    else:
        shapes.append(args[i]); i += 1
if lib:
    os.environ["LC_HIP_LIB"] = os.path.abspath(lib)
sys.path.insert(0, os.environ.get("LC_TREE") or os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from lidarcrafter_amd import ops as K  # noqa: E402


def unet_shapes(B):
    s = []
    for (c, h, w) in ((64, 32, 1024), (128, 16, 512), (256, 8, 256), (512, 4, 128)):
        s.append((B, c, c, h, w, 3))
    s += [(B, 32, 64, 32, 1024, 3), (B, 64, 128, 32, 1024, 3), (B, 128, 256, 16, 512, 3),
          (B, 256, 512, 8, 256, 3), (B, 512, 256, 4, 128, 3), (B, 512, 128, 8, 256, 3),
          (B, 256, 64, 16, 512, 3), (B, 128, 64, 32, 1024, 3), (B, 64, 2, 32, 1024, 3),
          (B, 512, 256, 4, 128, 1), (B, 512, 128, 8, 256, 1), (B, 256, 64, 16, 512, 1),
          (B, 128, 64, 32, 1024, 1), (B, 512, 1536, 4, 128, 1), (B, 256, 768, 4, 128, 1)]
    return s


todo = []
for sh in shapes:
    if sh.startswith("uncond"):
        todo += unet_shapes(int(sh[6:]))
    else:
        todo.append(tuple(int(v) for v in sh.split(":")))
dev = torch.device("cuda:0")
for (B, Ci, Co, H, W, ks) in todo:
    x = torch.randn(B, Ci, H, W, device=dev)
    w = torch.randn(Co, Ci, ks, ks, device=dev) / (Ci * ks * ks) ** 0.5
    b = torch.randn(Co, device=dev)
    pk = K.PackedConv()
    out = torch.empty(B, Co, H, W, device=dev)
    xin = x
    if ps and ks == 3 and K.can_presplit(Ci, 8):
        xin = K.groupnorm(x, 8, 1e-6, act_silu=True, split_for=pk)      # pre-split once, outside the graph
    kw = {}
    if gn and xin is x:
        kw["gn_coeffs"] = K.groupnorm_stats(x, 8, 1e-6)
    if res:
        kw["res"] = torch.randn(B, Co, H, W, device=dev)
    run = lambda: K.conv2d_ring(xin, pk, w, b, out=out, tile_cfg=cfg or 0, precision="f16x2",
                                emit_stats=emit, **kw)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20):
            run()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 20)
    ms = sorted(ts)[2]
    print(f"B{B} {Ci:4d}->{Co:4d} {H:2d}x{W:4d} k{ks} cfg{cfg or 0}{' ps' if xin is not x else ''}{' gn' if 'gn_coeffs' in kw else ''}{' res' if res else ''}{' emit' if emit else ''}: {ms*1e3:7.1f} us "
          f"{2.0*B*H*W*Co*Ci*ks*ks/ms/1e9:7.1f} TF", flush=True)
