"""Build, check and time the pre-split 1x1 conv candidate against the shipped 1x1 conv (GroupNorm -> projection, as the
qkv projections of the layout model run it).   python devtools/variants/ps1x1/run.py [B]        (on the GPU box)"""
import ctypes as C
import os
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(HERE))))
import torch  # noqa: E402

from lidarcrafter_amd import ops as K  # noqa: E402
from lidarcrafter_amd.testing import rel_l2  # noqa: E402

SO = os.path.join(HERE, "libcand_ps1x1.so")


def build():
    src = os.path.join(HERE, "conv1x1_ps.hip")
    if not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(src):
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", src, "-o", SO],
                       check=True)
    lib = C.CDLL(SO)
    vp, i64, i32, f32 = C.c_void_p, C.c_int64, C.c_int, C.c_float
    lib.cand_conv1x1_ps_fwd.restype = i32
    lib.cand_conv1x1_ps_fwd.argtypes = [vp, vp, vp, vp, vp, i64, vp, i64, i32, i32, i32, i32, f32, vp, vp, vp]
    return lib


def timed(f, n=20):
    for _ in range(3):
        f()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            f()
    g.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        g.replay()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / n)
    return min(ts) * 1e6


def main():
    if "--build-only" in sys.argv:
        build()
        return
    lib = build()
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    dev = torch.device("cuda:0")
    for (Ci, Co, H, W, with_res) in ((256, 768, 8, 256, False), (512, 1536, 4, 128, False), (256, 256, 8, 256, True),
                                     (512, 256, 8, 256, False), (128, 384, 16, 512, False), (64, 96, 5, 50, True)):
        x = torch.randn(B, Ci, H, W, device=dev) * 1.7
        w = torch.randn(Co, Ci, 1, 1, device=dev) / Ci ** 0.5
        b = torch.randn(Co, device=dev)
        res = torch.randn(B, Co, H, W, device=dev) if with_res else None
        gam, bet = torch.rand(Ci, device=dev) + 0.5, torch.randn(Ci, device=dev) * 0.1
        # shipped route: GroupNorm (fp32 out) -> 1x1 conv that splits on the fly
        pk0 = K.PackedConv()
        out0 = torch.empty(B, Co, H, W, device=dev)
        f0 = lambda: K.conv2d_ring(K.groupnorm(x, 32, 1e-5, gam, bet), pk0, w, b, res=res, out=out0, out_scale=0.5)
        # candidate: GroupNorm writes the pre-split planes for `pk`, the candidate kernel consumes them
        pk = K.PackedConv()
        wh, wl = pk.get_f16x2(w)
        out1 = torch.empty(B, Co, H, W, device=dev)
        def f1():
            sa = K.groupnorm(x, 32, 1e-5, gam, bet, split_for=pk)
            assert isinstance(sa, K.SplitAct), "the GroupNorm did not take the pre-split route for this shape"
            rc = lib.cand_conv1x1_ps_fwd(sa.buf.data_ptr(), wh.data_ptr(), wl.data_ptr(), b.data_ptr(),
                                         None if res is None else res.data_ptr(), Co * H * W, out1.data_ptr(), Co * H * W,
                                         B, Ci, Co, H * W, 0.5, pk.wmeta.data_ptr(), pk.range_ptr(dev),
                                         torch.cuda.current_stream().cuda_stream)      # (the capture stream inside a graph)
            assert rc == 0, rc
        f0(), f1()
        torch.cuda.synchronize()
        bad = K.range_poll(dev)
        if bad:                       # the default x_scale did not fit this input: the poll adjusted it, run again
            f0(), f1()
            torch.cuda.synchronize()
        ref = torch.nn.functional.conv2d(torch.nn.functional.group_norm(x.double(), 32, gam.double(), bet.double(), 1e-5),
                                         w.double(), b.double())
        if res is not None:
            ref = ref + res.double()
        ref = ref * 0.5
        e0, e1 = rel_l2(out0, ref), rel_l2(out1, ref)
        t0, t1 = timed(f0), timed(f1)
        # the GroupNorm alone, both forms, to separate the conv times
        g0 = timed(lambda: K.groupnorm(x, 32, 1e-5, gam, bet))
        g1 = timed(lambda: K.groupnorm(x, 32, 1e-5, gam, bet, split_for=pk))
        print(f"{B}:{Ci}:{Co}:{H}:{W} res={int(with_res)}: shipped {t0:.1f} us (norm {g0:.1f}) err {e0:.2e} | "
              f"candidate {t1:.1f} us (norm+split {g1:.1f}) err {e1:.2e} | conv alone {t0 - g0:.1f} -> {t1 - g1:.1f} us")
        assert e1 < 2e-6, "candidate result off"


if __name__ == "__main__":
    main()
