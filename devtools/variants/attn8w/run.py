"""Build, check and time the 8-wave attention candidate against its own 4-wave instantiation (= the shipped kernel's body)
and the shipped entry point.   python devtools/variants/attn8w/run.py        (on the GPU box)"""
import ctypes as C
import os
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(HERE))))
import torch  # noqa: E402

from lidarcrafter_amd import ops as K  # noqa: E402
from lidarcrafter_amd._lib import CmOperand  # noqa: E402

SO = os.path.join(HERE, "libcand_attn8w.so")


def build():
    src = os.path.join(HERE, "attention_8w.hip")
    if not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(src):
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", src, "-o", SO],
                       check=True)
    lib = C.CDLL(SO)
    vp, i64, i32, f32 = C.c_void_p, C.c_int64, C.c_int, C.c_float
    op = C.POINTER(CmOperand)
    lib.cand_attention_f16x2_fwd.restype = i32
    lib.cand_attention_f16x2_fwd.argtypes = [op] * 8 + [vp, i64, i64, i64] + [i32] * 8 + [f32, i32, vp]
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
    dev = torch.device("cuda:0")
    # (B, heads, d content, d positional, d_v, Lq = Lk0, Lk1): the two attention layers of the layout model at batch 8, the
    # self-attention of the unconditional model, a ragged case
    for (B, heads, dqk, dpos, dv, L, L2) in ((8, 8, 32, 32, 32, 2048, 13), (8, 16, 32, 32, 32, 512, 13), (8, 8, 32, 0, 32, 512, 0),
                                             (2, 4, 32, 32, 32, 300, 13)):
        g = torch.Generator(device="cpu").manual_seed(1)
        rn = lambda *s: torch.randn(*s, generator=g).to(dev)
        q, k, v = rn(B, heads * dqk, L), rn(B, heads * dqk, L), rn(B, heads * dv, L)
        qp = rn(B, heads * dpos, L) if dpos else None
        k2, v2 = (rn(B, heads * dqk, L2), rn(B, heads * dv, L2)) if L2 else (None, None)
        k2p = rn(B, heads * dpos, L2) if (dpos and L2) else None
        scale = float((dqk + dpos) ** -0.5)
        ref = K.attention_cm(q, k, v, heads, scale, k2=k2, v2=v2, q_pos=qp, k_pos=qp, k2_pos=k2p, precision="f16x2")

        def op(t, d):
            return None if t is None else C.byref(CmOperand(t.data_ptr(), t.stride(0), d * t.stride(1), t.stride(1)))

        outs = {}

        def run(waves):
            out = outs.setdefault(waves, torch.empty(B, heads * dv, L, device=dev))
            rc = lib.cand_attention_f16x2_fwd(op(q, dqk), op(qp, dpos), op(k, dqk), op(qp, dpos), op(v, dv), op(k2, dqk),
                                              op(k2p, dpos), op(v2, dv), out.data_ptr(), out.stride(0), dv * out.stride(1),
                                              out.stride(1), B, heads, L, L, L2, dqk, dpos, dv, scale, waves,
                                              torch.cuda.current_stream().cuda_stream)
            assert rc == 0, rc
            return out
        o4, o8 = run(4).clone(), run(8).clone()
        torch.cuda.synchronize()
        same4, same8 = torch.equal(o4, ref), torch.equal(o8, ref)
        t_ship = timed(lambda: K.attention_cm(q, k, v, heads, scale, k2=k2, v2=v2, q_pos=qp, k_pos=qp, k2_pos=k2p,
                                              precision="f16x2"))
        t4, t8 = timed(lambda: run(4)), timed(lambda: run(8))
        print(f"{B}x{heads} heads, d {dqk}+{dpos}/{dv}, {L}+{L2} keys: shipped {t_ship:.1f} us | 4 waves {t4:.1f} us "
              f"(bit-equal {same4}) | 8 waves {t8:.1f} us (bit-equal {same8}, max diff {float((o8 - ref).abs().max()):.1e})")
        assert same8, "the 8-wave block must give the shipped kernel's bits"


if __name__ == "__main__":
    main()
