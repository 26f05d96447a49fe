"""Forward + backward time of the training attention core: autograd.FlashAttention (HIP) vs torch ops.
    python devtools/attn_train_time.py B:h:dqk:dv:Lq:Lk ..."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from lidarcrafter_amd import autograd as AG  # noqa: E402

dev = torch.device("cuda:0")
for shape in [a for a in sys.argv[1:] if ":" in a]:
    B, h, dqk, dv, Lq, Lk = (int(v) for v in shape.split(":"))
    q = torch.randn(B, h, dqk, Lq, device=dev, requires_grad=True)
    k = torch.randn(B, h, dqk, Lk, device=dev, requires_grad=True)
    v = torch.randn(B, h, dv, Lk, device=dev, requires_grad=True)
    g = torch.randn(B, h, dv, Lq, device=dev)
    for mode in ("hip", "torch"):
        AG.TRAIN_ATTENTION = mode

        def step():
            o = AG.flash_attention(q, k, v, dqk ** -0.5)
            o.backward(g)
            q.grad = k.grad = v.grad = None

        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        flop = 2.0 * B * h * Lq * Lk * (dqk + dv) * (1 + 3.5)
        print(f"{shape} {mode:5s}: {dt * 1e3:8.3f} ms fwd+bwd  ({flop / dt / 1e12:6.1f} TFLOP/s algorithmic)", flush=True)
