import sys, torch
sys.path.insert(0, ".")
from lidarcrafter_amd import ops as K
dev = torch.device("cuda:0")
for (B, heads, dqk, dv, L, L2) in [(8, 8, 64, 32, 2048, 13), (8, 16, 64, 32, 512, 13), (8, 8, 64, 64, 512, 0), (8, 8, 32, 32, 2048, 0)]:
    q = torch.randn(B, heads*dqk, L, device=dev); k = torch.randn(B, heads*dqk, L, device=dev); v = torch.randn(B, heads*dv, L, device=dev)
    k2 = torch.randn(B, heads*dqk, L2, device=dev) if L2 else None; v2 = torch.randn(B, heads*dv, L2, device=dev) if L2 else None
    for prec in ("f32", "f16x2"):
        f = lambda: K.attention_cm(q, k, v, heads, dqk**-0.5, k2=k2, v2=v2, precision=prec)
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): o = f()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)/10
        fl = 2.0*B*heads*L*(L+L2)*(dqk+dv)
        print(f"B{B} h{heads} dqk{dqk} dv{dv} L{L}+{L2} {prec}: {ms*1e3:.1f} us {fl/ms/1e9:.1f} TF", flush=True)
    a = K.attention_cm(q, k, v, heads, dqk**-0.5, k2=k2, v2=v2, precision="f32"); bb = K.attention_cm(q, k, v, heads, dqk**-0.5, k2=k2, v2=v2, precision="f16x2")
    print("  rel diff", float((a-bb).norm()/a.norm()))
