"""One attention shape, f16x2 kernel, 20 launches -- the driver of a PMC pass (devtools/pmc_conv.sh with
PMC_SCRIPT=devtools/attn_one.py PMC_KERNEL=attn_h_kernel).  Default: the 2048 + 13 key layer of the layout model."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from lidarcrafter_amd import ops as K  # noqa: E402

B, heads, dqk, dv, L, L2 = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "8:8:64:32:2048:13").split(":"))
dev = torch.device("cuda:0")
q = torch.randn(B, heads * dqk, L, device=dev)
k = torch.randn(B, heads * dqk, L, device=dev)
v = torch.randn(B, heads * dv, L, device=dev)
k2 = torch.randn(B, heads * dqk, L2, device=dev) if L2 else None
v2 = torch.randn(B, heads * dv, L2, device=dev) if L2 else None
for _ in range(20):
    o = K.attention_cm(q, k, v, heads, dqk ** -0.5, k2=k2, v2=v2, precision="f16x2")
torch.cuda.synchronize()
print("ok", float(o.abs().mean()))
