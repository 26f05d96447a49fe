"""A few training steps of the C2 (or, with `cond`, the C3) denoiser -- for rocprofv3.
    python devtools/train_run.py [B] [steps] [cond]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from lidarcrafter_amd.testing import seeded_fill, synth_layout_batch  # noqa: E402
from lidargen.utils import inference  # noqa: E402
from lidargen.utils.configs import __all__ as C  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
S = int(sys.argv[2]) if len(sys.argv) > 2 else 4
cond = "cond" in sys.argv
dev = torch.device("cuda:0")
ddpm, model, _ = inference.load_model_duffusion_training(
    C["nuscenes-box-layout-v6" if cond else "nuscenes-unet-uncond"]())
seeded_fill(model, salt=100)
if cond:
    seeded_fill(ddpm.condition_model, salt=201)
ddpm = ddpm.train().to(dev)
opt = torch.optim.AdamW(ddpm.parameters(), lr=1e-4)
x0 = torch.randn(B, 2, 32, 1024, device=dev).clamp(-1, 1)
if cond:
    batch = {k: v.to(dev) for k, v in synth_layout_batch(B, 32, 1024, seed=53).items()}
    batch["x_0"] = x0
else:
    batch = x0
for _ in range(S):
    opt.zero_grad(set_to_none=True)
    loss = ddpm(batch)
    loss.backward()
    opt.step()
torch.cuda.synchronize()
print("ok", float(loss))
