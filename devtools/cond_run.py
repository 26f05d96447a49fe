"""Run a few layout-conditioned DDIM steps (nuscenes-box-layout-v6, C3 shape) -- for rocprofv3."""
import sys
import torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lidarcrafter_amd.testing import seeded_fill, synth_layout_batch
from lidargen.utils import inference
from lidargen.utils.configs import __all__ as C

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
S = int(sys.argv[2]) if len(sys.argv) > 2 else 6
dev = torch.device("cuda:0")
cfg = C["nuscenes-box-layout-v6"]()
ddpm, model, _ = inference.load_model_duffusion_training(cfg)
seeded_fill(model, salt=200), seeded_fill(ddpm.condition_model, salt=201)
ddpm = ddpm.eval().to(dev)
batch = {k: v.to(dev) for k, v in synth_layout_batch(B, 32, 1024, seed=53).items()}
rng = [torch.Generator().manual_seed(i) for i in range(B)]
if os.environ.get("LC_GN_TRACE"):
    import collections
    from lidarcrafter_amd import ops as K
    K.GN_TRACE = collections.Counter()
x = ddpm.sample(batch, B, S, progress=False, rng=rng, mode="ddim")
torch.cuda.synchronize()
print("ok", float(x.abs().mean()))
if os.environ.get("LC_GN_TRACE"):
    for k, v in sorted(K.GN_TRACE.items(), key=lambda kv: (kv[0][2], kv[0][0], kv[0][3])):
        print("gn lookup", k, v)
