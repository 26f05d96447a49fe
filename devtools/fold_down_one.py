"""One shape of the folded down-sampling stage, 30 launches (for rocprofv3 --pmc): python devtools/fold_down_one.py B Ci Co H W"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lidarcrafter_amd import ops as K  # noqa: E402
from lidarcrafter_amd.testing import seeded_fill, seeded_randn  # noqa: E402
from lidargen.models.unets import ops  # noqa: E402

B, Ci, Co, H, W = (int(v) for v in sys.argv[1:6])
dev = torch.device("cuda:0")
conv = seeded_fill(ops.Conv2d(Ci, Co, 3, 1, 1, ring=True), salt=Ci).to(dev)
x = seeded_randn(B, Ci, H, W, seed=Ci).to(dev)
with torch.no_grad():
    for _ in range(30):
        y = K.conv_down2(x, conv._packed, conv.weight, conv.bias, emit_stats=8)
torch.cuda.synchronize()
print("ok", float(y.abs().mean()))
