"""What do fewer MFMA passes of the f16x2 split cost in accuracy?  ("commit the
measurement that justifies 3 passes".)  Runs the C2 forward (batch 8), the C1 trajectory (10 DDIM
steps, batch 1) and the C2 run (50 DDIM steps, batch 8) against the reference's own outputs
(tests/golden/c2_b8.npz, trajectory.npz) with the product library (3 products: wh*xh + wl*xh + wh*xl)
and the developer builds of devtools/build_variants.sh (1 product; 2 products without wh*xl, i.e.
activations rounded to 11 bits; 2 products without wl*xh, i.e. weights rounded to 11 bits).
    python devtools/passes_error.py > gpurun_out/passes_error.json      (on the MI355X box)
One subprocess per library (LC_HIP_LIB)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CHILD = r'''
import json, sys, os
sys.path.insert(0, %r)
import numpy as np, torch
from lidarcrafter_amd.testing import seeded_fill, seeded_randn
from lidargen.utils import inference
from lidargen.utils.configs import __all__ as C
dev = torch.device("cuda:0")
cfg = C["nuscenes-unet-uncond"]()
ddpm, model, _ = inference.load_model_duffusion_training(cfg)
seeded_fill(model, salt=100)
ddpm = ddpm.eval().to(dev)
g = np.load(os.path.join(%r, "tests/golden/c2_b8.npz"))
gt = np.load(os.path.join(%r, "tests/golden/trajectory.npz"))
T = torch.from_numpy
def rel(a, b):
    a, b = a.double().cpu().flatten(1), b.double().flatten(1)
    return float(((a - b).norm(dim=1) / b.norm(dim=1)).max())
out = {}
x = seeded_randn(8, 2, 32, 1024, seed=81).to(dev)
with torch.no_grad():
    y = ddpm.model(x, T(g["lam"]).to(dev))
out["c2_forward_b8"] = rel(y[..., ::4], T(g["y_s4"]))
rng = [torch.Generator().manual_seed(0)]
xs = ddpm.sample(1, 10, progress=False, rng=rng, return_all=True, mode="ddim")
for k, i in (("c1_x1", 1), ("c1_x2", 2), ("c1_x10", 10)):
    out[k] = rel(xs[i], T(gt[k]))
rng = [torch.Generator().manual_seed(i) for i in range(8)]
xs = ddpm.sample(8, 50, progress=False, rng=rng, return_all=True, mode="ddim")
for i in (1, 25, 50):
    out["c2_x%%d" %% i] = rel(xs[i][..., ::4], T(g["x%%d_s4" %% i]))
print(json.dumps(out))
''' % (ROOT, ROOT, ROOT)

LIBS = {"3 products (product build)": None,
        "2 products: wh*xh + wl*xh (x rounded to 11 bits)": "devtools/variants/liblc_terms3.so",
        "2 products: wh*xh + wh*xl (w rounded to 11 bits)": "devtools/variants/liblc_terms5.so",
        "1 product: wh*xh (plain fp16 operands)": "devtools/variants/liblc_terms1.so"}
res = {"what": "max per-sample rel-L2 vs the reference's CPU run; north-star gate 1e-3 on frames, "
               "forward tests 2e-5",
       "rows": {}}
for name, lib in LIBS.items():
    env = dict(os.environ)
    if lib:
        path = os.path.join(ROOT, lib)
        if not os.path.exists(path):
            res["rows"][name] = "library not built"
            continue
        env["LC_HIP_LIB"] = path
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    try:
        res["rows"][name] = json.loads(r.stdout.strip().splitlines()[-1])
    except Exception:
        res["rows"][name] = {"error": (r.stderr or r.stdout)[-400:]}
print(json.dumps(res, indent=1))
