"""One-off soak: the full 1024-step DDPM schedule of config C1/C2 through the graph-replayed sampler --
finite, clipped, no range event, and sample 3 of a batch-8 run equals the same seed run alone."""
import sys, time, torch
sys.path.insert(0, ".")
from lidarcrafter_amd import ops as K
from lidarcrafter_amd.testing import seeded_fill
from lidargen.utils import inference
from lidargen.utils.configs import __all__ as C

dev = torch.device("cuda:0")
ddpm, model, _ = inference.load_model_duffusion_training(C["nuscenes-unet-uncond"]())
seeded_fill(model, salt=100)
ddpm = ddpm.eval().to(dev)
for mode, steps in (("ddpm", 1024), ("ddim", 256)):
    rng = [torch.Generator().manual_seed(i) for i in range(8)]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    x8 = ddpm.sample(8, steps, progress=False, rng=rng, mode=mode)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    x1 = ddpm.sample(1, steps, progress=False, rng=[torch.Generator().manual_seed(3)], mode=mode)
    d = float((x8[3:4] - x1).norm() / x1.norm())
    print(mode, steps, "steps:", f"{dt:.2f} s ({dt/steps*1e3:.2f} ms/step)", "finite", bool(torch.isfinite(x8).all()),
          "range", float(x8.min()), float(x8.max()), "batch-vs-single rel-L2", f"{d:.2e}", "range events", K.range_poll(dev))
