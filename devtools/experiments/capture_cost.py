import sys, time, torch
sys.path.insert(0, ".")
from lidarcrafter_amd.testing import seeded_fill
from lidargen.utils import inference
from lidargen.utils.configs import __all__ as C
dev = torch.device("cuda:0")
ddpm, model, _ = inference.load_model_duffusion_training(C["nuscenes-unet-uncond"]())
seeded_fill(model, salt=100); ddpm = ddpm.eval().to(dev)
for B in (2, 8):
    ddpm.sample(B, 4, progress=False, mode="ddim")
    for S in (2, 3, 8, 32):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ddpm.sample(B, S, progress=False, mode="ddim")
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"B={B} S={S}: {dt*1e3:.1f} ms total, {dt/S*1e3:.2f} ms/step")
