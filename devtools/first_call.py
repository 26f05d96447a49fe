"""Where the first sampling step of a process spends its time (one-time costs: code-object load, weight packing,
allocator growth, rocBLAS initialisation) -- python devtools/first_call.py [cond] [prepare]"""
import os
import sys
import time

T0 = time.perf_counter()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

cond = "cond" in sys.argv
prepare = "prepare" in sys.argv
out = {}


def lap(name, t):
    torch.cuda.synchronize()
    out[name] = round((time.perf_counter() - t) * 1e3, 1)
    return time.perf_counter()


t = time.perf_counter()
dev = torch.device("cuda:0")
torch.zeros(1, device=dev)
t = lap("torch_context_ms", t)
from lidarcrafter_amd import ops as K  # noqa: E402
from lidarcrafter_amd._lib import lib  # noqa: E402
from lidarcrafter_amd.testing import seeded_fill, synth_layout_batch  # noqa: E402
from lidargen.utils import inference  # noqa: E402
from lidargen.utils.configs import __all__ as C  # noqa: E402

lib()
t = lap("library_load_ms", t)
B = 8
ddpm, model, _ = inference.load_model_duffusion_training(C["nuscenes-box-layout-v6" if cond else "nuscenes-unet-uncond"]())
seeded_fill(model, salt=100)
if cond:
    seeded_fill(ddpm.condition_model, salt=201)
ddpm = ddpm.eval().to(dev)
t = lap("model_build_and_copy_ms", t)
if prepare:
    K.prepare_model(ddpm)
    t = lap("prepare_model_ms", t)
with torch.inference_mode():
    x_T = torch.randn(B, *ddpm.sampling_shape, device=dev)
    cdict = None
    if cond:
        batch = {k: v.to(dev) for k, v in synth_layout_batch(B, 32, 1024, seed=53).items()}
        t = time.perf_counter()
        for i in range(3):
            cdict = ddpm.get_network_condition(input_dict=batch, only_custom_condition=True)
            t = lap(f"layout_encoder_call{i + 1}_ms", t)
    t = time.perf_counter()
    st = ddpm.begin_sampling(B, 16, None, "ddim", 0.0, x_T=x_T, condition_dict=cdict)
    t = lap("begin_sampling_ms", t)
    for i in range(4):
        ddpm.sampling_step(st)
        t = lap(f"step{i + 1}_ms", t)
out["process_total_ms"] = round((time.perf_counter() - T0) * 1e3, 1)
print(("cond " if cond else "uncond ") + ("prepare " if prepare else "") + str(out))
