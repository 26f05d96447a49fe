"""Wall time of whole `ddpm.sample()` calls (what tools/generate and the evaluation harness pay per batch) against steps x the replayed
step: the difference is the per-call setup -- x_T draw, schedule tables, time features, first step eager, HIP-graph capture.
python devtools/sample_call_time.py [steps]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lidarcrafter_amd.testing import seeded_fill, synth_layout_batch  # noqa: E402
from lidargen.utils import inference  # noqa: E402
from lidargen.utils.configs import __all__ as CONFIGS  # noqa: E402


def main():
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    dev = torch.device("cuda:0")
    for name, B in (("nuscenes-unet-uncond", 8), ("nuscenes-unet-uncond", 1), ("nuscenes-box-layout-v6", 8)):
        ddpm, model, _ = inference.load_model_duffusion_training(CONFIGS[name]())
        seeded_fill(model, salt=100)
        if getattr(ddpm, "condition_model", None) is not None and not isinstance(ddpm.condition_model, torch.nn.Identity):
            seeded_fill(ddpm.condition_model, salt=101)
        ddpm = ddpm.eval().to(dev)
        cond = "layout" in name
        batch = {k: v.to(dev) for k, v in synth_layout_batch(B, 32, 1024, seed=83).items()} if cond else None
        for cache in (0, 4):                     # 0: every call pays an eager first step + a capture (rounds 1-5)
            ddpm.graph_cache_size = cache
            ts, outs = [], []
            for call in range(4):
                rng = [torch.Generator().manual_seed(i) for i in range(B)]
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                x = ddpm.sample(batch, B, S, progress=False, rng=rng, mode="ddim") if cond else \
                    ddpm.sample(B, S, progress=False, rng=rng, mode="ddim")
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) * 1e3)
                outs.append(x.clone())
            same = all(torch.equal(outs[0], o) for o in outs[1:])
            print(f"{name} B={B} {S} DDIM steps, graph cache {cache}: sample() calls {', '.join(f'{t:.1f}' for t in ts)} ms; "
                  f"repeat calls bit-equal: {same}", flush=True)


if __name__ == "__main__":
    main()
