"""Soak of the layout-conditioned path with everything of round 6 switched on (step graph kept across calls, operand / encoder graphs,
unit-form attention, paired resampler): N sample() calls of S DDPM steps over changing conditions and seeds, every call repeated later
in another order -- results must be finite, free of range events and bit-equal between the two visits.  python devtools/soak_c3.py [S] [N]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lidarcrafter_amd import ops as K  # noqa: E402
from lidarcrafter_amd.testing import seeded_fill, synth_layout_batch  # noqa: E402
from lidargen.utils import inference  # noqa: E402
from lidargen.utils.configs import __all__ as CONFIGS  # noqa: E402


def main():
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    dev = torch.device("cuda:0")
    ddpm, model, _ = inference.load_model_duffusion_training(CONFIGS["nuscenes-box-layout-v6"]())
    seeded_fill(model, salt=200), seeded_fill(ddpm.condition_model, salt=201)
    ddpm = ddpm.eval().to(dev)
    B = 4
    batches = [{k: v.to(dev) for k, v in synth_layout_batch(B, 32, 1024, seed=300 + i).items()} for i in range(N)]

    def run(i, mode):
        rng = [torch.Generator().manual_seed(1000 * i + j) for j in range(B)]
        return ddpm.sample(dict(batches[i]), B, S, progress=False, rng=rng, mode=mode)

    t0 = time.perf_counter()
    first = {}
    for i in range(N):
        for mode in ("ddpm", "ddim"):
            first[i, mode] = run(i, mode)
            assert torch.isfinite(first[i, mode]).all()
    for i in reversed(range(N)):
        for mode in ("ddim", "ddpm"):
            again = run(i, mode)
            assert torch.equal(again, first[i, mode]), (i, mode)
    assert not K.range_poll(dev)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"soak ok: {4 * N} calls x {S} steps at batch {B}, {dt:.1f} s, {dt / (4 * N * S) * 1e3:.2f} ms per step incl. per-call work; "
          f"second visits bit-equal, no range event; graphs: sampler {len(__import__('lidargen.models.diffusion.continuous_time', fromlist=['x'])._GRAPH_CACHES.get(ddpm, {}))}, "
          f"operands {bool(ddpm.model._prep['graph'])}, encoder {bool(ddpm.condition_model._core_graph['graph'])}")


if __name__ == "__main__":
    main()
