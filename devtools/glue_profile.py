"""Where a frame of the temporal loop spends its host time outside the denoising steps: the C4 sequence (bench_rows.sequence64's
setup) run with 3 steps per frame under cProfile, plus wall time per frame.  python devtools/glue_profile.py [frames]"""
import cProfile
import io
import os
import pstats
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "devtools"))
import bench_rows as R  # noqa: E402
from lidarcrafter_amd.testing import synth_scene_boxes, synth_temporal_inputs  # noqa: E402
from lidargen.dataset.custom_dataset import CustomDataset, DataConfig  # noqa: E402
from lidargen.models.diffusion import CondContinuousTimeGaussianDiffusion  # noqa: E402
from lidargen.utils import temporal  # noqa: E402
from lidargen.utils.lidar import LiDARUtility  # noqa: E402


def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    dev = torch.device("cuda:0")
    H, W = 64, 2048
    m0, e0 = R._cond_pair_64(10)
    m1, e1 = R._cond_pair_64(11)
    ddpm = CondContinuousTimeGaussianDiffusion(m0, e0, cond_mode="concat").eval().to(dev)
    auto = CondContinuousTimeGaussianDiffusion(m1, e1, cond_mode="concat").eval().to(dev)
    lu = LiDARUtility(resolution=(H, W), depth_format="log_depth", min_depth=1.45, max_depth=80.0,
                      ray_angles=m0.coords).to(dev)

    class Cfg(DataConfig):
        resolution = (H, W)

    K_ = 6
    sb = synth_scene_boxes(K_, seed=40)
    names = ["ego"] + [DataConfig.class_names[int(c) - 1] for c in sb[:, 7]]
    info = dict(gt_boxes=np.concatenate([np.zeros((1, 7)), sb[:, :7].astype(np.float64)]), gt_names=names)
    ds = CustomDataset([dict(info)], cfg=Cfg())
    batch = ds.collate_fn([ds[0]])
    batch["gt_fut_trajs"] = [synth_temporal_inputs(50, K=K_)[0]]

    def run(nf, ns):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        temporal.generate_sequence(ddpm, auto, lu, dict(batch), num_frames=nf, num_steps=ns, mode="ddpm",
                                   traj_length=16, rng=[torch.Generator().manual_seed(90)], data_cfg=Cfg())
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    run(2, 4)
    for ns in (3, 3, 8):
        print(f"{frames} frames x {ns} steps: {run(frames, ns) * 1e3 / frames:.2f} ms per frame", flush=True)
    pr = cProfile.Profile()
    pr.enable()
    run(frames, 3)
    pr.disable()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
    print(s.getvalue())


if __name__ == "__main__":
    main()
