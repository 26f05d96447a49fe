"""Sampling harness with the reference CLI's flags (tools/generate/generate.py:92-102 and the bulk
sampler tools/evaluation/sample_and_save_cond.py): `--cfg --ckpt --device --mode --batch_size
--sampling_steps`.

  python -m lidarcrafter_amd.cli --cfg nuscenes-unet-uncond --batch_size 8 --sampling_steps 50 \
         --mode ddim --out samples/
  torchrun --nproc-per-node 8 -m lidarcrafter_amd.cli --cfg nuscenes-box-layout-v6 ...

Without --ckpt (no checkpoints ship with the reference, README.md:62) the weights are the seeded
random initialisation used by the tests.  Layout-conditioned configs take a synthetic layout batch
(lidarcrafter_amd.testing.synth_layout_batch) unless --batch_pt points to a saved batch dict.
Output per rank-0: `<out>/samples.pt` = float32 [N,5,H,W] (metric depth, x, y, z, reflectance), the
tensor sample_and_save_cond.py:119-124,157-159 saves per sample."""
from __future__ import annotations

import argparse
import os
import time

import torch


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--cfg", default="nuscenes-unet-uncond")
    ap.add_argument("--ckpt", default=None)
    ap.add_argument("--device", default="cuda")
    ap.add_argument("--mode", choices=["ddpm", "ddim"], default="ddim")
    ap.add_argument("--batch_size", type=int, default=8, help="GLOBAL batch (sharded over ranks)")
    ap.add_argument("--sampling_steps", type=int, default=50)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--batch_pt", default=None, help="torch-saved layout batch dict (cond configs)")
    ap.add_argument("--out", default="samples")
    args = ap.parse_args(argv)

    import torch.distributed as dist

    from lidarcrafter_amd import parallel
    from lidarcrafter_amd.testing import seeded_fill, synth_layout_batch
    from lidargen.utils import inference
    from lidargen.utils.configs import __all__ as CONFIGS

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    # under torchrun (RANK / WORLD_SIZE exported) the process group is created for ONE rank too: the same RCCL
    # init / all-gather path runs on a single-GPU box as on the 8-GPU node (tests/test_rccl_single_gpu.py)
    dist_on = world > 1 or ("RANK" in os.environ and "WORLD_SIZE" in os.environ and args.device == "cuda")
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29534")
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    device = torch.device(args.device, torch.cuda.current_device()) if args.device == "cuda" \
        else torch.device(args.device)

    cfg = CONFIGS[args.cfg]()
    cfg.resume = args.ckpt
    built = inference.load_model_duffusion_training(cfg)
    ddpm, model, lidar_utils = built[:3]
    if args.ckpt is None:
        seeded_fill(ddpm, salt=100)
    ddpm, lidar_utils = ddpm.eval().to(device), lidar_utils.to(device)

    shard = parallel.shard_range(args.batch_size, rank, world)
    batch = None
    if hasattr(cfg, "condition_model"):
        H, W = cfg.data.resolution
        if args.batch_pt:
            full = torch.load(args.batch_pt, map_location="cpu")
        else:
            extra = cfg.condition_model.params["out_channels"] - 10
            full = synth_layout_batch(args.batch_size, H, W, seed=args.seed, n_extra=extra)
        batch = {k: v[shard.start:shard.stop].to(device) for k, v in full.items()}
    t0 = time.perf_counter()
    frames = parallel.sample_data_parallel(ddpm, args.batch_size, args.sampling_steps,
                                           batch_dict=batch, mode=args.mode, base_seed=args.seed,
                                           gather=False)
    out = lidar_utils.postprocess(frames.clamp(-1, 1))          # [b,5,H,W] fused epilogue
    out = parallel.gather_frames(out, args.batch_size)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if rank == 0:
        os.makedirs(args.out, exist_ok=True)
        torch.save(out.cpu(), os.path.join(args.out, "samples.pt"))
        print(f"{args.cfg}: {out.shape[0]} frames, {args.sampling_steps} {args.mode} steps, "
              f"{dt:.2f} s ({args.sampling_steps / dt:.1f} denoising-steps/s) -> {args.out}/samples.pt")
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
