"""Worker of tests/test_training.py::test_ddp_two_ranks (launched with torch.distributed.run, one rank per
GPU, RCCL): every rank runs loss.backward() of the SAME reduced denoiser on ITS shard of a 4-sample batch
under DistributedDataParallel; rank 0 also computes the unwrapped gradients of both shards and checks that
the all-reduced gradient is their mean (train_lidm_cond.py:139-141 is the reference's only collective)."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = int(os.environ.get("LOCAL_RANK", rank))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("nccl", rank=rank, world_size=world)
    from lidargen.models.diffusion import ContinuousTimeGaussianDiffusion
    from lidarcrafter_amd.testing import seeded_randn
    from tests.test_hip_parity import _uncond

    m = _uncond(16, (8, 64), dev).train()
    ddpm = ContinuousTimeGaussianDiffusion(m, torch.nn.Identity()).to(dev)
    x0 = seeded_randn(2 * world, 2, 8, 64, seed=41).clamp(-1, 1).to(dev)
    shard = lambda r: x0[2 * r:2 * r + 2]
    ref = None
    if rank == 0:                                   # unwrapped gradients of every shard, averaged
        acc = {}
        for r in range(world):
            ddpm.zero_grad(set_to_none=True)
            torch.manual_seed(100 + r)
            ddpm(shard(r)).backward()
            for k, p in ddpm.named_parameters():
                if p.grad is not None:
                    acc[k] = acc.get(k, 0) + p.grad.detach().clone() / world
        ref = acc
        ddpm.zero_grad(set_to_none=True)
    wrapped = torch.nn.parallel.DistributedDataParallel(ddpm, device_ids=[local], bucket_cap_mb=1)
    torch.manual_seed(100 + rank)                   # the same timestep / noise draws as the unwrapped pass
    wrapped(shard(rank)).backward()
    ok = True
    if rank == 0:
        worst = 0.0
        for k, p in ddpm.named_parameters():
            if p.grad is None:
                continue
            d = float((p.grad - ref[k]).abs().max()) / (float(ref[k].abs().max()) + 1e-30)
            worst = max(worst, d)
        ok = worst < 1e-5 and len(ref) > 100
        print(f"ddp2 worst relative deviation {worst:.3e} over {len(ref)} parameters: {'OK' if ok else 'FAIL'}")
    flag = torch.tensor([1 if ok else 0], device=dev)
    dist.broadcast(flag, 0)
    dist.destroy_process_group()
    sys.exit(0 if int(flag.item()) else 1)


if __name__ == "__main__":
    main()
