"""Data-parallel sampling over the GPUs of one node: one process per GPU, `torch.distributed`
with backend "nccl" (= RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The path shards over SAMPLES with no cross-sample coupling: weights are
replicated, rank r owns the contiguous block of global sample indices below and the per-sample
generators seeded with those indices, so every generated frame is independent of the number of
ranks.  There is no collective inside the denoising loop; the only message is one all-gather of
the finished frames (the reference instead `torch.save`s per-rank files,
tools/evaluation/sample_and_save_cond.py:157-159)."""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.distributed as dist


def shard_range(global_batch: int, rank: int, world: int) -> range:
    """Contiguous block of global sample indices owned by `rank` (sizes differ by at most 1)."""
    if not 0 <= rank < world:
        raise ValueError(f"rank {rank} outside world {world}")
    base, extra = divmod(global_batch, world)
    lo = rank * base + min(rank, extra)
    return range(lo, lo + base + (1 if rank < extra else 0))


def shard_generators(global_batch: int, rank: int, world: int, base_seed: int = 0,
                     device: str = "cpu") -> List[torch.Generator]:
    """One generator per LOCAL sample, seeded with base_seed + GLOBAL sample index."""
    return [torch.Generator(device=device).manual_seed(base_seed + i)
            for i in shard_range(global_batch, rank, world)]


def gather_frames(local: torch.Tensor, global_batch: int, group=None) -> torch.Tensor:
    """All-gather rank-local frames [b_r, ...] into [global_batch, ...] in global sample order.
    Shards may differ in size by one sample, so shorter shards are padded for the collective."""
    if not (dist.is_available() and dist.is_initialized()):
        return local
    world = dist.get_world_size(group)
    sizes = [len(shard_range(global_batch, r, world)) for r in range(world)]
    mx = max(sizes)
    pad = local
    if local.shape[0] < mx:
        pad = torch.cat([local, local.new_zeros((mx - local.shape[0], *local.shape[1:]))], 0)
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad.contiguous(), group=group)
    return torch.cat([b[:n] for b, n in zip(bufs, sizes)], 0)


@torch.inference_mode()
def sample_data_parallel(ddpm, global_batch: int, num_steps: int, *, batch_dict: Optional[dict] = None,
                         mode: str = "ddim", ddim_eta: float = 0.0, base_seed: int = 0,
                         gather: bool = True, group=None) -> torch.Tensor:
    """Rank-local `ddpm.sample` on this rank's shard, then one all-gather.  `batch_dict` (layout
    condition) must already hold this rank's shard of the batch."""
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rng = shard_generators(global_batch, rank, world, base_seed)
    b = len(rng)
    kw = dict(progress=False, rng=rng, mode=mode, ddim_eta=ddim_eta)
    x = ddpm.sample(batch_dict, b, num_steps, **kw) if batch_dict is not None else \
        ddpm.sample(b, num_steps, **kw)
    return gather_frames(x, global_batch, group) if gather else x
