"""N>1 path on CPU: world_size-2 gloo processes exercise the sharding + all-gather code that the
GPU job runs over RCCL (the denoiser itself needs the GPU and is covered by -m gpu tests)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from lidarcrafter_amd import parallel


def test_shard_range_partition():
    for gb in (1, 7, 8, 32, 33):
        for world in (1, 2, 4, 8):
            idx = [i for r in range(world) for i in parallel.shard_range(gb, r, world)]
            assert idx == list(range(gb))
            sizes = [len(parallel.shard_range(gb, r, world)) for r in range(world)]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        parallel.shard_range(8, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, gb, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from lidargen.models.diffusion import ContinuousTimeGaussianDiffusion

        class Stub(torch.nn.Module):
            resolution, in_channels = (4, 16), 2

        ddpm = ContinuousTimeGaussianDiffusion(Stub(), torch.nn.Identity())
        rng = parallel.shard_generators(gb, rank, world, base_seed=3)
        # x_T of this shard through the product's RNG contract (base.randn, CPU generators)
        x = ddpm.randn(len(rng), *ddpm.sampling_shape, rng=rng, device="cpu")
        x = x + 0.0 * rank
        full = parallel.gather_frames(x, gb)
        q.put((rank, full.numpy().copy()))   # by value: a shared-fd tensor dies with the worker
    finally:
        dist.destroy_process_group()


def _run_two_ranks(gb):
    """One attempt: (rank -> gathered tensor) or None when the rendezvous did not come up (a cold
    `import torch` in the spawned interpreters, or the probed port taken in between)."""
    import queue

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, gb, q)) for r in range(2)]
    [p.start() for p in procs]
    try:
        got = {r: torch.from_numpy(a) for r, a in (q.get(timeout=300) for _ in range(2))}
    except queue.Empty:
        got = None
    [p.join(timeout=60) for p in procs]
    for p in procs:
        if p.is_alive():
            p.kill()          # exactly the processes started above
            p.join()
    if got is None or not all(p.exitcode == 0 for p in procs):
        return None
    return got


@pytest.mark.parametrize("gb", [8, 5])
def test_two_rank_gather_is_shard_invariant(gb):
    got = _run_two_ranks(gb) or _run_two_ranks(gb)      # one retry on a failed rendezvous
    assert got is not None, "two-rank gloo run failed twice"
    # single-process draw of the whole batch with the same global seeds
    ref = torch.stack([torch.randn(2, 4, 16, generator=torch.Generator().manual_seed(3 + i))
                       for i in range(gb)])
    for r in range(2):
        assert got[r].shape == ref.shape
        assert torch.equal(got[r], ref)
