"""The collective path on hardware, with the one GPU the test box has: RCCL process group (`nccl` backend with
`device_id`), barrier, `all_reduce(MAX)` of the timing and the all-gather of the generated frames on DEVICE tensors --
what `bench.py --gpus N` and `python -m lidarcrafter_amd.cli` under torchrun execute on the 8-GPU node, here with
world size 1 (the reference's per-rank loop: tools/evaluation/sample_and_save_cond.py:33-38,157-159).
The world-size-2 logic (shard ranges, padded all-gather, shard-invariant frames) is covered on CPU by
tests/test_parallel_gloo.py."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _env(**kw):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    env.update({k: str(v) for k, v in kw.items()})
    return env


def test_bench_takes_the_rccl_path_with_one_rank():
    assert torch.cuda.is_available()
    env = _env(LC_BENCH_FORCE_DIST=1, MASTER_ADDR="127.0.0.1", MASTER_PORT=_free_port())
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "2",
                        "--repeat", "1", "--no-cpu-baseline", "--no-traffic", "--no-rows"],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 2 and d["value"] > 0
    assert d["config"]["parallelism"].startswith("dp1")
    assert d["verify"]["ok"] is True                     # the 50-step check against the reference fixture ran inside it
    assert d["verify"]["ranks_verified"] == 1 and d["verify"]["per_rank"][0]["samples"] == [0, 7]
    assert d["config"].get("collectives") == "rccl: barrier + all_reduce(MAX) + all_gather of the frames executed"


def test_bench_verifies_the_shard_of_any_rank():
    """rank r of `bench.py --gpus N` checks ITS shard (global samples 8r..8r+7) against the
    reference's run of those seeds -- the function every rank calls, here for rank 3 of 8 on the one GPU."""
    sys.path.insert(0, ROOT)
    import bench

    dev = torch.device("cuda:0")
    ddpm, _ = bench.build_ddpm(dev)
    res = bench.verify_against_reference(ddpm, 3, 8, dev)
    assert res["ok"] is True and res["samples"] == [24, 31] and res["against"].endswith("c2_shards.npz"), res
    assert 0 < res["max_rel_l2_per_sample"]["x50"] < 1e-3
    x3 = bench.x_T_for(3, ddpm.sampling_shape, 8)
    assert torch.equal(x3[2], torch.randn(*ddpm.sampling_shape, generator=torch.Generator().manual_seed(26)))


def test_cli_under_torchrun_one_rank(tmp_path):
    assert torch.cuda.is_available()
    out = tmp_path / "samples"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), "-m", "lidarcrafter_amd.cli", "--cfg", "nuscenes-unet-uncond",
           "--batch_size", "2", "--sampling_steps", "3", "--out", str(out)]
    p = subprocess.run(cmd, env=_env(), cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, (p.stdout + p.stderr)[-2000:]
    frames = torch.load(out / "samples.pt")
    assert frames.shape == (2, 5, 32, 1024) and torch.isfinite(frames).all()
    # the same frames without a process group (shard-invariance, world 1): bit-equal
    out2 = tmp_path / "plain"
    p2 = subprocess.run([sys.executable, "-m", "lidarcrafter_amd.cli", "--cfg", "nuscenes-unet-uncond", "--batch_size", "2",
                         "--sampling_steps", "3", "--out", str(out2)], env=_env(), cwd=ROOT, capture_output=True, text=True,
                        timeout=600)
    assert p2.returncode == 0, (p2.stdout + p2.stderr)[-2000:]
    assert torch.equal(frames, torch.load(out2 / "samples.pt"))
