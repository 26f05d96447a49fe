#!/usr/bin/env python
"""Benchmark of the denoising hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W [--repeat R]
  (N>1: one rank per GPU over RCCL -- under `python -m torch.distributed.run --nproc-per-node N`, as
   the driver launches it, or started plainly, in which case it re-executes itself under
   torch.distributed.run with N local ranks; --gpus must equal WORLD_SIZE)

A "step" = one reverse-diffusion step of the C2 workload of BASELINE.json (EfficientUNet
`nuscenes-unet-uncond`, 32x1024 range image, DDIM, batch 8 PER GPU, seeded random-init weights,
synthetic x_T): one denoiser forward + the fused x0/clamp/update kernel, inputs resident in HBM.
Weak scaling: every rank owns its own batch of 8 samples (global sample index = rank*8 + i); the
only communication is one RCCL all-gather of the finished frames after the loop.

Timing: W warmup steps, then R sweeps (default 10) of EXACTLY K steps, each sweep bracketed by a
barrier + torch.cuda.synchronize() on both sides and reduced with MAX over ranks; `value` and
`ms_per_step` come from the MEDIAN sweep, `timing` carries all sweeps and the spread (R = 1
reproduces the single-sweep protocol).  Before timing, one step of the bench configuration is
checked against the reference's own output (tests/golden/c2_b8.npz, `verify`).

Prints ONE JSON line (rank 0) with the driver's contract fields plus
  roofline     -- dominant kernel (3x3 ring conv, f16x2-split MFMA): algorithmic FLOPs / HIP-event time
  cpu_baseline -- the CPU oracle (oracle/, torch fp32) timed on this host on a bounded sample.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# multi-process GPU work on this driver stack needs dmabuf IPC (RCCL's hipIpcGetMemHandle fails
# otherwise); the images export it already, keep it if the launcher's environment was scrubbed
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

BATCH_PER_GPU = 8
RES = (32, 1024)
PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_F16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: bf16/f16 MFMA dense peak (~2.5 PF)
GFLOP_PER_SAMPLE_STEP = 116.6  # SURVEY.md §8(d), uncond 32x1024 (114.4 conv + 2.15 MHA)


def build_ddpm(device):
    from lidarcrafter_amd.testing import seeded_fill
    from lidargen.utils import inference
    from lidargen.utils.configs import __all__ as CONFIGS

    cfg = CONFIGS["nuscenes-unet-uncond"]()
    ddpm, model, lidar_utils = inference.load_model_duffusion_training(cfg)
    seeded_fill(model, salt=100)  # random-init weights of the named architecture (no checkpoints)
    return ddpm.eval().to(device), cfg


def x_T_for(rank: int, shape, world: int = 1):
    """x_T of this rank's shard: per-sample CPU generators seeded with the GLOBAL sample index."""
    from lidarcrafter_amd import parallel

    g = parallel.shard_generators(BATCH_PER_GPU * world, rank, world)
    return torch.stack([torch.randn(*shape, generator=r) for r in g])


def cpu_baseline(n_steps: int = 2):
    """Oracle leg: the CPU restatement (checked against the reference's golden vectors) on the
    same workload, bounded sample, all host cores torch gives us."""
    from lidarcrafter_amd.testing import seeded_fill
    from lidargen.utils import inference
    from lidargen.utils.configs import __all__ as CONFIGS
    from oracle import denoiser as D
    from oracle import diffusion as DF

    cfg = CONFIGS["nuscenes-unet-uncond"]()
    ddpm, model, _ = inference.load_model_duffusion_training(cfg)
    seeded_fill(model, salt=100)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    x = x_T_for(0, ddpm.sampling_shape)
    steps = torch.linspace(1.0, 0.0, 51)[None].repeat_interleave(BATCH_PER_GPU, 0)
    den = lambda a, lam: D.efficient_unet_forward(sd, a, lam)
    noise = torch.zeros_like(x)
    DF.p_step(den, x[:1], steps[:1, 0], steps[:1, 1], noise[:1], mode="ddim")  # warm
    t0 = time.perf_counter()
    for i in range(n_steps):
        x = DF.p_step(den, x, steps[:, i], steps[:, i + 1], noise, mode="ddim")
    dt = time.perf_counter() - t0
    cpu_model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith("model name"):
                    cpu_model = ln.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {"value": n_steps / dt, "unit": "denoising-steps/s (batch 8)",
            "cores": torch.get_num_threads(), "cpu_model": cpu_model, "kind": "port",
            "sample": f"{n_steps} DDIM steps of the C2 batch (8x 32x1024) through "
                      f"oracle.efficient_unet_forward + oracle p_step, torch fp32 CPU, "
                      f"{dt:.1f} s wall"}


def measure_hbm_traffic(timeout_s: int = 150):
    """HBM-side bytes per launch of the dominant kernel family, MEASURED in this run: this script
    re-executes itself for 4 steps under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and, in a
    separate pass, `--pmc WRITE_SIZE` (MI355X_MICROARCH.md HBM section: separate passes, units KB,
    FETCH_SIZE x2 on gfx950 -- 128-byte fabric reads are tallied as 64 bytes -- WRITE_SIZE x1).
    Returns (bytes_per_launch, detail) or (None, reason)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    tool = shutil.which("rocprofv3")
    if tool is None:
        return None, "rocprofv3 not on PATH"
    if any(os.environ.get(k) for k in ("ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB", "ROCPROFILER_LIBRARY_CTOR")):
        return None, "this process already runs under a profiler tool: nested PMC passes skipped"
    totals = {}
    tmp = tempfile.mkdtemp(prefix="lc_pmc_", dir=os.environ.get("TMPDIR", "/tmp"))
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            cmd = [tool, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", out, "-o", "p",
                   "--", sys.executable, os.path.abspath(__file__), "--steps", "4", "--warmup", "2",
                   "--repeat", "1", "--no-verify", "--no-cpu-baseline", "--no-roofline", "--no-rows"]
            env = dict(os.environ, TMPDIR=os.environ.get("TMPDIR", "/tmp"))
            r = subprocess.run(cmd, cwd=tmp, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                               timeout=timeout_s)
            files = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None, f"rocprofv3 --pmc {counter} pass failed (rc {r.returncode})"
            tot, n = 0.0, 0
            for row in csv.DictReader(open(files[0])):
                if row["Counter_Name"] != counter:
                    continue
                k = row["Kernel_Name"]
                # the 3x3 instantiations (HCfg<..., 3>), the tall level-0 kernel and the stride-2 conv of the folded down path
                if ("conv_f16x2" in k and ("Li3EEEE" in k or ", 3>" in k or "tall_kernel" in k)) or "conv_s2_shared_w" in k:
                    tot += float(row["Counter_Value"]) * 1024.0
                    n += 1
            if n == 0:
                return None, f"no 3x3 conv launches in the {counter} pass"
            totals[counter] = (tot, n)
    except Exception as e:   # a missing tool / a refused counter must not cost the bench line
        return None, f"PMC passes failed: {e!r}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    (f, nf), (w, nw) = totals["FETCH_SIZE"], totals["WRITE_SIZE"]
    read, write = 2.0 * f / nf, w / nw
    return read + write, {"read_bytes_per_launch": round(read), "write_bytes_per_launch": round(write),
                          "launches_counted": nf}


def verify_against_reference(ddpm, rank, world, device):
    """The full 50-step DDIM run of the BENCH configuration (batch 8, x_T from CPU generators seeded with the
    GLOBAL sample index -- exactly what the timed loop of this rank runs) against the reference's own CPU run of
    the same seeds.  EVERY rank checks ITS OWN shard: rank 0 against tests/golden/c2_b8.npz (samples 0..7, states
    1 / 25 / 50, every 4th column), rank r = 1..7 against tests/golden/c2_shards.npz (samples 8r..8r+7, final
    state, every 8th column; make_fixtures.py::sec_c2_shards).  Returns this rank's result dict."""
    import numpy as np

    if rank == 0:
        path, keys = os.path.join(ROOT, "tests", "golden", "c2_b8.npz"), {1: ("x1_s4", 4), 25: ("x25_s4", 4), 50: ("x50_s4", 4)}
    else:
        path, keys = os.path.join(ROOT, "tests", "golden", "c2_shards.npz"), {50: (f"shard{rank}_x50_s8", 8)}
    if rank > 7 or not os.path.exists(path):
        return {"rank": rank, "ok": None, "why": "no reference fixture for this shard"}
    g = np.load(path)
    x_T = x_T_for(rank, ddpm.sampling_shape, world).to(device)
    st = ddpm.begin_sampling(BATCH_PER_GPU, 50, rng=None, mode="ddim", ddim_eta=0.0, x_T=x_T)
    out = {}
    for i in range(1, 51):
        x = ddpm.sampling_step(st)
        if i in keys:
            name, stride = keys[i]
            a = x[..., ::stride].double().cpu().flatten(1)
            b = torch.from_numpy(g[name]).double().flatten(1)
            out[f"x{i}"] = float(((a - b).norm(dim=1) / b.norm(dim=1)).max())
    tol = 1e-3
    return {"rank": rank, "samples": [BATCH_PER_GPU * rank, BATCH_PER_GPU * rank + BATCH_PER_GPU - 1],
            "ok": all(v < tol for v in out.values()),
            "max_rel_l2_per_sample": {k: float(f"{v:.3g}") for k, v in out.items()},
            "against": os.path.relpath(path, ROOT)}


def verify_all_ranks(ddpm, rank, world, device, dist_on):
    """Every rank verifies its shard; rank 0 collects the per-rank results (one all_gather_object) and the run
    stops on every rank if any shard is off."""
    mine = verify_against_reference(ddpm, rank, world, device)
    results = [mine]
    if dist_on and world > 1:
        import torch.distributed as dist

        results = [None] * world
        dist.all_gather_object(results, mine)
    bad = [r for r in results if r["ok"] is False]
    res = {"ok": not bad and all(r["ok"] for r in results if r["ok"] is not None),
           "tolerance_rel_l2": 1e-3,
           "max_rel_l2_per_sample": results[0].get("max_rel_l2_per_sample"),
           "ranks_verified": sum(1 for r in results if r["ok"]), "per_rank": results,
           "against": "the reference's CPU run of C2 (batch 8, 50 DDIM steps, same seeds): tests/golden/c2_b8.npz "
                      "(rank 0: states 1 / 25 / 50), tests/golden/c2_shards.npz (ranks 1..7: final state of "
                      "global samples 8r..8r+7)"}
    if bad:
        raise SystemExit(f"bench.py: the bench configuration does not reproduce the reference: {res}")
    return res


def box_calibration(device, seconds: float = 0.2):
    """What THIS box sustains (the pool's boxes spread 229-281 steps/s for one tree): a bare
    v_mfma_f32_32x32x16_f16 loop on random fp16 operands (the matrix pipes run into the chip's power budget, and
    the sustained rate depends on the operand bits) and a streaming copy larger than the 256 MB Infinity Cache,
    ~`seconds` of GPU time each.  Not part of the timed region."""
    from lidarcrafter_amd import _lib

    L = _lib.lib()
    st = torch.cuda.current_stream(device).cuda_stream
    blocks = 512
    ops = (torch.rand(blocks * 512 * 4 * 8, device=device) * 2 - 1).to(torch.float16)
    sink = torch.empty(blocks * 512, device=device)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def mfma(iters):
        e0.record()
        fl = L.lc_calibrate_mfma_f16(ops.data_ptr(), blocks, iters, sink.data_ptr(), st)
        e1.record()
        e1.synchronize()
        assert fl > 0, fl
        return fl, e0.elapsed_time(e1) * 1e-3

    mfma(200)
    fl, t = mfma(2000)
    iters = max(2000, int(2000 * seconds / max(t, 1e-6) / 4))
    rates = []
    for _ in range(4):
        fl, t = mfma(iters)
        rates.append(fl / t / 1e12)
    n = 1 << 28                                             # 1 GiB of floats each way
    src = torch.empty(n, device=device).normal_()
    dst = torch.empty(n, device=device)
    reps = 0
    for k in range(2):
        e0.record()
        reps = 3 if k == 0 else max(3, int(seconds / max(tc, 1e-6)))
        for _ in range(reps):
            rc = L.lc_calibrate_stream_copy(src.data_ptr(), dst.data_ptr(), n, st)
            assert rc == 0, rc
        e1.record()
        e1.synchronize()
        tc = e0.elapsed_time(e1) * 1e-3 / reps
    # the same MFMA loop in launches of ~50 us, each followed by a 32 MiB -> 32 MiB copy (~15 us): the duty cycle of the
    # denoising step (matrix-bound launches between memory-bound passes) -- the chip's power management gives short bursts
    # more clock than the long launches above
    it_s = max(50, int(iters * 50e-6 / max(t, 1e-9)))
    m = 1 << 23

    def loop(with_mfma, nl=100):
        e0.record()
        fl_ = 0
        for _ in range(nl):
            if with_mfma:
                fl_ += L.lc_calibrate_mfma_f16(ops.data_ptr(), blocks, it_s, sink.data_ptr(), st)
            L.lc_calibrate_stream_copy(src.data_ptr(), dst.data_ptr(), m, st)
        e1.record()
        e1.synchronize()
        return fl_, e0.elapsed_time(e1) * 1e-3

    loop(True, 10)
    fl_b, t_both = loop(True)
    _, t_copy = loop(False)
    burst = fl_b / max(t_both - t_copy, 1e-9) / 1e12
    del src, dst
    mf = sorted(rates)[len(rates) // 2]
    best = max(mf, burst)
    return {"mfma_f16_tflops_random_operands": round(mf, 1),
            "mfma_f16_tflops_random_operands_short_launches": round(burst, 1),
            "three_product_ceiling_tflops": round(best / 3, 1),
            "stream_copy_tb_s": round(2 * 4 * n / tc / 1e12, 3),
            "how": f"lc_calibrate_mfma_f16: {blocks} blocks x 8 waves, 4 accumulators; long launches: {iters} x 16 MFMAs per "
                   f"wave, median of 4; short launches: 100 x ({it_s} x 16 MFMAs per wave + a 32 MiB copy), copy-only loop "
                   f"subtracted; ceiling = the larger of the two / 3 products; lc_calibrate_stream_copy: 1 GiB -> 1 GiB "
                   f"float4 copy, read + write bytes, {reps} launches"}


def extra_rows(device, steps_small: int = 20):
    """Same-process rows next to the headline: the C2 shape at batch 1 and batch 32 and the
    C3 shard (layout-conditioned denoiser, batch 8) -- ms per denoising step through the graph-replayed sampler,
    each behind its own check against the reference's outputs (tests/golden/c2_b8.npz, c3_b8.npz)."""
    import numpy as np

    from lidarcrafter_amd.testing import seeded_fill, seeded_randn, synth_layout_batch
    from lidargen.utils import inference
    from lidargen.utils.configs import __all__ as CONFIGS

    def rel(a, b):
        a, b = a.double().cpu().flatten(1), b.double().flatten(1)
        return float(((a - b).norm(dim=1) / b.norm(dim=1)).max())

    def timed(fn, n):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n

    rows = {}
    g = np.load(os.path.join(ROOT, "tests", "golden", "c2_b8.npz"))
    ddpm, _ = build_ddpm(device)
    x8 = seeded_randn(8, 2, 32, 1024, seed=81).to(device)
    lam8 = torch.from_numpy(g["lam"]).to(device)
    want = torch.from_numpy(g["y_s4"])
    for B, n in ((1, steps_small), (32, max(5, steps_small // 2))):
        with torch.no_grad():
            if B == 1:
                r = max(rel(ddpm.model(x8[k:k + 1].contiguous(), lam8[k:k + 1])[..., ::4], want[k:k + 1]) for k in (0, 5))
            else:
                r = rel(ddpm.model(x8.repeat(4, 1, 1, 1), lam8.repeat(4))[..., ::4], want.repeat(4, 1, 1, 1))
        if not r < 2e-5:
            raise SystemExit(f"bench.py rows: batch-{B} forward is {r} from the reference's output")
        gen = [torch.Generator().manual_seed(i) for i in range(B)]
        x_T = torch.stack([torch.randn(*ddpm.sampling_shape, generator=q) for q in gen]).to(device)
        st = ddpm.begin_sampling(B, n + 6, rng=None, mode="ddim", ddim_eta=0.0, x_T=x_T)
        dt = timed(lambda: ddpm.sampling_step(st), n)
        assert torch.isfinite(st["x"]).all()
        rows[f"uncond_32x1024_batch{B}"] = {
            "ms_per_step": round(dt * 1e3, 3), "sample_steps_per_s": round(B / dt, 1),
            "algorithmic_tflops": round(B * GFLOP_PER_SAMPLE_STEP / dt / 1e3, 1),
            "check": {"forward_max_rel_l2_vs_reference": float(f"{r:.3g}"), "tolerance": 2e-5,
                      "against": "tests/golden/c2_b8.npz y_s4 (samples of a batch are independent in the reference)"}}
    # whole `sample()` calls, as tools/generate and the bulk harness issue them (one per batch): the first call of a shape
    # runs its first step eagerly and captures the step's HIP graph, later calls replay that graph from step 0
    calls = []
    for _ in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        xs = ddpm.sample(8, 50, progress=False, rng=None, mode="ddim")
        torch.cuda.synchronize()
        calls.append(round((time.perf_counter() - t0) * 1e3, 2))
    assert torch.isfinite(xs).all()
    rows["uncond_32x1024_batch8_sample_call"] = {
        "ddim_steps": 50, "ms_per_call": calls,
        "note": "wall time of ddpm.sample(8, 50, mode='ddim') calls on one sampler, in order; 50 x the headline's "
                "ms_per_step is the floor; the range poll at the end of a call is one blocking 64 KB copy"}
    del ddpm
    g = np.load(os.path.join(ROOT, "tests", "golden", "c3_b8.npz"))
    cfg = CONFIGS["nuscenes-box-layout-v6"]()
    ddpm, model, _ = inference.load_model_duffusion_training(cfg)
    seeded_fill(model, salt=200), seeded_fill(ddpm.condition_model, salt=201)
    ddpm = ddpm.eval().to(device)
    batch = {k: v.to(device) for k, v in synth_layout_batch(8, 32, 1024, seed=83).items()}
    x = seeded_randn(8, 2, 32, 1024, seed=84).to(device)
    with torch.inference_mode():
        cond = ddpm.condition_model(batch)
        y = ddpm.model(x, {"time_condition": torch.from_numpy(g["lam"]).to(device), "other_condition": cond})
        r = rel(y[..., ::4], torch.from_numpy(g["y_s4"]))
        if not r < 2e-5:
            raise SystemExit(f"bench.py rows: C3 forward is {r} from the reference's output")
        gen = [torch.Generator().manual_seed(40 + i) for i in range(8)]
        x_T = ddpm.randn(8, *ddpm.sampling_shape, rng=gen, device=ddpm.device)
        cdict = ddpm.get_network_condition(input_dict=batch, only_custom_condition=True)
        n = max(5, steps_small // 2)
        st = ddpm.begin_sampling(8, n + 6, None, "ddim", 0.0, x_T=x_T, condition_dict=cdict)
        dt = timed(lambda: ddpm.sampling_step(st), n)
        assert torch.isfinite(st["x"]).all()
    rows["cond_layout_v6_32x1024_batch8"] = {
        "ms_per_step": round(dt * 1e3, 3), "sample_steps_per_s": round(8 / dt, 1),
        "algorithmic_tflops": round(8 * 255.9 / dt / 1e3, 1),
        "check": {"forward_max_rel_l2_vs_reference": float(f"{r:.3g}"), "tolerance": 2e-5,
                  "against": "tests/golden/c3_b8.npz y_s4 (C3 shard: nuscenes-box-layout-v6, batch 8)"}}
    return rows


def main():
    global BATCH_PER_GPU
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--repeat", type=int, default=10,
                    help="sweeps of the K timed steps (median reported; 1 = single sweep)")
    ap.add_argument("--no-verify", action="store_true",
                    help="skip the golden-vector check of the bench configuration")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-rows", action="store_true",
                    help="skip the same-process secondary rows (batch 1 / 32, C3) and the box calibration")
    ap.add_argument("--no-traffic", action="store_true",
                    help="skip the two rocprofv3 PMC passes that measure roofline.traffic")
    ap.add_argument("--batch", type=int, default=BATCH_PER_GPU,
                    help="samples per GPU (default 8 = the C2 workload; other values are "
                         "diagnostic and are named in config.workload)")
    ap.add_argument("--conv-precision", choices=["f32", "f16x2"], default=None,
                    help="override LC_CONV_PRECISION (default f16x2)")
    args = ap.parse_args()

    BATCH_PER_GPU = args.batch
    if args.gpus < 1 or args.repeat < 1 or args.steps < 1:
        raise SystemExit("bench.py: --gpus, --steps and --repeat must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # started plainly with --gpus N: become N ranks (one per GPU) under torch.distributed.run
        import socket
        import subprocess

        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
               f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1", "--master-port",
               str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with "
                         f"--nproc-per-node {args.gpus} (or without torchrun: bench.py re-executes "
                         "itself under torch.distributed.run)")
    # (LC_BENCH_FORCE_DIST=1: take the process-group / barrier / all-gather path with one rank too,
    #  to exercise it on a single-GPU box)
    dist_on = world > 1 or os.environ.get("LC_BENCH_FORCE_DIST") == "1"
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if dist_on:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("LOCAL_RANK", "0")
        dist.init_process_group("nccl", device_id=device)

    from lidarcrafter_amd import ops as K

    if args.conv_precision:
        K.set_conv_precision(args.conv_precision)
    ddpm, cfg = build_ddpm(device)
    # The sampler launches step 0 eagerly and captures its HIP graph at step 1; with fewer than two
    # warmup steps that one-off setup would land in the timed region, so it is run up front (and
    # reported as config.graph_setup_steps) -- the W warmup and K timed steps are all graph replays.
    verify = None
    if not args.no_verify and BATCH_PER_GPU == 8:
        verify = verify_all_ranks(ddpm, rank, world, device, dist_on)
    setup = max(0, 2 - args.warmup)
    total = setup + args.warmup + args.steps * args.repeat
    x_T = x_T_for(rank, ddpm.sampling_shape, world).to(device)
    st = ddpm.begin_sampling(BATCH_PER_GPU, total, rng=None, mode="ddim", ddim_eta=0.0, x_T=x_T)
    K.range_poll(device)
    for _ in range(setup):
        ddpm.sampling_step(st)

    def sync():
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        ddpm.sampling_step(st)
    sweeps = []
    for r in range(args.repeat):
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            ddpm.sampling_step(st)
        frames = st["x"]
        if dist_on and r == args.repeat - 1:
            # reassemble the generated frames once, after the loop (RCCL all-gather)
            from lidarcrafter_amd import parallel

            all_frames = parallel.gather_frames(frames, BATCH_PER_GPU * world)
            assert all_frames.shape[0] == BATCH_PER_GPU * world
        sync()
        dt_r = time.perf_counter() - t0
        if dist_on:
            tmax = torch.tensor([dt_r], device=device, dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt_r = float(tmax.item())
        sweeps.append(dt_r)
    frames = frames.clone()      # st["x"] is the model's resident buffer; later runs overwrite it
    assert torch.isfinite(frames).all()
    bad = K.range_poll(device)   # no conv layer of the timed steps ran on saturated fp16 operands
    assert not bad, f"timed steps are invalid: {bad}"
    srt = sorted(sweeps)
    dt = srt[len(srt) // 2] if len(srt) % 2 else 0.5 * (srt[len(srt) // 2 - 1] + srt[len(srt) // 2])

    # the box's calibration and the secondary rows come right behind the timed sweeps -- BEFORE the profiled pass and the
    # rocprofv3 PMC sub-runs of the roofline block (a row measured in the seconds after those sub-runs has come out 25 %
    # slow on one box: batch 1 1.81 ms in the line, 1.43 ms in every other process of the same call)
    calib = rows = None
    if rank == 0 and world == 1 and not args.no_rows:
        calib = box_calibration(device)
        if BATCH_PER_GPU == 8:
            rows = extra_rows(device)
    roof = None
    if rank == 0 and not args.no_roofline:
        # second, instrumented pass: HIP events around every launch of the dominant kernel
        n_prof = min(args.steps, 5)
        st2 = ddpm.begin_sampling(BATCH_PER_GPU, n_prof + 1, rng=None, mode="ddim", x_T=x_T)
        ddpm.sampling_step(st2)                      # one untimed step, then n_prof profiled ones
        torch.cuda.synchronize()
        K.PROFILE = []
        for _ in range(n_prof):
            ddpm.sampling_step(st2)
        torch.cuda.synchronize()
        rec, K.PROFILE = K.PROFILE, None
        # Every profiled step issues the same launch sequence.  A launch's time is the MINIMUM over the n_prof steps of the
        # HIP-event interval at its position in the sequence: the profiled steps run eagerly, and an interval also contains
        # whatever the stream waited for the host in front of that launch (a slow host inflated the family by 8 % on one
        # box of round 6); the minimum over steps drops those waits, the kernels themselves repeat within 1-2 %.
        per_step = len(rec) // n_prof
        assert per_step * n_prof == len(rec) and all(
            rec[i][0] == rec[i + k * per_step][0] for i in range(per_step) for k in range(1, n_prof)), "launch sequence differs between steps"
        fam, mean_t = {}, {}
        for i in range(per_step):
            name, work, _, _, rd, wr, executed = rec[i]
            ts_i = [rec[i + k * per_step][2].elapsed_time(rec[i + k * per_step][3]) * 1e-3 for k in range(n_prof)]
            dt_i = min(ts_i)
            mean_t[name] = mean_t.get(name, 0.0) + sum(ts_i)
            a = fam.setdefault(name, [0.0, 0.0, 0, 0.0, 0.0, 0.0])
            a[0] += work * n_prof
            a[1] += dt_i * n_prof
            a[2] += n_prof
            a[3] += rd * n_prof
            a[4] += wr * n_prof
            a[5] += executed * n_prof
        w, t, n, alg_rd, alg_wr, w_exec = fam["conv3x3"]
        achieved = w / t / 1e12
        split = K.CONV_PRECISION == "f16x2"
        peak = PEAK_F16_MFMA_TFLOPS if split else PEAK_F32_MFMA_TFLOPS
        # HBM-side bytes per launch of the dominant kernel family: measured HERE (two PMC passes of
        # this very script, N = 1 only), else the committed figure of an earlier profile, labelled
        traffic, tsource, tdetail = None, None, None
        if split and BATCH_PER_GPU == 8 and world == 1 and not args.no_traffic:
            traffic, tdetail = measure_hbm_traffic()
            if traffic is not None:
                traffic = round(traffic)
                tsource = ("MEASURED in this run: bench.py re-executed for 4 steps under rocprofv3 "
                           "--kernel-trace --pmc FETCH_SIZE and, separately, --pmc WRITE_SIZE; FETCH_SIZE "
                           "x2 (gfx950 correction), WRITE_SIZE x1, units KB; per 3x3 conv launch")
        if traffic is None:
            why = tdetail if isinstance(tdetail, str) else "--no-traffic / N > 1"
            for tfile in ("r03_hbm_traffic.json", "r02_hbm_traffic.json", "r01_i_hbm_traffic.json"):
                tpath = os.path.join(ROOT, "profiles", tfile)
                if split and BATCH_PER_GPU == 8 and os.path.exists(tpath):
                    traffic = round(json.load(open(tpath))["conv3x3_bytes_per_launch"])
                    tsource = (f"NOT measured in this run ({why}): read from profiles/{tfile} (rocprofv3 "
                               "--pmc FETCH_SIZE (x2 gfx950 correction) + WRITE_SIZE, separate passes, same "
                               "workload, devtools/hbm_traffic.py)")
                    break
            tdetail = None
        roof = {"bound": "mfma",
                "kernel": ("conv_f16x2_{tall,pipe,ps}_kernel<KS=3> + conv_s2_shared_w_kernel" if split else "conv_ring_kernel<KS=3>") +
                          " (all tile instantiations)",
                "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                "frac": round(achieved / peak, 4), "traffic": traffic,
                "traffic_source": tsource, "traffic_detail": tdetail,
                # from the layer shapes of the launches themselves: input + packed weights + residual (where the layer
                # reads one) read once, output written once
                "algorithmic_bytes_per_launch": round((alg_rd + alg_wr) / n),
                "algorithmic_read_bytes_per_launch": round(alg_rd / n),
                "algorithmic_write_bytes_per_launch": round(alg_wr / n),
                "traffic_read_write": (tdetail if isinstance(tdetail, dict) else None),
                "note": ("achieved counts ALGORITHMIC flops (2*MACs); the f16x2 split issues 3 "
                         "f16 MFMAs per product, so the attainable ceiling of this kernel is "
                         "peak/3 = 833 TFLOP/s" if split else
                         "fp32 MFMA (exact fp32), peak = fp32 matrix peak"),
                "launches_per_step": n // n_prof,
                "avg_launch_us": round(t / n * 1e6, 1),
                # (per launch position the minimum over the profiled steps; the plain mean of the same intervals, which also
                #  holds the stream's waits for the host in the eager profiled pass:)
                "avg_launch_us_mean_incl_host_waits": round(mean_t["conv3x3"] / n * 1e6, 1),
                "flop_per_launch": round(w / n),
                # the three down-sampling convs are evaluated as stride-2 convs of the FIR-pre-filtered input (a quarter of
                # the reference conv's multiply-adds): `achieved` counts the reference's algorithmic flops, these two what the
                # launches issue
                "achieved_executed": round(w_exec / t / 1e12, 2), "frac_executed": round(w_exec / t / 1e12 / peak, 4),
                "time_share_per_family_ms_per_step": {
                    k: round(v[1] / n_prof * 1e3, 3) for k, v in sorted(fam.items())}}
    if dist_on:
        # RCCL prints a version banner through C stdio; piped, it would be flushed at process exit, i.e.
        # AFTER rank 0's JSON line.  Every rank flushes its C streams before the last barrier, so the
        # JSON line is the last thing on stdout.
        import ctypes

        ctypes.CDLL(None).fflush(None)
        dist.barrier()

    if calib is not None:
        if roof is not None and K.CONV_PRECISION == "f16x2":
            roof["frac_of_box_ceiling"] = round(roof["achieved"] / calib["three_product_ceiling_tflops"], 4)
            roof["box_ceiling_note"] = ("achieved / (this box's bare-MFMA rate on random operands / 3 products): the "
                                        "fraction of what the matrix pipes of THIS box sustain that the 3x3 family "
                                        "delivers; `frac` stays against the 2.5 PFLOP/s data-sheet peak")
    if rank == 0:
        cpu = None if (args.no_cpu_baseline or world > 1) else cpu_baseline()   # N = 1 only (contract)
        steps_per_s = world * args.steps / dt
        line = {
            "metric": "denoising-steps/sec, nuScenes 32\u00d71024 range image, 1/2/4/8 GPU",   # BASELINE.json, verbatim
            "value": round(steps_per_s, 3),
            "unit": f"denoising-steps/s (each step = batch of {BATCH_PER_GPU} frames per GPU)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "timing": {"protocol": f"{args.repeat} sweeps of exactly {args.steps} steps, each bracketed "
                                   "by barrier + synchronize, MAX over ranks; value = median sweep",
                       "repeat": args.repeat, "timed_region_s": round(sum(sweeps), 4),
                       "sweep_ms_per_step": [round(t / args.steps * 1e3, 4) for t in sweeps],
                       "min_ms_per_step": round(srt[0] / args.steps * 1e3, 4),
                       "max_ms_per_step": round(srt[-1] / args.steps * 1e3, 4),
                       "spread_pct": round((srt[-1] - srt[0]) / dt * 100, 2)},
            "verify": verify,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": ("f16x2-split MFMA, fp32 accumulate (fp32-class accuracy)"
                      if K.CONV_PRECISION == "f16x2" else "f32"), "data": "synthetic",
            "config": {"workload": ("C2" if BATCH_PER_GPU == 8 else f"C2-shape at batch {BATCH_PER_GPU}") +
                                   ": EfficientUNet nuscenes-unet-uncond (31.1M params, seeded "
                                   f"random init), 32x1024, DDIM eta=0, batch {BATCH_PER_GPU} per GPU, "
                                   f"{total}-step schedule (" +
                                   (f"{setup} graph-setup + " if setup else "") +
                                   f"{args.warmup} warmup + {args.repeat} x {args.steps} timed)",
                       "graph_setup_steps": setup,
                       "batch_per_gpu": BATCH_PER_GPU, "global_batch": BATCH_PER_GPU * world,
                       "resolution": list(RES), "parallelism": f"dp{world} (no data-path collective)",
                       "collectives": ("rccl: barrier + all_reduce(MAX) + all_gather of the frames executed" if dist_on
                                       else "none (single process: no process group)"),
                       "rng": "x_T from per-sample CPU generators (global sample index); the timed DDIM steps run with "
                              "rng=None (eta = 0 uses no noise).  The parity mode -- a CPU generator per sample, "
                              "advanced every step as the reference does -- is timed as a row of its own: "
                              "devtools/bench_rows.py uncond_32x1024_cpu_generators (profiles/r04_rows.json)",
                       "sample_steps_per_s": round(steps_per_s * BATCH_PER_GPU, 2),
                       "algorithmic_tflops": round(
                           steps_per_s * BATCH_PER_GPU * GFLOP_PER_SAMPLE_STEP / 1e3, 2)},
            "roofline": roof, "cpu_baseline": cpu,
            "box_calibration": calib, "rows": rows,
        }
        print(json.dumps(line), flush=True)
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
