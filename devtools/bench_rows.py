"""Secondary measurements of SURVEY.md section 8(d) (not the driver's bench line): the C2 shape at
other batch sizes, the layout-conditioned C3 shape, the 64x2048 shape, projection and
points-in-boxes bandwidth.  Prints one JSON object; run on the MI355X box via gpurun:
    python devtools/bench_rows.py [--quick] > gpurun_out/rows.json
Everything is synthetic (seeded random weights / inputs), inputs resident in HBM before timing."""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0
GFLOP = {"uncond32": 116.6, "uncond64": 476.5, "cond32": 255.9, "cond64": 1468.9}   # SURVEY 8(d), per sample-step


def timed(fn, n, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def uncond(dev, B, res, steps, key):
    from lidarcrafter_amd.testing import seeded_fill
    from lidargen.utils import inference
    from lidargen.utils.configs import __all__ as C

    cfg = C["nuscenes-unet-uncond"]()
    if res != (32, 1024):
        cfg.data.resolution = res
    ddpm, model, _ = inference.load_model_duffusion_training(cfg)
    seeded_fill(model, salt=100)
    ddpm = ddpm.eval().to(dev)
    g = [torch.Generator().manual_seed(i) for i in range(B)]
    x_T = torch.stack([torch.randn(*ddpm.sampling_shape, generator=r) for r in g]).to(dev)
    st = ddpm.begin_sampling(B, steps + 3, rng=None, mode="ddim", ddim_eta=0.0, x_T=x_T)
    dt = timed(lambda: ddpm.sampling_step(st), steps, warm=3)
    assert torch.isfinite(st["x"]).all()
    return {"batch": B, "resolution": list(res), "ms_per_step": round(dt * 1e3, 3),
            "steps_per_s": round(1 / dt, 2), "sample_steps_per_s": round(B / dt, 1),
            "algorithmic_tflops": round(B * GFLOP[key] / dt / 1e3, 1),
            "frac_of_f16_mfma_peak": round(B * GFLOP[key] / dt / 1e3 / 2500.0, 4)}


def uncond_autocast(dev, B, steps):
    """The reference's bulk harness runs the sampler under torch.autocast(float16)
    (tools/evaluation/sample_and_save_cond.py:64,145): with LC_AUTOCAST_SINGLE_PRODUCT the convolutions then
    take ONE fp16 product per multiply (liblidarcrafter_hip_p1.so) -- fp16-autocast-class accuracy, NOT the
    fp32-class arithmetic of the benchmarked path.  Reported with the deviation of its frames from the
    three-product run on the same x_T."""
    from lidarcrafter_amd import ops as K
    from lidarcrafter_amd.testing import seeded_fill
    from lidargen.utils import inference
    from lidargen.utils.configs import __all__ as C

    ddpm, model, _ = inference.load_model_duffusion_training(C["nuscenes-unet-uncond"]())
    seeded_fill(model, salt=100)
    ddpm = ddpm.eval().to(dev)
    g = [torch.Generator().manual_seed(i) for i in range(B)]
    x_T = torch.stack([torch.randn(*ddpm.sampling_shape, generator=r) for r in g]).to(dev)
    def run10():
        st_ = ddpm.begin_sampling(B, 10, rng=None, mode="ddim", ddim_eta=0.0, x_T=x_T.clone())
        for _ in range(10):
            x_ = ddpm.sampling_step(st_)
        return x_.clone()

    ref = run10()
    old = K.AUTOCAST_SINGLE_PRODUCT
    K.AUTOCAST_SINGLE_PRODUCT = True
    try:
        with torch.autocast("cuda", dtype=torch.float16):
            assert K.conv_products() == 1
            st = ddpm.begin_sampling(B, steps + 3, rng=None, mode="ddim", ddim_eta=0.0, x_T=x_T)
            dt = timed(lambda: ddpm.sampling_step(st), steps, warm=3)
            assert torch.isfinite(st["x"]).all()
            got = run10()
            dev_frames = float(((got - ref).flatten(1).norm(dim=1) / ref.flatten(1).norm(dim=1)).max())
    finally:
        K.AUTOCAST_SINGLE_PRODUCT = old
    return {"batch": B, "resolution": [32, 1024], "ms_per_step": round(dt * 1e3, 3), "steps_per_s": round(1 / dt, 2),
            "conv_products": 1, "frames_rel_l2_vs_three_products_10_ddim_steps": dev_frames,
            "note": "fp16-autocast caller, single fp16 product per multiply; a precision-reduced row, not the "
                    "headline arithmetic"}


def cond(dev, B, steps):
    from lidarcrafter_amd.testing import seeded_fill, synth_layout_batch
    from lidargen.utils import inference
    from lidargen.utils.configs import __all__ as C

    cfg = C["nuscenes-box-layout-v6"]()
    ddpm, model, _ = inference.load_model_duffusion_training(cfg)
    seeded_fill(model, salt=200), seeded_fill(ddpm.condition_model, salt=201)
    ddpm = ddpm.eval().to(dev)
    batch = {k: v.to(dev) for k, v in synth_layout_batch(B, 32, 1024, seed=53).items()}
    rng = [torch.Generator().manual_seed(i) for i in range(B)]
    with torch.inference_mode():
        x_T = ddpm.randn(B, *ddpm.sampling_shape, rng=rng, device=ddpm.device)
        t_first = None
        for _ in range(3):   # the first call of a process also pays the one-time costs (weight
            torch.cuda.synchronize()   # packing, code-object load, rocBLAS init for the Linears)
            t0 = time.perf_counter()
            cdict = ddpm.get_network_condition(input_dict=batch, only_custom_condition=True)
            torch.cuda.synchronize()
            t_enc = time.perf_counter() - t0
            t_first = t_enc if t_first is None else t_first
        st = ddpm.begin_sampling(B, steps + 3, None, "ddim", 0.0, x_T=x_T, condition_dict=cdict)
        dt = timed(lambda: ddpm.sampling_step(st), steps, warm=3)
        x = st["x"]
    assert torch.isfinite(x).all()
    return {"batch": B, "resolution": [32, 1024], "ms_per_step": round(dt * 1e3, 3),
            "steps_per_s": round(1 / dt, 2), "sample_steps_per_s": round(B / dt, 1),
            "algorithmic_tflops": round(B * GFLOP["cond32"] / dt / 1e3, 1),
            "frac_of_f16_mfma_peak": round(B * GFLOP["cond32"] / dt / 1e3 / 2500.0, 4),
            "layout_encoder_ms_once_per_batch": round(t_enc * 1e3, 2),
            "layout_encoder_first_call_ms": round(t_first * 1e3, 2)}


def _cond_pair_64(cond_out):
    """The full-width architecture of nuscenes-box-layout-v6 (cond_out 10) / nuscenes-auto-reg-v2 (11 condition
    channels) at 64 x 2048 -- the constructor arguments of those configs with the C4 resolution."""
    from lidarcrafter_amd.testing import seeded_fill
    from lidargen.models.unets import __all__ as U
    from lidargen.utils.lidar import get_linear_ray_angles

    res = (64, 2048)
    m = U["layout_unet_v1"](
        in_channels=2 + cond_out, resolution=res, image_size=64, use_fp16=False, use_scale_shift_norm=True,
        out_channels=2, model_channels=64, encoder_channels=64, num_head_channels=32, num_heads=-1,
        num_heads_upsample=-1, num_res_blocks=2, num_attention_blocks=1, resblock_updown=True, attention_ds=[4, 8],
        channel_mult=[1, 2, 4, 8], dropout=0.1, use_checkpoint=False, use_positional_embedding_for_attention=True,
        attention_block_type="ObjectAwareCrossAttention")
    m.coords = get_linear_ray_angles(res[0], res[1], 10.0, -30.0)
    enc = U["layout_encoder"](
        feature_map_size=list(res), used_condition_types=["obj_class", "obj_bbox", "is_valid_obj"], layout_length=13,
        num_classes_for_layout_object=9, mask_size_for_layout_object=32, hidden_dim=64, output_dim=256, num_layers=6,
        num_heads=4, use_final_ln=True, use_positional_embedding=False, not_use_layout_fusion_module=False,
        resolution_to_attention=[4, 8], use_key_padding_mask=False, out_channels=cond_out)
    return seeded_fill(m, salt=200).eval(), seeded_fill(enc, salt=201).eval()


def cond64(dev, B, steps):
    """The denoising step of config C4's frames 1..4: nuscenes-auto-reg-v2 architecture (LayoutUnetV1, 11 condition
    channels, object-aware cross attention over 8192 + 13 and 2048 + 13 keys) at 64 x 2048, DDPM."""
    from lidarcrafter_amd.testing import synth_layout_batch
    from lidargen.models.diffusion import CondContinuousTimeGaussianDiffusion

    m, enc = _cond_pair_64(11)
    ddpm = CondContinuousTimeGaussianDiffusion(m, enc, cond_mode="concat").eval().to(dev)
    batch = {k: v.to(dev) for k, v in synth_layout_batch(B, 64, 2048, seed=91, n_extra=1).items()}
    rng = [torch.Generator().manual_seed(i) for i in range(B)]
    with torch.inference_mode():
        x_T = ddpm.randn(B, *ddpm.sampling_shape, rng=rng, device=ddpm.device)
        cdict = ddpm.get_network_condition(input_dict=batch, only_custom_condition=True)
        st = ddpm.begin_sampling(B, steps + 3, None, "ddpm", 0.0, x_T=x_T, condition_dict=cdict)
        dt = timed(lambda: ddpm.sampling_step(st), steps, warm=3)
        x = st["x"]
    assert torch.isfinite(x).all()
    return {"batch": B, "resolution": [64, 2048], "mode": "ddpm (device noise)", "ms_per_step": round(dt * 1e3, 3),
            "steps_per_s": round(1 / dt, 2), "sample_steps_per_s": round(B / dt, 1),
            "algorithmic_tflops": round(B * GFLOP["cond64"] / dt / 1e3, 1),
            "frac_of_f16_mfma_peak": round(B * GFLOP["cond64"] / dt / 1e3 / 2500.0, 4)}


def sequence64(dev, frames, steps):
    """Config C4 itself at B = 1: frame 0 with the box-layout-v6 architecture, frames 1.. with auto-reg-v2, both at
    64 x 2048, DDPM with the per-sample CPU generator (the parity mode), temporal glue on the device."""
    import numpy as np
    from lidarcrafter_amd.testing import synth_scene_boxes, synth_temporal_inputs
    from lidargen.dataset.custom_dataset import CustomDataset, DataConfig
    from lidargen.models.diffusion import CondContinuousTimeGaussianDiffusion
    from lidargen.utils import temporal
    from lidargen.utils.lidar import LiDARUtility

    H, W = 64, 2048
    m0, e0 = _cond_pair_64(10)
    m1, e1 = _cond_pair_64(11)
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
        fr, _ = temporal.generate_sequence(ddpm, auto, lu, dict(batch), num_frames=nf, num_steps=ns, mode="ddpm",
                                           traj_length=16, rng=[torch.Generator().manual_seed(90)], data_cfg=Cfg())
        torch.cuda.synchronize()
        assert all(torch.isfinite(f).all() for f in fr)
        return time.perf_counter() - t0

    run(2, 4)
    t_full = run(frames, steps)
    t_glue = run(frames, 3)
    per_step = (t_full - t_glue) / (frames * (steps - 3))
    return {"batch": 1, "frames": frames, "resolution": [H, W], "steps_per_frame": steps, "mode": "ddpm, CPU generator",
            "seconds": round(t_full, 3), "ms_per_denoising_step": round(per_step * 1e3, 3),
            "algorithmic_tflops": round(GFLOP["cond64"] / per_step / 1e3, 1),
            "frac_of_f16_mfma_peak": round(GFLOP["cond64"] / per_step / 1e3 / 2500.0, 4),
            "glue_ms_per_frame_upper_bound": round((t_glue - 3 * frames * per_step) / frames * 1e3, 2)}


def uncond_generators(dev, B, steps, mode):
    """C2 with the per-sample CPU-generator list every parity test uses (bench.py times rng=None): the draws of a step
    are made on the host, staged in pinned memory and copied once per step (DDPM; DDIM eta = 0 only ADVANCES the
    generators, as the reference does)."""
    from lidarcrafter_amd.testing import seeded_fill
    from lidargen.utils import inference
    from lidargen.utils.configs import __all__ as C

    ddpm, model, _ = inference.load_model_duffusion_training(C["nuscenes-unet-uncond"]())
    seeded_fill(model, salt=100)
    ddpm = ddpm.eval().to(dev)
    rng = [torch.Generator().manual_seed(i) for i in range(B)]
    st = ddpm.begin_sampling(B, steps + 3, rng=rng, mode=mode, ddim_eta=0.0)
    dt = timed(lambda: ddpm.sampling_step(st), steps, warm=3)
    assert torch.isfinite(st["x"]).all()
    return {"batch": B, "mode": mode, "rng": "one CPU torch.Generator per sample (parity mode)",
            "ms_per_step": round(dt * 1e3, 3), "steps_per_s": round(1 / dt, 2)}


def sequence(dev, B, frames, steps):
    """C4-shaped run at 32x1024: frame 0 with nuscenes-box-layout-v6, frames 1.. with
    nuscenes-auto-reg-v2 (DDPM), the temporal glue resident on the device."""
    import numpy as np
    from lidarcrafter_amd.testing import seeded_fill, synth_scene_boxes, synth_temporal_inputs
    from lidargen.dataset.custom_dataset import CustomDataset, DataConfig
    from lidargen.utils import inference, temporal
    from lidargen.utils.configs import __all__ as C

    def build(name, s0):
        ddpm, model, lu = inference.load_model_duffusion_training(C[name]())
        seeded_fill(model, salt=s0), seeded_fill(ddpm.condition_model, salt=s0 + 1)
        return ddpm.eval().to(dev), lu.to(dev)

    ddpm, lu = build("nuscenes-box-layout-v6", 200)
    auto, _ = build("nuscenes-auto-reg-v2", 300)
    K_ = 6
    infos = []
    for b in range(B):
        sb = synth_scene_boxes(K_, seed=40 + b)
        names = ["ego"] + [DataConfig.class_names[int(c) - 1] for c in sb[:, 7]]
        infos.append(dict(gt_boxes=np.concatenate([np.zeros((1, 7)), sb[:, :7].astype(np.float64)]),
                          gt_names=names, gt_fut_trajs=synth_temporal_inputs(50 + b, K=K_)[0]))
    ds = CustomDataset([dict(d) for d in infos])
    batch = ds.collate_fn([ds[i] for i in range(B)])
    batch["gt_fut_trajs"] = [d["gt_fut_trajs"] for d in infos]

    def run(nf, ns):
        rng = [torch.Generator().manual_seed(90 + i) for i in range(B)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fr, _ = temporal.generate_sequence(ddpm, auto, lu, dict(batch), num_frames=nf, num_steps=ns,
                                           mode="ddpm", traj_length=16, rng=rng)
        torch.cuda.synchronize()
        assert all(torch.isfinite(f).all() for f in fr)
        return time.perf_counter() - t0

    run(2, 4)                                        # warm (weight packs, graphs, allocator)
    t_full = run(frames, steps)
    t_glue = run(frames, 3) - 0.0                    # 3 steps per frame: mostly glue + launch cost
    per_step = (t_full - t_glue) / (frames * (steps - 3))
    return {"batch": B, "frames": frames, "steps_per_frame": steps, "mode": "ddpm",
            "seconds": round(t_full, 3), "ms_per_denoising_step": round(per_step * 1e3, 3),
            "glue_ms_per_frame_upper_bound": round((t_glue - 3 * frames * per_step) / frames * 1e3, 2),
            "note": "glue = projection, box rasterisation, transforms, points-in-boxes, compaction, "
                    "CustomDataset item + collate, RNG draws on the host for DDPM noise"}


def pipeline_metrics(dev, B, n_batches, steps):
    """C5-shaped flow on one GPU: sample two sets of frames (different seeds), post-process to
    points, BEV histograms, JSD / MMD between the sets and the chamfer distance of one pair --
    everything on the device (random-init weights: the VALUES are meaningless, the flow is real)."""
    from lidarcrafter_amd import ops as K
    from lidarcrafter_amd.testing import seeded_fill
    from lidargen.metrics import bev, chamfer
    from lidargen.utils import inference
    from lidargen.utils.configs import __all__ as C

    ddpm, model, lu = inference.load_model_duffusion_training(C["nuscenes-unet-uncond"]())
    seeded_fill(model, salt=100)
    ddpm, lu = ddpm.eval().to(dev), lu.to(dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sets, clouds = [], []
    for s in range(2):
        hs = []
        for k in range(n_batches):
            rng = [torch.Generator().manual_seed(1000 * s + k * B + i) for i in range(B)]
            fr = lu.postprocess(ddpm.sample(B, steps, progress=False, rng=rng, mode="ddim").clamp(-1, 1))
            for b in range(B):
                pts, keep = K.image_to_points(fr[b, 1:4].contiguous(), fr[b, 4].contiguous(), None, 1.0,
                                              min_norm=1e-2)
                pts = K.compact_points(pts, keep)
                hs.append(bev.point_cloud_to_histogram(pts))
                if b == 0 and k == 0:
                    clouds.append(pts[:, :3].contiguous())
        sets.append(torch.stack(hs))
    torch.cuda.synchronize()
    t_gen = time.perf_counter() - t0
    t0 = time.perf_counter()
    jsd, mmd = bev.compute_jsd_2d(sets[0], sets[1]), bev.compute_mmd_2d(sets[0], sets[1])
    cd = chamfer.compute_pairwise_cd(clouds[0], clouds[1])
    torch.cuda.synchronize()
    return {"frames_per_set": B * n_batches, "ddim_steps": steps, "generate_and_histogram_s": round(t_gen, 3),
            "metrics_s": round(time.perf_counter() - t0, 4), "bev_jsd": float(jsd), "bev_mmd": float(mmd),
            "chamfer": float(cd)}


def projection(dev, N):
    from lidarcrafter_amd import ops as K
    from lidarcrafter_amd.testing import synth_points

    pts = torch.from_numpy(synth_points(N, seed=3)).to(dev)
    H, W = 32, 1024
    dt = timed(lambda: K.project_points(pts, H, W, 10.0, -30.0, 1.45, 80.0), 50, warm=5)
    algo = 16 * N + 8 * N + 24 * H * W
    return {"points": N, "us": round(dt * 1e6, 1), "algorithmic_bytes": algo,
            "GBps": round(algo / dt / 1e9, 1), "frac_of_hbm_peak": round(algo / dt / 1e9 / HBM_PEAK_GBS, 4),
            "note": "two launches (scatter, gather that re-empties the cached z-buffer) + the image / winner allocations"}


def pib(dev, N, nbox):
    from lidarcrafter_amd import ops as K
    from lidarcrafter_amd.testing import synth_boxes, synth_points

    p = synth_points(N, seed=4)
    pts = torch.from_numpy(p[:, :3].copy()).to(dev)
    boxes = torch.from_numpy(synth_boxes(nbox, p, seed=5)).to(dev)
    dt = timed(lambda: K.points_in_boxes_mask(pts, boxes, 1e-2), 50, warm=5)
    algo = 12 * N + 4 * N * nbox
    return {"points": N, "boxes": nbox, "us": round(dt * 1e6, 1), "algorithmic_bytes": algo,
            "GBps": round(algo / dt / 1e9, 1), "frac_of_hbm_peak": round(algo / dt / 1e9 / HBM_PEAK_GBS, 4)}


def train_step(dev, B):
    """One training step of the C2 denoiser through the HIP autograd Functions (forward + backward +
    AdamW), tools/train/train_lidm.py:214-265 at batch B."""
    from lidarcrafter_amd import autograd as AGm
    from lidarcrafter_amd.testing import seeded_fill
    from lidargen.utils import inference
    from lidargen.utils.configs import __all__ as C

    ddpm, model, _ = inference.load_model_duffusion_training(C["nuscenes-unet-uncond"]())
    seeded_fill(model, salt=100)
    ddpm = ddpm.train().to(dev)
    opt = torch.optim.AdamW(ddpm.parameters(), lr=1e-4)
    x0 = torch.randn(B, 2, 32, 1024, device=dev).clamp(-1, 1)

    def step():
        opt.zero_grad(set_to_none=True)
        loss = ddpm(x0)
        loss.backward()
        opt.step()

    dt = timed(step, 5, warm=2)
    flop = 3 * B * GFLOP["uncond32"] * 1e9                       # forward + dX + dW
    return {"batch": B, "ms_per_step": round(dt * 1e3, 2), "samples_per_s": round(B / dt, 2),
            "algorithmic_tflops": round(flop / dt / 1e12, 1),
            "train_conv_precision": AGm.TRAIN_CONV_PRECISION, "train_wgrad_precision": AGm.TRAIN_WGRAD_PRECISION,
            "note": "forward + backward + AdamW; forward / dX convolutions in the precision named by "
                    "train_conv_precision (lidarcrafter_amd.autograd.TRAIN_CONV_PRECISION, env "
                    "LC_TRAIN_CONV_PRECISION; default f16x2 split), weight gradient by train_wgrad_precision "
                    "(LC_TRAIN_WGRAD_PRECISION; default f16x2 split, exact fp32 where the shape has no whole "
                    "2 x 32 pixel tiles)"}


def train_step_cond(dev, B):
    """One training step of the layout-conditioned denoiser + layout encoder (box-layout-v6, 70 M
    parameters), tools/train/train_lidm_cond.py:259-322 at batch B: ddpm(batch) -> backward -> AdamW."""
    from lidarcrafter_amd.testing import seeded_fill, synth_layout_batch
    from lidargen.utils import inference
    from lidargen.utils.configs import __all__ as C

    from lidarcrafter_amd import autograd as AGm
    ddpm, model, _ = inference.load_model_duffusion_training(C["nuscenes-box-layout-v6"]())
    seeded_fill(model, salt=200), seeded_fill(ddpm.condition_model, salt=201)
    ddpm = ddpm.train().to(dev)
    opt = torch.optim.AdamW(ddpm.parameters(), lr=1e-4)
    batch = {k: v.to(dev) for k, v in synth_layout_batch(B, 32, 1024, seed=53).items()}
    batch["x_0"] = torch.randn(B, 2, 32, 1024, device=dev).clamp(-1, 1)

    def step():
        opt.zero_grad(set_to_none=True)
        loss = ddpm(batch)
        loss.backward()
        opt.step()

    dt = timed(step, 5, warm=2)
    flop = 3 * B * GFLOP["cond32"] * 1e9
    return {"batch": B, "ms_per_step": round(dt * 1e3, 2), "samples_per_s": round(B / dt, 2),
            "algorithmic_tflops": round(flop / dt / 1e12, 1),
            "train_conv_precision": AGm.TRAIN_CONV_PRECISION, "train_wgrad_precision": AGm.TRAIN_WGRAD_PRECISION,
            "note": "denoiser + layout encoder, forward + backward + AdamW, dropout as configured; convolutions "
                    "as in train_step_c2; attention core = autograd.FlashAttention (HIP flash forward + f16x2-split flash backward; LC_TRAIN_ATTENTION, LC_TRAIN_ATTN_{FWD,BWD}_PRECISION), small dense layers are torch ops on the device"}


def object_branch(dev, n_obj, steps):
    """Foreground-object branch: `steps` DDPM steps of PointUNet over n_obj x [1024, 4] point sets."""
    from lidarcrafter_amd.testing import seeded_fill, synth_object_batch, synth_text_features
    from lidargen.utils import inference
    from lidargen.utils.configs import __all__ as C

    ddpm, model = inference.load_model_object_duffusion_training(C["nuscenes-object"]())
    seeded_fill(model, salt=300), seeded_fill(ddpm.condition_model, salt=301)
    ddpm = ddpm.eval().to(dev)
    ddpm.condition_model.set_text_features(synth_text_features(), dev)
    batch = {k: v.to(dev) for k, v in synth_object_batch(n_obj, seed=95).items()}
    ddpm.sample(batch, n_obj, 8, progress=False, mode="ddpm")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    x = ddpm.sample(batch, n_obj, steps, progress=False, mode="ddpm")
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert torch.isfinite(x).all()
    return {"objects": n_obj, "steps": steps, "seconds": round(dt, 3), "us_per_step": round(dt / steps * 1e6, 1)}


def voxel_scatter(dev, n_sweeps, N):
    """pcd2bev_sum ('32': 1200 x 1200 voxels of 5 cm) over n_sweeps sweeps of N points, and
    sparse_quantize of one 4 M-point cloud."""
    from lidarcrafter_amd import ops as K
    from lidarcrafter_amd.testing import synth_points
    from lidargen.metrics import metric_utils as M

    sweeps = [torch.from_numpy(synth_points(N, seed=i)).to(dev) for i in range(n_sweeps)]
    dt = timed(lambda: M.pcd2bev_sum("32", sweeps), 5, warm=1)
    big = torch.from_numpy(synth_points(1 << 22, seed=99)[:, :3].copy()).to(dev)
    dq = timed(lambda: K.sparse_quantize(big, 0.1, return_index=True, return_inverse=True), 5, warm=1)
    return {"pcd2bev_sum": {"sweeps": n_sweeps, "points_per_sweep": N, "ms": round(dt * 1e3, 3),
                            "Mpoints_per_s": round(n_sweeps * N / dt / 1e6, 1)},
            "sparse_quantize": {"points": 1 << 22, "ms": round(dq * 1e3, 3),
                                "Mpoints_per_s": round((1 << 22) / dq / 1e6, 1)}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--only", default="", help="comma-separated row names (default: all)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    out = {"device": torch.cuda.get_device_name(0)}
    q = args.quick
    rows = {
        "uncond_32x1024": lambda: [uncond(dev, B, (32, 1024), 20 if B <= 8 else 8, "uncond32")
                                   for B in ((1, 8) if q else (1, 2, 8, 32))],
        "cond_layout_v6_32x1024": lambda: [cond(dev, B, 10) for B in ((8,) if q else (1, 8))],
        "uncond_64x2048": lambda: [] if q else [uncond(dev, 4, (64, 2048), 6, "uncond64")],
        "cond_autoreg_v2_64x2048": lambda: [cond64(dev, B, 6) for B in ((1,) if q else (1, 2))],
        "temporal_sequence_c4_64x2048": lambda: [sequence64(dev, 5, 8 if q else 16)],
        "uncond_32x1024_cpu_generators": lambda: [uncond_generators(dev, 8, 20, m) for m in ("ddim", "ddpm")],
        "uncond_32x1024_fp16_autocast_single_product": lambda: [uncond_autocast(dev, 8, 20)],
        "temporal_sequence_32x1024": lambda: [sequence(dev, 2, 5, 16 if q else 32)],
        "pipeline_metrics_c5_shape": lambda: [pipeline_metrics(dev, 8, 1 if q else 2, 8)],
        "train_step_c2": lambda: [train_step(dev, B) for B in ((2,) if q else (2, 8))],
        "train_step_c3": lambda: [train_step_cond(dev, B) for B in ((2,) if q else (2, 8))],
        "object_branch": lambda: [object_branch(dev, 10, 256)],
        "voxel_scatter": lambda: [voxel_scatter(dev, 16, 34720)],
        "projection": lambda: [projection(dev, N) for N in (34720, 131072, 1 << 22)],
        "points_in_boxes_mask": lambda: [pib(dev, N, nb) for N, nb in ((34720, 13), (1 << 22, 13))],
    }
    only = [r for r in args.only.split(",") if r]
    for name, fn in rows.items():
        if only and name not in only:
            continue
        res = fn()
        if res:
            out[name] = res
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
