"""Generate golden vectors by running the REFERENCE (imported read-only from /root/reference).

Run in the build container only:   python tests/golden/make_fixtures.py [section ...]
Outputs tests/golden/*.npz (committed).  Inputs are regenerated from seeds by the tests
(lidarcrafter_amd.testing.seeded_randn / seeded_fill), so the files hold mostly OUTPUTS.
Nothing here travels to the GPU box except the .npz data files.
"""
import os
import sys
import warnings

import numpy as np
import torch

warnings.filterwarnings("ignore")
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import _ref_import as R  # noqa: E402

R.install()
from lidarcrafter_amd.testing import seeded_fill, seeded_randn, synth_points  # noqa: E402

torch.set_num_threads(int(os.environ.get("LC_FIXTURE_THREADS", "8")))
torch.manual_seed(0)


def save(name, **arrays):
    out = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}.npz  {os.path.getsize(path)/1024:.1f} KiB  keys={list(out)}")


# ------------------------------------------------------------------------------------------
def sec_ops():
    ops = R.ref("models.unets.ops")
    enc = R.ref("models.unets.encoding")
    lidar = R.ref("utils.lidar")
    eu = R.ref("models.unets.efficient_unet")
    out = {}
    with torch.no_grad():
        # ring conv 3x3 and plain 1x1 (ops.py:149-173)
        c3 = seeded_fill(ops.Conv2d(5, 7, 3, 1, 1, ring=True), salt=1)
        x = seeded_randn(2, 5, 4, 16, seed=11)
        out["conv3_y"] = c3(x)
        c1 = seeded_fill(ops.Conv2d(5, 7, 1, 1, 0), salt=2)
        out["conv1_y"] = c1(x)
        # resample (ops.py:52-146)
        x = seeded_randn(2, 3, 4, 16, seed=12)
        out["down_y"] = ops.Resample(down=2, ring=True)(x)
        out["up_y"] = ops.Resample(up=2, ring=True)(x)
        # GN + AdaGN (ops.py:176-200)
        x = seeded_randn(2, 16, 4, 8, seed=13)
        gn = seeded_fill(torch.nn.GroupNorm(8, 16, 1e-6), salt=3)
        out["gn_y"] = gn(x)
        ada = seeded_fill(ops.AdaGN(32, 16, 8, 1e-6), salt=4)
        emb = seeded_randn(2, 32, seed=14)
        out["adagn_y"] = ada(x, emb)
        # sinusoid of log-SNR values (ops.py:14-29)
        lam = torch.tensor([-15.0, -3.25, 0.0, 7.5, 15.0])
        out["sin_y"] = ops.SinusoidalPositionalEmbedding(64)(lam)
        # Fourier features on linear ray angles (encoding.py:120-146, lidar.py:22-32)
        coords = lidar.get_linear_ray_angles(8, 64, 10.0, -30.0)
        out["coords_8x64"] = coords
        out["fourier_8x64"] = enc.FourierFeatures((8, 64))(coords)
        # blocks (efficient_unet.py:28-115)
        x = seeded_randn(2, 32, 4, 8, seed=15)
        temb = seeded_randn(2, 64, seed=16)
        rb = seeded_fill(eu.ResidualBlock(32, 32, 64, 8, 1e-6, ring=True), salt=5)
        out["rb_same_y"] = rb(x, temb)
        rb2 = seeded_fill(eu.ResidualBlock(32, 16, 64, 8, 1e-6, ring=True), salt=6)
        out["rb_skip_y"] = rb2(x, temb)
        sa = seeded_fill(eu.SelfAttentionBlock(32, 4, 1e-6, 8), salt=7).eval()
        out["sa_y"] = sa(x)
    save("ops", **out)


def sec_fold_down():
    """Block.downsample of the reference's EfficientUNet (efficient_unet.py:132-135): ops.Conv2d(3x3, ring) followed by
    ops.Resample(down=2) -- the pair the HIP path evaluates as one stride-2 conv behind a FIR pre-filter
    (lidarcrafter_amd/csrc/conv_f16x2_s2.hip).  Shapes: the image's top and bottom rows in one tile (H = 4), whole and
    ragged 64-channel output blocks, an input with a mean (the bias / border factor 7/8 shows)."""
    ops = R.ref("models.unets.ops")
    out = {}
    with torch.no_grad():
        for tag, (B, Ci, Co, H, W, salt) in {"a": (2, 16, 32, 8, 128, 31), "b": (1, 32, 72, 4, 256, 32),
                                             "c": (1, 64, 128, 16, 128, 33)}.items():
            conv = seeded_fill(ops.Conv2d(Ci, Co, 3, 1, 1, ring=True), salt=salt)
            x = seeded_randn(B, Ci, H, W, seed=300 + salt) + 0.3
            out[f"{tag}_y"] = ops.Resample(down=2, ring=True)(conv(x))
    save("fold_down", **out)


def sec_fold_up():
    """Block.upsample of the reference's EfficientUNet (efficient_unet.py:143-145) = the inner pair of LayoutUnetV1's
    up-sampling ResBlock (layout_unet_v1.py:219-235): ops.Resample(up=2) followed by ops.Conv2d(3x3, ring) -- the pair the
    HIP path evaluates at the LOW resolution (lidarcrafter_amd/csrc/upfold.hip: nine 1x1 tap planes + one combine pass).
    Shapes: one and several 128-column segments, H = 1 (both H borders in one row pair), Ci != Co, an input with a mean
    (the bias and the dropped border taps show).  Every second column is stored."""
    ops = R.ref("models.unets.ops")
    out = {}
    with torch.no_grad():
        for tag, (B, Ci, Co, H, W, salt) in {"a": (2, 32, 32, 4, 128, 41), "b": (1, 64, 40, 1, 256, 42),
                                             "c": (1, 128, 128, 8, 128, 43), "d": (2, 32, 16, 2, 384, 44)}.items():
            conv = seeded_fill(ops.Conv2d(Ci, Co, 3, 1, 1, ring=True), salt=salt)
            x = seeded_randn(B, Ci, H, W, seed=400 + salt) + 0.3
            out[f"{tag}_y"] = conv(ops.Resample(up=2, ring=True)(x))[..., ::2]
    save("fold_up", **out)


def _build_uncond(eu, base, res):
    m = eu.EfficientUNet(2, res, base_channels=base, temb_channels=None,
                         channel_multiplier=(1, 2, 4, 8), num_residual_blocks=(3, 3, 3, 3),
                         gn_num_groups=8, gn_eps=1e-6, attn_num_heads=8,
                         coords_encoding="fourier_features", ring=True)
    lidar = R.ref("utils.lidar")
    m.coords = lidar.get_linear_ray_angles(res[0], res[1], 10.0, -30.0)  # inference.py:281-282
    return seeded_fill(m, salt=100).eval()


def sec_unet_small():
    eu = R.ref("models.unets.efficient_unet")
    m = _build_uncond(eu, 16, (8, 64))
    x = seeded_randn(2, 2, 8, 64, seed=21)
    lam = torch.tensor([-4.0, 2.5])
    with torch.no_grad():
        y = m(x, lam)
    save("unet_small", y=y, nkeys=len(m.state_dict()))


def sec_unet_full():
    eu = R.ref("models.unets.efficient_unet")
    m = _build_uncond(eu, 64, (32, 1024))
    x = seeded_randn(1, 2, 32, 1024, seed=22)
    lam = torch.tensor([-1.5])
    with torch.no_grad():
        y = m(x, lam)
    keys = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    save("unet_full", y=y, keys=np.array(sorted(f"{k}:{s}" for k, s in keys.items())))


def sec_diffusion():
    df = R.ref("models.diffusion")
    ct = R.ref("models.diffusion.continuous_time")

    class Stub(torch.nn.Module):
        resolution = (4, 16)
        in_channels = 2

        def __init__(self):
            super().__init__()
            self.pred = None

        def forward(self, x, c):
            return self.pred.clone()  # the reference clamps x_0 (== prediction) in place

    out = {}
    for S in (10, 50):
        t = torch.linspace(1.0, 0.0, S + 1)
        lam = ct._log_snr_schedule_cosine(t)[:, 0, 0, 0]
        a, s = ct._log_snr_to_alpha_sigma(lam)
        out[f"lam_{S}"], out[f"alpha_{S}"], out[f"sigma_{S}"] = lam, a, s
    out["lam_linear_10"] = ct._log_snr_schedule_linear(torch.linspace(1.0, 0.0, 11))[:, 0, 0, 0]
    x_t = seeded_randn(3, 2, 4, 16, seed=31)
    pred = seeded_randn(3, 2, 4, 16, seed=32)
    step_t = torch.tensor([1.0, 0.6, 0.1])
    step_s = torch.tensor([0.9, 0.5, 0.0])
    for obj in ("eps", "v", "x_0"):
        stub = Stub()
        stub.pred = pred
        ddpm = df.ContinuousTimeGaussianDiffusion(stub, torch.nn.Identity(), prediction_type=obj)
        for mode, eta in (("ddpm", 0.0), ("ddim", 0.0), ("ddim", 0.5)):
            rng = [torch.Generator().manual_seed(100 + i) for i in range(3)]
            out[f"pstep_{obj}_{mode}_{eta}"] = ddpm.p_step(x_t, step_t, step_s, rng=rng,
                                                           mode=mode, ddim_eta=eta)
    # discrete tables + p_step (discrete_time.py)
    for kind in ("linear", "cosine", "sigmoid"):
        stub = Stub()
        stub.pred = pred
        dd = df.DiscreteTimeGaussianDiffusion(stub, None, num_training_steps=50,
                                              noise_schedule=kind)
        out[f"disc_{kind}_beta"] = dd.beta[:, 0, 0, 0]
        out[f"disc_{kind}_abar"] = dd.alpha_bar[:, 0, 0, 0]
        steps = torch.tensor([49, 20, 0])
        for mode in ("ddpm", "ddim"):
            rng = [torch.Generator().manual_seed(200 + i) for i in range(3)]
            out[f"disc_{kind}_{mode}"] = dd.p_step(x_t, steps, rng=rng, mode=mode)
    save("diffusion", **out)


def sec_trajectory():
    """C1: 32x1024, DDIM 10 steps, B=1, seeded weights, per-sample CPU generator;
    plus the same loop on the reduced model with all states kept."""
    eu = R.ref("models.unets.efficient_unet")
    df = R.ref("models.diffusion")
    m = _build_uncond(eu, 16, (8, 64))
    ddpm = df.ContinuousTimeGaussianDiffusion(m, torch.nn.Identity()).eval()
    out = {}
    for mode in ("ddim", "ddpm"):
        rng = [torch.Generator().manual_seed(i) for i in range(2)]
        out[f"small_{mode}"] = ddpm.sample(2, 10, progress=False, rng=rng, return_all=True, mode=mode)
    m = _build_uncond(eu, 64, (32, 1024))
    ddpm = df.ContinuousTimeGaussianDiffusion(m, torch.nn.Identity()).eval()
    rng = [torch.Generator().manual_seed(0)]
    xs = ddpm.sample(1, 10, progress=False, rng=rng, return_all=True, mode="ddim")
    out["c1_x1"], out["c1_x2"], out["c1_x10"] = xs[1], xs[2], xs[10]
    out["c1_means"] = xs.flatten(1).mean(1)
    out["c1_abs_means"] = xs.flatten(1).abs().mean(1)
    save("trajectory", **out)


def sec_lidar():
    lidar = R.ref("utils.lidar")
    cm = R.ref("dataset.transforms_3d.common")
    out = {}
    ang = lidar.get_linear_ray_angles(8, 64, 10.0, -30.0)
    lu = lidar.LiDARUtility((8, 64), "log_depth", 1.45, 80.0, ray_angles=ang)
    metric = seeded_randn(2, 1, 8, 64, seed=41).abs() * 40
    out["convert_depth"] = lu.convert_depth(metric)
    nz = torch.rand(2, 1, 8, 64, generator=torch.Generator().manual_seed(42))
    out["revert_depth"] = lu.revert_depth(nz)
    out["to_xyz"] = lu.to_xyz(metric)
    # projection on a seeded synthetic sweep: azimuth U(-pi,pi), elevation
    # U(-30.5,10.5) deg, range log-U(0.8,95) m, intensity U(0,255)
    for tag, N, H, W, seed in (("a", 4096, 16, 256, 0), ("b", 34720, 32, 1024, 1)):
        pts = synth_points(N, seed)
        img = cm.load_points_as_images(points=pts, scan_unfolding=False, H=H, W=W,
                                       min_depth=1.45, max_depth=80.0, fov_up=10.0, fov_down=-30.0)
        out[f"proj_{tag}_depth"] = img[..., 4]
        out[f"proj_{tag}_mask"] = img[..., 5].astype(np.uint8)
        out[f"proj_{tag}_x"] = img[..., 0]
        out[f"proj_{tag}_i"] = img[..., 3]
    save("lidar", **out)


def synth_layout_batch(B, H, W, seed, n_extra=0):
    """Synthetic layout condition batch."""
    from lidarcrafter_amd.testing import synth_layout_batch as f
    return f(B, H, W, seed, n_extra)


def _build_cond(res, image_size, model_channels, cond_out=10, unet_kw=None, enc_kw=None):
    lu = R.ref("models.unets.layout_unet_v1")
    le = R.ref("models.unets.layout_encoder")
    lidar = R.ref("utils.lidar")
    import io, contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        m = lu.LayoutUnetV1(in_channels=2 + cond_out, resolution=res, image_size=image_size,
                            use_fp16=False, use_scale_shift_norm=True, out_channels=2,
                            model_channels=model_channels, encoder_channels=64,
                            num_head_channels=32, num_heads=-1, num_heads_upsample=-1,
                            num_res_blocks=2, num_attention_blocks=1, resblock_updown=True,
                            attention_ds=[4, 8], channel_mult=[1, 2, 4, 8], dropout=0.1,
                            use_checkpoint=False, use_positional_embedding_for_attention=True,
                            attention_block_type="ObjectAwareCrossAttention", **(unet_kw or {}))
    m.coords = lidar.get_linear_ray_angles(res[0], res[1], 10.0, -30.0)
    enc = le.LayoutTransformerEncoder(
        feature_map_size=list(res), used_condition_types=["obj_class", "obj_bbox", "is_valid_obj"],
        layout_length=13, num_classes_for_layout_object=9, mask_size_for_layout_object=32,
        hidden_dim=64, output_dim=model_channels * 4, num_layers=6, num_heads=4, use_final_ln=True,
        use_positional_embedding=False, not_use_layout_fusion_module=False,
        resolution_to_attention=[4, 8], out_channels=cond_out,
        **{"use_key_padding_mask": False, **(enc_kw or {})})
    return seeded_fill(m, salt=200).eval(), seeded_fill(enc, salt=201).eval()


def sec_cond_small():
    m, enc = _build_cond((8, 64), 8, 32)
    batch = synth_layout_batch(2, 8, 64, seed=51)
    x = seeded_randn(2, 2, 8, 64, seed=52)
    lam = torch.tensor([-3.0, 1.5])
    with torch.no_grad():
        cond = enc(batch)
        y = m(x, {"time_condition": lam, "other_condition": cond})
    out = {"y": y}
    for k in ("xf_proj", "xf_out", "obj_class_embedding", "obj_bbox_embedding",
              "image_patch_bbox_embedding_for_resolution2",
              "image_patch_bbox_embedding_for_resolution1"):
        out["cond_" + k] = cond[k]
    out["keys_unet"] = np.array(sorted(f"{k}:{tuple(v.shape)}" for k, v in m.state_dict().items()))
    out["keys_enc"] = np.array(sorted(f"{k}:{tuple(v.shape)}" for k, v in enc.state_dict().items()))
    save("cond_small", **out)


# The ObjectAwareCrossAttention constructor options no shipped configuration sets (layout_unet_v1.py:367-376, 402-412)
COND_OPTION_VARIANTS = {
    "nf": ({"norm_first": True, "norm_for_obj_embedding": True, "channels_scale_for_positional_embedding": 0.5,
            "use_key_padding_mask": True}, {"use_key_padding_mask": True}),
    "mask": ({"use_key_padding_mask": True}, {}),
    "scale": ({"channels_scale_for_positional_embedding": 0.5}, {}),
}


def sec_cond_options():
    """Reduced LayoutUnetV1 + layout encoder with the unshipped attention options switched on, same seeded weights /
    inputs as cond_small (the reference prints the padding mask in this path: stdout is swallowed)."""
    import io, contextlib
    out = {}
    for tag, (ukw, ekw) in COND_OPTION_VARIANTS.items():
        m, enc = _build_cond((8, 64), 8, 32, unet_kw=ukw, enc_kw=ekw)
        batch = synth_layout_batch(2, 8, 64, seed=51)
        x = seeded_randn(2, 2, 8, 64, seed=52)
        lam = torch.tensor([-3.0, 1.5])
        with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
            cond = enc(batch)
            y = m(x, {"time_condition": lam, "other_condition": cond})
        out[f"{tag}_y"] = y
        out[f"{tag}_xf_out"] = cond["xf_out"]
        out[f"{tag}_keys_unet"] = np.array(sorted(f"{k}:{tuple(v.shape)}" for k, v in m.state_dict().items()))
        assert (1 - batch["is_valid_obj"]).bool().any(), "the batch must hold padded layout slots"
    save("cond_options", **out)


def sec_cond_full():
    """box-layout-v6 shapes: 32x1024, model_channels 64 (70.1 M params), B=1."""
    df = R.ref("models.diffusion")
    m, enc = _build_cond((32, 1024), 32, 64)
    batch = synth_layout_batch(1, 32, 1024, seed=53)
    x = seeded_randn(1, 2, 32, 1024, seed=54)
    lam = torch.tensor([-0.5])
    with torch.no_grad():
        cond = enc(batch)
        y = m(x, {"time_condition": lam, "other_condition": cond})
    keys = np.array(sorted(f"{k}:{tuple(v.shape)}" for k, v in m.state_dict().items()))
    out = {"y": y, "keys_unet": keys, "nparams": sum(p.numel() for p in m.parameters())}
    # 3-step conditional DDIM trajectory through the reference sampler (C3 at B=1)
    ddpm = df.CondContinuousTimeGaussianDiffusion(m, enc, cond_mode="concat").eval()
    rng = [torch.Generator().manual_seed(7)]
    xs = ddpm.sample(batch, 1, 3, progress=False, rng=rng, return_all=True, mode="ddim")
    out["traj_x1"], out["traj_x3"] = xs[1], xs[3]
    save("cond_full", **out)


def _strided(x, step=4):
    """Every `step`-th column of a [..., W] tensor (fixture size: the headline tensors are 2 MB each;
    samples and rows are all kept, so batch-index / tile-selection mistakes cannot hide)."""
    return x[..., ::step].contiguous()


def sec_c2():
    """Config C2 at the BENCHMARKED shape (EfficientUNet 32x1024, batch 8): one forward
    at per-sample log-SNRs and a 50-step DDIM run through the reference sampler
    (continuous_time.py:237-260), states 1 / 25 / 50.  Columns ::4 of every sample are stored."""
    eu = R.ref("models.unets.efficient_unet")
    df = R.ref("models.diffusion")
    m = _build_uncond(eu, 64, (32, 1024))
    x = seeded_randn(8, 2, 32, 1024, seed=81)
    lam = torch.linspace(-6.0, 6.0, 8)
    with torch.no_grad():
        y = m(x, lam)
    out = {"lam": lam, "y_s4": _strided(y), "y_norm": y.flatten(1).norm(dim=1)}
    ddpm = df.ContinuousTimeGaussianDiffusion(m, torch.nn.Identity()).eval()
    rng = [torch.Generator().manual_seed(i) for i in range(8)]
    xs = ddpm.sample(8, 50, progress=False, rng=rng, return_all=True, mode="ddim")
    for i in (1, 25, 50):
        out[f"x{i}_s4"] = _strided(xs[i])
        out[f"x{i}_norm"] = xs[i].flatten(1).norm(dim=1)
    save("c2_b8", **out)


def sec_c2_shards():
    """The C2 run of `sec_c2` for the shards a multi-GPU bench owns beyond rank 0: global samples
    8 ... 63 (rank r of `bench.py --gpus N` draws x_T from generators seeded 8 r + i,
    lidarcrafter_amd.parallel.shard_generators).  Per shard: the final state of the reference's
    50-step DDIM run, every 8th column of every sample and row, plus the full-frame norms."""
    eu = R.ref("models.unets.efficient_unet")
    df = R.ref("models.diffusion")
    m = _build_uncond(eu, 64, (32, 1024))
    ddpm = df.ContinuousTimeGaussianDiffusion(m, torch.nn.Identity()).eval()
    out = {}
    for r in range(1, 8):
        rng = [torch.Generator().manual_seed(8 * r + i) for i in range(8)]
        x = ddpm.sample(8, 50, progress=False, rng=rng, mode="ddim")
        out[f"shard{r}_x50_s8"] = _strided(x, 8)
        out[f"shard{r}_x50_norm"] = x.flatten(1).norm(dim=1)
        print(f"  shard {r} done", flush=True)
    save("c2_shards", **out)


def sec_c3():
    """Config C3 shard (box-layout-v6, 32x1024, batch 8 per GPU): LayoutUnetV1 forward at per-sample
    log-SNRs on a synthetic layout batch + a 2-step DDIM run through the reference conditional
    sampler (continuous_time_cond.py:255-281).  Columns ::4 stored."""
    df = R.ref("models.diffusion")
    m, enc = _build_cond((32, 1024), 32, 64)
    batch = synth_layout_batch(8, 32, 1024, seed=83)
    x = seeded_randn(8, 2, 32, 1024, seed=84)
    lam = torch.linspace(-5.0, 5.0, 8)
    with torch.no_grad():
        cond = enc(batch)
        y = m(x, {"time_condition": lam, "other_condition": cond})
    out = {"lam": lam, "y_s4": _strided(y), "y_norm": y.flatten(1).norm(dim=1)}
    ddpm = df.CondContinuousTimeGaussianDiffusion(m, enc, cond_mode="concat").eval()
    rng = [torch.Generator().manual_seed(40 + i) for i in range(8)]
    xs = ddpm.sample(batch, 8, 2, progress=False, rng=rng, return_all=True, mode="ddim")
    out["x2_s4"] = _strided(xs[2])
    out["x2_norm"] = xs[2].flatten(1).norm(dim=1)
    save("c3_b8", **out)


def sec_c4():
    """Config C4 shapes: the layout-conditioned denoiser at 64x2048 (`image_size=64`,
    `feature_map_size=[64,2048]`; attention over 8192 / 2048 tokens + 13 layout keys), width
    reduced to model_channels=32 so the reference finishes in a minute on 8 cores; both the
    box-layout (10 cond channels) and the auto-regressive (11 channels, `autoregressive_cond`)
    input forms (layout_encoder.py:298-302).  B=1, columns ::4 stored."""
    out = {}
    for tag, cond_out, n_extra, seed in (("box", 10, 0, 85), ("ar", 11, 1, 87)):
        m, enc = _build_cond((64, 2048), 64, 32, cond_out=cond_out)
        batch = synth_layout_batch(1, 64, 2048, seed=seed, n_extra=n_extra)
        x = seeded_randn(1, 2, 64, 2048, seed=seed + 1)
        lam = torch.tensor([0.75])
        with torch.no_grad():
            cond = enc(batch)
            y = m(x, {"time_condition": lam, "other_condition": cond})
        out[f"{tag}_y_s4"] = _strided(y)
        out[f"{tag}_y_norm"] = y.flatten(1).norm(dim=1)
    save("c4_64x2048", **out)


def sec_c4_full():
    """Config C4 at FULL WIDTH (round 3): `nuscenes-auto-reg-v2` (model_channels=64, 11 condition
    channels, `autoregressive_cond`) at 64x2048 (`image_size=64`, `feature_map_size=[64,2048]`:
    attention over 8192 / 2048 image tokens + 13 layout keys), B=1, one forward through the
    reference's LayoutUnetV1 + layout encoder (~1.5 TFLOP on the CPU).  Columns ::4 stored."""
    m, enc = _build_cond((64, 2048), 64, 64, cond_out=11)
    batch = synth_layout_batch(1, 64, 2048, seed=91, n_extra=1)
    x = seeded_randn(1, 2, 64, 2048, seed=92)
    lam = torch.tensor([-0.25])
    with torch.no_grad():
        cond = enc(batch)
        y = m(x, {"time_condition": lam, "other_condition": cond})
    save("c4_full", y_s4=_strided(y), y_norm=y.flatten(1).norm(dim=1),
         nparams=sum(p.numel() for p in m.parameters()))


def _ref_lidar_utils(res, coords):
    lidar = R.ref("utils.lidar")
    return lidar.LiDARUtility(resolution=res, depth_format="log_depth", min_depth=1.45, max_depth=80.0,
                              ray_angles=coords)


def _ref_preprocess_condition_mask(lu, condition_mask, num_classes=9):
    """tools/evaluation/sample_and_save_cond.py:106-117 (a closure of the caller script: restated,
    every op is the reference's own LiDARUtility / torch)."""
    import torch.nn.functional as F
    one_hot = F.one_hot(condition_mask[:, 0, ...].long(), num_classes=num_classes).permute(0, 3, 1, 2)
    depth = lu.convert_depth(condition_mask[:, 1, ...].unsqueeze(1))
    return torch.cat([one_hot.float(), depth], dim=1)


def _ref_postprocess(lu, sample):
    """tools/evaluation/sample_and_save_cond.py:119-124."""
    sample = lu.denormalize(sample)
    depth, rflct = sample[:, [0]], sample[:, [1]]
    depth = lu.revert_depth(depth)
    return torch.cat([depth, lu.to_xyz(depth), rflct], dim=1)


def _scene(seed, K_):
    from lidarcrafter_amd.testing import synth_scene_boxes
    names_all = ('car', 'truck', 'construction_vehicle', 'bus', 'trailer', 'motorcycle', 'bicycle', 'pedestrian')
    sb = synth_scene_boxes(K_, seed=seed)
    names = ["ego"] + [names_all[int(c) - 1] for c in sb[:, 7]]
    gt_boxes = np.concatenate([np.zeros((1, 7)), sb[:, :7].astype(np.float64)])
    return gt_boxes, names


def sec_c5_flow():
    """Config C5 COMPOSED on the reference (round 3): user boxes -> object-branch item
    (pipe_related.conduct_obj_data_dict :200-202 -> CustomNuscObjectDataset) -> the object sampler
    flow of tools/vis_tools/functions/object_sampler.py:23-45 (collate, squeeze, 4-step 'ddpm' run
    of CondContinuousLayoutGaussianDiffusion1D, NuscDataset.unscaled_objs_3d) -> background:
    pipe_related.get_mask_cond_single :220-227 -> collate -> preprocess_condition_mask -> 4-step
    DDIM run of the full-width box-layout-v6 denoiser -> postprocess; merged cloud = [background
    rows outside the condition mask | object rows]; lidargen/metrics/bev.py histograms, JSD / MMD
    against a set of seeded sweeps.  Two scenes, batch 2."""
    import importlib
    import importlib.util
    from lidarcrafter_amd.testing import synth_text_features

    _import_pipe_related()
    cd = _import_custom_dataset()
    cd.CustomNuscObjectDataset.scene_graph_assigner = None
    sys.modules.pop("ref_vis_utils.pipe_related", None)
    pr = importlib.import_module("ref_vis_utils.pipe_related")
    spec = importlib.util.spec_from_file_location("ref_bev", R.REF + "/lidargen/metrics/bev.py")
    bev = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bev)
    df = R.ref("models.diffusion")
    pu = R.ref("models.unets.point_unet")
    oe = R.ref("models.unets.encoders.object_gen_encoder")
    d1 = R.ref("models.diffusion.continuous_time_1d_cond")

    om = seeded_fill(pu.PointUNet(point_dim=4, cond_dims=768), salt=300).eval()
    oenc = seeded_fill(oe.ObjectGenEncoder(num_class=8), salt=301).eval()
    oenc.obj_text_feat = synth_text_features()
    oenc.prepare_called = True
    oddpm = d1.CondContinuousLayoutGaussianDiffusion1D(om, oenc, clip_sample=False).eval()
    m, enc = _build_cond((32, 1024), 32, 64)
    ddpm = df.CondContinuousTimeGaussianDiffusion(m, enc, cond_mode="concat").eval()
    lu = _ref_lidar_utils((32, 1024), m.coords)

    out, items, hists = {}, [], []
    scenes = [_scene(70, 4), _scene(71, 5)]
    obj_rows = []
    for s_, (gt_boxes, names) in enumerate(scenes):
        t = f"s{s_}_"
        custom = pr.conduct_obj_data_dict([dict(gt_boxes=gt_boxes.copy(), gt_names=list(names))])
        out[t + "fg_encoding_box"], out[t + "fg_class"] = custom["fg_encoding_box"], custom["fg_class"]
        ods = cd.CustomNuscObjectDataset(custom_box_infos=[custom])
        batch = ods.collate_fn([dict(custom)])
        batch["fg_encoding_box"] = batch["fg_encoding_box"].squeeze(0)
        batch["fg_class"] = batch["fg_class"].squeeze(0)
        n = batch["fg_encoding_box"].shape[0]
        rng = [torch.Generator().manual_seed(700 + 10 * s_ + i) for i in range(n)]
        gen = oddpm.sample(batch_dict=batch, batch_size=n, num_steps=4, mode="ddpm", return_all=False,
                           rng=rng, progress=False)
        out[t + "gen"] = gen
        rows = ods.unscaled_objs_3d(0, custom, gen.detach().cpu().numpy().copy())
        out[t + "obj_rows"] = rows
        obj_rows.append(rows)
        items.append(pr.get_mask_cond_single([dict(gt_boxes=gt_boxes.copy(), gt_names=list(names))]))
    ds = cd.CustomDataset(custom_box_infos=[])
    batch = ds.collate_fn(items)
    out["condition_mask_class"] = batch["condition_mask"][:, 0].numpy().astype(np.uint8)
    out["condition_mask_depth"] = batch["condition_mask"][:, 1]
    batch["concat_cond"] = _ref_preprocess_condition_mask(lu, batch["condition_mask"])
    rng = [torch.Generator().manual_seed(800 + i) for i in range(2)]
    x = ddpm.sample(batch_dict=batch, batch_size=2, num_steps=4, mode="ddim", rng=rng, progress=False).clamp(-1, 1)
    frames = _ref_postprocess(lu, x)
    out["x_s4"], out["x_norm"] = _strided(x), x.flatten(1).norm(dim=1)
    out["frames_s4"] = _strided(frames)
    for b in range(2):
        fr = frames[b].numpy()
        keep = ~(batch["condition_mask"][b, 0].numpy() > 0)[None]
        xyz, inten = fr[1:4] * keep, fr[4:5] * keep
        bg = np.stack([xyz[0], xyz[1], xyz[2], inten[0]], axis=-1).reshape(-1, 4)
        bg = bg[np.linalg.norm(bg[:, :3], axis=1) > 1e-2]
        merged = np.concatenate([bg, obj_rows[b][:, :4].astype(np.float32)], axis=0)
        out[f"s{b}_n_bg"] = np.array([bg.shape[0]])
        hists.append(bev.point_cloud_to_histogram(torch.from_numpy(merged[:, :3].copy())))
    set_a = torch.stack(hists)
    set_b = torch.stack([bev.point_cloud_to_histogram(torch.from_numpy(synth_points(30000, seed=900 + i)[:, :3]))
                         for i in range(3)])
    out["hist_a"] = set_a.numpy().astype(np.uint16)
    out["jsd"], out["mmd"] = np.float64(bev.compute_jsd_2d(set_a, set_b)), np.float64(bev.compute_mmd_2d(set_a, set_b))
    save("c5_flow", **out)


def sec_c4_seq():
    """Config C4 as a SEQUENCE on the reference (round 3): frame 0 of
    tools/evaluation/sample_and_save_temporal.py:198-262 at 64x2048 with the FULL-WIDTH
    box-layout-v6 architecture (image_size 64, feature_map_size [64,2048]): the reference's own
    CustomDataset item at resolution (64, 2048) -> collate -> preprocess_condition_mask -> 2-step
    'ddpm' run of CondContinuousTimeGaussianDiffusion under per-sample CPU generators ->
    clamp -> postprocess.  B=1.  (Frames 1-4 of the device loop are checked in the GPU test by
    feeding each HIP frame to the oracle glue, which is pinned bit-exactly on the reference.)"""
    _import_pipe_related()
    cd = _import_custom_dataset()
    df = R.ref("models.diffusion")
    m, enc = _build_cond((64, 2048), 64, 64)
    ddpm = df.CondContinuousTimeGaussianDiffusion(m, enc, cond_mode="concat").eval()
    lu = _ref_lidar_utils((64, 2048), m.coords)
    gt_boxes, names = _scene(72, 5)
    cfg = cd.DataConfig(resolution=(64, 2048))
    ds = cd.CustomDataset([dict(gt_boxes=gt_boxes.copy(), gt_names=list(names))], cfg=cfg)
    batch = ds.collate_fn([ds[0]])
    out = {"condition_mask_class": batch["condition_mask"][:, 0].numpy().astype(np.uint8),
           "condition_mask_depth_s4": _strided(batch["condition_mask"][:, 1])}
    batch["concat_cond"] = _ref_preprocess_condition_mask(lu, batch["condition_mask"])
    rng = [torch.Generator().manual_seed(810)]
    xs = ddpm.sample(batch_dict=batch, batch_size=1, num_steps=2, mode="ddpm", rng=rng, progress=False,
                     return_all=True)
    x = xs[-1].clamp(-1, 1)
    out["x1_s16"], out["x1_norm"] = _strided(xs[1], 16), xs[1].flatten(1).norm(dim=1)
    out["x_s4"], out["x_norm"] = _strided(x), x.flatten(1).norm(dim=1)
    out["frame_s8"] = _strided(_ref_postprocess(lu, x), 8)
    save("c4_seq", **out)


def sec_rng_state():
    """Generator state after `sample()` (base.py:73-96, continuous_time.py:226-231: DDIM eta=0 still
    draws randn_like every step): 4 numbers drawn from each per-sample generator AFTER a 3-step
    run of the reduced model, for ddim and ddpm."""
    eu = R.ref("models.unets.efficient_unet")
    df = R.ref("models.diffusion")
    m = _build_uncond(eu, 16, (8, 64))
    ddpm = df.ContinuousTimeGaussianDiffusion(m, torch.nn.Identity()).eval()
    out = {}
    for mode in ("ddim", "ddpm"):
        rng = [torch.Generator().manual_seed(500 + i) for i in range(2)]
        out[f"{mode}_x"] = ddpm.sample(2, 3, progress=False, rng=rng, mode=mode)
        out[f"{mode}_next"] = torch.stack([torch.randn(4, generator=g) for g in rng])
        one = torch.Generator().manual_seed(510)
        ddpm.sample(2, 3, progress=False, rng=one, mode=mode)
        out[f"{mode}_next_one"] = torch.randn(4, generator=one)
    save("rng_state", **out)


def sec_render():
    """lidargen/utils/render.py: bilinear_rasterizer :83-142, estimate_surface_normal :145-236,
    colorize :239-246 (the functions that do not call kornia; the module is imported with an EMPTY
    stand-in for the absent `kornia` import line only -- make_Rt / render_point_clouds, which do call
    it, are not pinned), and lidargen/utils/training.py:7-24 (LR multipliers)."""
    import types
    import matplotlib.cm as cm

    k = types.ModuleType("kornia")
    k.geometry = types.ModuleType("kornia.geometry")
    k.geometry.conversions = types.ModuleType("kornia.geometry.conversions")
    k.geometry.conversions.axis_angle_to_rotation_matrix = None
    sys.modules.update({"kornia": k, "kornia.geometry": k.geometry,
                        "kornia.geometry.conversions": k.geometry.conversions})
    rd = R.ref("utils.render")
    tr = R.ref("utils.training")
    out = {}
    g = torch.Generator().manual_seed(90)
    coords = torch.rand(2, 500, 2, generator=g) * 40 - 4          # some outside the 32x32 image
    vals = torch.rand(2, 500, 3, generator=g)
    out["splat"] = rd.bilinear_rasterizer(coords, vals, (32, 32))
    pts = seeded_randn(2, 3, 16, 64, seed=91).cumsum(-1)
    out["normal_closest"] = rd.estimate_surface_normal(pts, d=2, mode="closest")
    out["normal_mean"] = rd.estimate_surface_normal(pts, d=1, mode="mean")
    v = torch.rand(2, 1, 8, 16, generator=g) * 1.2 - 0.1
    out["color_turbo"] = rd.colorize(v)
    out["color_viridis"] = rd.colorize(v[:, 0], cm.viridis)
    opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=1.0)
    sch = tr.get_cosine_schedule_with_warmup(opt, num_warmup_steps=5, num_training_steps=40)
    lrs = []
    for _ in range(45):
        lrs.append(opt.param_groups[0]["lr"])
        opt.step()
        sch.step()
    out["lr"] = np.array(lrs)
    save("render", **out)


def sec_caller_names():
    """NAMES ONLY: every `lidargen.*` module / attribute the reference's caller scripts touch
    (tools/generate/generate.py, generate_cond.py, tools/train/train_lidm.py, train_lidm_cond.py,
    tools/evaluation/sample_and_save_cond.py, sample_and_save_temporal.py), extracted with `ast`;
    committed as tests/golden/caller_names.json and resolved against this build by a CPU test."""
    import ast
    import json

    scripts = ["tools/generate/generate.py", "tools/generate/generate_cond.py",
               "tools/train/train_lidm.py", "tools/train/train_lidm_cond.py",
               "tools/evaluation/sample_and_save_cond.py",
               "tools/evaluation/sample_and_save_temporal.py"]
    res = {}
    for sc in scripts:
        tree = ast.parse(open(os.path.join(R.REF, sc)).read())
        alias, names = {}, set()
        for node in ast.walk(tree):
            if isinstance(node, ast.ImportFrom) and node.module and node.module.split(".")[0] == "lidargen":
                for a in node.names:
                    names.add(f"{node.module}.{a.name}")
                    alias[a.asname or a.name] = f"{node.module}.{a.name}"
            elif isinstance(node, ast.Import):
                for a in node.names:
                    if a.name.split(".")[0] == "lidargen":
                        names.add(a.name)
                        alias[a.asname or a.name] = a.name
        for node in ast.walk(tree):
            if isinstance(node, ast.Attribute) and isinstance(node.value, ast.Name) and \
                    node.value.id in alias and not alias[node.value.id].endswith("__all__"):
                names.add(f"{alias[node.value.id]}.{node.attr}")
        # attribute chains on the objects the factory returns (cfg.training.lr, ddpm.sample, ...)
        roots = {"cfg": "cfg", "auto_cfg": "cfg", "ddpm": "ddpm", "auto_ddpm": "ddpm",
                 "lidar_utils": "lidar_utils", "model": "model"}
        for node in ast.walk(tree):
            if isinstance(node, ast.Attribute):
                chain, cur = [], node
                while isinstance(cur, ast.Attribute):
                    chain.append(cur.attr)
                    cur = cur.value
                if isinstance(cur, ast.Name) and cur.id in roots:
                    chain = list(reversed(chain))
                    keep = chain[:2] if roots[cur.id] == "cfg" else chain[:1]
                    names.add("@" + roots[cur.id] + "." + ".".join(keep))
        res[sc] = sorted(names)
    path = os.path.join(HERE, "caller_names.json")
    json.dump(res, open(path, "w"), indent=1)
    print("caller_names.json", {k: len(v) for k, v in res.items()})


def _grad_digest(named_params):
    """Per parameter: L2 norm of the gradient and its first 8 entries (flattened) -- enough to pin
    every parameter's gradient without storing megabytes."""
    names, norms, heads = [], [], []
    for k, p_ in named_params:
        if p_.grad is None:
            continue
        g = p_.grad.detach().double().flatten()
        names.append(k)
        norms.append(float(g.norm()))
        h = torch.zeros(8, dtype=torch.float64)
        h[: min(8, g.numel())] = g[:8]
        heads.append(h)
    return np.array(names), np.array(norms), torch.stack(heads).numpy()


def sec_train():
    """Training step of the REFERENCE modules under torch autograd (tools/train/train_lidm.py:214-265,
    train_lidm_cond.py:259-322): loss of one denoising step at fixed timesteps / noise and the
    gradient of every parameter -- reduced EfficientUNet, reduced LayoutUnetV1 + layout encoder
    (eval mode: dropout off)."""
    eu = R.ref("models.unets.efficient_unet")
    out = {}
    m = _build_uncond(eu, 16, (8, 64))
    for p_ in m.parameters():
        p_.requires_grad_(True)
    x_t = seeded_randn(2, 2, 8, 64, seed=61)
    noise = seeded_randn(2, 2, 8, 64, seed=62)
    lam = torch.tensor([-2.5, 1.0])
    loss = ((m(x_t, lam) - noise) ** 2).mean()
    loss.backward()
    out["u_loss"] = loss.detach()
    out["u_names"], out["u_norms"], out["u_heads"] = _grad_digest(m.named_parameters())
    mc, enc = _build_cond((8, 64), 8, 32)
    for p_ in list(mc.parameters()) + list(enc.parameters()):
        p_.requires_grad_(True)
    batch = synth_layout_batch(2, 8, 64, seed=51)
    x_t = seeded_randn(2, 2, 8, 64, seed=63)
    noise = seeded_randn(2, 2, 8, 64, seed=64)
    lam = torch.tensor([-1.5, 2.0])
    cond = enc(batch)
    loss = ((mc(x_t, {"time_condition": lam, "other_condition": cond}) - noise) ** 2).mean()
    loss.backward()
    out["c_loss"] = loss.detach()
    out["c_names"], out["c_norms"], out["c_heads"] = _grad_digest(
        [("unet." + k, v) for k, v in mc.named_parameters()] + [("enc." + k, v) for k, v in enc.named_parameters()])
    # foreground-object branch (tools/train/train_object.py): PointUNet + ObjectGenEncoder
    from lidarcrafter_amd.testing import synth_object_batch, synth_text_features

    pu = R.ref("models.unets.point_unet")
    oe = R.ref("models.unets.encoders.object_gen_encoder")
    mo = seeded_fill(pu.PointUNet(point_dim=4, cond_dims=768), salt=300).eval()
    eo = seeded_fill(oe.ObjectGenEncoder(num_class=8), salt=301).eval()
    eo.obj_text_feat = synth_text_features()
    eo.prepare_called = True
    for p_ in list(mo.parameters()) + list(eo.parameters()):
        p_.requires_grad_(True)
    ob = synth_object_batch(3, seed=95)
    x_t = seeded_randn(3, 1024, 4, seed=65)
    noise = seeded_randn(3, 1024, 4, seed=66)
    loss = ((mo(x_t, {"time_condition": torch.tensor([-6.0, 0.5, 9.0]), "other_condition": eo(ob)}) - noise) ** 2).mean()
    loss.backward()
    out["o_loss"] = loss.detach()
    out["o_names"], out["o_norms"], out["o_heads"] = _grad_digest(
        [("unet." + k, v) for k, v in mo.named_parameters()] + [("enc." + k, v) for k, v in eo.named_parameters()])
    save("train", **out)


def sec_object():
    """Foreground-object branch: ObjectGenEncoder (encoders/object_gen_encoder.py:7-88) with
    synthetic class text features, PointUNet (point_unet.py:14-71) forward, and 4-step DDPM / DDIM
    runs of CondContinuousLayoutGaussianDiffusion1D (continuous_time_1d_cond.py:9-91,
    clip_sample=False like option_nusc_object.py)."""
    from lidarcrafter_amd.testing import synth_object_batch, synth_text_features

    pu = R.ref("models.unets.point_unet")
    oe = R.ref("models.unets.encoders.object_gen_encoder")
    d1 = R.ref("models.diffusion.continuous_time_1d_cond")
    m = seeded_fill(pu.PointUNet(point_dim=4, cond_dims=768), salt=300).eval()
    enc = seeded_fill(oe.ObjectGenEncoder(num_class=8), salt=301).eval()
    enc.obj_text_feat = synth_text_features()
    enc.prepare_called = True
    batch = synth_object_batch(3, seed=95)
    out = {}
    with torch.no_grad():
        cond = enc(batch)
        out["cond"] = cond
        x = seeded_randn(3, 1024, 4, seed=96)
        lam = torch.tensor([-6.0, 0.5, 9.0])
        out["unet_y"] = m(x, {"time_condition": lam, "other_condition": cond})
    out["keys"] = np.array(sorted(f"{k}:{tuple(v.shape)}" for k, v in
                                  list(m.state_dict().items()) + [("enc." + k, v) for k, v in
                                                                  enc.state_dict().items()]))
    ddpm = d1.CondContinuousLayoutGaussianDiffusion1D(m, enc, clip_sample=False).eval()
    for mode in ("ddpm", "ddim"):
        rng = [torch.Generator().manual_seed(600 + i) for i in range(3)]
        out[f"traj_{mode}"] = ddpm.sample(batch, 3, 4, progress=False, rng=rng, return_all=True,
                                          mode=mode)
    save("object", **out)


def sec_voxel():
    """lidargen/metrics/metric_utils.py: ravel_hash :28-40, sparse_quantize :43-66, pcd2bev_sum
    :233-258.  The module itself cannot be imported (its package __init__ pulls torchsparse and the
    pretrained-extractor registry), so the three numpy-only functions are compiled from the file's
    AST IN MEMORY (nothing of the source is stored) and run on seeded sweeps; the 1200 x 1200
    volumes are stored sparsely (flat index + count)."""
    import ast
    import math
    from itertools import repeat
    from typing import List, Tuple, Union

    src = open(R.REF + "/lidargen/metrics/metric_utils.py").read()
    tree = ast.parse(src)
    keep = [n for n in tree.body if isinstance(n, ast.FunctionDef)
            and n.name in ("ravel_hash", "sparse_quantize", "pcd2bev_sum")]
    ns = dict(np=np, math=math, repeat=repeat, List=List, Tuple=Tuple, Union=Union, VOXEL_SIZE=0.05,
              DATA_CONFIG={"64": {"x": [-50, 50], "y": [-50, 50], "z": [-3, 1]},
                           "32": {"x": [-30, 30], "y": [-30, 30], "z": [-3, 6]}})
    exec(compile(ast.Module(body=keep, type_ignores=[]), "<metric_utils>", "exec"), ns)
    out = {}
    pts = synth_points(20000, seed=11)
    for tag, cols, vs in (("2d", 2, 0.5), ("3d", 3, (0.4, 0.4, 0.2))):
        c, idx, inv = ns["sparse_quantize"](pts[:, :cols], vs, return_index=True, return_inverse=True)
        out[f"sq_{tag}_coords"], out[f"sq_{tag}_index"], out[f"sq_{tag}_inverse"] = c, idx, inv
    out["hash"] = ns["ravel_hash"](np.floor(pts[:500, :3] / 0.7).astype(np.int32))
    sets = [[synth_points(30000, seed=20 + 10 * k + i) for i in range(3)] for k in range(2)]
    sets[0][0][:6, :2] = [[30.0, 0.0], [-30.0, 0.0], [29.999998, 1.0], [-29.999998, 1.0],
                          [0.0, 29.999998], [0.025, 0.05]]                   # range edges, bin edges
    vols = ns["pcd2bev_sum"]("32", sets[0], sets[1])
    for k, v in enumerate(vols):
        nz = np.flatnonzero(v)
        out[f"bev{k}_idx"], out[f"bev{k}_cnt"] = nz.astype(np.int32), v.flat[nz].astype(np.uint8)
        out[f"bev{k}_shape"] = np.array(v.shape)
    save("voxel", **out)


def synth_boxes(n, pts, seed):
    from lidarcrafter_amd.testing import synth_boxes as f
    return f(n, pts, seed)


def sec_boxes():
    """points_in_boxes_cpu of the REFERENCE C++ (oracle/_ref build of roiaware_pool3d.cpp) through
    the reference's Python wrapper semantics (boxes[:, 3:6] += 0.2 in place, MARGIN 1e-2)."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import build_c

    build_c.build_ref()
    ref = build_c.load_ref()
    pts = synth_points(32768, 5)[:, :3].copy()
    bx = synth_boxes(12, pts, 6)
    infl = bx.copy()
    infl[:, 3:6] += 0.2
    out = torch.zeros(12, len(pts), dtype=torch.int32)
    ref.points_in_boxes_cpu(torch.from_numpy(infl), torch.from_numpy(pts), out)
    save("boxes", mask_packed=np.packbits(out.numpy().astype(np.uint8), axis=1),
         count=int(out.sum()))


def sec_layout_cond():
    """convert_boxes_to_2d (dataset/transforms_3d/common.py:99-181) on seeded float32 boxes."""
    cm = R.ref("dataset.transforms_3d.common")
    from lidarcrafter_amd.testing import synth_scene_boxes
    out = {}
    for tag, n, H, W, seed in (("a", 9, 32, 1024, 0), ("b", 13, 32, 1024, 1), ("c", 5, 64, 2048, 2)):
        boxes = synth_scene_boxes(n, seed)
        c2d, mask, wmap = cm.convert_boxes_to_2d(boxes.copy(), H=H, W=W, min_depth=1.45,
                                                 max_depth=80.0, fov_up=10.0, fov_down=-30.0)
        out[f"{tag}_corners2d"] = c2d.astype(np.float64)
        out[f"{tag}_class"] = mask[0].astype(np.uint8)
        out[f"{tag}_depth"] = mask[1]
    save("layout_cond", **out)


def sec_temporal():
    """tools/vis_tools/utils/common.py (numpy only: imported by file path) warp_lidar_future :59-112,
    warp_boxes_future :115-172, compute_inter_frame_transforms :174-222, and
    lidargen/dataset/utils.py rotate_points_along_z :37-59, on seeded inputs."""
    import importlib.util
    from lidarcrafter_amd.testing import synth_points, synth_temporal_inputs

    spec = importlib.util.spec_from_file_location(
        "ref_vis_common", R.REF + "/tools/vis_tools/utils/common.py")
    vc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(vc)
    du = R.ref("dataset.utils")
    trajs, boxes = synth_temporal_inputs()
    a = np.insert(trajs, 0, 0, axis=1)
    acc = np.cumsum(a, axis=1)
    a = acc[:, 1:] - acc[:, :-1]
    ego_xy, obj_xy = np.cumsum(a[0], axis=0), np.cumsum(a[1:], axis=1)
    P = synth_points(800, seed=7)
    out = dict(trajs=trajs, boxes=boxes, ego_xy=ego_xy, obj_xy=obj_xy,
               warp_lidar=vc.warp_lidar_future(P=P, future_xy=ego_xy, z0=0.0),
               warp_lidar64=vc.warp_lidar_future(P=P.astype(np.float64), future_xy=ego_xy, z0=0.0),
               warp_boxes=vc.warp_boxes_future(boxes0=boxes, traj_obj=obj_xy, traj_ego=ego_xy, z_e=0.0),
               Ts=vc.compute_inter_frame_transforms(future_xy=ego_xy, z0=0.0))
    ang = np.array([0.3, -2.1], np.float64)
    pts = np.stack([P[:300], P[300:600]])
    out["rot"] = du.rotate_points_along_z(pts, ang)
    save("temporal", **out)


def _import_pipe_related():
    """The reference's tools/vis_tools/utils/pipe_related.py, imported for the functions that are
    plain numpy + the reference's own points_in_boxes_cpu.  Its module-level imports cannot all be
    satisfied in this image:
      * lidargen.ops.roiaware_pool3d.roiaware_pool3d_utils  -- imported FOR REAL; the compiled
        extension it wraps is the reference's own roiaware_pool3d.cpp built by oracle/build_c.py;
      * lidargen.dataset.custom_dataset (-> nuscenes_dataset -> loguru / clip / pyquaternion) and
        lidargen.metrics.models.ptv3.model (spconv, flash-attn): absent third-party packages.  The
        two module names are bound to EMPTY placeholders only so that the import statement passes;
        every function that touches them (refine_next_frame_points, get_next_frame_points,
        get_mask_cond*, build_point_segmenter) stays UNPINNED and is never called here."""
    import importlib
    import types

    root = os.path.dirname(os.path.dirname(HERE))
    sys.path.insert(0, root)
    from oracle import build_c

    build_c.build()
    ext = build_c.load_ref()
    assert ext is not None, "oracle/_ref build of the reference's roiaware_pool3d.cpp is missing"
    for name in ("lidargen.ops", "lidargen.ops.roiaware_pool3d", "lidargen.metrics",
                 "lidargen.metrics.models", "lidargen.metrics.models.ptv3", "ref_vis_utils"):
        m = types.ModuleType(name)
        sub = "tools/vis_tools/utils" if name == "ref_vis_utils" else name.replace(".", "/")
        m.__path__ = [R.REF + "/" + sub]
        m.__package__ = name
        sys.modules[name] = m
    sys.modules["lidargen.ops.roiaware_pool3d.roiaware_pool3d_cuda"] = ext
    for name, attrs in (("lidargen.dataset.custom_dataset", ("CustomDataset", "CustomNuscObjectDataset")),
                        ("lidargen.metrics.models.ptv3.model", ("PTv3",))):
        ph = types.ModuleType(name)
        for a_ in attrs:
            setattr(ph, a_, None)
        sys.modules[name] = ph
    return importlib.import_module("ref_vis_utils.pipe_related")


def sec_pipe():
    """pipe_related.py of the reference: interp_trajs_numpy :229-241, remove_ego_points :11-13,
    get_temporal_boxes_3d :28-95 (float64 flow), delete_fg_points :282-288, on the seeded first-frame
    scene of tests/_scenes.py at 8x256 (inputs are regenerated by the test; a checksum guards them)."""
    pr = _import_pipe_related()
    from tests._scenes import temporal_scene

    out = {}
    tr = np.cumsum(seeded_randn(3, 7, 2, seed=61).numpy().astype(np.float64), axis=1)
    out["interp_in"], out["interp_16"], out["interp_5"] = tr, pr.interp_trajs_numpy(tr, M=16), pr.interp_trajs_numpy(tr, M=5)
    pts = synth_points(500, seed=62)
    out["ego_removed_idx"] = np.flatnonzero(
        (pr.remove_ego_points(np.concatenate([pts, np.arange(500, dtype=np.float32)[:, None]], 1))[:, 4:5] >= 0)[:, 0])
    out["ego_removed"] = pr.remove_ego_points(pts.copy())
    for seed in (0, 1):
        first, _, _ = temporal_scene(seed, H=8, W=256, n_pts=6000)
        ref_first = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in first.items()}
        bg, fut_bg, boxes, fut_boxes, Ts, obj_pts, obj_int = pr.get_temporal_boxes_3d(ref_first, M=None)
        t = f"s{seed}_"
        out[t + "in_sum"] = np.array([float(np.abs(first["xyz"]).sum()), float(first["reflectance"].sum()),
                                      float(first["condition_mask"].sum()), float(np.abs(first["gt_fut_trajs"]).sum())])
        out[t + "bg"], out[t + "boxes"], out[t + "fut_boxes"], out[t + "Ts"] = bg, boxes, fut_boxes, Ts
        out[t + "fut_bg_first"], out[t + "fut_bg_last"] = fut_bg[0], fut_bg[-1]
        out[t + "obj_n"] = np.array([p.shape[0] for p in obj_pts])
        out[t + "obj_pts"] = np.concatenate(obj_pts, 0)
        out[t + "obj_int"] = np.concatenate(obj_int, 0)
        comb = np.concatenate([fut_bg[0], bg], axis=0)
        out[t + "delete_fg"] = pr.delete_fg_points(comb.copy(), fut_boxes[:, 0].copy())
        bgM = pr.get_temporal_boxes_3d({k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in first.items()}, M=9)
        out[t + "M9_fut_boxes"], out[t + "M9_Ts"] = bgM[3], bgM[4]
    save("pipe", **out)


def _import_custom_dataset():
    """The reference's lidargen/dataset/custom_dataset.py + nuscenes_dataset.py + base_dataset.py,
    imported FOR REAL (round 3) so that `CustomDataset.__getitem__` / `NuscDataset.pre_process`
    and, through them, `pipe_related.refine_next_frame_points` / `get_next_frame_points` run as the
    reference wrote them.  Module-level imports that this image cannot satisfy are bound to EMPTY
    placeholders; none of them is touched by the pinned functions (task 'layout_cond' /
    'autoregressive_generation', no augmentor, no pkl, no scene graph):
      * loguru (logger), clip, pyquaternion (Quaternion): absent third-party packages, used only by
        the nuScenes file loaders / `SceneGraphAssigner.build_clip` / pose code;
      * lidargen.dataset.augmentor.data_augmentor (-> SharedArray, iou3d_nms CUDA extension):
        `DatasetBase.__init__` constructs it only when `cfg.data_augmentor` is set (DataConfig has none);
      * `CustomDataset.scene_graph_assigner = None` is set on the class: `NuscDataset.__init__`
        (nuscenes_dataset.py:24-33) builds a `SceneGraphAssigner` (CLIP ViT-B/32 download onto
        'cuda') only `if not hasattr(self, 'scene_graph_assigner')`; it is read by task
        'layout_generation' alone (custom_dataset.py:84-88)."""
    import importlib
    import types

    for name in ("lidargen.dataset.augmentor",):
        m = types.ModuleType(name)
        m.__path__ = [R.REF + "/" + name.replace(".", "/")]
        m.__package__ = name
        sys.modules[name] = m
    for name, attrs in (("loguru", ("logger",)), ("clip", ()), ("pyquaternion", ("Quaternion",)),
                        ("lidargen.dataset.augmentor.data_augmentor", ("DataAugmentor",))):
        ph = types.ModuleType(name)
        for a_ in attrs:
            setattr(ph, a_, None)
        sys.modules[name] = ph
    sys.modules.pop("lidargen.dataset.custom_dataset", None)      # drop sec_pipe's placeholder
    cd = importlib.import_module("lidargen.dataset.custom_dataset")
    cd.CustomDataset.scene_graph_assigner = None
    return cd


def sec_pipe_next():
    """Round 3: the last three unpinned temporal-glue functions, run on the REFERENCE's own modules:
    CustomDataset.__getitem__ (custom_dataset.py:57-89, both tasks, float32 and float64 points,
    boxes only), pipe_related.refine_next_frame_points :271-281 and get_next_frame_points :243-269
    chained over two future frames exactly like sample_and_save_temporal.py:296-327 (generated
    frame stood in by a seeded point set).  Scenes: tests/_scenes.temporal_scene at the default
    DataConfig resolution 32x1024 (refine_next_frame_points cannot use another one)."""
    _import_pipe_related()                       # package nodes + the real roiaware_pool3d_utils
    cd = _import_custom_dataset()
    import importlib
    sys.modules.pop("ref_vis_utils.pipe_related", None)
    pr = importlib.import_module("ref_vis_utils.pipe_related")
    assert pr.CustomDataset is cd.CustomDataset
    from tests._scenes import temporal_scene

    out = {}
    img_keys = ("xyz", "reflectance", "depth", "mask", "condition_mask", "scene_loss_weight_map")
    box_keys = ("gt_boxes", "scaled_gt_boxes", "gt_boxes_2d", "fg_encoding_box", "is_valid_obj")
    for seed in (0, 1):
        first, pts, _ = temporal_scene(seed)
        t = f"s{seed}_"
        out[t + "in_sum"] = np.array([float(np.abs(pts).sum()), float(np.abs(first["gt_boxes"]).sum()),
                                      float(np.abs(first["gt_fut_trajs"]).sum())])
        info = lambda p: dict(points=p, gt_boxes=first["gt_boxes"].copy(), gt_names=list(first["gt_names"]))
        # ---- the item, task 'layout_cond' (float32 points) ----
        it = cd.CustomDataset([info(pts.copy())])[0]
        for k in (img_keys if seed == 0 else ("xyz", "reflectance", "condition_mask")) + box_keys:
            out[t + "item_" + k] = it[k]          # seed 1: `depth` / `mask` follow from xyz
        out[t + "item_keys"] = np.array(sorted(it.keys()))
        assert "points" not in it
        # ---- float64 points (what the glue hands over) and task 'autoregressive_generation' ----
        p64 = pts.astype(np.float64) * 1.0000001
        ds = cd.CustomDataset([info(p64.copy())])
        it64 = ds[0]
        setattr(ds, "task", "autoregressive_generation")
        ds.data = [info(p64.copy())]
        itar = ds[0]
        out[t + "itemar_keys"] = np.array(sorted(itar.keys()))
        assert np.array_equal(itar["autoregressive_cond"], np.concatenate([it64["depth"], it64["reflectance"]]))
        if seed == 0:
            out[t + "item64_xyz"] = it64["xyz"]
            out[t + "itemar_cond"] = itar["autoregressive_cond"]
        else:   # digest only (fixture size): position-weighted sums are order / cell sensitive
            wts = np.arange(1, 2 * 32 * 1024 + 1, dtype=np.float64).reshape(2, 32, 1024)
            out[t + "itemar_cond_digest"] = np.array([float((itar["autoregressive_cond"] * wts).sum()),
                                                      float(np.abs(it64["xyz"]).sum(dtype=np.float64))])
        # ---- boxes only (no 'points'): the first-frame item of the bulk harness ----
        itb = cd.CustomDataset([dict(gt_boxes=first["gt_boxes"].copy(), gt_names=list(first["gt_names"]))])[0]
        out[t + "itemb_keys"] = np.array(sorted(itb.keys()))
        out[t + "itemb_condition_mask"] = itb["condition_mask"]
        # ---- the frame chain (sample_and_save_temporal.py:261-327) ----
        ref_first = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in first.items()}
        ref_first["xyz"], ref_first["reflectance"], ref_first["condition_mask"] = (
            it["xyz"].copy(), it["reflectance"].copy(), it["condition_mask"].copy())
        _, fut_bg, _, fut_boxes, Ts, obj_pts, obj_int = pr.get_temporal_boxes_3d(ref_first)
        cur = np.stack([it["xyz"][0], it["xyz"][1], it["xyz"][2], it["reflectance"][0]], -1).reshape(-1, 4)
        out[t + "fut_boxes"], out[t + "Ts"] = fut_boxes, Ts
        for f in range(2):
            moved = (Ts[f] @ np.hstack((cur[:, :3], np.ones((cur.shape[0], 1)))).T).T
            moved[:, 3] = cur[:, 3]
            ref_bg = pr.refine_next_frame_points([dict(
                points=moved, gt_boxes=np.concatenate([np.zeros((1, 7)), fut_boxes[:, f]]),
                gt_names=list(first["gt_names"]))])
            nxt = pr.get_next_frame_points(cur, obj_pts, obj_int, fut_boxes[:, f],
                                           list(first["gt_names"]), Ts[f])
            # get_next_frame_points = [refined background | re-posed objects]: the refined rows are
            # stored once, as the head of `next`
            assert ref_bg.dtype == np.float32 and np.array_equal(ref_bg, nxt[:ref_bg.shape[0]])
            out[t + f"refine{f}_n"] = np.array([ref_bg.shape[0]])
            out[t + f"next{f}"] = nxt
            # the generated frame of step f is stood in by a seeded sweep (float32, like samples)
            gen = synth_points(32 * 1024, seed=300 + 10 * seed + f)
            comb = np.concatenate([fut_bg[f], gen], axis=0)
            gt_boxes = np.concatenate([np.zeros((1, 7), np.float32), fut_boxes[:, f]], axis=0)
            cur = pr.delete_fg_points(comb, gt_boxes[1:, :7])
            out[t + f"cur{f}_n"] = np.array([cur.shape[0]])
            out[t + f"cur{f}_sum"] = np.array([np.abs(cur).sum()])
    save("pipe_next", **out)


def sec_bev():
    """lidargen/metrics/bev.py (torch + scipy only: imported by file path): histograms of seeded
    sweeps, the bin edges torch.histogramdd used, JSD / MMD between two sets of sweeps."""
    import importlib.util
    from lidarcrafter_amd.testing import synth_points

    spec = importlib.util.spec_from_file_location("ref_bev", R.REF + "/lidargen/metrics/bev.py")
    bev = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bev)
    sets = []
    for base, scale in ((0, 1.0), (100, 0.8)):
        hs = []
        for i in range(5):
            pts = synth_points(6000, seed=base + i)[:, :3] * np.float32(scale)
            if i == 0:
                pts[:7, 0] = [80.0, -80.0, 79.99999, 0.0, 1.6, -1.6, 80.00001]   # edges of the range
                pts[:7, 1] = [0.5, 0.5, 0.5, 80.0, 0.0, 0.0, 0.5]
                pts[:7, 2] = 0.0
            hs.append(bev.point_cloud_to_histogram(torch.from_numpy(pts)))
        sets.append(torch.stack(hs))
    e = torch.histogramdd(torch.empty(0, 2), bins=100, range=[-80.0, 80.0, -80.0, 80.0]).bin_edges[0]
    small = bev.point_cloud_to_histogram(torch.from_numpy(synth_points(3000, seed=5)[:, :3] * np.float32(0.02)),
                                         min_depth=1e-6, max_depth=1e3, field_size=2.0)
    save("bev", hist_a=sets[0].numpy().astype(np.uint16), hist_b=sets[1].numpy().astype(np.uint16),
         edges=e.numpy(), hist_small=small.numpy().astype(np.uint16),
         jsd=np.float64(bev.compute_jsd_2d(sets[0], sets[1])),
         mmd=np.float64(bev.compute_mmd_2d(sets[0], sets[1])))


def sec_sampler_extras():
    """The sampler entry points not covered by `diffusion` / `trajectory`: q_step_from_x_0, q_step,
    RePaint (continuous_time.py:262-330), conditional inpaint (continuous_time_cond.py:283-353)
    and the training-loss VALUE p_loss (base.py:124-143) for every objective / loss weighting the
    reference offers, on the reduced models with seeded weights."""
    eu = R.ref("models.unets.efficient_unet")
    df = R.ref("models.diffusion")
    out = {}
    m = _build_uncond(eu, 16, (8, 64))
    ddpm = df.ContinuousTimeGaussianDiffusion(m, torch.nn.Identity()).eval()
    x0 = seeded_randn(2, 2, 8, 64, seed=71).clamp(-1, 1)
    steps = torch.tensor([0.8, 0.3])
    rng = [torch.Generator().manual_seed(300 + i) for i in range(2)]
    xt, noise = ddpm.q_step_from_x_0(x0, steps, rng=rng)
    out["q0_xt"], out["q0_noise"] = xt, noise
    rng = [torch.Generator().manual_seed(310 + i) for i in range(2)]
    out["q_step"] = ddpm.q_step(xt, torch.tensor([0.9, 0.5]), steps, rng=rng)
    mask = (seeded_randn(2, 1, 8, 64, seed=72) > 0).float().expand(-1, 2, -1, -1).contiguous()
    out["mask"] = mask
    with torch.no_grad():
        rng = [torch.Generator().manual_seed(320 + i) for i in range(2)]
        out["repaint"] = ddpm.repaint(x0, mask, num_steps=4, num_resample_steps=2, jump_length=2,
                                      progress=False, rng=rng, return_all=True)
        for obj, lt, msw in (("eps", "l2", True), ("v", "l2", True), ("x_0", "l2", True),
                             ("eps", "l1", False), ("v", "huber", True)):
            d2 = df.ContinuousTimeGaussianDiffusion(m, torch.nn.Identity(), prediction_type=obj,
                                                    loss_type=lt, min_snr_loss_weight=msw).eval()
            torch.manual_seed(5)
            out[f"loss_{obj}_{lt}_{int(msw)}"] = d2.p_loss(x0, steps)
    mc, enc = _build_cond((8, 64), 8, 32)
    dc = df.CondContinuousTimeGaussianDiffusion(mc, enc, cond_mode="concat").eval()
    batch = synth_layout_batch(2, 8, 64, seed=73)
    with torch.no_grad():
        rng = [torch.Generator().manual_seed(330 + i) for i in range(2)]
        out["inpaint"] = dc.inpaint(x0, mask, batch, num_steps=3, num_resample_steps=1, jump_length=2,
                                    progress=False, rng=rng, return_all=True)
    save("sampler_extras", **out)


def sec_schedules():
    """The log-SNR schedule variants of continuous_time.py:18-58 (linear, cosine, shifted,
    interpolated) as the reference evaluates them (float32 tensors, python-float constants)."""
    ct = R.ref("models.diffusion.continuous_time")
    t = torch.linspace(1.0, 0.0, 33)
    out = {"t": t,
           "linear": ct._log_snr_schedule_linear(t)[:, 0, 0, 0],
           "cosine": ct._log_snr_schedule_cosine(t)[:, 0, 0, 0],
           "cosine_m10_12": ct._log_snr_schedule_cosine(t, logsnr_min=-10, logsnr_max=12)[:, 0, 0, 0],
           "shifted_64_32": ct._log_snr_schedule_cosine_shifted(t, 64, 32)[:, 0, 0, 0],
           "shifted_32_128": ct._log_snr_schedule_cosine_shifted(t, 32, 128)[:, 0, 0, 0],
           # the reference multiplies t [N] by a [N,1,1,1] tensor here, so for N > 1 it returns an
           # [N,1,1,N] outer product (unusable for batches); evaluated one t at a time = the intent
           "interp_64_32_256": torch.cat([ct._log_snr_schedule_cosine_interpolated(
               t[i:i + 1], 64, 32, 256)[:, 0, 0, 0] for i in range(len(t))])}
    save("schedules", **out)


def sec_discrete_trajectory():
    """DiscreteTimeGaussianDiffusion.sample (discrete_time.py:182-201) on the reduced UNet: all
    states 0, 1, 25, 50 of a 50-step run (T = 50), ddpm and ddim."""
    eu = R.ref("models.unets.efficient_unet")
    df = R.ref("models.diffusion")
    m = _build_uncond(eu, 16, (8, 64))
    out = {}
    for kind in ("linear", "cosine"):
        dd = df.DiscreteTimeGaussianDiffusion(m, None, num_training_steps=50, noise_schedule=kind).eval()
        for mode in ("ddpm", "ddim"):
            rng = [torch.Generator().manual_seed(400 + i) for i in range(2)]
            xs = dd.sample(2, 50, progress=False, rng=rng, return_all=True, mode=mode)
            assert torch.isfinite(xs).all()
            out[f"{kind}_{mode}"] = xs[[0, 1, 25, 50]]
    save("discrete_trajectory", **out)


SECTIONS = {k[4:]: v for k, v in list(globals().items()) if k.startswith("sec_")}

if __name__ == "__main__":
    names = sys.argv[1:] or list(SECTIONS)
    for n in names:
        print(f"== {n}")
        SECTIONS[n]()
