"""Up-path fold (round 6, third part): `Resample(up=2) -> Conv2d(3x3, ring)` of the reference -- EfficientUNet's
Block.upsample (efficient_unet.py:143-145) and the inner pair of LayoutUnetV1's up-sampling ResBlock
(layout_unet_v1.py:219-235) -- evaluated at the LOW resolution: one 1x1 projection to the nine tap planes + one combine
pass (lidarcrafter_amd/csrc/upfold.hip, ops.conv_up2).  Checked against
  * the reference's own modules (tests/golden/fold_up.npz, make_fixtures.py::sec_fold_up),
  * the oracle's resample_up2 + conv_ring on more shapes, <= 2e-6 rel-L2 like every f16x2 convolution, the first and last
    output rows on their own (the two dropped border taps),
  * the unfolded HIP route (LC_FOLD_UP=0's order: resampling pass, then the conv at the high resolution),
and the per-channel statistics entries the combine pass leaves for the GroupNorm behind it.
The CPU part restates the kernel's algebra in float64 (nine planes, per-axis FIR phases, the two dropped (row, ky) terms)
and pins it on the oracle and on the reference's fixture: the fold is proven before any GPU runs it."""
import numpy as np
import pytest
import torch

from lidarcrafter_amd.testing import rel_l2, seeded_fill, seeded_randn

T = torch.from_numpy
SHAPES = {"a": (2, 32, 32, 4, 128, 41), "b": (1, 64, 40, 1, 256, 42), "c": (1, 128, 128, 8, 128, 43),
          "d": (2, 32, 16, 2, 384, 44)}


def folded_model(x, w, b):
    """What csrc/upfold.hip computes, in the precision of `x`: P[t] = W[:, :, ky, kx] . x (t = 3 ky + kx) at the low
    resolution; per low-resolution row the horizontal sums over kx of the up-sampled planes' columns s + kx - 1, then the
    vertical sums over ky of the up-sampled rows r + ky - 1, rows outside [0, 2H) dropped."""
    B, Ci, H, W = x.shape
    Co = w.shape[0]
    P = torch.einsum("toc,bchw->btohw", w.permute(2, 3, 0, 1).reshape(9, Co, Ci), x)      # [B, 9, Co, H, W]

    def up_w(p):      # [..., H, W] -> [..., H, 2W], ring
        ev = 0.25 * torch.roll(p, 1, -1) + 0.75 * p
        od = 0.75 * p + 0.25 * torch.roll(p, -1, -1)
        return torch.stack([ev, od], -1).reshape(*p.shape[:-1], 2 * W)

    def up_h(p):      # [..., H, W'] -> [..., 2H, W'], zeros outside
        z = torch.zeros_like(p[..., :1, :])
        pm, pp = torch.cat([z, p[..., :-1, :]], -2), torch.cat([p[..., 1:, :], z], -2)
        ev = 0.25 * pm + 0.75 * p
        od = 0.75 * p + 0.25 * pp
        return torch.stack([ev, od], -2).reshape(*p.shape[:-2], 2 * H, p.shape[-1])

    out = torch.zeros((B, Co, 2 * H, 2 * W), dtype=x.dtype)
    for ky in range(3):
        hz = sum(torch.roll(up_w(P[:, 3 * ky + kx]), -(kx - 1), -1) for kx in range(3))      # column s + kx - 1 (ring)
        U = up_h(hz)                                                                         # [B, Co, 2H, 2W]
        d = ky - 1                                                                           # row r + d, zero outside
        if d == 0:
            out += U
        elif d > 0:
            out[:, :, :-1] += U[:, :, 1:]
        else:
            out[:, :, 1:] += U[:, :, :-1]
    return out + (b.view(1, -1, 1, 1) if b is not None else 0.0)


def _ref_layer(Ci, Co, salt):
    from lidargen.models.unets import ops

    return seeded_fill(ops.Conv2d(Ci, Co, 3, 1, 1, ring=True), salt=salt)


# ------------------------------------------------------------------------------------------------ CPU
@pytest.mark.parametrize("tag", sorted(SHAPES))
def test_folded_algebra_and_oracle_vs_reference_golden(golden, tag):
    from oracle import denoiser as D

    B, Ci, Co, H, W, salt = SHAPES[tag]
    conv = _ref_layer(Ci, Co, salt)
    x = seeded_randn(B, Ci, H, W, seed=400 + salt) + 0.3
    w, b = conv.weight.detach(), conv.bias.detach()
    want = T(golden("fold_up")[f"{tag}_y"])
    ora = D.conv_ring(D.resample_up2(x), w, b)
    assert rel_l2(ora[..., ::2], want) < 1e-6                        # the oracle is pinned on the reference
    got = folded_model(x.double(), w.double(), b.double())
    assert rel_l2(got[..., ::2], want) < 1e-6
    assert float((got - ora.double()).abs().max()) < 1e-5 * float(ora.abs().max())
    for r in (0, 1, 2 * H - 2, 2 * H - 1):                           # the border rows on their own
        assert rel_l2(got[:, :, r], ora[:, :, r].double()) < 1e-6, r


def test_combine_kernel_coefficients_restated():
    """The coefficient table of up2_combine9_kernel (csrc/upfold.hip: hz0 / hz1 per low-resolution row, the three-row window
    with its two dropped border terms) restated lane by lane in float64, against the plane-wise model above."""
    torch.manual_seed(0)
    B, Ci, Co, H, W = 1, 4, 3, 5, 8
    x = torch.randn(B, Ci, H, W, dtype=torch.float64)
    w = torch.randn(Co, Ci, 3, 3, dtype=torch.float64)
    b = torch.randn(Co, dtype=torch.float64)
    want = folded_model(x, w, b)
    P = torch.einsum("toc,bchw->btohw", w.permute(2, 3, 0, 1).reshape(9, Co, Ci), x)[0]          # [9, Co, H, W]

    def hz0(p0m, p0, p1m, p1, p2, p2p):      # output column 2j
        return .75 * p0m + .25 * p0 + .25 * p1m + .75 * p1 + .75 * p2 + .25 * p2p

    def hz1(p0m, p0, p1, p1p, p2, p2p):      # output column 2j + 1
        return .25 * p0m + .75 * p0 + .75 * p1 + .25 * p1p + .25 * p2 + .75 * p2p

    out = torch.zeros(Co, 2 * H, 2 * W, dtype=torch.float64)
    j0 = torch.arange(0, W, 2)                                    # a lane owns columns j0, j0 + 1
    jm, jp = (j0 - 1) % W, (j0 + 2) % W

    def horiz(i):                                                 # -> [3 ky][4 columns][Co, lanes]
        if i < 0 or i >= H:
            return [[torch.zeros(Co, len(j0), dtype=torch.float64)] * 4 for _ in range(3)]
        res = []
        for ky in range(3):
            t0, t1, t2 = 3 * ky, 3 * ky + 1, 3 * ky + 2
            L, A, Bv, R = (lambda t: P[t, :, i, jm]), (lambda t: P[t, :, i, j0]), (lambda t: P[t, :, i, j0 + 1]), (lambda t: P[t, :, i, jp])
            res.append([hz0(L(t0), A(t0), L(t1), A(t1), A(t2), Bv(t2)), hz1(L(t0), A(t0), A(t1), Bv(t1), A(t2), Bv(t2)),
                        hz0(A(t0), Bv(t0), A(t1), Bv(t1), Bv(t2), R(t2)), hz1(A(t0), Bv(t0), Bv(t1), R(t1), Bv(t2), R(t2))])
        return res

    for i in range(H):
        Hm, H0, Hp = horiz(i - 1), horiz(i), horiz(i + 1)
        ft, fb = float(i > 0), float(i + 1 < H)
        for c in range(4):
            top = ft * (.25 * H0[0][c] + .75 * Hm[0][c]) + .25 * Hm[1][c] + .75 * H0[1][c] + .75 * H0[2][c] + .25 * Hp[2][c]
            bot = fb * (.75 * Hp[2][c] + .25 * H0[2][c]) + .25 * Hm[0][c] + .75 * H0[0][c] + .75 * H0[1][c] + .25 * Hp[1][c]
            out[:, 2 * i, 2 * j0 + c] = top + b[:, None]
            out[:, 2 * i + 1, 2 * j0 + c] = bot + b[:, None]
    assert float((out - want[0]).abs().max()) < 1e-12


def test_up9_weight_layout_and_eligibility():
    from lidarcrafter_amd import ops as K

    w = seeded_randn(5, 3, 3, 3, seed=9)
    w9 = K.up9_weight(w)
    assert tuple(w9.shape) == (45, 3, 1, 1)
    for ky in range(3):
        for kx in range(3):
            assert torch.equal(w9[(3 * ky + kx) * 5:(3 * ky + kx + 1) * 5, :, 0, 0], w[:, :, ky, kx])
    assert K.can_fold_up(128, 128, 16, 512) and K.can_fold_up(512, 512, 4, 128) and K.can_fold_up(256, 256, 32, 1024)
    assert not K.can_fold_up(64, 64, 16, 512)         # below LC_FOLD_UP_MIN_CI: bound by the nine planes' bytes
    assert not K.can_fold_up(128, 128, 8, 64)         # no whole 128-column segment
    assert not K.can_fold_up(144, 128, 8, 128)        # the pre-split 1x1 kernel needs Ci % 32 == 0


def test_abi_exports_and_slots():
    from lidarcrafter_amd import _lib

    h = _lib.lib()
    assert h.lc_up2_combine9_stats_slots(16, 512) == 16 * 4 and h.lc_up2_combine9_stats_slots(4, 128) == 4
    assert h.lc_up2_combine9_stats_slots(4, 100) == 0 and h.lc_up2_combine9_stats_slots(0, 128) == 0
    assert h.lc_up2_combine9_fwd(None, 0, None, None, 0, 1, 1, 1, 128, None, None) == -1      # LC_EINVAL
    assert h.lc_split_act_fwd(None, 0, None, 1, 16, 1, 4, None, None) == -1


# ------------------------------------------------------------------------------------------------ GPU
@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


def _layer(Ci, Co, salt, dev):
    return _ref_layer(Ci, Co, salt).to(dev)


def _fold(x, conv, out=None, emit_stats=False, norm=None):
    """ops.conv_up2 of `x` through a fresh PackedConv (min-Ci gate lifted by the caller where needed)."""
    from lidarcrafter_amd import ops as K

    pk = K.PackedConv("test.up9")
    xs = K.split_act(x, pk)
    return K.conv_up2(xs, pk, K.up9_weight(conv.weight), conv.bias, out=out, emit_stats=emit_stats)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", sorted(SHAPES))
def test_fold_up_vs_reference_golden(dev, golden, tag, monkeypatch):
    from lidarcrafter_amd import ops as K

    monkeypatch.setattr(K, "FOLD_UP_MIN_CI", 32)
    B, Ci, Co, H, W, salt = SHAPES[tag]
    conv = _layer(Ci, Co, salt, dev)
    x = (seeded_randn(B, Ci, H, W, seed=400 + salt) + 0.3).to(dev)
    y = _fold(x, conv)
    want = T(golden("fold_up")[f"{tag}_y"])
    assert tuple(y.shape) == (B, Co, 2 * H, 2 * W)
    assert rel_l2(y[..., ::2], want) < 2e-6, rel_l2(y[..., ::2], want)
    for r in (0, 2 * H - 1):          # the rows whose border tap is dropped must not hide behind the interior
        assert rel_l2(y[:, :, r, ::2], want[:, :, r]) < 4e-6, (r, rel_l2(y[:, :, r, ::2], want[:, :, r]))


@pytest.mark.gpu
@pytest.mark.parametrize("B,Ci,Co,H,W", [(2, 32, 64, 4, 128), (8, 512, 512, 4, 128), (3, 96, 40, 3, 256),
                                         (8, 256, 256, 8, 256), (2, 128, 128, 16, 512), (1, 128, 128, 32, 1024),
                                         (1, 64, 8, 1, 128)])
def test_fold_up_vs_oracle_and_unfolded_route(dev, B, Ci, Co, H, W, monkeypatch):
    from lidarcrafter_amd import ops as K
    from oracle import denoiser as D

    monkeypatch.setattr(K, "FOLD_UP_MIN_CI", 32)
    conv = _layer(Ci, Co, 70 + Ci, dev)
    x = seeded_randn(B, Ci, H, W, seed=11 * Ci + H) * 1.3 - 0.2
    # a channel slice of a wider buffer as the input (batch-strided), a channel slice of a concat buffer as the output
    wide = torch.empty((B, Ci + 16, H, W), device=dev)
    xd = wide[:, 16:]
    xd.copy_(x.to(dev))
    y = _fold(xd, conv)
    unfolded = conv(K.resample2x(xd, up=True))
    assert rel_l2(y, unfolded) < 2e-6, rel_l2(y, unfolded)
    if B * Ci * Co * H * W <= 2 * 128 * 128 * 16 * 512:
        ref = D.conv_ring(D.resample_up2(x), conv.weight.detach().cpu(), conv.bias.detach().cpu())
        assert rel_l2(y, ref) < 2e-6, rel_l2(y, ref)
        for r in (0, 1, 2 * H - 2, 2 * H - 1):
            assert rel_l2(y[:, :, r], ref[:, :, r]) < 4e-6, r
    obuf = torch.full((B, Co + 8, 2 * H, 2 * W), 7.0, device=dev)
    _fold(xd, conv, out=obuf[:, 8:])
    assert torch.equal(obuf[:, 8:], y) and bool((obuf[:, :8] == 7.0).all())


@pytest.mark.gpu
@pytest.mark.parametrize("B,C,H,W", [(2, 32, 5, 128), (1, 128, 16, 512), (3, 48, 2, 64)])
def test_split_act_is_the_apply_passes_split(dev, B, C, H, W):
    """lc_split_act_fwd writes exactly the planes the GroupNorm apply + split pass writes for an identity normalisation:
    the pre-split 3x3 conv of both operands gives bit-equal results; hi + lo reproduces x * x_scale to 2^-22."""
    from lidarcrafter_amd import ops as K

    x = (seeded_randn(B, C, H, W, seed=5 + C) * 2.0 + 0.1).to(dev)
    pk = K.PackedConv("test.split")
    xs = K.split_act(x, pk)
    n = B * 2 * (C // 8) * H * W
    planes = xs.buf[:n].view(B, 2, C // 8, H * W, 8).float()
    back = (planes[:, 0] + planes[:, 1]).permute(0, 1, 3, 2).reshape(B, C, H, W) / pk.x_scale
    assert float((back - x).abs().max()) <= 2.0 ** -21 * float(x.abs().max())
    assert not K.range_poll(dev)


@pytest.mark.gpu
@pytest.mark.parametrize("B,Ci,Co,H,W,G", [(2, 128, 128, 8, 256, 32), (1, 64, 256, 4, 128, 32), (2, 32, 64, 2, 384, 8)])
def test_fold_up_statistics_feed_groupnorm(dev, B, Ci, Co, H, W, G, monkeypatch):
    """The per-channel entries of the combine pass against the statistics pass on the same tensor, through the
    consumers the models use (apply, apply + split)."""
    from lidarcrafter_amd import ops as K
    from oracle import denoiser as D

    monkeypatch.setattr(K, "FOLD_UP_MIN_CI", 32)
    conv = _layer(Ci, Co, 80 + Ci, dev)
    x = (seeded_randn(B, Ci, H, W, seed=600 + Ci) + 0.25).to(dev)
    y = _fold(x, conv, emit_stats=True)
    h = y._lc_gnstats[(0, Co)]
    assert h.unit == 1 and tuple(h.buf.shape) == (B, Co, H * (W // 128), 4)
    e = h.buf.double()
    assert float(e[..., 1].sum()) == B * Co * 4 * H * W                     # every stored value counted once
    tot = (e[..., 0] * e[..., 1] + e[..., 2]).sum(-1)                       # sum per (sample, channel)
    want = y.double().sum((2, 3))
    assert float((tot - want).abs().max()) < 1e-2 * max(1.0, float(want.abs().max()))
    sq = (e[..., 3] + 2 * e[..., 0] * e[..., 2] + e[..., 1] * e[..., 0] ** 2).sum(-1)
    want2 = (y.double() ** 2).sum((2, 3))
    assert float(((sq - want2) / want2).abs().max()) < 1e-4
    ga, be = (1 + 0.1 * seeded_randn(Co, seed=78)).to(dev), (0.1 * seeded_randn(Co, seed=79)).to(dev)
    y2 = y.clone()
    assert not getattr(y2, "_lc_gnstats", None)
    a1, a2 = K.groupnorm(y, G, 1e-6, ga, be, act_silu=True), K.groupnorm(y2, G, 1e-6, ga, be, act_silu=True)
    assert rel_l2(a1, a2) < 2e-6, rel_l2(a1, a2)
    ref = torch.nn.functional.silu(D.group_norm(y.cpu(), G, ga.cpu(), be.cpu(), 1e-6))
    assert rel_l2(a1, ref) < 2e-6
    # ... and through the pre-split apply pass feeding a 3x3 conv
    c2 = _layer(Co, 64, 99, dev)
    s1 = K.groupnorm(y, G, 1e-6, ga, be, act_silu=True, split_for=c2._packed)
    assert isinstance(s1, K.SplitAct)
    z1 = c2(s1)
    z2 = c2(a2)
    assert rel_l2(z1, z2) < 2e-6, rel_l2(z1, z2)


@pytest.mark.gpu
@pytest.mark.parametrize("B,C,H,W", [(4, 64, 4, 256), (6, 32, 2, 128), (8, 128, 16, 512)])
def test_fold_up_in_slabs_of_samples_is_bit_equal(dev, B, C, H, W, monkeypatch):
    """A batch whose nine planes would not stay inside the Infinity Cache runs as slabs of samples through the same two
    kernels (ops.FOLD_UP_SLAB_MB): same bits, same statistics entries."""
    from lidarcrafter_amd import ops as K

    monkeypatch.setattr(K, "FOLD_UP_MIN_CI", 32)
    conv = _layer(C, C, 120 + C, dev)
    x = (seeded_randn(B, C, H, W, seed=700 + C) * 0.8 + 0.1).to(dev)
    monkeypatch.setattr(K, "FOLD_UP_SLAB_MB", 0)
    y0 = _fold(x, conv, emit_stats=True)
    per_sample_mb = 4.0 * 13 * C * H * W / 2 ** 20
    for slab_samples in (1, 2, 3):
        monkeypatch.setattr(K, "FOLD_UP_SLAB_MB", max(1, int(per_sample_mb * slab_samples + 0.999)))
        y1 = _fold(x, conv, emit_stats=True)
        assert torch.equal(y0, y1), slab_samples
        assert torch.equal(y0._lc_gnstats[(0, C)].buf, y1._lc_gnstats[(0, C)].buf), slab_samples


@pytest.mark.gpu
def test_fold_up_range_safety(dev, monkeypatch):
    """Operands the default pre-scale cannot hold (|x| ~ 3e4): the split pass publishes max |x * scale| into the layer's
    range record, the poll re-derives the scale, and the recomputed result is fp32-class."""
    from lidarcrafter_amd import ops as K
    from oracle import denoiser as D

    monkeypatch.setattr(K, "FOLD_UP_MIN_CI", 32)
    conv = _layer(32, 64, 91, dev)
    x = seeded_randn(1, 32, 4, 128, seed=92) * 3e4
    xd = x.to(dev)
    pk = K.PackedConv("test.range")
    w9 = K.up9_weight(conv.weight)
    K.range_poll(dev)
    K.conv_up2(K.split_act(xd, pk), pk, w9, conv.bias)
    assert K.range_poll(dev), "the saturated layer must be reported"
    y = K.conv_up2(K.split_act(xd, pk), pk, w9, conv.bias)
    assert not K.range_poll(dev)
    ref = D.conv_ring(D.resample_up2(x), conv.weight.detach().cpu(), conv.bias.detach().cpu())
    assert rel_l2(y, ref) < 2e-6, rel_l2(y, ref)


@pytest.mark.gpu
def test_models_take_the_folded_route(dev, monkeypatch):
    """EfficientUNet's Block.upsample and LayoutUnetV1's up-sampling ResBlocks go through ops.conv_up2 where the shape
    allows, and the whole denoisers agree with the unfolded route (LC_FOLD_UP=0) to fp32-class accuracy."""
    from lidarcrafter_amd import ops as K
    from tests.test_hip_parity import _uncond

    monkeypatch.setattr(K, "FOLD_UP_MIN_CI", 32)
    m = _uncond(32, (8, 1024), dev)
    x = seeded_randn(2, 2, 8, 1024, seed=21).to(dev)
    lam = torch.tensor([-4.0, 2.5], device=dev)
    calls = []
    orig = K.conv_up2
    monkeypatch.setattr(K, "conv_up2", lambda *a, **k: (calls.append(a[0].shape), orig(*a, **k))[1])
    with torch.no_grad():
        y1 = m(x, lam).clone()
    assert len(calls) >= 2, f"Block.upsample did not take the folded route: {calls}"
    monkeypatch.setattr(K, "FOLD_UP", False)
    n = len(calls)
    with torch.no_grad():
        y0 = m(x, lam).clone()
    assert len(calls) == n
    assert rel_l2(y1, y0) < 5e-6, rel_l2(y1, y0)


@pytest.mark.gpu
def test_layout_model_takes_the_folded_route(dev, monkeypatch):
    """LayoutUnetV1 (box-layout-v6 at 32 x 1024): its three up-sampling ResBlocks go through ops.conv_up2 (the GroupNorm apply
    pass writes the low-resolution operand pre-split, x is up-sampled beside it), and the forward agrees with the reference's
    order of operations (LC_FOLD_UP=0) to fp32-class accuracy.  The reference-pinned goldens of this model
    (test_cond_full_golden, test_c3_b8_golden) run through the fold by default."""
    from lidarcrafter_amd import ops as K
    from lidarcrafter_amd.testing import synth_layout_batch
    from lidargen.utils import inference
    from lidargen.utils.configs import __all__ as C

    ddpm, model, _ = inference.load_model_duffusion_training(C["nuscenes-box-layout-v6"]())
    seeded_fill(model, salt=200), seeded_fill(ddpm.condition_model, salt=201)
    ddpm = ddpm.eval().to(dev)
    batch = {k: v.to(dev) for k, v in synth_layout_batch(2, 32, 1024, seed=53).items()}
    calls = []
    orig = K.conv_up2
    monkeypatch.setattr(K, "conv_up2", lambda *a, **k: (calls.append(a[0].shape), orig(*a, **k))[1])
    x = seeded_randn(2, 2, 32, 1024, seed=54).to(dev)
    with torch.no_grad():
        cond = ddpm.condition_model(batch)
        tc = {"time_condition": torch.tensor([-0.5, 1.0], device=dev), "other_condition": cond}
        y1 = ddpm.model(x, tc).clone()
    assert sorted(s[1] for s in calls) == [128, 256, 512], calls
    n = len(calls)
    monkeypatch.setattr(K, "FOLD_UP", False)
    with torch.no_grad():
        y0 = ddpm.model(x, tc).clone()
    assert len(calls) == n
    assert rel_l2(y1, y0) < 5e-6, rel_l2(y1, y0)


@pytest.mark.gpu
@pytest.mark.parametrize("B,C,H,W", [(2, 32, 4, 128), (3, 64, 1, 256), (8, 128, 16, 512), (2, 32, 5, 384)])
def test_fold_up_with_the_skip_paths_upsampling_in_the_same_launch(dev, B, C, H, W, monkeypatch):
    """lc_up2_combine9_xup_fwd: y as without it (same bits, same entries), and the second output bit-identical to
    lc_resample2x_fwd of the second tensor -- in one slab and in slabs of samples."""
    from lidarcrafter_amd import ops as K

    monkeypatch.setattr(K, "FOLD_UP_MIN_CI", 32)
    conv = _layer(C, C, 130 + C, dev)
    a = (seeded_randn(B, C, H, W, seed=800 + C) * 0.7).to(dev)
    wide = torch.empty((B, C + 8, H, W), device=dev)
    x = wide[:, 8:]                                   # batch-strided second tensor
    x.copy_((seeded_randn(B, C, H, W, seed=900 + C) * 1.5 + 0.3).to(dev))
    pk = K.PackedConv("test.xup")
    w9 = K.up9_weight(conv.weight)
    y0 = K.conv_up2(K.split_act(a, pk), pk, w9, conv.bias, emit_stats=True)
    want = K.resample2x(x, up=True)
    per_sample_mb = 4.0 * 13 * C * H * W / 2 ** 20
    for slab in (0, max(1, int(per_sample_mb + 0.999))):
        monkeypatch.setattr(K, "FOLD_UP_SLAB_MB", slab)
        y1, x1 = K.conv_up2(K.split_act(a, pk), pk, w9, conv.bias, emit_stats=True, up_also=x)
        assert torch.equal(y0, y1) and torch.equal(y0._lc_gnstats[(0, C)].buf, y1._lc_gnstats[(0, C)].buf), slab
        assert torch.equal(x1, want), (slab, float((x1 - want).abs().max()))
