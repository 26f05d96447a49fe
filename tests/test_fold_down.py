"""Block.downsample folded (round 6): `Conv2d(3x3, ring) -> Resample(down=2)` of the reference's EfficientUNet
(efficient_unet.py:132-135, ops.py:52-173) evaluated as ONE stride-2 convolution of the FIR-pre-filtered input
(lidarcrafter_amd/csrc/conv_f16x2_s2.hip, ops.conv_down2).  `pytest -m gpu`.  Checked against
  * the reference's own modules (tests/golden/fold_down.npz, make_fixtures.py::sec_fold_down),
  * the oracle's conv_ring + resample_down2 on more shapes (the image's first and last row in one tile, ragged output
    channel blocks, batch-strided inputs, persistent blocks), <= 2e-6 rel-L2 like every f16x2 convolution,
  * the unfolded HIP route (LC_FOLD_DOWN=0's order: full-resolution conv, then the resampling pass),
and the statistics entries the stride-2 conv leaves for the GroupNorm behind it."""
import numpy as np
import pytest
import torch

from lidarcrafter_amd.testing import rel_l2, seeded_fill, seeded_randn

pytestmark = pytest.mark.gpu
T = torch.from_numpy


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


def _layer(Ci, Co, salt, dev):
    from lidargen.models.unets import ops

    return seeded_fill(ops.Conv2d(Ci, Co, 3, 1, 1, ring=True), salt=salt).to(dev)


@pytest.mark.parametrize("tag,B,Ci,Co,H,W,salt", [("a", 2, 16, 32, 8, 128, 31), ("b", 1, 32, 72, 4, 256, 32),
                                                   ("c", 1, 64, 128, 16, 128, 33)])
def test_fold_down_vs_reference_golden(dev, golden, tag, B, Ci, Co, H, W, salt):
    from lidarcrafter_amd import ops as K

    conv = _layer(Ci, Co, salt, dev)
    x = (seeded_randn(B, Ci, H, W, seed=300 + salt) + 0.3).to(dev)
    y = K.conv_down2(x, conv._packed, conv.weight, conv.bias)
    want = T(golden("fold_down")[f"{tag}_y"])
    assert tuple(y.shape) == tuple(want.shape)
    assert rel_l2(y, want) < 2e-6, rel_l2(y, want)
    # every row on its own: the border rows (bias factor, variant rows) must not hide behind the interior
    for r in (0, H // 2 - 1):
        assert rel_l2(y[:, :, r], want[:, :, r]) < 4e-6, (r, rel_l2(y[:, :, r], want[:, :, r]))


@pytest.mark.parametrize("B,Ci,Co,H,W", [(2, 16, 64, 4, 128), (1, 64, 128, 32, 1024), (3, 48, 40, 6, 256),
                                         (2, 128, 256, 16, 512), (8, 256, 512, 8, 256), (1, 16, 8, 64, 2048)])
def test_fold_down_vs_oracle_and_unfolded_route(dev, B, Ci, Co, H, W):
    from lidarcrafter_amd import ops as K
    from oracle import denoiser as D

    conv = _layer(Ci, Co, 40 + Ci, dev)
    x = seeded_randn(B, Ci, H, W, seed=7 * Ci + H) * 1.3 - 0.2
    # a channel slice of a wider buffer (the down conv reads the skip half of a concat buffer)
    wide = torch.empty((B, Ci + 16, H, W), device=dev)
    xd = wide[:, 16:]
    xd.copy_(x.to(dev))
    y = K.conv_down2(xd, conv._packed, conv.weight, conv.bias)
    unfolded = K.resample2x(conv(xd.contiguous()), up=False)
    assert rel_l2(y, unfolded) < 2e-6, rel_l2(y, unfolded)
    if B * Ci * Co * H * W <= 2 * 128 * 256 * 16 * 512:
        ref = D.resample_down2(D.conv_ring(x, conv.weight.detach().cpu(), conv.bias.detach().cpu()))
        assert rel_l2(y, ref) < 2e-6, rel_l2(y, ref)
        for r in (0, H // 2 - 1):
            assert rel_l2(y[:, :, r], ref[:, :, r]) < 4e-6, r
    # into a channel slice of a wider output buffer
    obuf = torch.full((B, Co + 8, H // 2, W // 2), 7.0, device=dev)
    K.conv_down2(xd, conv._packed, conv.weight, conv.bias, out=obuf[:, 8:])
    assert torch.equal(obuf[:, 8:], y) and bool((obuf[:, :8] == 7.0).all())


@pytest.mark.parametrize("B,Ci,Co,H,W,G", [(2, 64, 128, 8, 256, 8), (1, 16, 32, 16, 128, 8), (2, 32, 64, 4, 384, 32)])
def test_fold_down_statistics_feed_groupnorm(dev, B, Ci, Co, H, W, G):
    """The entries of the stride-2 conv's epilogue (octets, or quads for 4 channels per group) against the statistics
    pass on the same tensor, through the consumers the model uses (apply, apply + split)."""
    from lidarcrafter_amd import ops as K
    from oracle import denoiser as D

    conv = _layer(Ci, Co, 60 + Ci, dev)
    x = (seeded_randn(B, Ci, H, W, seed=500 + Ci) + 0.25).to(dev)
    cpg = Co // G
    unit = 8 if cpg % 8 == 0 else 4
    y = K.conv_down2(x, conv._packed, conv.weight, conv.bias, emit_stats=unit)
    h = y._lc_gnstats[(0, Co)]
    assert h.unit == unit and tuple(h.buf.shape) == (B, Co // unit, h.slots, 4)
    e = h.buf.double()
    assert float(e[..., 1].sum()) == B * Co * (H // 2) * (W // 2)          # every stored value counted once
    tot = (e[..., 0] * e[..., 1] + e[..., 2]).sum(-1)                       # sum per (sample, channel unit)
    want = y.double().view(B, Co // unit, unit * (H // 2) * (W // 2)).sum(-1)
    assert float((tot - want).abs().max()) < 1e-2 * max(1.0, float(want.abs().max()))
    ga, be = (1 + 0.1 * seeded_randn(Co, seed=78)).to(dev), (0.1 * seeded_randn(Co, seed=79)).to(dev)
    y2 = y.clone()
    assert not getattr(y2, "_lc_gnstats", None)
    a1, a2 = K.groupnorm(y, G, 1e-6, ga, be, act_silu=True), K.groupnorm(y2, G, 1e-6, ga, be, act_silu=True)
    assert rel_l2(a1, a2) < 2e-6, rel_l2(a1, a2)
    ref = torch.nn.functional.silu(D.group_norm(y.cpu(), G, ga.cpu(), be.cpu(), 1e-6))
    assert rel_l2(a1, ref) < 2e-6


def test_fold_down_range_safety(dev):
    """Inputs the default pre-scale cannot hold (|x| ~ 3e4: x * 16 saturates fp16): the pre-filter publishes max |F * scale|
    into the layer's range record, the poll re-derives the scale, and the recomputed result is fp32-class."""
    from lidarcrafter_amd import ops as K
    from oracle import denoiser as D

    conv = _layer(32, 64, 91, dev)
    x = seeded_randn(1, 32, 8, 128, seed=92) * 3e4
    xd = x.to(dev)
    K.range_poll(dev)
    y = K.conv_down2(xd, conv._packed, conv.weight, conv.bias)
    bad = K.range_poll(dev)
    assert bad, "the saturated layer must be reported"
    y = K.conv_down2(xd, conv._packed, conv.weight, conv.bias)
    assert not K.range_poll(dev)
    ref = D.resample_down2(D.conv_ring(x, conv.weight.detach().cpu(), conv.bias.detach().cpu()))
    assert rel_l2(y, ref) < 2e-6, rel_l2(y, ref)


def test_block_takes_the_folded_route(dev, monkeypatch):
    """EfficientUNet's Block.downsample goes through ops.conv_down2 where the shape allows, and the whole denoiser
    agrees with the unfolded route (LC_FOLD_DOWN=0) to fp32-class accuracy."""
    from lidarcrafter_amd import ops as K
    from tests.test_hip_parity import _uncond

    m = _uncond(16, (8, 128), dev)
    x = seeded_randn(2, 2, 8, 128, seed=21).to(dev)
    lam = torch.tensor([-4.0, 2.5], device=dev)
    calls = []
    orig = K.conv_down2
    monkeypatch.setattr(K, "conv_down2", lambda *a, **k: (calls.append(a[0].shape), orig(*a, **k))[1])
    with torch.no_grad():
        y1 = m(x, lam).clone()
    assert len(calls) >= 1, "no Block took the folded route"
    monkeypatch.setattr(K, "FOLD_DOWN", False)
    n = len(calls)
    with torch.no_grad():
        y0 = m(x, lam).clone()
    assert len(calls) == n
    assert rel_l2(y1, y0) < 5e-6, rel_l2(y1, y0)
