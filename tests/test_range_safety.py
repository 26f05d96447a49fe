"""Range safety of the f16x2-split convolution (VERDICT r01 "weak #2").  fp16 saturates at 65504
and loses its low bits below 2^-14: the kernels therefore pre-scale x and w by powers of two held
in device memory, track max |x * x_scale| per layer, and `ops.range_poll` moves a layer's scale
and reports the launch INVALID when its operands saturated (or vanished) -- callers recompute.
These tests drive magnitudes 1e-6 ... 1e8 through the conv and hold the result to the same
fp32-class bound as at unit scale (<= 2e-6 rel-L2 vs fp64); silent clipping must be impossible.
`pytest -m gpu`."""
import warnings

import pytest
import torch

from lidarcrafter_amd.testing import rel_l2, seeded_randn

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


def ring_conv_f64(x, w):
    xp = torch.nn.functional.pad(torch.cat([x[..., -1:], x, x[..., :1]], -1).double(), (0, 0, 1, 1))
    return torch.nn.functional.conv2d(xp, w.double())


def checked_conv(K, x, pk, w, **kw):
    """conv + poll until no layer reports invalid operands; returns (y, number of invalid passes)."""
    n_bad = 0
    for _ in range(5):
        y = K.conv2d_ring(x, pk, w, precision="f16x2", **kw)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)
            bad = K.range_poll(x.device)
        if not bad:
            return y, n_bad
        n_bad += 1
    raise AssertionError(f"pre-scale did not converge: {bad}")


@pytest.mark.parametrize("xs", [1e-6, 1e-4, 1e-2, 1.0, 1e3, 1e4, 1e5, 1e8])
@pytest.mark.parametrize("ws", [1e-5, 1e-3, 1.0, 1e3])
@pytest.mark.parametrize("cfg", [0, 3, 23])
def test_conv_any_magnitude(dev, xs, ws, cfg):
    from lidarcrafter_amd import ops as K

    x = seeded_randn(2, 128, 8, 64, seed=40) * xs
    w = seeded_randn(64, 128, 3, 3, seed=41) * ws / 34.0
    ref = ring_conv_f64(x, w)
    pk = K.PackedConv("range_test")
    K.range_poll(dev)
    y, n_bad = checked_conv(K, x.to(dev), pk, w.to(dev), tile_cfg=cfg)
    e = rel_l2(y, ref)
    assert e < 2e-6, (xs, ws, cfg, e, pk.x_scale)
    # magnitudes the default pre-scale (16) cannot hold MUST have been reported, not clipped
    amax = float(x.abs().max())
    if amax * 16 >= 65504 or amax * 16 < 2 ** -3:
        assert n_bad >= 1, "saturated / vanishing operands went unreported"
    # further calls at the same magnitude are clean, and bit-stable once the scale has settled (a
    # first poll may move a merely sub-optimal scale without invalidating what was computed)
    y2, n_bad2 = checked_conv(K, x.to(dev), pk, w.to(dev), tile_cfg=cfg)
    y3, n_bad3 = checked_conv(K, x.to(dev), pk, w.to(dev), tile_cfg=cfg)
    assert n_bad2 == 0 and n_bad3 == 0 and rel_l2(y2, ref) < 2e-6 and torch.equal(y2, y3)


def test_conv_outlier_dynamic_range(dev):
    """One 1e5 spike in an O(1) tensor: the scale follows the spike, the bulk keeps fp32-class
    accuracy relative to the output (error floor 2^-25 / (max|x| * x_scale) relative to max|x|)."""
    from lidarcrafter_amd import ops as K

    x = seeded_randn(1, 64, 8, 64, seed=42)
    x[0, 3, 2, 10] = 1.0e5
    w = seeded_randn(64, 64, 3, 3, seed=43) / 24.0
    pk = K.PackedConv("spike")
    K.range_poll(dev)
    y, n_bad = checked_conv(K, x.to(dev), pk, w.to(dev))
    assert n_bad >= 1
    assert rel_l2(y, ring_conv_f64(x, w)) < 2e-6


def test_fused_groupnorm_input_is_tracked_after_the_norm(dev):
    """With the GroupNorm fused into the conv's staging the split sees the NORMALISED value: a 1e6
    input tensor must not trip the range check, a 1e6 AdaGN scale must."""
    from lidarcrafter_amd import ops as K

    x = seeded_randn(2, 64, 8, 64, seed=44) * 1.0e6
    w = seeded_randn(64, 64, 3, 3, seed=45) / 24.0
    pk = K.PackedConv("fused")
    K.range_poll(dev)
    xd = x.to(dev)
    K.conv2d_ring(xd, pk, w.to(dev), precision="f16x2", gn_coeffs=K.groupnorm_stats(xd, 8, 1e-6))
    assert K.range_poll(dev) == []
    scale = torch.full((2, 64), 1.0e6, device=dev)
    shift = torch.zeros((2, 64), device=dev)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        K.conv2d_ring(xd, pk, w.to(dev), precision="f16x2",
                      gn_coeffs=K.groupnorm_stats(xd, 8, 1e-6, scale=scale, shift=shift))
        bad = K.range_poll(dev)
    assert len(bad) == 1 and bad[0].invalid and bad[0].layer == "fused"


def test_model_forward_recomputes_instead_of_clipping(dev):
    """A denoiser whose first conv sees 1e5-scale inputs: the standalone forward is range-checked
    (ops.range_checked) and returns the fp32-class result; the sampler does the same per run."""
    from oracle import denoiser as D
    from tests.test_hip_parity import _uncond

    m = _uncond(16, (8, 64), dev)
    with torch.no_grad():
        m.in_conv.weight.mul_(1.0e-5)          # keep the network's activations in range ...
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    x = seeded_randn(2, 2, 8, 64, seed=46) * 1.0e5     # ... with an input 1e5 times too large
    lam = torch.tensor([-2.0, 1.0])
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        with torch.no_grad():
            y = m(x.to(dev), lam.to(dev))
    assert any("in_conv" in str(w_.message) for w_ in rec), [str(w_.message) for w_ in rec]
    r = rel_l2(y, D.efficient_unet_forward(sd, x, lam))
    assert r < 2e-5, r


def test_f32_kernel_needs_no_range_state(dev):
    from lidarcrafter_amd import ops as K

    x = seeded_randn(1, 32, 4, 64, seed=47) * 1.0e7
    w = seeded_randn(32, 32, 3, 3, seed=48) / 17.0
    K.range_poll(dev)
    y = K.conv2d_ring(x.to(dev), K.PackedConv(), w.to(dev), precision="f32")
    assert K.range_poll(dev) == []
    assert rel_l2(y, ring_conv_f64(x, w)) < 1e-6


@pytest.mark.parametrize("mag", [1e-12, 3e-5, 0.7, 1.0, 4097.0, 6.5e4, 1e9])
def test_range_from_tensor_device_side(dev, mag):
    """lc_range_from_tensor (training): the record is set on the device to the power of two that
    puts max|x| into [2^12, 2^13); a conv through it is fp32-class at any magnitude, with no poll."""
    import math

    from lidarcrafter_amd import ops as K

    x = seeded_randn(2, 32, 4, 64, seed=50) * mag
    w = seeded_randn(32, 32, 3, 3, seed=51) / 17.0
    pk = K.PackedConv("train.range")
    K.range_poll(dev)
    xd = x.to(dev)
    K.range_from_tensor(xd, pk)
    y = K.conv2d_ring(xd, pk, w.to(dev), precision="f16x2")
    assert rel_l2(y, ring_conv_f64(x, w)) < 2e-6
    rec = pk._arena.buf.view(-1, 4)[pk._slot].cpu()
    amax = float(x.abs().max())
    assert float(rec[0]) == 2.0 ** (13 - math.frexp(amax)[1])      # amax * scale in [2^12, 2^13)
    assert 4096.0 <= amax * float(rec[0]) < 8192.0
    assert float(rec[1]) == 1.0 / float(rec[0]) and float(rec[3]) == 0.0
    assert K.range_poll(dev) == []                                  # nothing for the host to repair


def test_range_from_tensor_zero_input(dev):
    from lidarcrafter_amd import ops as K

    pk = K.PackedConv("train.zero")
    x = torch.zeros(1, 16, 4, 64, device=dev)
    K.range_from_tensor(x, pk)
    y = K.conv2d_ring(x, pk, torch.ones(16, 16, 3, 3, device=dev), precision="f16x2")
    assert float(y.abs().max()) == 0.0
    assert float(pk._arena.buf.view(-1, 4)[pk._slot][0]) == 16.0     # the default scale


def test_deepcopied_model_keeps_the_range_guarantee(dev):
    """`copy.deepcopy(model)` after a forward (ema_pytorch's EMA(ddpm) in the reference
    trainers; any `torch.save(model)` round trip) must leave the copy's convolutions with their OWN
    slots in the registered arena -- a saturating input to the copy is reported and recomputed, and
    the original's records are untouched."""
    import copy
    import io

    from lidarcrafter_amd import ops as K
    from oracle import denoiser as D
    from tests.test_hip_parity import _uncond

    m = _uncond(16, (8, 64), dev)
    with torch.no_grad():
        m.in_conv.weight.mul_(1.0e-5)
    x = seeded_randn(2, 2, 8, 64, seed=46)
    lam = torch.tensor([-2.0, 1.0])
    with torch.no_grad():
        m(x.to(dev), lam.to(dev))                                  # slots allocated, caches built
    pk0 = m.in_conv._packed
    assert pk0._arena is not None
    m2 = copy.deepcopy(m)
    pk2 = m2.in_conv._packed
    assert pk2 is not pk0 and pk2._arena is None and pk2.wh is None  # fresh, lazily allocated
    buf = io.BytesIO()
    torch.save(m, buf)
    buf.seek(0)
    m3 = torch.load(buf, weights_only=False)
    assert m3.in_conv._packed._arena is None
    sd = {k: v.detach().cpu().clone() for k, v in m2.state_dict().items()}
    big = x * 1.0e5
    for mc in (m2, m3):
        with warnings.catch_warnings(record=True) as rec:
            warnings.simplefilter("always")
            with torch.no_grad():
                y = mc(big.to(dev), lam.to(dev))
        assert any("in_conv" in str(w_.message) for w_ in rec), [str(w_.message) for w_ in rec]
        assert rel_l2(y, D.efficient_unet_forward(sd, big, lam)) < 2e-5
        pk = mc.in_conv._packed
        assert pk._arena is K._arena(dev) and pk._slot != pk0._slot
    assert pk0.x_scale == 16.0                                      # the original never saw `big`


def test_training_forward_does_not_poll(dev, monkeypatch):
    """the range-checked wrapper leaves the autograd path alone (no blocking
    device->host copy per training step, no repeated graph)."""
    from lidarcrafter_amd import ops as K
    from tests.test_hip_parity import _uncond

    m = _uncond(16, (8, 64), dev).train()
    calls = []
    real = K.range_poll
    monkeypatch.setattr(K, "range_poll", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    x = seeded_randn(2, 2, 8, 64, seed=46).to(dev)
    y = m(x, torch.tensor([-2.0, 1.0], device=dev))
    y.square().mean().backward()
    assert calls == []
    with torch.no_grad():
        m.eval()(x, torch.tensor([-2.0, 1.0], device=dev))
    assert len(calls) == 1


def test_single_product_mode_is_opt_in_and_close(dev):
    """ops.conv_products(): 3 unless asked otherwise; 1 (liblidarcrafter_hip_p1.so: one fp16 product per
    multiply) only through set_conv_products(1) or, with AUTOCAST_SINGLE_PRODUCT, inside fp16 autocast.  Both conv
    kernels (fp32 input, pre-split input) then differ from the three-product result by the 11-bit operand
    rounding -- about 1e-3 -- and by nothing more."""
    from lidarcrafter_amd import ops as K

    assert K.conv_products() == 3
    x = seeded_randn(2, 64, 8, 256, seed=401).to(dev)
    w = (seeded_randn(64, 64, 3, 3, seed=402) / 24.0).to(dev)
    b = seeded_randn(64, seed=403).to(dev)
    pk = K.PackedConv()
    y3 = K.conv2d_ring(x, pk, w, b)
    sa = K.groupnorm(x, 8, 1e-6, act_silu=True, split_for=pk)
    z3 = K.conv2d_ring(sa, pk, w, b)
    old = K.set_conv_products(1)
    try:
        assert K.conv_products() == 1
        y1 = K.conv2d_ring(x, pk, w, b)
        z1 = K.conv2d_ring(K.groupnorm(x, 8, 1e-6, act_silu=True, split_for=pk), pk, w, b)
    finally:
        K.set_conv_products(old)
    for a1, a3 in ((y1, y3), (z1, z3)):
        r = rel_l2(a1, a3)
        assert 1e-5 < r < 3e-3, r                      # rounded operands: different, and only that different
    assert torch.equal(K.conv2d_ring(x, pk, w, b), y3)          # back on the product path, bit for bit
    K.AUTOCAST_SINGLE_PRODUCT = True
    try:
        assert K.conv_products() == 3
        with torch.autocast("cuda", dtype=torch.float16):
            assert K.conv_products() == 1
        with torch.autocast("cuda", dtype=torch.bfloat16):
            assert K.conv_products() == 3
    finally:
        K.AUTOCAST_SINGLE_PRODUCT = False
