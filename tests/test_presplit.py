"""Pre-split activations (DESIGN.md "conv_f16x2_ps_kernel"): the GroupNorm apply pass writes fp16
hi / lo planes for ONE consumer conv, the conv stages them with LDS-DMA.  Parity of both halves of
that hand-over against the fp32 route and the oracle, incl. ragged planes, persistent tiles, the
statistics-emitting epilogue and the range-safety contract (which moves to the producer).
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


def decode(sa):
    """SplitAct -> fp64 [B, C, H, W] of (hi + lo) / x_scale."""
    B, C, H, W = sa.shape
    u = sa.buf.view(B, 2, C // 8, H, W, 8).double()
    v = (u[:, 0] + u[:, 1]).permute(0, 1, 4, 2, 3).reshape(B, C, H, W)
    return v / sa.packed.x_scale


@pytest.mark.parametrize("B,C,H,W,G", [(2, 64, 8, 64, 8), (1, 128, 16, 512, 8), (2, 256, 5, 50, 32),
                                       (1, 32, 3, 7, 2), (3, 512, 4, 128, 32),
                                       # GroupNorm32 at the narrow widths of the layout denoiser: 4 / 2 / 1
                                       # channels per group, an octet spans several groups
                                       (2, 128, 16, 64, 32), (1, 64, 8, 64, 32), (2, 32, 4, 64, 32),
                                       # concatenated inputs: 6 / 12 / 3 channels per group
                                       (2, 192, 8, 64, 32), (1, 384, 4, 128, 32), (1, 96, 4, 64, 32)])
@pytest.mark.parametrize("mode", ["plain", "adagn"])
@pytest.mark.parametrize("route", ["two_pass", "producer_stats"])
def test_groupnorm_split_matches_fp32_route(dev, B, C, H, W, G, mode, route):
    from lidarcrafter_amd import ops as K

    assert K.can_presplit(C, G)
    x = (seeded_randn(B, C, H, W, seed=1) * 3 + 0.5).to(dev)
    kw = {}
    if mode == "plain":
        kw = dict(gamma=seeded_randn(C, seed=2).to(dev), beta=seeded_randn(C, seed=3).to(dev))
    else:
        kw = dict(scale=seeded_randn(B, C, seed=4).to(dev), shift=seeded_randn(B, C, seed=5).to(dev))
    if route == "producer_stats":        # x := output of a stats-emitting conv
        if W % 32 or H % 2:
            pytest.skip("the producer conv needs a pipelined tile shape")
        w = (seeded_randn(C, C, 3, 3, seed=6) / (3 * C ** 0.5)).to(dev)
        # octet entries for groups of whole octets, else what the models ask for (layout_unet_v1._stats_unit): quads
        # (this fp32-input producer writes the finer pairs) at 4 / 12 per group, pairs below.  The pre-split apply pass
        # folds them for whole-octet groups and for 2 / 4 channels per group (round 5); 1 / 3 / 6 / 12: statistics pass
        cpg = C // G
        x = K.conv2d_ring(x, K.PackedConv(), w, emit_stats=True if cpg % 8 == 0 else (4 if cpg % 4 == 0 else 2))
        assert (K._find_stats(x, G, octet_groups=True) is not None) == (cpg % 8 == 0 or cpg in (2, 4))
    pk = K.PackedConv("consumer")
    ref = K.groupnorm(x.clone(), G, 1e-6, act_silu=True, **kw)
    sa = K.groupnorm(x, G, 1e-6, act_silu=True, split_for=pk, **kw)
    assert isinstance(sa, K.SplitAct) and sa.shape == (B, C, H, W) and sa.packed is pk
    r = rel_l2(decode(sa), ref)
    assert r < 4e-7, r            # hi + lo carry 22 bits of the fp32 value


@pytest.mark.parametrize("B,Ci,Co,H,W", [(2, 64, 64, 8, 128), (1, 128, 256, 8, 256), (2, 256, 96, 4, 128),
                                         (1, 64, 2, 32, 256), (2, 48, 64, 5, 50), (1, 512, 512, 4, 128),
                                         (1, 16, 32, 3, 70)])
@pytest.mark.parametrize("cfg", [0, 12, 13, 15, 22, 23, 25, 28, 223, 423, 225, 212])
def test_conv_presplit_vs_oracle(dev, B, Ci, Co, H, W, cfg):
    """GroupNorm(split) -> 3x3 ring conv on the LDS-DMA kernel vs the CPU oracle of the same chain."""
    from lidarcrafter_amd import ops as K
    from oracle import denoiser as D

    G = 8 if Ci % 64 == 0 else 2
    if not K.can_presplit(Ci, G):
        pytest.skip("shape not eligible")
    x = seeded_randn(B, Ci, H, W, seed=11)
    w = seeded_randn(Co, Ci, 3, 3, seed=12) / (Ci * 9) ** 0.5
    b = seeded_randn(Co, seed=13)
    res = seeded_randn(B, Co, H, W, seed=14)
    gamma, beta = seeded_randn(Ci, seed=15), seeded_randn(Ci, seed=16)
    a_ref = D.silu(D.group_norm(x, G, gamma, beta, 1e-6))
    ref = (D.conv_ring(a_ref, w, b) + res) * 0.7071
    pk = K.PackedConv("ps")
    sa = K.groupnorm(x.to(dev), G, 1e-6, gamma.to(dev), beta.to(dev), act_silu=True, split_for=pk)
    y = K.conv2d_ring(sa, pk, w.to(dev), b.to(dev), res=res.to(dev), out_scale=0.7071, tile_cfg=cfg)
    assert rel_l2(y, ref) < 2e-6, rel_l2(y, ref)
    # emit_stats: the next GroupNorm from the epilogue's octet statistics == the two-pass result
    if Co % 8 == 0:
        y2 = K.conv2d_ring(sa, pk, w.to(dev), b.to(dev), res=res.to(dev), out_scale=0.7071,
                           tile_cfg=cfg, emit_stats=True)
        assert torch.equal(y2, y)
        if Co % 64 == 0:
            g1 = K.groupnorm(y2, 8, 1e-6)
            g2 = K.groupnorm(y.clone(), 8, 1e-6)
            assert rel_l2(g1, g2) < 2e-6
        # quad entries (round 5; not from the split-K reduction): every entry against the stored output, then the
        # consumers at 4 channels per group -- apply, pre-split apply (one group per wave), a conv's fused input norm
        y4 = K.conv2d_ring(sa, pk, w.to(dev), b.to(dev), res=res.to(dev), out_scale=0.7071,
                           tile_cfg=cfg, emit_stats=4)
        assert torch.equal(y4, y)
        d = getattr(y4, "_lc_gnstats", None)
        if cfg != 0 or B * H * W >= 4096:
            assert d, "no quad entries from the pre-split kernel"
        if d and Co % 16 == 0:
            h = d[(0, Co)]
            assert h.unit == 4 and tuple(h.buf.shape) == (B, Co // 4, h.slots, 4)
            e = h.buf.double()
            n, tot = e[..., 1], e[..., 0] * e[..., 1] + e[..., 2]
            assert bool((n.sum(-1) == 4.0 * H * W).all())
            assert float((tot.sum(-1) - y.double().view(B, Co // 4, -1).sum(-1)).abs().max()) < 2e-3 * (H * W) ** 0.5
            G4 = Co // 4
            yc = y.clone()
            assert rel_l2(K.groupnorm(y4, G4, 1e-6, act_silu=True), K.groupnorm(yc, G4, 1e-6, act_silu=True)) < 2e-6
            pk2 = K.PackedConv("next")
            s1 = K.groupnorm(y4, G4, 1e-6, act_silu=True, split_for=pk2)
            assert isinstance(s1, K.SplitAct)
            assert rel_l2(decode(s1), K.groupnorm(yc, G4, 1e-6, act_silu=True)) < 4e-7
            w2 = (seeded_randn(64, Co, 3, 3, seed=17) / (Co * 9) ** 0.5).to(dev)
            c1 = K.conv2d_ring(y4, K.PackedConv(), w2, None, gn_coeffs=K.groupnorm_stats(y4, G4, 1e-6), gn_silu=True)
            c2 = K.conv2d_ring(yc, K.PackedConv(), w2, None, gn_coeffs=K.groupnorm_stats(yc, G4, 1e-6), gn_silu=True)
            assert rel_l2(c1, c2) < 2e-6


def test_presplit_belongs_to_one_layer(dev):
    from lidarcrafter_amd import ops as K

    x = seeded_randn(1, 64, 4, 64, seed=21).to(dev)
    w = seeded_randn(64, 64, 3, 3, seed=22).to(dev) / 24
    pk, other = K.PackedConv("a"), K.PackedConv("b")
    sa = K.groupnorm(x, 8, 1e-6, split_for=pk)
    with pytest.raises(ValueError):
        K.conv2d_ring(sa, other, w)


def test_presplit_range_contract_moves_to_the_producer(dev):
    """A 1e6 AdaGN scale saturates fp16 in the PRODUCER: it must publish that into the consumer
    layer's range record, the poll names the layer, and the re-run is exact."""
    from lidarcrafter_amd import ops as K
    from oracle import denoiser as D

    x = seeded_randn(2, 64, 8, 64, seed=31)
    w = seeded_randn(64, 64, 3, 3, seed=32) / 24
    scale = torch.full((2, 64), 3.0e5)
    shift = torch.zeros(2, 64)
    a_ref = D.group_norm(x, 8, None, None, 1e-6) * (1 + scale[:, :, None, None]) + shift[:, :, None, None]
    ref = D.conv_ring(a_ref, w, None)
    pk = K.PackedConv("ps_range")
    K.range_poll(dev)
    n_bad = 0
    for _ in range(4):
        sa = K.groupnorm(x.to(dev), 8, 1e-6, scale=scale.to(dev), shift=shift.to(dev), split_for=pk)
        y = K.conv2d_ring(sa, pk, w.to(dev))
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)
            bad = K.range_poll(dev)
        if not bad:
            break
        assert bad[0].layer == "ps_range"
        n_bad += 1
    assert n_bad == 1 and rel_l2(y, ref) < 2e-6


@pytest.mark.parametrize("B,Ci,Co,H,W", [(1, 512, 512, 4, 128), (1, 256, 256, 4, 128), (2, 256, 96, 4, 64),
                                         (1, 64, 64, 5, 50), (1, 512, 256, 4, 128)])
def test_conv_presplit_split_k(dev, B, Ci, Co, H, W):
    """Small grids: the K range of a tile is divided over several blocks (lc_conv2d_ring_f16x2_ps_fwd
    with splitk_part) and finished by lc_splitk_reduce -- same result as the single-block path to
    fp32-class accuracy, deterministic, and the reduce pass's GroupNorm statistics drive the next
    GroupNorm to the two-pass result."""
    from lidarcrafter_amd import ops as K
    from oracle import denoiser as D

    ks = K.splitk_factor(B, Ci, Co, H, W)
    assert ks >= 2, "shape chosen to take the split-K route"
    x = seeded_randn(B, Ci, H, W, seed=41)
    w = seeded_randn(Co, Ci, 3, 3, seed=42) / (Ci * 9) ** 0.5
    b = seeded_randn(Co, seed=43)
    res = seeded_randn(B, Co, H, W, seed=44)
    a_ref = D.silu(D.group_norm(x, 8, None, None, 1e-6))
    ref = (D.conv_ring(a_ref, w, b) + res) * 0.7071
    pk = K.PackedConv("splitk")
    sa = K.groupnorm(x.to(dev), 8, 1e-6, act_silu=True, split_for=pk)
    y = K.conv2d_ring(sa, pk, w.to(dev), b.to(dev), res=res.to(dev), out_scale=0.7071, emit_stats=True)
    assert rel_l2(y, ref) < 2e-6, rel_l2(y, ref)
    y2 = K.conv2d_ring(sa, pk, w.to(dev), b.to(dev), res=res.to(dev), out_scale=0.7071)
    assert torch.equal(y, y2)                                     # deterministic, stats do not change y
    old = K.SPLITK
    K.SPLITK = False
    try:
        y1 = K.conv2d_ring(sa, pk, w.to(dev), b.to(dev), res=res.to(dev), out_scale=0.7071)
    finally:
        K.SPLITK = old
    assert rel_l2(y, y1) < 1e-6
    if Co % 64 == 0:
        assert K._find_stats(y, 8) is not None
        assert rel_l2(K.groupnorm(y, 8, 1e-6), K.groupnorm(y.clone(), 8, 1e-6)) < 2e-6


@pytest.mark.parametrize("B,Ci,Co,H,W,with_res", [(8, 256, 768, 8, 256, False), (2, 512, 1536, 4, 128, False),
                                                  (2, 256, 256, 1, 2048, True), (3, 64, 96, 5, 50, True),
                                                  (1, 128, 640, 1, 37, False)])
def test_conv1x1_presplit_matches_fp32_route(dev, B, Ci, Co, H, W, with_res):
    """GroupNorm32 -> 1x1 projection with the activation handed over pre-split (lc_conv1x1_f16x2_ps_fwd) vs the fp32
    hand-over and float64: ragged pixel counts, output-channel tails, token shapes (H = 1), residual + scale."""
    from lidarcrafter_amd import ops as K

    x = (seeded_randn(B, Ci, H, W, seed=1) * 2 + 0.3).to(dev)
    w = (seeded_randn(Co, Ci, 1, 1, seed=2) / Ci ** 0.5).to(dev)
    b = seeded_randn(Co, seed=3).to(dev)
    res = seeded_randn(B, Co, H, W, seed=4).to(dev) if with_res else None
    gam, bet = (seeded_randn(Ci, seed=5) * 0.2 + 1).to(dev), (seeded_randn(Ci, seed=6) * 0.1).to(dev)
    pk0, pk1 = K.PackedConv("fp32-in"), K.PackedConv("presplit-in")
    y0 = K.conv2d_ring(K.groupnorm(x, 32, 1e-5, gam, bet), pk0, w, b, res=res, out_scale=0.5)
    sa = K.groupnorm(x, 32, 1e-5, gam, bet, split_for=pk1)
    assert isinstance(sa, K.SplitAct)
    y1 = K.conv2d_ring(sa, pk1, w, b, res=res, out_scale=0.5)
    assert not K.range_poll(dev)
    ref = torch.nn.functional.conv2d(
        torch.nn.functional.group_norm(x.double(), 32, gam.double(), bet.double(), 1e-5), w.double(), b.double())
    ref = ((ref + res.double()) if with_res else ref) * 0.5
    assert rel_l2(y1, ref) < 2e-6, rel_l2(y1, ref)
    assert rel_l2(y1, y0) < 2e-6
    # into a caller's strided output (a slice of a wider buffer), as the modules pass `out=`
    wide = torch.zeros(B, Co + 8, H, W, device=dev)
    K.conv2d_ring(K.groupnorm(x, 32, 1e-5, gam, bet, split_for=pk1), pk1, w, b, res=res, out=wide[:, 4:4 + Co], out_scale=0.5)
    assert torch.equal(wide[:, 4:4 + Co], y1) and float(wide[:, :4].abs().max()) == 0.0
