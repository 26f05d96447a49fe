"""Host logic of the training graph that needs no GPU (lidarcrafter_amd/autograd.py): the hand-over tag a producer pass
puts on a tensor (partial maxima of |.| for the consuming conv's range record) -- when it is trusted, when it is not,
what survives views / dropout, and that the autograd engine hands a tagged gradient to the next backward node as the
same Python object (the assumption the backward hand-over rests on); TrainWeightPlan's job record layout."""
import struct

import torch

from lidarcrafter_amd import autograd as AG


def test_tag_is_trusted_only_for_the_very_tensor():
    t = torch.randn(2, 3, 4, 5)
    slot = torch.tensor([1.0, 7.5, 3.0])
    assert AG._amax_of(t) is None
    AG._tag_amax(t, slot)
    got = AG._amax_of(t)
    assert got is not None and got[0] is slot and got[1] == 1.0
    t.add_(1.0)                                   # an in-place edit (what the engine does when it accumulates gradients)
    assert AG._amax_of(t) is None
    u = torch.randn(4, 4)
    AG._tag_amax(u, slot, 2.0)
    u.data = torch.randn(4, 4)                    # same object, other storage
    assert AG._amax_of(u) is None


def test_tag_survives_views_and_dropout_with_a_bound():
    t = torch.randn(2, 8, 6)
    slot = torch.tensor([float(t.abs().max())])
    AG._tag_amax(t, slot)
    v = AG._tok(t)                                # [B, C, 1, L] view
    m = AG._amax_of(v)
    assert v.shape == (2, 8, 1, 6) and m is not None and m[0] is slot and m[1] == 1.0
    assert AG._amax_of(AG._tok(torch.randn(2, 8, 6))) is None          # nothing to carry
    d = AG.dropout(t, 0.2, True)
    md = AG._amax_of(d)
    assert md is not None and abs(md[1] - 1.25) < 1e-12
    assert float(d.abs().max()) <= float(slot.max()) * md[1] * (1 + 1e-6)
    assert AG.dropout(t, 0.0, True) is t and AG.dropout(t, 0.3, False) is t
    d2 = AG.dropout(d, 0.5, True)                 # bounds multiply
    assert abs(AG._amax_of(d2)[1] - 2.5) < 1e-12
    t.mul_(2.0)
    assert AG._amax_of(t) is None and AG._amax_of(AG._carry_amax(t, t.view(2, 48))) is None


class _Producer(torch.autograd.Function):
    """backward tags the gradient it returns, as GroupNormAct.backward does"""

    @staticmethod
    def forward(ctx, x):
        return x * 2.0

    @staticmethod
    def backward(ctx, dy):
        dx = dy * 2.0
        AG._tag_amax(dx, torch.tensor([float(dx.abs().max())]))
        return dx


class _Consumer(torch.autograd.Function):
    """backward looks for the tag on the gradient it receives, as ConvRing.backward does"""
    seen = []

    @staticmethod
    def forward(ctx, x):
        return x + 1.0

    @staticmethod
    def backward(ctx, dy):
        _Consumer.seen.append((AG._amax_of(dy), float(dy.abs().max())))
        return dy


def test_engine_hands_the_tagged_gradient_over_and_in_place_accumulation_is_caught():
    # single consumer of the conv output: the producer's gradient arrives as the tagged object
    _Consumer.seen.clear()
    x = torch.randn(3, 4, requires_grad=True)
    _Producer.apply(_Consumer.apply(x)).sum().backward()
    tag, amax = _Consumer.seen[0]
    assert tag is not None and float(tag[0].max()) == amax
    # two consumers: whatever the engine does to combine the two gradients (a new tensor, or an in-place add into one of
    # them), the node never sees a tag that does not describe the tensor it got
    _Consumer.seen.clear()
    x = torch.randn(3, 4, requires_grad=True)
    h = _Consumer.apply(x)
    (_Producer.apply(h).sum() + _Producer.apply(h * 3.0).sum()).backward()
    tag, amax = _Consumer.seen[0]
    assert tag is None or float(tag[0].max()) == amax


def test_weight_pack_job_record_layout():
    """ops.TrainWeightPlan packs lc_weight_pack_job records by hand: 7 pointers + 4 ints, no padding (72 bytes)."""
    assert struct.calcsize("<7Q4i") == 72
    import re
    from pathlib import Path
    hdr = (Path(__file__).resolve().parents[1] / "include" / "lidarcrafter_hip.h").read_text()
    body = re.search(r"typedef struct lc_weight_pack_job \{(.*?)\} lc_weight_pack_job;", hdr, re.S).group(1)
    decls = [d.strip() for d in body.split(";") if d.strip()]
    assert decls == ["const float* w", "void *fwd_hi, *fwd_lo", "float* fwd_meta", "void *dx_hi, *dx_lo", "float* dx_meta",
                     "int Co, Ci, ks, reserved"], decls


def test_prepare_model_leaves_a_cpu_model_alone():
    """ops.prepare_model (called by inference.setup_model for a GPU model) packs nothing and loads nothing for weights
    that are not on a GPU -- and there is still no CPU path behind it: the forward of such a model raises."""
    import pytest

    from lidarcrafter_amd import ops as K
    from lidargen.models.unets import EfficientUNet

    m = EfficientUNet(2, (8, 64), base_channels=16, coords_encoding="fourier_features", num_residual_blocks=(1, 1, 1, 1),
                      gn_num_groups=8, gn_eps=1e-6, attn_num_heads=8, ring=True)
    assert K.prepare_model(m) == 0
    with pytest.raises(Exception):
        m(torch.zeros(1, 2, 8, 64), torch.zeros(1))
