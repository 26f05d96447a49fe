"""Host logic of the graphs kept across runs (no GPU): the key under which a sampler may replay an earlier run's captured
step, the packing of a run's per-step parameters, the eligibility rules of the unit-form attention and the redirect of a
prepared condition to its static view.  The GPU side -- bit-equality with the uncached paths, capture counts -- is
tests/test_graph_cache.py."""
import torch

from lidarcrafter_amd import ops as K


def _sampler():
    from lidargen.models.diffusion import ContinuousTimeGaussianDiffusion
    from lidargen.models.unets import EfficientUNet
    from lidargen.utils.lidar import get_linear_ray_angles

    m = EfficientUNet(2, (8, 64), base_channels=16, coords_encoding="fourier_features", num_residual_blocks=(1, 1, 1, 1),
                      gn_num_groups=8, gn_eps=1e-6, attn_num_heads=8, ring=True)
    m.coords = get_linear_ray_angles(8, 64, 10.0, -30.0)
    return ContinuousTimeGaussianDiffusion(m, torch.nn.Identity()).eval()


def _state(ddpm, B=2, S=5, tf=True):
    x = torch.zeros(B, 2, 8, 64)
    return dict(x=x, B=B, n=S, i=0, cond=None, needs_noise=False, obj=0, mid=1,
                lam=torch.arange(S * B, dtype=torch.float32).reshape(S, B),
                coef=torch.arange(S * B * 8, dtype=torch.float32).reshape(S, B, 8),
                tf=(torch.ones(S * B, 7), 2 * torch.ones(S * B, 10)) if tf else None)


def test_graph_key_names_what_a_captured_step_depends_on(monkeypatch):
    ddpm = _sampler()
    st = _state(ddpm)
    k0 = ddpm._graph_key(st)
    assert k0 == ddpm._graph_key(st) and hash(k0) is not None
    assert ddpm._graph_key(_state(ddpm, B=1)) != k0                                  # batch
    st2 = _state(ddpm)
    st2["needs_noise"] = True
    assert ddpm._graph_key(st2) != k0                                                 # DDPM / DDIM eta > 0: a noise operand
    st2 = _state(ddpm)
    st2["mid"] = 0
    assert ddpm._graph_key(st2) != k0                                                 # another update kernel
    assert ddpm._graph_key(_state(ddpm, tf=False)) != k0                              # time features hoisted or not
    assert ddpm._graph_key(_state(ddpm, S=9))[:2] == k0[:2]                           # the step count is NOT part of it
    monkeypatch.setattr(K, "PRODUCER_GN_STATS", not K.PRODUCER_GN_STATS)
    assert ddpm._graph_key(st) != k0                                                  # a routing switch of ops
    monkeypatch.undo()
    assert ddpm._graph_key(st) == k0
    with torch.no_grad():
        ddpm.model.in_conv.bias.add_(1.0)                                             # weights moved in place (optimizer step)
    assert ddpm._graph_key(st) != k0
    k1 = ddpm._graph_key(st)
    ddpm.model.in_conv.weight = torch.nn.Parameter(ddpm.model.in_conv.weight.detach().clone())   # a parameter REPLACED
    assert ddpm._graph_key(st) != k1
    # a condition the denoiser does not expose through graph_operands(): the run keeps its graph to itself
    st3 = _state(ddpm)
    st3["cond"] = {"other_condition": {"anything": torch.zeros(1)}}
    assert ddpm._graph_key(st3) is None
    st3["cond"] = {"other_condition": torch.zeros(2, 5)}                              # a tensor: kept in the entry's buffer
    assert ddpm._graph_key(st3) is not None and ddpm._graph_key(st3) != k0


def test_fingerprint_covers_every_parameter_and_buffer():
    ddpm = _sampler()
    fp = ddpm._weights_fingerprint()
    n = sum(1 for _ in ddpm.model.parameters()) + sum(1 for _ in ddpm.model.buffers())
    assert len(fp) == n and len(set(p for p, _ in fp)) == n                           # every tensor once, none shared


def test_pack_rows_layout():
    ddpm = _sampler()
    st = _state(ddpm, B=3, S=4)
    table, offs, widths, shapes = ddpm._pack_rows(st)
    assert widths == [3, 24, 21, 30] and all(o % 4 == 0 for o in offs)                # every view 16-byte aligned
    assert shapes == [(3,), (3, 8), (3, 7), (3, 10)]
    assert table.shape == (4, offs[-1] + 32)
    for i in range(4):
        assert torch.equal(table[i, offs[0]:offs[0] + 3], st["lam"][i])
        assert torch.equal(table[i, offs[1]:offs[1] + 24].view(3, 8), st["coef"][i])
        assert torch.equal(table[i, offs[2]:offs[2] + 21].view(3, 7), st["tf"][0][3 * i:3 * i + 3])
        assert torch.equal(table[i, offs[3]:offs[3] + 30].view(3, 10), st["tf"][1][3 * i:3 * i + 3])


def test_unit_form_eligibility():
    ok = K.AttnUnits.eligible
    assert ok(8, 2048, 13, 32, 32, 32) and ok(16, 512, 13, 32, 32, 32) and ok(4, 64, 0, 64, 0, 8) and ok(2, 128, 32, 40, 16, 24)
    assert not ok(8, 2000, 13, 32, 32, 32)          # image keys in whole 32-key tiles
    assert not ok(8, 2048, 33, 32, 32, 32)          # at most one further tile
    assert not ok(8, 2048, 13, 32, 32, 64)          # values: 32 channels per head
    assert not ok(8, 2048, 13, 20, 32, 32)          # channels in whole octets
    assert not ok(8, 2048, 13, 16, 16, 16)          # <= 32 q/k channels: the 32-channel kernel of attention.hip is the cheaper one
    assert K.qkv_units_ok(256, 8, 2048) and K.qkv_units_ok(512, 16, 512)
    assert not K.qkv_units_ok(192, 6, 512) and not K.qkv_units_ok(256, 4, 512) and not K.qkv_units_ok(256, 8, 500)


def test_static_condition_redirect():
    """LayoutUnetV1._static_condition: only THE condition prepare_condition last saw (same three tensors) is swapped for
    its static view; any other dict passes through untouched."""
    from lidargen.models.unets.layout_unet_v1 import LayoutUnetV1

    m = LayoutUnetV1.__new__(LayoutUnetV1)
    torch.nn.Module.__init__(m)
    keys = LayoutUnetV1._PREP_KEYS
    a = {k: torch.zeros(2, 4, 3) for k in keys}
    a["concat_cond"] = torch.zeros(1)
    assert m._static_condition(a) is a                                                # nothing prepared
    static = dict(a)
    static.update({k: torch.ones(2, 4, 3) for k in keys})
    m.__dict__["_prep"] = dict(lay=static, src=tuple(a[k] for k in keys),
                               src_id=tuple((a[k].data_ptr(), tuple(a[k].shape)) for k in keys))
    assert m._static_condition(a) is static and m._static_condition(dict(a)) is static
    b = {k: torch.zeros(2, 4, 3) for k in keys}
    assert m._static_condition(b) is b                                                # another condition
    assert m._static_condition({"xf_out": a["xf_out"]}) is not static                 # not a layout condition at all
