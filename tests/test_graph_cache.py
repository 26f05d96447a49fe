"""The captured denoising step kept across `sample()` calls (continuous_time.py::_graph_key): a later run of a known
shape replays the earlier run's HIP graph from its first step.  Every claim here is bit-equality with the same sampler
run WITHOUT the cache (`graph_cache_size = 0`: first step eager, one capture per run -- the path every golden test of
rounds 1-5 pinned on the reference), plus the count of captures.  `pytest -m gpu`."""
import pytest
import torch

from lidarcrafter_amd.testing import seeded_fill, synth_layout_batch, synth_object_batch, synth_text_features

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


def _gens(n, seed):
    return [torch.Generator().manual_seed(seed + i) for i in range(n)]


class _Captures:
    """Counts the captures a sampler makes."""

    def __init__(self, ddpm, monkeypatch):
        self.n = 0
        inner = ddpm._capture

        def counted(st):
            self.n += 1
            return inner(st)

        monkeypatch.setattr(ddpm, "_capture", counted)


def _uncond_ddpm(dev, base=16, res=(8, 64)):
    from lidargen.models.diffusion import ContinuousTimeGaussianDiffusion
    from tests.test_hip_parity import _uncond

    return ContinuousTimeGaussianDiffusion(_uncond(base, res, dev), torch.nn.Identity()).eval().to(dev)


@pytest.mark.parametrize("mode", ["ddim", "ddpm"])
def test_uncond_runs_share_one_capture(dev, monkeypatch, mode):
    ddpm = _uncond_ddpm(dev)
    run = lambda seed, steps=6: ddpm.sample(2, steps, progress=False, rng=_gens(2, seed), mode=mode)
    monkeypatch.setattr(ddpm, "graph_cache_size", 0)
    ref = {s: run(s) for s in (0, 10)}
    ref9 = run(10, 9)
    monkeypatch.setattr(ddpm, "graph_cache_size", 4)
    cap = _Captures(ddpm, monkeypatch)
    assert torch.equal(run(0), ref[0]) and cap.n == 1           # capture at step 1, as before
    assert torch.equal(run(10), ref[10]) and cap.n == 1         # replayed from step 0
    assert torch.equal(run(0), ref[0]) and cap.n == 1
    assert torch.equal(run(10, 9), ref9) and cap.n == 1         # another step count: same graph, another table
    assert not torch.equal(ref[0], ref[10])


def test_key_follows_batch_mode_route_and_weights(dev, monkeypatch):
    from lidarcrafter_amd import ops as K

    ddpm = _uncond_ddpm(dev)
    cap = _Captures(ddpm, monkeypatch)
    run = lambda B=2, mode="ddim": ddpm.sample(B, 5, progress=False, rng=_gens(B, 3), mode=mode)
    a = run()
    run(), run()
    assert cap.n == 1
    run(mode="ddpm")
    assert cap.n == 2                                           # another update kernel + a noise operand
    run(B=1)
    assert cap.n == 3
    monkeypatch.setattr(K, "PRODUCER_GN_STATS", not K.PRODUCER_GN_STATS)
    b = run()
    assert cap.n == 4                                           # a routing switch: never a graph of the other route
    monkeypatch.undo()
    cap = _Captures(ddpm, monkeypatch)
    # new weights in place (an optimizer step, load_state_dict): the packed copies are rebuilt, the old graph reads
    # the old ones -- the key must move
    with torch.no_grad():
        w = ddpm.model.in_conv.weight
        w.mul_(1.5)
    c = run()
    assert cap.n == 1 and not torch.equal(c, a)
    fresh = _uncond_ddpm(dev)
    with torch.no_grad():
        fresh.model.in_conv.weight.mul_(1.5)
    assert torch.equal(c, fresh.sample(2, 5, progress=False, rng=_gens(2, 3), mode="ddim"))
    assert torch.isfinite(b).all()


def test_layout_conditions_share_one_capture(dev, monkeypatch):
    """Two different layout conditions: the second run's attention operands are written into the first run's tensors
    (ObjectAwareCrossAttention._refill), its concat channels into the resident input buffer (prepare_condition)."""
    from lidargen.models.diffusion import CondContinuousTimeGaussianDiffusion
    from tests.test_oracle_vs_golden import build_cond_pair

    m, enc = build_cond_pair((8, 64), 8, 32)
    ddpm = CondContinuousTimeGaussianDiffusion(m, enc, cond_mode="concat").eval().to(dev)
    batches = {}
    for s in (51, 57):
        b = {k: v.to(dev) for k, v in synth_layout_batch(2, 8, 64, seed=s).items()}
        b["concat_cond"] = torch.randn(2, 10, 8, 64, generator=torch.Generator().manual_seed(s)).to(dev)
        batches[s] = b
    import lidargen.models.unets.layout_unet_v1 as LU

    run = lambda s, seed: ddpm.sample(dict(batches[s]), 2, 5, progress=False, rng=_gens(2, seed), mode="ddim")
    # the reference runs: no graph kept across runs, the condition operands computed eagerly per run (rounds 1-5)
    monkeypatch.setattr(ddpm, "graph_cache_size", 0)
    monkeypatch.setattr(LU, "PREPARE_GRAPH", False)
    ref = {(s, seed): run(s, seed) for s, seed in ((51, 0), (57, 0), (57, 4))}
    assert not torch.equal(ref[51, 0], ref[57, 0])              # the condition matters
    assert ddpm.model.__dict__.get("_prep") is None
    monkeypatch.setattr(LU, "PREPARE_GRAPH", True)
    monkeypatch.setattr(ddpm, "graph_cache_size", 4)
    cap = _Captures(ddpm, monkeypatch)
    assert torch.equal(run(51, 0), ref[51, 0]) and cap.n == 1
    assert torch.equal(run(57, 0), ref[57, 0]) and cap.n == 1   # new condition, old graph
    assert torch.equal(run(57, 4), ref[57, 4]) and cap.n == 1
    assert torch.equal(run(51, 0), ref[51, 0]) and cap.n == 1
    assert ddpm.model._prep["graph"] is not None and ddpm.model._prep["graph"] is not False     # operands by one graph
    # the image-side positional operand is kept across conditions while the weights behind it stand still
    # (layout_encoder._patch_embedding's tag): move one of them -- operand and graph key must follow
    with torch.no_grad():
        ddpm.condition_model.obj_bbox_2d_embedding.weight.mul_(1.25)
    moved = run(57, 4)
    assert cap.n == 2 and not torch.equal(moved, ref[57, 4])
    m2, enc2 = build_cond_pair((8, 64), 8, 32)
    fresh = CondContinuousTimeGaussianDiffusion(m2, enc2, cond_mode="concat").eval().to(dev)
    fresh.graph_cache_size = 0
    with torch.no_grad():
        fresh.condition_model.obj_bbox_2d_embedding.weight.mul_(1.25)
    assert torch.equal(moved, fresh.sample(dict(batches[57]), 2, 5, progress=False, rng=_gens(2, 4), mode="ddim"))


def test_masked_layout_keys_are_never_shared(dev, monkeypatch):
    """Per-sample key sets (use_key_padding_mask) change shape with the condition: such a run keeps its graph to
    itself, as before."""
    from lidargen.models.diffusion import CondContinuousTimeGaussianDiffusion
    from tests.test_oracle_vs_golden import COND_OPTION_VARIANTS, build_cond_pair

    ukw, ekw = COND_OPTION_VARIANTS["mask"]
    m, enc = build_cond_pair((8, 64), 8, 32, unet_kw=ukw, enc_kw=ekw)
    ddpm = CondContinuousTimeGaussianDiffusion(m, enc, cond_mode="concat").eval().to(dev)
    b = {k: v.to(dev) for k, v in synth_layout_batch(2, 8, 64, seed=51).items()}
    b["concat_cond"] = torch.zeros(2, 10, 8, 64, device=dev)
    cap = _Captures(ddpm, monkeypatch)
    x1 = ddpm.sample(dict(b), 2, 5, progress=False, rng=_gens(2, 0), mode="ddim")
    x2 = ddpm.sample(dict(b), 2, 5, progress=False, rng=_gens(2, 0), mode="ddim")
    assert cap.n == 2 and torch.equal(x1, x2)


def test_object_branch_conditions_share_one_capture(dev, monkeypatch):
    """The 1-D sampler: no resident input buffer (the state lives in the cache entry), a tensor condition (kept in the
    entry's buffer)."""
    from lidargen.utils import inference
    from lidargen.utils.configs import __all__ as C

    ddpm, model = inference.load_model_object_duffusion_training(C["nuscenes-object"]())
    seeded_fill(model, salt=300), seeded_fill(ddpm.condition_model, salt=301)
    ddpm = ddpm.eval().to(dev)
    ddpm.condition_model.set_text_features(synth_text_features(), dev)
    batches = {s: {k: v.to(dev) for k, v in synth_object_batch(3, seed=s).items()} for s in (95, 97)}
    run = lambda s, seed: ddpm.sample(batches[s], 3, 6, progress=False, rng=_gens(3, seed), mode="ddpm")
    monkeypatch.setattr(ddpm, "graph_cache_size", 0)
    ref = {(s, seed): run(s, seed) for s, seed in ((95, 600), (97, 600), (97, 7))}
    assert not torch.equal(ref[95, 600], ref[97, 600])
    monkeypatch.setattr(ddpm, "graph_cache_size", 4)
    cap = _Captures(ddpm, monkeypatch)
    for key in ((95, 600), (97, 600), (97, 7), (95, 600)):
        assert torch.equal(run(*key), ref[key]), key
    assert cap.n == 1


def test_two_samplers_interleaved(dev):
    """generate_sequence alternates two samplers (frame 0 / frames 1..): each replays its own graph, the scratch of the
    statistics pass and the range arena are shared."""
    a, b = _uncond_ddpm(dev), _uncond_ddpm(dev, base=32)
    ra = a.sample(2, 5, progress=False, rng=_gens(2, 1), mode="ddim")
    rb = b.sample(2, 5, progress=False, rng=_gens(2, 1), mode="ddim")
    for _ in range(2):
        assert torch.equal(a.sample(2, 5, progress=False, rng=_gens(2, 1), mode="ddim"), ra)
        assert torch.equal(b.sample(2, 5, progress=False, rng=_gens(2, 1), mode="ddim"), rb)


def test_sampler_copies_and_pickles_without_its_graphs(dev):
    import copy
    import io

    ddpm = _uncond_ddpm(dev)
    x = ddpm.sample(2, 5, progress=False, rng=_gens(2, 1), mode="ddim")
    twin = copy.deepcopy(ddpm)                                  # the reference trainers' EMA(ddpm)
    assert torch.equal(twin.sample(2, 5, progress=False, rng=_gens(2, 1), mode="ddim"), x)
    buf = io.BytesIO()
    torch.save(ddpm, buf)


def test_layout_encoder_core_by_graph(dev, monkeypatch):
    """The layout-dependent core of the encoder as one replayed graph inside sampling runs (inference mode): same bits as
    the eager core, fresh result tensors per call, weights followed."""
    import lidargen.models.unets.layout_encoder as LE
    from tests.test_oracle_vs_golden import build_cond_pair

    _, enc = build_cond_pair((8, 64), 8, 32)
    enc = enc.to(dev)
    batches = [{k: v.to(dev) for k, v in synth_layout_batch(2, 8, 64, seed=s).items()} for s in (51, 57, 63)]
    keys = ("xf_proj", "xf_out", "obj_class_embedding", "obj_bbox_embedding", "key_padding_mask")
    with torch.inference_mode():
        monkeypatch.setattr(LE, "CORE_GRAPH", False)
        ref = [enc(b) for b in batches]
        monkeypatch.setattr(LE, "CORE_GRAPH", True)
        got = [enc(b) for b in batches] + [enc(batches[0])]
        assert enc._core_graph["graph"]
        for g, r in zip(got, ref + [ref[0]]):
            assert set(g) == set(r)
            for k in keys:
                assert torch.equal(g[k], r[k]), k
        assert got[1]["xf_out"].data_ptr() != got[2]["xf_out"].data_ptr()          # several conditions alive at once
        assert not torch.equal(got[1]["xf_out"], got[2]["xf_out"])
        with torch.inference_mode(False), torch.no_grad():
            enc.transformer_proj.weight.mul_(2.0)
        moved = enc(batches[0])
        monkeypatch.setattr(LE, "CORE_GRAPH", False)
        assert torch.equal(moved["xf_proj"], enc(batches[0])["xf_proj"]) and not torch.equal(moved["xf_proj"], ref[0]["xf_proj"])


def test_precomputed_conditions_reused_in_any_order(dev, monkeypatch):
    """Condition dicts computed once and sampled from repeatedly, interleaved (A, B, A, B): the static inputs of the
    operand graph never leak one condition into a dict that named another (the caller's dicts are left untouched)."""
    import lidargen.models.unets.layout_unet_v1 as LU
    from lidargen.models.diffusion import CondContinuousTimeGaussianDiffusion
    from tests.test_oracle_vs_golden import build_cond_pair

    m, enc = build_cond_pair((8, 64), 8, 32)
    ddpm = CondContinuousTimeGaussianDiffusion(m, enc, cond_mode="concat").eval().to(dev)
    batches = [{k: v.to(dev) for k, v in synth_layout_batch(2, 8, 64, seed=s).items()} for s in (51, 57)]

    def run(cdict, seed):
        with torch.inference_mode():
            x_T = ddpm.randn(2, *ddpm.sampling_shape, rng=_gens(2, seed), device=ddpm.device)
            st = ddpm.begin_sampling(2, 5, None, "ddim", 0.0, x_T=x_T, condition_dict=cdict)
            for _ in range(5):
                ddpm.sampling_step(st)
            return st["x"].clone()

    with torch.inference_mode():
        conds = [ddpm.get_network_condition(input_dict=b, only_custom_condition=True) for b in batches]
    ptrs = [{k: v.data_ptr() for k, v in c["other_condition"].items() if isinstance(v, torch.Tensor)} for c in conds]
    monkeypatch.setattr(ddpm, "graph_cache_size", 0)
    monkeypatch.setattr(LU, "PREPARE_GRAPH", False)
    ref = [run(conds[0], 1), run(conds[1], 1)]
    assert not torch.equal(ref[0], ref[1])
    monkeypatch.setattr(LU, "PREPARE_GRAPH", True)
    monkeypatch.setattr(ddpm, "graph_cache_size", 4)
    for i in (0, 1, 0, 1, 1, 0):
        assert torch.equal(run(conds[i], 1), ref[i]), i
    assert ddpm.model._prep["graph"]
    for c, p in zip(conds, ptrs):
        assert {k: v.data_ptr() for k, v in c["other_condition"].items() if isinstance(v, torch.Tensor)} == p
