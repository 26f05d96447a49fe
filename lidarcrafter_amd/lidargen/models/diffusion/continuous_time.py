"""Continuous-time Gaussian diffusion (https://arxiv.org/abs/2107.00630) on the HIP hot path.

API mirror of the reference's lidargen/models/diffusion/continuous_time.py:66-330
(constructor, log_snr, get_*, q_step*, p_step, sample, repaint), with the sampling loop
re-designed for the GPU:
  * all schedule scalars of a run are tabulated once on the host ([S,B,8], schedules.py) --
    the reference launches ~20 tiny kernels per step for them;
  * the time-embedding MLP and every AdaGN projection are evaluated for ALL steps in one batch
    before the loop (they depend on the step index only);
  * x0-prediction, clamp and the DDPM/DDIM update are one fused kernel (lc_pstep_fwd) that
    writes x_s straight into the denoiser's resident input buffer;
  * no host<->device synchronisation inside the loop (CPU generators excepted, see base.randn).
"""
from __future__ import annotations

import os
import weakref
from typing import Literal

import torch
from torch import nn
from tqdm.auto import tqdm

from lidarcrafter_amd import ops as K

from . import base, schedules


# sampler -> {key: captured step}: outside the module's __dict__, so that copy.deepcopy(ddpm) (the reference trainers'
# EMA wrapper) and torch.save(ddpm) never meet a graph object; entries die with their sampler
_GRAPH_CACHES = weakref.WeakKeyDictionary()


class ContinuousTimeGaussianDiffusion(base.GaussianDiffusion):
    # Replay one captured HIP graph per denoising step (~170 kernel launches -> 1 graph launch +
    # 4 tiny parameter copies).  Matters when the step is launch-bound (batch 1-2: 2.5 -> ~1.6 ms).
    use_hip_graph = os.environ.get("LC_HIP_GRAPH", "1") != "0"

    # The reference draws randn_like() in p_step even for DDIM eta=0 where it is multiplied by 0
    # (continuous_time.py:229).  With EXPLICIT generators the draw is made here too (on the
    # generator's own device, result discarded, nothing uploaded), so that a generator shared by
    # several sample() calls -- generate_sequence passes one `rng` to every frame -- ends in the
    # state a seeded reference run leaves it in.  rng=None (the process-global generator, whose
    # stream differs between CPU and GPU anyway) is not advanced.  False skips the draw always.
    # Inside `sample()` the draws are DEFERRED: the loop only counts them and `finish_sampling`
    # makes them after the last step has been enqueued, i.e. while the GPU is still working through
    # the replayed steps (K separate draws of the same shape leave a generator in the same state
    # wherever they happen) -- B full-frame CPU randn calls per step would otherwise sit on the
    # critical path of a ~1-4 ms step.
    advance_rng_when_unused = True

    def __init__(self, model: nn.Module, condition_model: nn.Module = None,
                 prediction_type: Literal["eps", "v", "x_0"] = "eps", loss_type="l2",
                 noise_schedule: str = "cosine", min_snr_loss_weight: bool = True,
                 min_snr_gamma: float = 5.0, sampling_resolution=None, clip_sample: bool = True,
                 clip_sample_range: float = 1, image_d: float = None, noise_d_low: float = None,
                 noise_d_high: float = None):
        self.image_d, self.noise_d_low, self.noise_d_high = image_d, noise_d_low, noise_d_high
        super().__init__(model=model, condition_model=condition_model, sampling="ddpm",
                         prediction_type=prediction_type, loss_type=loss_type,
                         num_training_steps=None, noise_schedule=noise_schedule,
                         min_snr_loss_weight=min_snr_loss_weight, min_snr_gamma=min_snr_gamma,
                         sampling_resolution=sampling_resolution, clip_sample=clip_sample,
                         clip_sample_range=clip_sample_range)

    # ---- schedule ------------------------------------------------------------------------------
    def setup_parameters(self) -> None:
        self._schedule = schedules.make_schedule(self.noise_schedule, self.image_d,
                                                 self.noise_d_low, self.noise_d_high)

    def log_snr(self, t: torch.Tensor) -> torch.Tensor:
        return self._schedule(t)[:, None, None, None]

    def sample_timesteps(self, batch_size: int, device) -> torch.Tensor:
        return torch.rand(batch_size, device=device, dtype=torch.float32)

    def get_network_condition(self, steps):
        return self.log_snr(steps)[:, 0, 0, 0]

    def get_target(self, x_0, step_t, noise):
        if self.objective == "eps":
            return noise
        if self.objective == "x_0":
            return x_0
        if self.objective == "v":
            alpha, sigma = schedules.alpha_sigma(self.log_snr(step_t))
            return alpha * noise - sigma * x_0
        raise ValueError(f"invalid objective {self.objective}")

    def get_loss_weight(self, steps):
        snr = self.log_snr(steps).exp()
        clipped = snr.clamp(max=self.min_snr_gamma) if self.min_snr_loss_weight else snr.clone()
        if self.objective == "eps":
            return clipped / snr
        if self.objective == "x_0":
            return clipped
        if self.objective == "v":
            return clipped / (snr + 1)
        raise ValueError(f"invalid objective {self.objective}")

    # ---- forward process (elementwise, off the hot path) ---------------------------------------
    def q_step_from_x_0(self, x_0, step_t, rng=None):
        noise = self.randn_like(x_0, rng=rng)
        alpha, sigma = schedules.alpha_sigma(self.log_snr(step_t))
        return x_0 * alpha + noise * sigma, noise

    def q_step(self, x_s, step_t, step_s, rng=None):
        a_t, s_t = schedules.alpha_sigma(self.log_snr(step_t))
        a_s, s_s = schedules.alpha_sigma(self.log_snr(step_s))
        a_ts = a_t / a_s
        var = s_t.pow(2) - a_ts.pow(2) * s_s.pow(2)
        return x_s * a_ts + var.sqrt() * self.randn_like(x_s, rng=rng)

    # ---- reverse process -----------------------------------------------------------------------
    def _objective_id(self):
        if self.objective not in schedules.OBJECTIVES:
            raise ValueError(f"invalid objective {self.objective}")
        return schedules.OBJECTIVES[self.objective]

    def _clip(self):
        return float(self.clip_sample_range) if self.clip_sample else 0.0

    def _noise_for(self, x_t, rng, mode, ddim_eta, st=None):
        if mode == "ddpm" or ddim_eta != 0.0:
            return self.randn_like(x_t, rng=rng)
        if self.advance_rng_when_unused and rng is not None:
            if st is not None:                       # inside a sampling run: see finish_sampling
                st["unused_draws"] = st.get("unused_draws", 0) + 1
            else:
                self._discard_draws(x_t, rng, 1)
        return None

    @staticmethod
    def _discard_draws(x_t, rng, count):
        gens = [rng] if isinstance(rng, torch.Generator) else rng
        shape = x_t.shape if isinstance(rng, torch.Generator) else x_t.shape[1:]
        for _ in range(count):
            for g in gens:
                torch.randn(*shape, generator=g, device=g.device, dtype=x_t.dtype)

    def finish_sampling(self, st: dict) -> None:
        """Make the generator draws the loop deferred (DDIM eta = 0 with explicit generators).
        `sample()` calls it; users of the step-wise API call it after their last `sampling_step`
        when they share the generators with later runs."""
        n = st.pop("unused_draws", 0)
        if n:
            self._discard_draws(st["x"], st["rng"], n)

    def _predict(self, x_t, log_snr_t, time_features=None):
        if time_features is not None:
            return self.model(x_t, log_snr_t, time_features=time_features)
        return self.model(x_t, log_snr_t)

    @torch.compiler.disable
    @torch.inference_mode()
    def p_step(self, x_t, step_t, step_s, rng=None, mode: Literal["ddpm", "ddim"] = "ddpm",
               ddim_eta: float = 0.0):
        if mode not in schedules.MODES:
            raise ValueError(f"invalid mode {mode}")
        lam_t = self._schedule(step_t.float())
        lam_s = self._schedule(step_s.float())
        coef = schedules.step_coefficients(lam_t, lam_s, mode, ddim_eta, self._clip())
        pred = self._predict(x_t, lam_t.to(x_t.device))
        noise = self._noise_for(x_t, rng, mode, ddim_eta)
        return K.pstep(x_t, pred, noise, coef.to(x_t.device), self._objective_id(),
                       schedules.MODES[mode])

    def _plan(self, batch_size, num_steps, mode, ddim_eta, device, other_condition=None):
        """Host-side tables for a whole run: log-SNR rows [S*B] and coefficients [S,B,8]."""
        if mode not in schedules.MODES:
            raise ValueError(f"invalid mode {mode}")
        t = torch.linspace(1.0, 0.0, num_steps + 1)
        lam = self._schedule(t)
        coef = schedules.step_coefficients(lam[:-1], lam[1:], mode, ddim_eta, self._clip())
        coef = coef[:, None, :].expand(num_steps, batch_size, 8).contiguous().to(device)
        lam_rows = lam[:-1, None].expand(num_steps, batch_size).contiguous().to(device)
        tf_all = None
        if hasattr(self.model, "time_features"):
            if isinstance(other_condition, dict):
                tf_all = self.model.time_features(lam_rows.reshape(-1), other_condition)
            elif other_condition is None:
                tf_all = self.model.time_features(lam_rows.reshape(-1))
        return lam_rows, coef, tf_all

    def _resident_x(self, x):
        """Place x in the denoiser's persistent input buffer when it has one."""
        if hasattr(self.model, "_input_buffer"):
            slot = self.model._input_buffer(x.shape[0], x)[:, : x.shape[1]]
            return K.copy_into(slot, x)
        return x.clone()

    # ---- the sampling loop, exposed step-wise (bench.py times exactly K calls of `sampling_step`)
    @torch.compiler.disable
    @torch.inference_mode()
    def begin_sampling(self, batch_size: int, num_steps: int, rng=None,
                       mode: Literal["ddpm", "ddim"] = "ddpm", ddim_eta: float = 0.0, x_T=None,
                       condition_dict: dict = None):
        """Draw x_T (or take it), build the per-run tables, make x resident.  Returns a state dict
        consumed by `sampling_step`."""
        x = x_T if x_T is not None else self.randn(batch_size, *self.sampling_shape, rng=rng,
                                                   device=self.device)
        x0 = x.clone()
        other = None if condition_dict is None else condition_dict["other_condition"]
        lam_rows, coef, tf_all = self._plan(batch_size, num_steps, mode, ddim_eta, self.device,
                                            other)
        xr = self._resident_x(x)
        if isinstance(other, dict) and hasattr(self.model, "prepare_condition"):
            self.model.prepare_condition(other)   # step-invariant attention operands, once (after the resident input
        return dict(x=xr, x_T=x0, i=0, n=num_steps, B=batch_size, rng=rng,   # buffer of this batch size exists)
                    cond=condition_dict, graph=None,
                    needs_noise=(mode == "ddpm" or ddim_eta != 0.0),
                    mode=mode, eta=ddim_eta, lam=lam_rows, coef=coef, tf=tf_all,
                    obj=self._objective_id(), mid=schedules.MODES[mode])

    def _step_body(self, st, lam, tf, coef, noise):
        x = st["x"]
        if st["cond"] is not None:
            st["cond"].update(dict(time_condition=lam))      # mutates, like the reference
            pred = self._predict_cond(x, st["cond"], tf)
        else:
            pred = self._predict(x, lam, tf)
        K.pstep(x, pred, noise, coef, st["obj"], st["mid"], out=x)

    # ---- one captured HIP graph per step, kept ACROSS runs ----------------------------------------------------------
    # A run used to pay its first step eagerly plus a capture of the ~110-230 nodes of a step: ~20 ms per `sample()` call
    # -- 10 % of a 50-step batch of the bulk harness, and all of the "glue" of the temporal loop (one run per frame).  The
    # graph of a step only depends on addresses and routes, so it is kept on the sampler under a key that names all of
    # them (`_graph_key`) and a later run of the same shape replays it from its first step.
    graph_cache_size = 4

    def _weights_fingerprint(self):
        """(address, version) of every parameter and buffer below the denoiser and the condition model.  A plain walk over
        `_modules` (nn.Module.parameters() builds a dotted name per tensor: 2.8 ms for the layout model, once per run)."""
        fp, seen = [], set()
        stack = [self.model] + ([self.condition_model] if isinstance(self.condition_model, nn.Module) else [])
        while stack:
            m = stack.pop()
            if id(m) in seen:
                continue
            seen.add(id(m))
            for d in (m._parameters, m._buffers):
                for t in d.values():
                    if t is not None:
                        try:
                            v = t._version
                        except RuntimeError:        # inference tensors carry no version counter
                            v = -1
                        fp.append((t.data_ptr(), v))
            stack.extend(c for c in m._modules.values() if c is not None)
        return tuple(fp)

    def _graph_key(self, st):
        """Everything a captured step reads by address or was routed by; None = this run's graph cannot serve another
        run (a condition the denoiser does not expose through `graph_operands`)."""
        cond, x = st["cond"], st["x"]
        other = None if cond is None else cond["other_condition"]
        if other is None:
            csig = None
        elif isinstance(other, torch.Tensor):       # concatenated condition: kept in a buffer of the cache entry
            csig = ("tensor", tuple(other.shape), other.dtype)
        elif isinstance(other, dict) and hasattr(self.model, "graph_operands"):
            ops_ = self.model.graph_operands()
            if ops_ is None:
                return None
            csig = ("operands",) + tuple((t.data_ptr(), tuple(t.shape)) for t in ops_)
        else:
            return None
        tfsig = None if st["tf"] is None else tuple((tuple(a.shape[1:]), a.dtype) for a in st["tf"])
        # a denoiser with a resident input buffer: the step works in place at that address; otherwise the state lives in
        # a tensor of the cache entry (a hit copies x_T into it)
        xptr = x.data_ptr() if hasattr(self.model, "_input_buffer") else None
        return (st["B"], tuple(x.shape), xptr, x.device, st["needs_noise"], st["obj"], st["mid"],
                getattr(self, "cond_mode", None), csig, tfsig, K.route_signature(), self._weights_fingerprint())

    def _pack_rows(self, st):
        """The per-step parameters of a run (log-SNR, update coefficients, time features) as one [S, P] table: a replay
        is preceded by a single device copy of row i into the static row the graph reads."""
        B, x, S = st["B"], st["x"], st["n"]
        parts = [st["lam"].reshape(S, -1), st["coef"].reshape(S, -1)]
        shapes = [tuple(st["lam"].shape[1:]), tuple(st["coef"].shape[1:])]
        if st["tf"] is not None:
            for a in st["tf"]:
                parts.append(a.reshape(S, -1))
                shapes.append((B,) + tuple(a.shape[1:]))
        pad = lambda n: (n + 3) // 4 * 4                       # keep every view 16-byte aligned
        widths = [p.shape[1] for p in parts]
        offs, P = [], 0
        for w in widths:
            offs.append(P)
            P += pad(w)
        table = torch.zeros((S, P), device=x.device, dtype=torch.float32)
        for p, o, w in zip(parts, offs, widths):
            table[:, o:o + w] = p
        return table, offs, widths, shapes

    def _capture(self, st):
        """Record one step (denoiser forward + fused update) into a HIP graph that reads its per-step parameters from
        ONE static row."""
        x = st["x"]
        table, offs, widths, shapes = self._pack_rows(st)
        P = table.shape[1]
        row = torch.empty((P,), device=x.device, dtype=torch.float32)
        views = [row[o:o + w].view(shp) for o, w, shp in zip(offs, widths, shapes)]
        g = dict(table=table, row=row, lam=views[0], coef=views[1], x=x, P=P,
                 tf=None if st["tf"] is None else tuple(views[2:]),
                 noise=torch.empty_like(x).contiguous() if st["needs_noise"] else None, other=None)
        cond = st["cond"]
        if cond is not None and isinstance(cond["other_condition"], torch.Tensor):
            g["other"] = cond["other_condition"].clone()       # the step reads the condition from the entry's buffer
            cond["other_condition"] = g["other"]
        row.copy_(table[st["i"]])
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            self._step_body(st, g["lam"], g["tf"], g["coef"], g["noise"])
        g["graph"] = graph
        return g

    def _cached_graph(self, st):
        """The graph an earlier run captured under this run's key, bound to this run's table; else None."""
        cache = _GRAPH_CACHES.get(self)
        if not cache:
            return None
        key = self._graph_key(st)
        if key is None or key not in cache:
            return None
        shared = cache.pop(key)
        cache[key] = shared                                    # most recently used last
        table = self._pack_rows(st)[0]
        if table.shape[1] != shared["P"]:
            return None
        cond = st["cond"]
        if shared["other"] is not None:
            shared["other"].copy_(cond["other_condition"])
            cond["other_condition"] = shared["other"]
        if cond is not None:
            cond.update(dict(time_condition=shared["lam"]))    # what an eager first step leaves in the caller's dict
        if shared["x"].data_ptr() != st["x"].data_ptr():
            st["x"] = K.copy_into(shared["x"], st["x"])
        return dict(shared, table=table)

    def _remember_graph(self, st, g):
        key = self._graph_key(st) if self.graph_cache_size > 0 else None   # (now: the eager first step may have packed
        if key is None:                                                     # weights and assigned range slots)
            return
        cache = _GRAPH_CACHES.setdefault(self, {})
        cache[key] = g
        while len(cache) > self.graph_cache_size:
            cache.pop(next(iter(cache)))

    @torch.compiler.disable
    @torch.inference_mode()
    def sampling_step(self, st: dict) -> torch.Tensor:
        """One reverse step: denoiser forward + fused x0/clamp/update, in place on the resident x."""
        with K.defer_range_checks():     # callers of the step-wise API poll (ops.range_poll) themselves
            return self._sampling_step(st)

    def _sampling_step(self, st: dict) -> torch.Tensor:
        i, B, x = st["i"], st["B"], st["x"]
        tf = None if st["tf"] is None else tuple(a[i * B:(i + 1) * B] for a in st["tf"])
        noise = self._noise_for(x, st["rng"], st["mode"], st["eta"], st)
        graphable = (self.use_hip_graph and x.is_cuda and K.PROFILE is None and st["n"] > 2)
        if graphable and st.get("graph") is None and i == 0 and self.graph_cache_size > 0:
            st["graph"] = self._cached_graph(st)                 # a later run of a known shape: replay from step 0
        if graphable and (i >= 1 or st.get("graph") is not None):
            if st.get("graph") is None:
                try:
                    st["graph"] = self._capture(st)      # step 0 ran eagerly: caches are warm
                    self._remember_graph(st, st["graph"])
                except Exception as e:                    # capture is an optimisation only:
                    import warnings                       # keep launching the same kernels eagerly

                    warnings.warn(f"HIP graph capture failed ({e!r}); continuing with eager launches")
                    st["graph"] = False
            g = st["graph"]
            if g is False:
                self._step_body(st, st["lam"][i], tf, st["coef"][i], noise)
                st["i"] = i + 1
                return x
            g["row"].copy_(g["table"][i])
            if g["noise"] is not None:
                g["noise"].copy_(noise)
            g["graph"].replay()
        else:
            self._step_body(st, st["lam"][i], tf, st["coef"][i], noise)
        st["i"] = i + 1
        return x

    @torch.inference_mode()
    def sample(self, batch_size: int, num_steps: int, progress: bool = True, rng=None,
               return_all: bool = False, mode: Literal["ddpm", "ddim"] = "ddpm",
               ddim_eta: float = 0.0):
        def run():
            st = self.begin_sampling(batch_size, num_steps, rng, mode, ddim_eta)
            out = [st["x_T"]] if return_all else None
            for _ in tqdm(range(num_steps), desc="sampling", leave=False, disable=not progress):
                x = self.sampling_step(st)
                if return_all:
                    out.append(x.clone())
            self.finish_sampling(st)
            return torch.stack(out) if return_all else st["x"].clone()

        # the conv range records are polled ONCE, after the loop; a run in which a layer's fp16
        # operands saturated is repeated with the same draws (ops.run_range_safe)
        return K.run_range_safe(run, rng, self.device, "ContinuousTimeGaussianDiffusion.sample")

    @torch.compiler.disable
    @torch.inference_mode()
    def repaint(self, known, mask, num_steps, num_resample_steps: int = 1, jump_length: int = 1,
                progress: bool = True, rng=None, return_all: bool = False):
        """RePaint (https://arxiv.org/abs/2201.09865), reference continuous_time.py:262-330."""
        assert num_resample_steps > 0 and jump_length > 0
        B = known.shape[0]
        x_t = self.randn(B, *self.sampling_shape, rng=rng, device=self.device)
        steps = torch.linspace(1, 0, num_steps + 1, device=self.device)[None].repeat_interleave(B, 0)
        out = [x_t] if return_all else None
        x_s = x_t
        for i in tqdm(range(num_steps), desc="RePaint", leave=False, disable=not progress):
            for j in range(num_resample_steps):
                interp = torch.linspace(0, 1, jump_length + 1, device=self.device)
                r = steps[:, [i]] + interp[None] * (steps[:, [i + 1]] - steps[:, [i]])
                x = x_t
                for k in range(jump_length):
                    known_s, _ = self.q_step_from_x_0(known, r[:, k + 1], rng=rng)
                    unknown_s = self.p_step(x, r[:, k], r[:, k + 1], rng=rng)
                    x = mask * known_s + (1 - mask) * unknown_s
                x_s = x
                if return_all:
                    out.append(x_s)
                if i == num_steps - 1 or j == num_resample_steps - 1:
                    x_t = x
                    break
                for k in range(jump_length, 0, -1):
                    x = self.q_step(x, r[:, k - 1], r[:, k], rng=rng)
                x_t = x
        return torch.stack(out) if return_all else x_s
