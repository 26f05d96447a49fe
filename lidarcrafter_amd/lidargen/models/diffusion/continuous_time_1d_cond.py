"""Conditional diffusion over per-object point sets -- API mirror of the reference's
lidargen/models/diffusion/continuous_time_1d_cond.py:9-91
(`CondContinuousLayoutGaussianDiffusion1D`): the continuous-time process of
`CondContinuousTimeGaussianDiffusion` on x [B, N = 1024 points, C = 4] instead of [B, C, H, W]
(`sampling_shape` = (N, C)), denoiser `PointUNet`, condition `ObjectGenEncoder`.

The sampling loop is the inherited one (schedule tables on the host once per run, the fused
x0 / clamp / update kernel lc_pstep_fwd, one replayed HIP graph per step); the per-sample schedule
scalars broadcast over [N, C] exactly as the reference's `squeeze_(-1)` forms do, so the point set
is simply viewed as a one-row image [B, 1, N, C] for the elementwise update."""
from __future__ import annotations

import contextlib

from typing import Literal

import torch

from lidarcrafter_amd import ops as K

from . import schedules
from .continuous_time_cond import CondContinuousTimeGaussianDiffusion


class CondContinuousLayoutGaussianDiffusion1D(CondContinuousTimeGaussianDiffusion):
    def __init__(self, model, condition_model=None, prediction_type="eps", loss_type="l2",
                 noise_schedule="cosine", min_snr_loss_weight=True, min_snr_gamma=5,
                 sampling_resolution=None, clip_sample=True, clip_sample_range=1, image_d=None,
                 noise_d_low=None, noise_d_high=None, cond_mode=None):
        super().__init__(model, condition_model, prediction_type, loss_type, noise_schedule,
                         min_snr_loss_weight, min_snr_gamma, sampling_resolution, clip_sample,
                         clip_sample_range, image_d, noise_d_low, noise_d_high, cond_mode=cond_mode)
        self.sampling_shape = (self.sampling_shape[1], self.sampling_shape[0])      # (N, C)

    # x [B, N, C] <-> the [B, 1, N, C] view the elementwise kernels take
    @staticmethod
    def _img(x):
        return x.reshape(x.shape[0], 1, x.shape[-2], x.shape[-1])

    def q_step_from_x_0(self, x_0, step_t, rng=None):
        noise = self.randn_like(x_0, rng=rng)
        alpha, sigma = schedules.alpha_sigma(self.log_snr(step_t))
        return x_0 * alpha.squeeze(-1) + noise * sigma.squeeze(-1), noise

    def _predict_cond(self, x_t, condition_dict, time_features=None):
        pts = x_t.reshape(x_t.shape[0], x_t.shape[-2], x_t.shape[-1])
        pred = self.model(pts, condition_dict)
        return self._img(pred) if x_t.dim() == 4 else pred

    @torch.compiler.disable
    @torch.inference_mode()
    def p_step(self, x_t, condition_dict: dict, step_t, step_s, rng=None,
               mode: Literal["ddpm", "ddim"] = "ddpm", ddim_eta: float = 0.0):
        if mode not in schedules.MODES:
            raise ValueError(f"invalid mode {mode}")
        lam_t = self._schedule(step_t.float())
        lam_s = self._schedule(step_s.float())
        coef = schedules.step_coefficients(lam_t, lam_s, mode, ddim_eta, self._clip())
        condition_dict.update(dict(time_condition=lam_t.to(x_t.device)))   # mutates, like the ref
        pred = self._predict_cond(x_t, condition_dict)
        noise = self._noise_for(x_t, rng, mode, ddim_eta)
        y = K.pstep(self._img(x_t.float().contiguous()), self._img(pred),
                    None if noise is None else self._img(noise.contiguous()),
                    coef.to(x_t.device), self._objective_id(), schedules.MODES[mode])
        return y.reshape(x_t.shape)

    @torch.compiler.disable
    @torch.inference_mode()
    def begin_sampling(self, batch_size, num_steps, rng=None, mode="ddpm", ddim_eta=0.0, x_T=None,
                       condition_dict=None):
        st = super().begin_sampling(batch_size, num_steps, rng, mode, ddim_eta,
                                    x_T=None if x_T is None else self._img(x_T),
                                    condition_dict=condition_dict)
        return st

    def randn(self, *shape, rng=None, **kwargs):
        # the sampler asks for [B, N, C] (and for noise like the [B, 1, N, C] state): draw in the
        # reference's shape -- the same numbers in the same order -- then view
        if len(shape) == 4 and shape[1] == 1:
            return self._img(super().randn(shape[0], shape[2], shape[3], rng=rng, **kwargs))
        return super().randn(*shape, rng=rng, **kwargs)

    @torch.inference_mode()
    def sample(self, batch_dict: dict, batch_size: int, num_steps: int, progress: bool = True,
               rng=None, return_all: bool = False, mode: Literal["ddpm", "ddim"] = "ddpm",
               ddim_eta: float = 0.0):
        out = super().sample(batch_dict, batch_size, num_steps, progress=progress, rng=rng,
                             return_all=return_all, mode=mode, ddim_eta=ddim_eta)
        return out.squeeze(2) if return_all else out.squeeze(1)

    def p_loss(self, input_dict: dict, steps, loss_mask=None):
        x_0 = input_dict["x_0"]
        loss_mask = torch.ones_like(x_0) if loss_mask is None else loss_mask
        x_t, noise = self.q_step_from_x_0(x_0, steps)
        from lidarcrafter_amd import autograd as AG

        train = AG.training_active(self.model) or AG.training_active(self.condition_model)
        with (contextlib.nullcontext() if train else torch.no_grad()):
            condition = self.get_network_condition(steps, input_dict)
            prediction = self.model(x_t, condition)
        return self._masked_loss(prediction, self.get_target(x_0, steps, noise), loss_mask, steps)
