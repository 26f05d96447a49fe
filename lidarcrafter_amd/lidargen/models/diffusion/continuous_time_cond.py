"""Conditional continuous-time diffusion -- API mirror of the reference's
lidargen/models/diffusion/continuous_time_cond.py (get_network_condition :146-152,
p_step :206-253, sample :255-281, inpaint :283-353, p_loss/forward :414-456).
The condition dict is built ONCE per batch; per step only `time_condition` changes."""
from __future__ import annotations

import contextlib

from typing import Literal

import torch
from torch import nn
from tqdm.auto import tqdm

from lidarcrafter_amd import ops as K

from . import continuous_time, schedules


class CondContinuousTimeGaussianDiffusion(continuous_time.ContinuousTimeGaussianDiffusion):
    def __init__(self, model: nn.Module, condition_model: nn.Module = None, *args,
                 cond_mode: str = None, w_loss_weight: bool = False, **kwargs):
        super().__init__(model, condition_model, *args, **kwargs)
        self.cond_mode = cond_mode
        if self.cond_mode == "concat":
            # reference :107-112 -- the denoiser's in_channels include the concat condition
            self.sampling_shape = (self.model.in_channels - condition_model.out_channels,
                                   *self.sampling_shape[1:])
        self.w_loss_weight = w_loss_weight

    def get_network_condition(self, steps=None, input_dict=None, only_custom_condition=False):
        other = self.condition_model(input_dict)
        if only_custom_condition:
            return dict(other_condition=other)
        return dict(time_condition=self.log_snr(steps)[:, 0, 0, 0], other_condition=other)

    def _predict_cond(self, x_t, condition_dict, time_features=None):
        other = condition_dict["other_condition"]
        kw = {} if time_features is None else {"time_features": time_features}
        if self.cond_mode == "concat" and isinstance(other, torch.Tensor):
            td = dict(time_condition=condition_dict["time_condition"])
            return self.model(torch.cat([x_t, other], dim=1), td, **kw)
        return self.model(x_t, condition_dict, **kw)

    @torch.compiler.disable
    @torch.inference_mode()
    def p_step(self, x_t, condition_dict: dict, step_t, step_s, rng=None,
               mode: Literal["ddpm", "ddim"] = "ddpm", ddim_eta: float = 0.0):
        if mode not in schedules.MODES:
            raise ValueError(f"invalid mode {mode}")
        lam_t = self._schedule(step_t.float())
        lam_s = self._schedule(step_s.float())
        coef = schedules.step_coefficients(lam_t, lam_s, mode, ddim_eta, self._clip())
        condition_dict.update(dict(time_condition=lam_t.to(x_t.device)))  # mutates, like the ref
        pred = self._predict_cond(x_t, condition_dict)
        noise = self._noise_for(x_t, rng, mode, ddim_eta)
        return K.pstep(x_t, pred, noise, coef.to(x_t.device), self._objective_id(),
                       schedules.MODES[mode])

    @torch.inference_mode()
    def sample(self, batch_dict: dict, batch_size: int, num_steps: int, progress: bool = True,
               rng=None, return_all: bool = False, mode: Literal["ddpm", "ddim"] = "ddpm",
               ddim_eta: float = 0.0):
        def run():
            x_T = self.randn(batch_size, *self.sampling_shape, rng=rng, device=self.device)
            condition_dict = self.get_network_condition(input_dict=batch_dict,
                                                        only_custom_condition=True)
            st = self.begin_sampling(batch_size, num_steps, rng, mode, ddim_eta, x_T=x_T,
                                     condition_dict=condition_dict)
            out = [st["x_T"]] if return_all else None
            for _ in tqdm(range(num_steps), desc="sampling", leave=False, disable=not progress):
                x = self.sampling_step(st)
                if return_all:
                    out.append(x.clone())
            self.finish_sampling(st)
            return torch.stack(out) if return_all else st["x"].clone()

        return K.run_range_safe(run, rng, self.device, "CondContinuousTimeGaussianDiffusion.sample")

    @torch.compiler.disable
    @torch.inference_mode()
    def inpaint(self, known, mask, batch_dict: dict, num_steps: int, num_resample_steps: int = 1,
                jump_length: int = 1, progress: bool = True, rng=None, return_all: bool = False):
        assert num_resample_steps > 0 and jump_length > 0
        B = known.shape[0]
        x_t = self.randn(B, *self.sampling_shape, rng=rng, device=self.device)
        cond = self.get_network_condition(input_dict=batch_dict, only_custom_condition=True)
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
                    unknown_s = self.p_step(x, cond, r[:, k], r[:, k + 1], rng=rng)
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

    def p_loss(self, input_dict: dict, steps, loss_mask=None):
        x_0 = input_dict["x_0"]
        loss_mask = torch.ones_like(x_0) if loss_mask is None else loss_mask
        x_t, noise = self.q_step_from_x_0(x_0, steps)
        from lidarcrafter_amd import autograd as AG

        # grad mode on + trainable parameters: the denoiser (and the layout encoder feeding it) build
        # an autograd graph, `ddpm(batch).backward()` works as in tools/train/train_lidm_cond.py
        train = AG.training_active(self.model) or AG.training_active(self.condition_model)
        with (contextlib.nullcontext() if train else torch.no_grad()):
            condition = self.get_network_condition(steps, input_dict)
            prediction = self._predict_cond(x_t, condition)
        return self._masked_loss(prediction, self.get_target(x_0, steps, noise), loss_mask, steps)

    def forward(self, input_dict: dict):
        x_0 = input_dict["x_0"]
        steps = self.sample_timesteps(x_0.shape[0], x_0.device)
        loss_mask = None
        if self.w_loss_weight:
            w = input_dict.get("scene_loss_weight_map", None)
            if w is not None:
                loss_mask = w.unsqueeze(1).repeat(1, x_0.shape[1], 1, 1)
        return self.p_loss(input_dict, steps, loss_mask)
