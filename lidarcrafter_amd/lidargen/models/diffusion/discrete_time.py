"""Discrete-time DDPM/DDIM (https://arxiv.org/abs/2006.11239) -- API mirror of the reference's
lidargen/models/diffusion/discrete_time.py:52-201.  Secondary: no shipped config selects it
(SURVEY.md §2-1b); the denoiser forward is the same HIP path, the B-element table lookups and the
update run as elementwise tensor ops."""
from __future__ import annotations

from typing import Literal

import torch
import torch.nn.functional as F
from tqdm.auto import tqdm

from lidarcrafter_amd import ops as K

from . import base, schedules


class DiscreteTimeGaussianDiffusion(base.GaussianDiffusion):
    def setup_parameters(self) -> None:
        assert self.num_training_steps is not None
        beta = schedules.beta_schedule(self.noise_schedule, self.num_training_steps)
        beta = beta[:, None, None, None]
        alpha_bar = torch.cumprod(1 - beta, dim=0)
        alpha_bar_prev = F.pad(alpha_bar[:-1], (0,) * 6 + (1, 0), value=1.0)
        self.register_buffer("beta", beta.float())
        self.register_buffer("alpha_bar", alpha_bar.float())
        self.register_buffer("alpha_bar_prev", alpha_bar_prev.float())
        self.register_buffer("snr", (alpha_bar / (1 - alpha_bar)).float())

    def sample_timesteps(self, batch_size, device):
        return torch.randint(0, self.num_training_steps, (batch_size,), device=device,
                             dtype=torch.long)

    def get_network_condition(self, steps):
        return steps

    def get_target(self, x_0, steps, noise):
        if self.objective == "eps":
            return noise
        if self.objective == "x_0":
            return x_0
        if self.objective == "v":
            ab = self.alpha_bar[steps]
            return ab.sqrt() * noise - (1 - ab).sqrt() * x_0
        raise ValueError(f"invalid objective {self.objective}")

    def get_loss_weight(self, timesteps):
        snr = self.snr[timesteps]
        clipped = snr.clamp(max=self.min_snr_gamma) if self.min_snr_loss_weight else snr.clone()
        if self.objective == "eps":
            return clipped / snr
        if self.objective == "x_0":
            return clipped
        if self.objective == "v":
            return clipped / (snr + 1)
        raise ValueError(f"invalid objective {self.objective}")

    def q_step_from_x_0(self, x_0, steps, rng=None):
        noise = self.randn_like(x_0, rng=rng)
        ab = self.alpha_bar[steps]
        return ab.sqrt() * x_0 + (1 - ab).sqrt() * noise, noise

    @torch.compiler.disable
    @torch.inference_mode()
    def p_step(self, x_t, steps, rng=None, mode: Literal["ddpm", "ddim"] = "ddim", eta: float = 0.0):
        beta, ab, abp = self.beta[steps], self.alpha_bar[steps], self.alpha_bar_prev[steps]
        alpha = 1 - beta
        pred = self.model(x_t, steps)
        if self.objective == "eps":
            x_0 = ab.rsqrt() * x_t - (ab.reciprocal() - 1).sqrt() * pred
        elif self.objective == "x_0":
            x_0 = pred
        elif self.objective == "v":
            x_0 = ab.sqrt() * x_t - (1 - ab).sqrt() * pred
        else:
            raise ValueError(f"invalid objective {self.objective}")
        if self.clip_sample:
            x_0 = x_0.clamp(-self.clip_sample_range, self.clip_sample_range)
        if mode == "ddpm":
            mean = abp.sqrt() * beta / (1 - ab) * x_0 + (1 - abp) * alpha.sqrt() / (1 - ab) * x_t
            var = (beta * (1 - abp) / (1 - ab)).clamp(min=1e-20)
            nz = self.randn_like(x_t, rng=rng)
            nz[steps == 0] *= 0
            return mean + (0.5 * var.log()).exp() * nz
        if mode == "ddim":
            var = (1 - abp) / (1 - ab) * (1 - ab / abp)
            sd = eta * var.sqrt()
            eps = (x_t - ab.sqrt() * x_0) / (1 - ab).sqrt()
            x_s = abp.sqrt() * x_0 + (1 - abp - sd ** 2).sqrt() * eps
            if eta > 0:
                nz = self.randn_like(x_t, rng=rng)
                nz[steps == 0] *= 0
                x_s = x_s + sd * nz
            return x_s
        raise ValueError(f"invalid mode {mode}")

    @torch.inference_mode()
    def sample(self, batch_size, num_steps, progress=True, rng=None, return_all=False,
               mode: Literal["ddpm", "ddim"] = "ddpm"):
        def run():
            x = self.randn(batch_size, *self.sampling_shape, rng=rng, device=self.device)
            out = [x] if return_all else None
            for ts in tqdm(list(reversed(range(num_steps))), desc="sampling", leave=False,
                           disable=not progress):
                steps = torch.full((batch_size,), ts, device=self.device).long()
                x = self.p_step(x, steps, rng=rng, mode=mode)
                if return_all:
                    out.append(x)
            return torch.stack(out) if return_all else x

        return K.run_range_safe(run, rng, self.device, "DiscreteTimeGaussianDiffusion.sample")
