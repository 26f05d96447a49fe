"""Discrete-time DDPM/DDIM (https://arxiv.org/abs/2006.11239) -- API mirror of the reference's
lidargen/models/diffusion/discrete_time.py:52-201.  Secondary: no shipped config selects it
(SURVEY.md §2-1b); the denoiser forward is the same HIP path and the update is the same fused
kernel as the continuous-time samplers (`lc_pstep_fwd`, modes 2 / 3): the per-sample coefficients are
tabulated on the host with the reference's fp32 torch expressions (bit-identical scalars), one
[B, 8] row per step."""
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

    def _tables(self):
        """CPU copies of the schedule buffers (the coefficient algebra runs on the host, in fp32,
        with the reference's own expressions)."""
        t = getattr(self, "_cpu_tables", None)
        if t is None or t[0] is not self.beta:
            t = (self.beta, self.beta.detach().cpu().reshape(-1), self.alpha_bar.detach().cpu().reshape(-1),
                 self.alpha_bar_prev.detach().cpu().reshape(-1))
            self._cpu_tables = t
        return t[1:]

    def step_coefficients(self, steps: torch.Tensor, mode: str, eta: float = 0.0) -> torch.Tensor:
        """[B] integer steps -> [B, 8] coefficient rows of `lc_pstep_fwd` modes 2 (ddpm) / 3 (ddim):
        reference discrete_time.py:126-180, every scalar computed by the same fp32 expression."""
        if mode not in ("ddpm", "ddim"):
            raise ValueError(f"invalid mode {mode}")
        beta_t, ab_t, abp_t = self._tables()
        idx = steps.detach().cpu().long()
        beta, ab, abp = beta_t[idx], ab_t[idx], abp_t[idx]
        alpha = 1 - beta
        zero = torch.zeros_like(ab)
        if self.objective == "eps":
            A, Bc = ab.rsqrt(), (ab.reciprocal() - 1).sqrt()
        elif self.objective == "v":
            A, Bc = ab.sqrt(), (1 - ab).sqrt()
        elif self.objective == "x_0":
            A, Bc = zero, zero
        else:
            raise ValueError(f"invalid objective {self.objective}")
        clip = torch.full_like(ab, float(self.clip_sample_range) if self.clip_sample else 0.0)
        live = (idx != 0).float()                                  # `nz[steps == 0] *= 0`
        if mode == "ddpm":
            c2 = abp.sqrt() * beta / (1 - ab)
            c3 = (1 - abp) * alpha.sqrt() / (1 - ab)
            var = (beta * (1 - abp) / (1 - ab)).clamp(min=1e-20)
            c4 = (0.5 * var.log()).exp() * live
            rows = [A, Bc, c2, c3, c4, zero, clip, zero]
        else:
            var = (1 - abp) / (1 - ab) * (1 - ab / abp)
            sd = eta * var.sqrt()
            rows = [A, Bc, ab.sqrt(), (1 - ab).sqrt(), abp.sqrt(), (1 - abp - sd ** 2).sqrt(), clip, sd * live]
        return torch.stack(rows, dim=1).float().contiguous()

    @torch.compiler.disable
    @torch.inference_mode()
    def p_step(self, x_t, steps, rng=None, mode: Literal["ddpm", "ddim"] = "ddim", eta: float = 0.0,
               coef: torch.Tensor = None):
        if coef is None:
            coef = self.step_coefficients(steps, mode, eta).to(x_t.device)
        pred = self.model(x_t, steps)
        noise = self.randn_like(x_t, rng=rng) if (mode == "ddpm" or eta > 0) else None
        return K.pstep(x_t, pred, noise, coef, schedules.OBJECTIVES[self.objective],
                       2 if mode == "ddpm" else 3)

    @torch.inference_mode()
    def sample(self, batch_size, num_steps, progress=True, rng=None, return_all=False,
               mode: Literal["ddpm", "ddim"] = "ddpm"):
        def run():
            x = self.randn(batch_size, *self.sampling_shape, rng=rng, device=self.device)
            out = [x] if return_all else None
            order = list(reversed(range(num_steps)))
            # all coefficient rows of the run in ONE host tabulation + ONE upload
            table = self.step_coefficients(torch.tensor(order).repeat_interleave(batch_size), mode)
            table = table.view(num_steps, batch_size, 8).to(self.device)
            for i, ts in enumerate(tqdm(order, desc="sampling", leave=False, disable=not progress)):
                steps = torch.full((batch_size,), ts, device=self.device).long()
                x = self.p_step(x, steps, rng=rng, mode=mode, coef=table[i])
                if return_all:
                    out.append(x)
            return torch.stack(out) if return_all else x

        return K.run_range_safe(run, rng, self.device, "DiscreteTimeGaussianDiffusion.sample")
