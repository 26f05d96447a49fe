"""Noise schedules and per-step coefficient tables (host side, fp32, computed ONCE per sample()).

Formulas follow the reference: continuous_time.py:14-63 (log-SNR schedules, alpha/sigma),
:220-231 (DDPM / DDIM update coefficients); discrete_time.py:12-78 (beta tables).
The sampling loop of the reference recomputes these B-element tensors with ~20 tiny device
kernels per step; here a [S, B, 8] table is built on the host with the same fp32 torch ops
(bit-identical to the CPU reference) and uploaded once, and the device-side update is one fused
kernel (lc_pstep_fwd).
"""
from __future__ import annotations

import math
from functools import partial

import torch
from torch.special import expm1


def _log(t, eps=1e-20):
    return torch.log(t.clamp(min=eps))


def log_snr_linear(t):
    return -_log(expm1(1e-4 + 10 * (t ** 2)))


def log_snr_cosine(t, logsnr_min=-15.0, logsnr_max=15.0):
    t_min = math.atan(math.exp(-0.5 * logsnr_max))
    t_max = math.atan(math.exp(-0.5 * logsnr_min))
    return -2 * _log(torch.tan(t_min + t * (t_max - t_min)))


def log_snr_cosine_shifted(t, image_d, noise_d, logsnr_min=-15.0, logsnr_max=15.0):
    return log_snr_cosine(t, logsnr_min, logsnr_max) + 2 * math.log(noise_d / image_d)


def log_snr_cosine_interpolated(t, image_d, noise_d_low, noise_d_high, logsnr_min=-15.0,
                                logsnr_max=15.0):
    lo = log_snr_cosine_shifted(t, image_d, noise_d_low, logsnr_min, logsnr_max)
    hi = log_snr_cosine_shifted(t, image_d, noise_d_high, logsnr_min, logsnr_max)
    return t * lo + (1 - t) * hi


def make_schedule(name, image_d=None, noise_d_low=None, noise_d_high=None):
    if name == "linear":
        return log_snr_linear
    if name == "cosine":
        return log_snr_cosine
    if name == "cosine_shifted":
        assert image_d is not None and noise_d_low is not None
        return partial(log_snr_cosine_shifted, image_d=image_d, noise_d=noise_d_low)
    if name == "cosine_interpolated":
        assert image_d is not None and noise_d_low is not None and noise_d_high is not None
        return partial(log_snr_cosine_interpolated, image_d=image_d, noise_d_low=noise_d_low,
                       noise_d_high=noise_d_high)
    raise ValueError(f"invalid beta schedule: {name}")


def alpha_sigma(log_snr):
    return log_snr.sigmoid().sqrt(), (-log_snr).sigmoid().sqrt()


OBJECTIVES = {"eps": 0, "v": 1, "x_0": 2}
MODES = {"ddpm": 0, "ddim": 1}


def step_coefficients(log_snr_t, log_snr_s, mode: str, ddim_eta: float, clip: float):
    """log_snr_t/s: [...] fp32 -> coef [..., 8] =
    (alpha_t, sigma_t, alpha_s, sigma_s, k0, k1, clip, 0); see include/lidarcrafter_hip.h."""
    a_t, s_t = alpha_sigma(log_snr_t)
    a_s, s_s = alpha_sigma(log_snr_s)
    if mode == "ddpm":
        k0 = -expm1(log_snr_t - log_snr_s)
        k1 = s_s * k0.sqrt()
    elif mode == "ddim":
        k0 = ddim_eta * s_s / s_t * (1 - a_t ** 2 / a_s ** 2).sqrt()
        k1 = (1 - a_s ** 2 - k0 ** 2).sqrt()
    else:
        raise ValueError(f"invalid mode {mode}")
    z = torch.zeros_like(a_t)
    return torch.stack([a_t, s_t, a_s, s_s, k0, k1, z + float(clip), z], dim=-1).contiguous()


# ------------------------------------------------------------------------------- discrete time
def beta_schedule(kind: str, steps: int) -> torch.Tensor:
    if kind == "linear":
        scale = 1000 / steps
        return torch.linspace(scale * 0.0001, scale * 0.02, steps, dtype=torch.float64)
    t = torch.linspace(0, steps, steps + 1, dtype=torch.float64) / steps
    if kind == "cosine":
        s = 0.008
        abar = torch.cos((t + s) / (1 + s) * math.pi * 0.5) ** 2
    elif kind == "sigmoid":
        start, end, tau = -3, 3, 1
        v0, v1 = torch.tensor(start / tau).sigmoid(), torch.tensor(end / tau).sigmoid()
        abar = (-((t * (end - start) + start) / tau).sigmoid() + v1) / (v1 - v0)
    else:
        raise ValueError(f"invalid beta schedule {kind}")
    abar = abar / abar[0]
    return torch.clip(1 - (abar[1:] / abar[:-1]), 0, 0.999)
