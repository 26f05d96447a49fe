"""CPU restatement of the reference diffusion process (continuous + discrete time).
TEST INFRASTRUCTURE -- see oracle/__init__.py.

Follows /root/reference/lidargen/models/diffusion/continuous_time.py:
  :22-29 cosine log-SNR schedule, :61-63 alpha/sigma, :195-234 p_step, :237-260 sample;
base.py:73-96 RNG contract; discrete_time.py:57-78 tables, :126-180 p_step, :182-201 sample.
`denoise(x_t, cond)` is any callable (oracle UNet, or a stub returning a fixed prediction).
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def log_snr_cosine(t: torch.Tensor, lo: float = -15.0, hi: float = 15.0) -> torch.Tensor:
    t_min = math.atan(math.exp(-0.5 * hi))
    t_max = math.atan(math.exp(-0.5 * lo))
    return -2 * torch.log(torch.tan(t_min + t * (t_max - t_min)).clamp(min=1e-20))


def log_snr_linear(t: torch.Tensor) -> torch.Tensor:
    return -torch.log(torch.special.expm1(1e-4 + 10 * (t ** 2)).clamp(min=1e-20))


def alpha_sigma(log_snr: torch.Tensor):
    return log_snr.sigmoid().sqrt(), (-log_snr).sigmoid().sqrt()


def randn(shape, rng, device="cpu"):
    """base.py:73-89: None -> global, Generator -> one stream, list -> one stream per sample."""
    if rng is None:
        return torch.randn(*shape, device=device)
    if isinstance(rng, torch.Generator):
        return torch.randn(*shape, generator=rng, device=device)
    assert len(rng) == shape[0]
    return torch.stack([torch.randn(*shape[1:], generator=r, device=device) for r in rng])


def x0_from_prediction(x_t, pred, alpha_t, sigma_t, objective):
    if objective == "eps":
        return (x_t - sigma_t * pred) / alpha_t
    if objective == "v":
        return alpha_t * x_t - sigma_t * pred
    if objective == "x_0":
        return pred
    raise ValueError(objective)


@torch.no_grad()
def p_step(denoise, x_t, step_t, step_s, noise, *, mode="ddpm", ddim_eta=0.0, objective="eps",
           clip=True, clip_range=1.0, schedule=log_snr_cosine):
    """One reverse step.  `noise` is the randn_like draw the reference makes in BOTH modes."""
    lt = schedule(step_t)[:, None, None, None]
    ls = schedule(step_s)[:, None, None, None]
    a_t, s_t = alpha_sigma(lt)
    a_s, s_s = alpha_sigma(ls)
    pred = denoise(x_t, lt[:, 0, 0, 0])
    x0 = x0_from_prediction(x_t, pred, a_t, s_t, objective)
    if clip:
        x0 = x0.clamp(-clip_range, clip_range)
    if mode == "ddpm":
        c = -torch.special.expm1(lt - ls)
        mean = a_s * (x_t * (1 - c) / a_t + c * x0)
        return mean + s_s * c.sqrt() * noise
    if mode == "ddim":
        c1 = ddim_eta * s_s / s_t * (1 - a_t ** 2 / a_s ** 2).sqrt()
        c2 = (1 - a_s ** 2 - c1 ** 2).sqrt()
        eps = (x_t - a_t * x0) / s_t
        return a_s * x0 + c1 * noise + c2 * eps
    raise ValueError(mode)


@torch.no_grad()
def sample(denoise, shape, num_steps, rng, *, mode="ddpm", ddim_eta=0.0, objective="eps",
           return_all=False, schedule=log_snr_cosine, clip=True):
    """continuous_time.py:237-260.  shape = (B, C, H, W)."""
    B = shape[0]
    x = randn(shape, rng)
    out = [x]
    steps = torch.linspace(1.0, 0.0, num_steps + 1)[None].repeat_interleave(B, dim=0)
    for i in range(num_steps):
        noise = randn(tuple(x.shape), rng)
        x = p_step(denoise, x, steps[:, i], steps[:, i + 1], noise, mode=mode,
                   ddim_eta=ddim_eta, objective=objective, schedule=schedule, clip=bool(clip))
        out.append(x)
    return torch.stack(out) if return_all else x


def q_step_from_x_0(x_0, steps, rng, schedule=log_snr_cosine):
    """continuous_time.py:171-178: (x_t, noise) with x_t = alpha x_0 + sigma noise."""
    noise = randn(tuple(x_0.shape), rng)
    a, s = alpha_sigma(schedule(steps)[:, None, None, None])
    return x_0 * a + noise * s, noise


def q_step(x_s, step_t, step_s, rng, schedule=log_snr_cosine):
    """continuous_time.py:180-193: q(z_t | z_s), 0 < s < t < 1."""
    a_t, s_t = alpha_sigma(schedule(step_t)[:, None, None, None])
    a_s, s_s = alpha_sigma(schedule(step_s)[:, None, None, None])
    a_ts = a_t / a_s
    var = s_t.pow(2) - a_ts.pow(2) * s_s.pow(2)
    return x_s * a_ts + var.sqrt() * randn(tuple(x_s.shape), rng)


@torch.no_grad()
def repaint(denoise, known, mask, num_steps, num_resample_steps, jump_length, rng, *, shape,
            return_all=False, schedule=log_snr_cosine):
    """continuous_time.py:262-330 (RePaint): the draw order is x_T, then per inner step the
    known-region noise (q_step_from_x_0) before the reverse step's noise, then the re-noising."""
    B = known.shape[0]
    x_t = randn(shape, rng)
    steps = torch.linspace(1, 0, num_steps + 1)[None].repeat_interleave(B, 0)
    out = [x_t]
    x_s = x_t
    for i in range(num_steps):
        for j in range(num_resample_steps):
            interp = torch.linspace(0, 1, jump_length + 1)
            r = steps[:, [i]] + interp[None] * (steps[:, [i + 1]] - steps[:, [i]])
            x = x_t
            for k in range(jump_length):
                known_s, _ = q_step_from_x_0(known, r[:, k + 1], rng, schedule)
                noise = randn(tuple(x.shape), rng)
                unknown_s = p_step(denoise, x, r[:, k], r[:, k + 1], noise, mode="ddpm",
                                   schedule=schedule)
                x = mask * known_s + (1 - mask) * unknown_s
            x_s = x
            out.append(x_s)
            if i == num_steps - 1 or j == num_resample_steps - 1:
                x_t = x
                break
            for k in range(jump_length, 0, -1):
                x = q_step(x, r[:, k - 1], r[:, k], rng, schedule)
            x_t = x
    return torch.stack(out) if return_all else x_s


def loss_weight(steps, objective, min_snr=True, gamma=5.0, schedule=log_snr_cosine):
    """continuous_time.py:155-169."""
    snr = schedule(steps).exp()
    clipped = snr.clamp(max=gamma) if min_snr else snr.clone()
    return {"eps": clipped / snr, "x_0": clipped, "v": clipped / (snr + 1)}[objective]


@torch.no_grad()
def p_loss(denoise, x_0, steps, noise_rng, objective="eps", loss_type="l2", min_snr=True,
           schedule=log_snr_cosine):
    """base.py:124-143 + continuous_time.py:140-153 (targets): masked mean of the per-pixel
    criterion (mask = ones), weighted per sample."""
    x_t, noise = q_step_from_x_0(x_0, steps, noise_rng, schedule)
    lam = schedule(steps)
    pred = denoise(x_t, lam)
    a, s = alpha_sigma(lam[:, None, None, None])
    target = {"eps": noise, "x_0": x_0, "v": a * noise - s * x_0}[objective]
    crit = {"l2": torch.nn.MSELoss(reduction="none"), "l1": torch.nn.L1Loss(reduction="none"),
            "huber": torch.nn.SmoothL1Loss(reduction="none")}[loss_type]
    loss = crit(pred, target).flatten(1).sum(1, keepdim=True)
    loss = loss / (torch.ones_like(x_0).flatten(1).sum(1, keepdim=True) + 1e-8)
    # the reference multiplies loss [B,1] by a weight shaped [B,1,1,1]: the product broadcasts to
    # [B,1,B,1], i.e. mean(loss) * mean(weight) -- kept as is (base.py:141-143)
    w = loss_weight(steps, objective, min_snr, schedule=schedule)[:, None, None, None]
    return (loss * w).mean()


# ----------------------------------------------------------------------------- discrete time
def beta_table(kind: str, T: int) -> torch.Tensor:
    """discrete_time.py:12-49 (fp64)."""
    if kind == "linear":
        sc = 1000 / T
        return torch.linspace(sc * 1e-4, sc * 0.02, T, dtype=torch.float64)
    t = torch.linspace(0, T, T + 1, dtype=torch.float64) / T
    if kind == "cosine":
        ab = torch.cos((t + 0.008) / 1.008 * math.pi * 0.5) ** 2
    elif kind == "sigmoid":
        start, end, tau = -3, 3, 1
        vs, ve = torch.tensor(start / tau).sigmoid(), torch.tensor(end / tau).sigmoid()
        ab = (-((t * (end - start) + start) / tau).sigmoid() + ve) / (ve - vs)
    else:
        raise ValueError(kind)
    ab = ab / ab[0]
    return torch.clip(1 - ab[1:] / ab[:-1], 0, 0.999)


def discrete_tables(kind: str, T: int):
    beta = beta_table(kind, T)[:, None, None, None]
    ab = torch.cumprod(1 - beta, dim=0)
    abp = F.pad(ab[:-1], (0,) * 6 + (1, 0), value=1.0)
    return beta.float(), ab.float(), abp.float(), (ab / (1 - ab)).float()


@torch.no_grad()
def p_step_discrete(denoise, x_t, steps, noise, tables, *, mode="ddim", eta=0.0,
                    objective="eps", clip=True, clip_range=1.0):
    """discrete_time.py:126-180.  `noise` only used for ddpm / eta>0."""
    beta_t, ab_t, abp_t, _ = tables
    beta, ab, abp = beta_t[steps], ab_t[steps], abp_t[steps]
    alpha = 1 - beta
    pred = denoise(x_t, steps)
    if objective == "eps":
        x0 = ab.rsqrt() * x_t - (ab.reciprocal() - 1).sqrt() * pred
    elif objective == "x_0":
        x0 = pred
    elif objective == "v":
        x0 = ab.sqrt() * x_t - (1 - ab).sqrt() * pred
    else:
        raise ValueError(objective)
    if clip:
        x0 = x0.clamp(-clip_range, clip_range)
    if mode == "ddpm":
        mean = abp.sqrt() * beta / (1 - ab) * x0 + (1 - abp) * alpha.sqrt() / (1 - ab) * x_t
        var = (beta * (1 - abp) / (1 - ab)).clamp(min=1e-20)
        nz = noise.clone()
        nz[steps == 0] *= 0
        return mean + (0.5 * var.log()).exp() * nz
    var = (1 - abp) / (1 - ab) * (1 - ab / abp)
    sd = eta * var.sqrt()
    eps = (x_t - ab.sqrt() * x0) / (1 - ab).sqrt()
    x_s = abp.sqrt() * x0 + (1 - abp - sd ** 2).sqrt() * eps
    if eta > 0:
        nz = noise.clone()
        nz[steps == 0] *= 0
        x_s = x_s + sd * nz
    return x_s
