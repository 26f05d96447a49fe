"""Deterministic synthetic weights / inputs shared by tests, bench and the fixture generator.

There are no pretrained checkpoints in this environment, and a freshly
constructed denoiser has 82 zero-initialised tensors (reference `ops.zero_out`
lidargen/models/unets/ops.py:9-11, `zero_module` nn.py:79-85) so it would output exactly 0.
`seeded_fill` therefore overwrites EVERY parameter with values that depend only on the
parameter's state_dict key and shape -- not on construction order or torch's global RNG --
so the reference modules (fixture generator) and this repo's modules get bit-identical
weights as long as their state_dict keys/shapes agree (which is itself the checkpoint contract).
"""
from __future__ import annotations

import zlib

import torch


def _gen_for(key: str, salt: int) -> torch.Generator:
    g = torch.Generator(device="cpu")
    g.manual_seed((zlib.crc32(key.encode()) ^ (salt * 2654435761)) & 0x7FFFFFFF)
    return g


@torch.no_grad()
def seeded_fill(module: torch.nn.Module, salt: int = 0) -> torch.nn.Module:
    """Fill all parameters of `module` in place, keyed by their state_dict name."""
    for key, p in module.named_parameters():
        g = _gen_for(key, salt)
        r = torch.randn(p.shape, generator=g, dtype=torch.float32)
        if p.ndim >= 2:
            fan_in = max(1, p.numel() // p.shape[0])
            v = r / fan_in ** 0.5
        elif key.endswith("weight"):  # norm gains
            v = 1.0 + 0.1 * r
        else:  # biases
            v = 0.1 * r
        p.copy_(v.to(device=p.device, dtype=p.dtype))
    return module


def seeded_randn(*shape: int, seed: int) -> torch.Tensor:
    g = torch.Generator(device="cpu").manual_seed(seed)
    return torch.randn(*shape, generator=g, dtype=torch.float32)


def rel_l2(a: torch.Tensor, b: torch.Tensor) -> float:
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def synth_points(N: int, seed: int):
    """Synthetic LiDAR sweep: azimuth U(-pi,pi), elevation U(-30.5,10.5) deg,
    range log-U(0.8,95) m, intensity U(0,255) -> float32 [N,4]."""
    import numpy as np

    g = np.random.default_rng(seed)
    az = g.uniform(-np.pi, np.pi, N)
    el = np.deg2rad(g.uniform(-30.5, 10.5, N))
    r = np.exp(g.uniform(np.log(0.8), np.log(95.0), N))
    return np.stack([r * np.cos(el) * np.cos(az), r * np.cos(el) * np.sin(az), r * np.sin(el),
                     g.uniform(0, 255, N)], axis=1).astype(np.float32)
