"""Coordinate encodings (mirror of the reference's lidargen/models/unets/encoding.py:80-146).

These are step-invariant: the reference recomputes FourierFeatures(coords) on every forward
(efficient_unet.py:283-286); here the encoded map is computed ONCE per coords tensor with the
reference's own formula on the host and cached on the device, and the denoiser writes it once
into its persistent input buffer.
"""
from __future__ import annotations

import math

import torch
from torch import nn


def generate_polar_coords(H: int, W: int, device="cpu") -> torch.Tensor:
    phi = (0.5 - torch.arange(H, device=device) / H) * torch.pi
    theta = (1 - torch.arange(W, device=device) / W) * 2 * torch.pi - torch.pi
    phi, theta = torch.meshgrid([phi, theta], indexing="ij")
    return torch.stack([phi, theta])[None]


class FourierFeatures(nn.Module):
    def __init__(self, resolution):
        super().__init__()
        self.resolution = resolution
        self.L_h = int(math.ceil(math.log2(resolution[0])))
        self.L_w = int(math.ceil(math.log2(resolution[1])))
        n = self.L_h + self.L_w
        freqs = torch.zeros(n, 2)
        freqs[: self.L_h, 0] = torch.arange(self.L_h).exp2()
        freqs[self.L_h:, 1] = torch.arange(self.L_w).exp2()
        self.register_buffer("freqs", freqs[..., None, None])
        self.register_buffer("phase", torch.zeros(n))
        self.extra_ch = 2 * n
        self._cache = None

    def forward(self, coords: torch.Tensor) -> torch.Tensor:
        """-> [1, extra_ch, H, W] on coords.device (cached per coords version)."""
        key = (coords.data_ptr(), coords._version, str(coords.device))
        if self._cache is None or self._cache[0] != key:
            c = coords.detach().float().cpu()
            f = self.freqs.detach().float().cpu()[:, :, 0, 0]           # [n, 2]
            ang = torch.einsum("nk,bkhw->bnhw", f, c) + self.phase.detach().cpu()[None, :, None, None]
            enc = torch.cat([ang.sin(), ang.cos()], dim=1).contiguous()
            self._cache = (key, enc.to(coords.device), coords)   # holds coords: see key
        return self._cache[1]

    def extra_repr(self):
        return f"shape={self.resolution}, num_freqs={self.extra_ch}, L=({self.L_h}, {self.L_w})"
