"""Building blocks of the range-image denoisers, HIP-backed.

Mirror of the reference's `lidargen/models/unets/ops.py` (class names, constructor arguments,
parameter/buffer names -> identical state_dict keys), but `forward` launches the gfx950 kernels
of lidarcrafter_amd through the C ABI instead of ATen:
  Conv2d   (ref ops.py:149-173 + Pad :32-49)  -> lc_conv2d_ring_fwd (halo built in LDS, no pad copy)
  Resample (ref ops.py:52-146)                -> lc_resample2x_fwd  (closed-form FIR, one pass)
  AdaGN    (ref ops.py:176-200)               -> lc_groupnorm_stats/apply (+scale/shift, +SiLU)
  SinusoidalPositionalEmbedding (ref :14-29)  -> lc_sinusoid_fwd
Their `forward` is the inference path (no autograd graph); training takes the composition of the same
layers in lidarcrafter_amd/autograd.py, which the denoisers' `forward` selects when
grad mode is on.
"""
from __future__ import annotations

import numpy as np
import torch
from torch import nn

from lidarcrafter_amd import ops as K


def zero_out(m: nn.Module) -> None:
    for p in m.parameters():
        p.data.zero_()


class SinusoidalPositionalEmbedding(nn.Module):
    def __init__(self, channels: int, max_period: int = 10_000):
        super().__init__()
        self.channels, self.max_period = channels, max_period

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        assert x.dim() == 1
        return K.sinusoid(x.float(), self.channels, float(self.max_period))

    def extra_repr(self):
        return f"dim={self.channels} max_period={self.max_period}"


class Pad(nn.Module):
    """Kept for API parity only: padding is fused into the conv / resample kernels."""

    def __init__(self, padding, ring=False, mode="constant"):
        super().__init__()
        self.padding, self.ring, self.mode = padding, ring, mode

    def forward(self, h):
        raise RuntimeError("Pad is fused into lc_conv2d_ring_fwd; it is never run on its own")


class Conv2d(nn.Conv2d):
    """3x3 (ring=True: W circular, H zeros) or 1x1 convolution; weights stay OIHW in the
    state_dict, a packed copy for the MFMA kernel is cached per weight version."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=1, bias=True,
                 ring=False):
        super().__init__(in_channels, out_channels, kernel_size, stride=stride, padding=0, bias=bias)
        ks = self.kernel_size[0]
        if stride != 1 or ks not in (1, 3) or (ks == 3 and (padding != 1 or not ring)) or \
                (ks == 1 and padding != 0):
            raise NotImplementedError(
                "HIP conv supports 3x3/pad1/ring and 1x1/pad0 (all convs on the denoiser path)")
        self.ring = ring
        self._packed = K.PackedConv()

    def forward(self, x, res=None, out=None, out_scale: float = 1.0, gn_coeffs=None,
                gn_silu: bool = True, emit_stats: bool = False):
        """emit_stats: also leave GroupNorm statistics of the output for a following (unfused)
        GroupNorm, so that it needs no statistics pass (ops.conv2d_ring)."""
        return K.conv2d_ring(x, self._packed, self.weight, self.bias, res=res, out=out,
                             out_scale=out_scale, gn_coeffs=gn_coeffs, gn_silu=gn_silu,
                             emit_stats=emit_stats)


class Resample(nn.Module):
    """x2 FIR up/down-sampling with the [1,3,3,1] window (buffer `kernel` kept for checkpoints)."""

    def __init__(self, up=1, down=1, window=(1, 3, 3, 1), ring=True, normalize=True,
                 direction="hw", mode="constant"):
        super().__init__()
        if tuple(window) != (1, 3, 3, 1) or not ring or direction != "hw" or not normalize or \
                (up, down) not in ((2, 1), (1, 2)):
            raise NotImplementedError("HIP Resample: x2 up or down, window [1,3,3,1], ring, 'hw'")
        self.up, self.down = up, down
        kernel = torch.tensor(window, dtype=torch.float32)
        kernel /= kernel.sum()
        kernel *= float(up * up) ** (kernel.ndim / 2)
        self.register_buffer("kernel", kernel)

    def forward(self, h, out=None):
        return K.resample2x(h, up=self.up == 2, out=out)

    def extra_repr(self):
        return f"up={self.up}, down={self.down}, ring=True"


class GroupNorm(nn.GroupNorm):
    """nn.GroupNorm with an optional fused SiLU (GN -> SiLU is one kernel pair here)."""

    def forward(self, x, act_silu: bool = False, out=None, split_for=None):
        """split_for: the `_packed` of the 3x3 conv that consumes the result (pre-split output,
        ops.groupnorm)."""
        return K.groupnorm(x, self.num_groups, self.eps, self.weight, self.bias,
                           act_silu=act_silu, out=out, split_for=split_for)

    def coeffs(self, x):
        """Statistics only: the next conv derives the per-(b, channel) rows in its prologue and
        applies them while staging its input."""
        return K.groupnorm_stats(x, self.num_groups, self.eps, self.weight, self.bias)


class AdaGN(nn.GroupNorm):
    def __init__(self, emb_channels, out_channels, num_groups, eps=1e-5):
        super().__init__(num_groups, out_channels, eps=eps, affine=False)
        # index 1 keeps the reference key `proj.1.weight`
        self.proj = nn.Sequential(nn.SiLU(), nn.Linear(emb_channels, out_channels * 2))

    def scale_shift(self, emb):
        ss = K.linear(emb, self.proj[1].weight, self.proj[1].bias, act_in=True)
        C = self.num_channels
        return ss[:, :C], ss[:, C:]

    def forward(self, x, emb=None, scale_shift=None, act_silu: bool = False, out=None,
                split_for=None):
        scale, shift = scale_shift if scale_shift is not None else self.scale_shift(emb)
        return K.groupnorm(x, self.num_groups, self.eps, None, None, scale, shift,
                           act_silu=act_silu, out=out, split_for=split_for)

    def coeffs(self, x, emb=None, scale_shift=None):
        scale, shift = scale_shift if scale_shift is not None else self.scale_shift(emb)
        return K.groupnorm_stats(x, self.num_groups, self.eps, None, None, scale, shift)


class ConditionalSequential(nn.Sequential):
    def forward(self, x, condition):
        for module in self:
            x = module(x, condition)
        return x


SQRT1_2 = float(1 / np.sqrt(2))
