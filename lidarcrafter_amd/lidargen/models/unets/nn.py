"""Small helpers of the guided-diffusion style UNet -- API mirror of the reference's
lidargen/models/unets/nn.py (GroupNorm32 :17-19, conv_nd :22-31, conv_nd_range :34-44,
linear :46-50, zero_module :79-85, normalization :104-111), HIP-backed where they compute."""
from __future__ import annotations

import torch
import torch.nn as nn

from lidarcrafter_amd import ops as K

from . import ops


class SiLU(nn.Module):
    """Marker only: SiLU is fused into the preceding GroupNorm kernel on this path."""

    def forward(self, x):
        raise RuntimeError("SiLU is fused into lc_groupnorm_apply; never run on its own")


class GroupNorm32(nn.GroupNorm):
    """fp32 GroupNorm on [B,C,H,W] or [B,C,L] (+ optional scale/shift, + optional SiLU)."""

    def forward(self, x, scale=None, shift=None, act_silu: bool = False, out=None, split_for=None):
        """split_for: the `_packed` of the 3x3 conv consuming the result (pre-split output where the
        shape allows: whole channel octets per group, i.e. C >= 256 with 32 groups)."""
        if x.dim() == 4 and split_for is not None and out is None:
            return K.groupnorm(x, self.num_groups, self.eps, self.weight, self.bias, scale, shift,
                               act_silu=act_silu, split_for=split_for)
        if x.dim() == 3 and split_for is not None and out is None:      # tokens for a pre-split 1x1 projection
            return K.groupnorm(_tok4(x), self.num_groups, self.eps, self.weight, self.bias, scale, shift,
                               act_silu=act_silu, split_for=split_for)
        if x.dim() == 3:
            B, C, L = x.shape
            o4 = None if out is None else out.view(B, C, 1, L)
            y = K.groupnorm(_tok4(x), self.num_groups, self.eps, self.weight,
                            self.bias, scale, shift, act_silu=act_silu, out=o4)
            return y.view(B, C, L)
        return K.groupnorm(x, self.num_groups, self.eps, self.weight, self.bias, scale, shift,
                           act_silu=act_silu, out=out)


def _tok4(x3):
    """[B,C,L] (batch stride arbitrary, rows dense) -> [B,C,1,L] view that keeps the producer statistics."""
    B, C, L = x3.shape
    if x3.stride(2) == 1 and x3.stride(1) == L:
        return K.alias(torch.as_strided(x3, (B, C, 1, L), (x3.stride(0), L, L, 1)), x3)
    return x3.reshape(B, C, 1, L)


def gn32_coeffs(norm: "GroupNorm32", x, scale=None, shift=None):
    """Statistics pass of a GroupNorm32 on [B,C,H,W] / [B,C,L] -> rows for the fused conv input."""
    if x.dim() == 3:
        x = _tok4(x)
    return K.groupnorm_stats(x, norm.num_groups, norm.eps, norm.weight, norm.bias, scale, shift)


class PointwiseConv1d(nn.Conv1d):
    """nn.Conv1d(kernel_size=1) parameters ([Co, Ci, 1]) driven through the 1x1 MFMA conv."""

    def __init__(self, in_channels, out_channels, kernel_size=1):
        assert kernel_size == 1
        super().__init__(in_channels, out_channels, 1)
        self._packed = K.PackedConv()

    def forward(self, x, res=None, out=None, gn_coeffs=None, gn_silu=False, emit_stats=False):
        r4 = None if res is None else _tok4(res)
        o4 = None if out is None else _tok4(out)
        if isinstance(x, K.SplitAct):                  # [B, C, 1, L] tokens written pre-split for THIS layer
            B, _, _, L = x.shape
            y = K.conv2d_ring(x, self._packed, self.weight, self.bias, res=r4, out=o4)
            return K.alias(y.view(B, -1, L), y)
        B, C, L = x.shape
        y = K.conv2d_ring(_tok4(x), self._packed, self.weight, self.bias, res=r4,
                          out=o4, gn_coeffs=gn_coeffs, gn_silu=gn_silu, emit_stats=emit_stats)
        return K.alias(y.view(B, -1, L), y)


def conv_nd(dims, *args, **kwargs):
    if dims == 1:
        return PointwiseConv1d(*args, **kwargs)
    if dims == 2:
        kwargs.setdefault("padding", 0)
        return ops.Conv2d(*args, **kwargs)
    raise ValueError(f"unsupported dimensions: {dims}")


def conv_nd_range(dims, *args, **kwargs):
    if dims == 2:
        return ops.Conv2d(*args, **kwargs)
    return conv_nd(dims, *args, **kwargs)


def linear(*args, **kwargs):
    return nn.Linear(*args, **kwargs)


def zero_module(module):
    for p in module.parameters():
        p.detach().zero_()
    return module


def normalization(channels):
    return GroupNorm32(32, channels)
