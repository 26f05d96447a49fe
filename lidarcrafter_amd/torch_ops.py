"""`torch.library` registration of the HIP hot-path ops (namespace `lidarcrafter`), SURVEY.md §8b.

The reference's bulk harness wraps the sampler in `torch.compile` and runs it under fp16 autocast
(tools/evaluation/sample_and_save_cond.py:64,145).  Two things keep such callers working:

  * the samplers and denoiser forwards of this build are `torch.compiler.disable`d: they already
    replay ONE captured HIP graph per step, so Dynamo has nothing to gain from tracing them, and a
    `torch.compile(ddpm.sample)` wrapper simply calls them;
  * the kernels themselves are exposed here as functional custom ops with FAKE (meta)
    implementations, so user code that composes them inside its own compiled region traces cleanly:
    Dynamo / AOT see opaque ops with known output shapes instead of ctypes calls and `data_ptr()`.

Functional forms (no `out=` aliasing, packed weights cached per weight tensor); all compute in
fp32 -- half-precision inputs (autocast) are up-cast at the op boundary (ops._entry):

    torch.ops.lidarcrafter.conv2d_ring(x, weight, bias?, res?, out_scale)      [B,Co,H,W]
    torch.ops.lidarcrafter.groupnorm(x, G, eps, gamma?, beta?, scale?, shift?, act_silu)
    torch.ops.lidarcrafter.resample2x(x, up)
    torch.ops.lidarcrafter.attention(q, k, v, heads, scale)                     channel-major [B,C,L]
    torch.ops.lidarcrafter.pstep(x_t, pred, noise?, coef, objective, mode)
    torch.ops.lidarcrafter.project_points(points, H, W, fov_up, fov_down, min_depth, max_depth)
    torch.ops.lidarcrafter.points_in_boxes(points, boxes, margin)
"""
from __future__ import annotations

import weakref
from typing import Optional, Tuple

import torch
from torch import Tensor

from . import ops as K

_packed = weakref.WeakKeyDictionary()   # weight tensor object -> PackedConv


def _packed_for(weight: Tensor) -> K.PackedConv:
    pk = _packed.get(weight)
    if pk is None:
        pk = _packed[weight] = K.PackedConv("torch.ops.lidarcrafter.conv2d_ring")
    return pk


@torch.library.custom_op("lidarcrafter::conv2d_ring", mutates_args=(), device_types="cuda")
def conv2d_ring(x: Tensor, weight: Tensor, bias: Optional[Tensor], res: Optional[Tensor],
                out_scale: float) -> Tensor:
    """(conv(x, W) + bias [+ res]) * out_scale; 3x3: W circular / H zero padding, 1x1: plain
    (reference ops.Conv2d + ops.Pad, lidargen/models/unets/ops.py:32-49,149-173)."""
    return K.conv2d_ring(x, _packed_for(weight), weight, bias, res=res, out_scale=out_scale)


@conv2d_ring.register_fake
def _(x, weight, bias, res, out_scale):
    return x.new_empty((x.shape[0], weight.shape[0], x.shape[2], x.shape[3]), dtype=torch.float32)


@torch.library.custom_op("lidarcrafter::groupnorm", mutates_args=(), device_types="cuda")
def groupnorm(x: Tensor, G: int, eps: float, gamma: Optional[Tensor], beta: Optional[Tensor],
              scale: Optional[Tensor], shift: Optional[Tensor], act_silu: bool) -> Tensor:
    """GroupNorm (+affine) (+AdaGN (1+scale) h + shift) (+SiLU), reference ops.py:176-200."""
    return K.groupnorm(x, G, eps, gamma, beta, scale, shift, act_silu=act_silu)


@groupnorm.register_fake
def _(x, G, eps, gamma, beta, scale, shift, act_silu):
    return x.new_empty(x.shape, dtype=torch.float32)


@torch.library.custom_op("lidarcrafter::resample2x", mutates_args=(), device_types="cuda")
def resample2x(x: Tensor, up: bool) -> Tensor:
    """x2 FIR up / down-sampling, window [1,3,3,1], ring (reference ops.Resample, ops.py:52-146)."""
    return K.resample2x(x, up=up)


@resample2x.register_fake
def _(x, up):
    B, C, H, W = x.shape
    return x.new_empty((B, C, 2 * H, 2 * W) if up else (B, C, H // 2, W // 2), dtype=torch.float32)


@torch.library.custom_op("lidarcrafter::attention", mutates_args=(), device_types="cuda")
def attention(q: Tensor, k: Tensor, v: Tensor, heads: int, scale: float) -> Tensor:
    """softmax(scale * q^T k) v per head on channel-major operands [B, heads*d, L]."""
    return K.attention_cm(q, k, v, heads, scale)


@attention.register_fake
def _(q, k, v, heads, scale):
    return q.new_empty((q.shape[0], v.shape[1], q.shape[2]), dtype=torch.float32)


@torch.library.custom_op("lidarcrafter::pstep", mutates_args=(), device_types="cuda")
def pstep(x_t: Tensor, pred: Tensor, noise: Optional[Tensor], coef: Tensor, objective: int,
          mode: int) -> Tensor:
    """Fused x0 estimate + clamp + DDPM/DDIM update (reference continuous_time.py:195-234)."""
    return K.pstep(x_t, pred, noise, coef, objective, mode)


@pstep.register_fake
def _(x_t, pred, noise, coef, objective, mode):
    return x_t.new_empty(x_t.shape, dtype=torch.float32)


@torch.library.custom_op("lidarcrafter::project_points", mutates_args=(), device_types="cuda")
def project_points(points: Tensor, H: int, W: int, fov_up: float, fov_down: float,
                   min_depth: float, max_depth: float) -> Tuple[Tensor, Tensor]:
    """Spherical projection + nearest-point z-buffer: image [H,W,6], winner index int32 [H,W]
    (reference load_points_as_images, dataset/transforms_3d/common.py:26-91)."""
    img, win = K.project_points(points, H, W, fov_up, fov_down, min_depth, max_depth)[:2]
    return img, win


@project_points.register_fake
def _(points, H, W, fov_up, fov_down, min_depth, max_depth):
    return (points.new_empty((H, W, 6), dtype=torch.float32),
            points.new_empty((H, W), dtype=torch.int32))


@torch.library.custom_op("lidarcrafter::points_in_boxes", mutates_args=(), device_types="cuda")
def points_in_boxes(points: Tensor, boxes: Tensor, margin: float) -> Tensor:
    """int32 [N_box, M] inside mask (reference points_in_boxes_cpu, roiaware_pool3d.cpp:143-168)."""
    return K.points_in_boxes_mask(points, boxes, margin)


@points_in_boxes.register_fake
def _(points, boxes, margin):
    return points.new_empty((boxes.shape[0], points.shape[0]), dtype=torch.int32)


OPS = ("conv2d_ring", "groupnorm", "resample2x", "attention", "pstep", "project_points",
       "points_in_boxes")
