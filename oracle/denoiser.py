"""Functional torch-CPU restatement of the reference range-image denoisers.

Everything is driven by a flat state_dict `sd` (reference key names) plus a key prefix, so the
same code checks reference checkpoints, the reference modules (fixtures) and this repo's
modules.  TEST INFRASTRUCTURE -- see oracle/__init__.py.

Reference files restated here (all under /root/reference/lidargen/models/unets/):
  ops.py:14-29 sinusoid, :32-49 Pad, :52-146 Resample, :149-173 Conv2d, :176-200 AdaGN
  encoding.py:120-146 FourierFeatures
  efficient_unet.py:28-58 SelfAttentionBlock, :61-115 ResidualBlock, :118-190 Block,
                    :274-300 EfficientUNet.forward
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------- primitives
def pad_ring(x: torch.Tensor, p: int = 1) -> torch.Tensor:
    """ops.py:32-49 with ring=True: W circular, then H zeros."""
    if p == 0:
        return x
    x = torch.cat([x[..., -p:], x, x[..., :p]], dim=-1)
    return F.pad(x, (0, 0, p, p))


def conv_ring(x, w, b=None):
    """ops.py:149-173.  3x3 -> ring pad 1; 1x1 -> no pad."""
    k = w.shape[-1]
    return F.conv2d(pad_ring(x, (k - 1) // 2), w, b)


def conv_sd(sd, pre, x):
    return conv_ring(x, sd[pre + ".weight"], sd.get(pre + ".bias"))


def _shift_w(x, d):  # value at column j+d, circular
    return torch.roll(x, shifts=-d, dims=-1)


def _shift_h(x, d):  # value at row i+d, zero outside
    H = x.shape[-2]
    out = torch.zeros_like(x)
    if d > 0:
        out[..., : H - d, :] = x[..., d:, :]
    elif d < 0:
        out[..., -d:, :] = x[..., : H + d, :]
    else:
        out = x.clone()
    return out


def resample_down2(x: torch.Tensor) -> torch.Tensor:
    """ops.py:52-146 with down=2, window [1,3,3,1], ring=True, in closed form:
    y[i,j] = sum_{a,b in 0..3} k[a]k[b] xpad[2i+a-1, 2j+b-1],  k=[1,3,3,1]/8,
    W circular, H zero-padded.  Horizontal pass first, like the reference."""
    k = (0.125, 0.375, 0.375, 0.125)
    hz = sum(k[b] * _shift_w(x, b - 1) for b in range(4))[..., 0::2]
    vt = sum(k[a] * _shift_h(hz, a - 1) for a in range(4))[..., 0::2, :]
    return vt


def resample_up2(x: torch.Tensor) -> torch.Tensor:
    """ops.py:52-146 with up=2 (zero insertion, kernel*2 per axis):
    y[2i] = 1/4 x[i-1] + 3/4 x[i],  y[2i+1] = 3/4 x[i] + 1/4 x[i+1] per axis."""
    B, C, H, W = x.shape
    ev = 0.25 * _shift_w(x, -1) + 0.75 * x
    od = 0.75 * x + 0.25 * _shift_w(x, 1)
    hz = torch.stack([ev, od], dim=-1).reshape(B, C, H, 2 * W)
    ev = 0.25 * _shift_h(hz, -1) + 0.75 * hz
    od = 0.75 * hz + 0.25 * _shift_h(hz, 1)
    return torch.stack([ev, od], dim=-2).reshape(B, C, 2 * H, 2 * W)


def group_norm(x, G, w=None, b=None, eps=1e-5):
    return F.group_norm(x, G, w, b, eps)


def silu(x):
    return x * torch.sigmoid(x)


def sinusoid(log_snr: torch.Tensor, channels: int, max_period: float = 10_000.0):
    """ops.py:14-29 -- the input is the log-SNR, not a timestep."""
    half = channels // 2
    f = torch.exp((-math.log(max_period) / (half - 1)) * torch.arange(half, dtype=torch.float32))
    a = log_snr[:, None].float() * f[None]
    return torch.cat([a.sin(), a.cos()], dim=-1)


def fourier_features(coords: torch.Tensor, H: int, W: int) -> torch.Tensor:
    """encoding.py:120-146: 2^k frequencies on elevation (ceil(log2 H) of them) then azimuth
    (ceil(log2 W)); output cat[sin, cos] -> [1, 2(Lh+Lw), H, W]."""
    Lh, Lw = math.ceil(math.log2(H)), math.ceil(math.log2(W))
    el, az = coords[:, 0:1], coords[:, 1:2]
    ang = torch.cat(
        [el * (2.0 ** k) for k in range(Lh)] + [az * (2.0 ** k) for k in range(Lw)], dim=1
    )
    return torch.cat([ang.sin(), ang.cos()], dim=1)


def linear_ray_angles(H, W, fov_up, fov_down):
    """utils/lidar.py:22-32."""
    el = (1 - torch.arange(H) / H) * (fov_up - fov_down) + fov_down
    az = (1 - torch.arange(W) / W) * 360.0 - 180.0
    el, az = torch.meshgrid(el, az, indexing="ij")
    return torch.stack([el, az])[None].deg2rad()


def time_mlp(sd, pre, log_snr, channels):
    h = sinusoid(log_snr, channels)
    h = F.linear(h, sd[pre + ".1.weight"], sd[pre + ".1.bias"])
    return F.linear(silu(h), sd[pre + ".3.weight"], sd[pre + ".3.bias"])


# ----------------------------------------------------------------------------- EfficientUNet
def mha_block(sd, pre, x, heads, G, eps):
    """efficient_unet.py:28-58: (x + out_proj(MHA(GN(x)))) * scale, tokens = H*W row-major."""
    B, C, H, W = x.shape
    h = group_norm(x, G, sd[pre + ".norm.weight"], sd[pre + ".norm.bias"], eps)
    t = h.flatten(2).transpose(1, 2)  # B L C
    qkv = F.linear(t, sd[pre + ".attn.in_proj_weight"], sd[pre + ".attn.in_proj_bias"])
    q, k, v = qkv.chunk(3, dim=-1)
    d = C // heads
    sp = lambda z: z.reshape(B, -1, heads, d).transpose(1, 2)  # B h L d
    q, k, v = sp(q), sp(k), sp(v)
    a = torch.softmax((q @ k.transpose(-1, -2)) / math.sqrt(d), dim=-1) @ v
    a = a.transpose(1, 2).reshape(B, H * W, C)
    a = F.linear(a, sd[pre + ".attn.out_proj.weight"], sd[pre + ".attn.out_proj.bias"])
    a = a.transpose(1, 2).reshape(B, C, H, W)
    return (x + a) * sd[pre + ".scale"]


def residual_block(sd, pre, x, temb, G, eps):
    """efficient_unet.py:61-115."""
    h = silu(group_norm(x, G, sd[pre + ".norm1.weight"], sd[pre + ".norm1.bias"], eps))
    h = conv_sd(sd, pre + ".conv1", h)
    ss = F.linear(silu(temb), sd[pre + ".norm2.proj.1.weight"], sd[pre + ".norm2.proj.1.bias"])
    scale, shift = ss[:, :, None, None].chunk(2, dim=1)
    h = group_norm(h, G, None, None, eps) * (1 + scale) + shift  # ops.py:176-200
    h = conv_sd(sd, pre + ".conv2", silu(h))
    skip = conv_sd(sd, pre + ".skip", x) if (pre + ".skip.weight") in sd else x
    return (skip + h) * sd[pre + ".scale"]


def unet_block(sd, pre, h, temb, G, eps, heads):
    """efficient_unet.py:118-190; structure discovered from the keys present."""
    if (pre + ".downsample.0.weight") in sd:
        h = resample_down2(conv_sd(sd, pre + ".downsample.0", h))
    i = 0
    while f"{pre}.residual_blocks.{i}.scale" in sd:
        h = residual_block(sd, f"{pre}.residual_blocks.{i}", h, temb, G, eps)
        i += 1
    if (pre + ".self_attn_block.scale") in sd:
        h = mha_block(sd, pre + ".self_attn_block", h, heads, G, eps)
    if (pre + ".upsample.1.weight") in sd:
        h = conv_sd(sd, pre + ".upsample.1", resample_up2(h))
    return h


@torch.no_grad()
def efficient_unet_forward(sd, x, log_snr, *, gn_num_groups=8, gn_eps=1e-6, attn_num_heads=8,
                           prefix=""):
    """efficient_unet.py:274-300 (coords_encoding='fourier_features')."""
    p = prefix
    B, _, H, W = x.shape
    base = sd[p + "time_embedding.1.weight"].shape[1]
    if log_snr.ndim == 0:
        log_snr = log_snr[None].repeat(B)
    temb = time_mlp(sd, p + "time_embedding", log_snr.float(), base)
    cenc = fourier_features(sd[p + "coords"], H, W).expand(B, -1, -1, -1)
    G, eps, nh = gn_num_groups, gn_eps, attn_num_heads
    h = conv_sd(sd, p + "in_conv", torch.cat([x, cenc], dim=1))
    h1 = unet_block(sd, p + "d_block1", h, temb, G, eps, nh)
    h2 = unet_block(sd, p + "d_block2", h1, temb, G, eps, nh)
    h3 = unet_block(sd, p + "d_block3", h2, temb, G, eps, nh)
    h4 = unet_block(sd, p + "d_block4", h3, temb, G, eps, nh)
    h = unet_block(sd, p + "u_block4", h4, temb, G, eps, nh)
    h = unet_block(sd, p + "u_block3", torch.cat([h, h3], 1), temb, G, eps, nh)
    h = unet_block(sd, p + "u_block2", torch.cat([h, h2], 1), temb, G, eps, nh)
    h = unet_block(sd, p + "u_block1", torch.cat([h, h1], 1), temb, G, eps, nh)
    return conv_sd(sd, p + "out_conv", h)
