"""EfficientUNet (R2DM) range-image denoiser on the gfx950 kernels.

API / state_dict mirror of the reference's `lidargen/models/unets/efficient_unet.py`
(SelfAttentionBlock :28-58, ResidualBlock :61-115, Block :118-190, EfficientUNet :193-300),
restructured around the HIP hot path:

  * GN -> SiLU and AdaGN -> SiLU are one stats + one apply launch each; the ring padding lives in
    the conv kernel; `(skip(x) + h) * scale` is the conv2 epilogue (no add / mul passes).
  * all 24 AdaGN projections of a forward are ONE dense launch on a concatenated weight.
  * torch.cat([h, skip]) never runs: each skip tensor is written by its producer directly into
    the channel slice of a pre-concatenated buffer (batch-strided views).
  * FourierFeatures(coords) is step-invariant: computed once, resident in the input buffer.
"""
from __future__ import annotations

from typing import Iterable, Literal

import numpy as np
import torch
from torch import nn

from lidarcrafter_amd import autograd as AG
from lidarcrafter_amd import ops as K

from . import encoding, ops


def _n_tuple(x, N):
    if isinstance(x, Iterable):
        assert len(x) == N
        return tuple(x)
    return (x,) * N


class _MHAParams(nn.Module):
    """Parameter container with nn.MultiheadAttention's names/shapes (packed in_proj)."""

    def __init__(self, embed_dim: int, num_heads: int):
        super().__init__()
        assert embed_dim % num_heads == 0
        self.embed_dim, self.num_heads = embed_dim, num_heads
        self.in_proj_weight = nn.Parameter(torch.empty(3 * embed_dim, embed_dim))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * embed_dim))
        self.out_proj = nn.Linear(embed_dim, embed_dim)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.zeros_(self.out_proj.bias)


class SelfAttentionBlock(nn.Module):
    def __init__(self, in_channels, num_heads, gn_eps=1e-6, gn_num_groups=8,
                 scale=1 / np.sqrt(2)):
        super().__init__()
        self.norm = ops.GroupNorm(gn_num_groups, in_channels, gn_eps)
        self.attn = _MHAParams(in_channels, num_heads)
        self.attn.out_proj.apply(ops.zero_out)
        self.register_buffer("scale", torch.tensor(scale).float())
        self._scale_f = float(scale)
        self._pk_in, self._pk_out = K.PackedConv(), K.PackedConv()

    def forward(self, x, out=None):
        B, C, H, W = x.shape
        heads = self.attn.num_heads
        # in_proj on channel-major tokens == 1x1 conv on NCHW (GroupNorm fused into its staging)
        w_in = self.attn.in_proj_weight[:, :, None, None]
        if K.fuse_gn(3 * C):
            qkv = K.conv2d_ring(x, self._pk_in, w_in, self.attn.in_proj_bias,
                                gn_coeffs=self.norm.coeffs(x), gn_silu=False)
        elif K.presplit_1x1(C, 3 * C, self.norm.num_groups):
            # the GroupNorm writes its result pre-split for the projection (no fp32 copy, no split per 64-channel output
            # block): lc_conv1x1_f16x2_ps_fwd with its 16-byte store form -- 9.6 + 54 -> ~37 us at 512 -> 1536 @ 8 x 4 x 128
            qkv = K.conv2d_ring(self.norm(x, split_for=self._pk_in), self._pk_in, w_in, self.attn.in_proj_bias)
        else:
            qkv = K.conv2d_ring(self.norm(x), self._pk_in, w_in, self.attn.in_proj_bias)
        t = qkv.view(B, 3 * C, H * W)
        o = K.attention_cm(t[:, :C], t[:, C:2 * C], t[:, 2 * C:], heads,
                           scale=1.0 / float(np.sqrt(C // heads)))
        # out_proj + residual + 1/sqrt(2) in the conv epilogue (+ octet statistics of what it stores: the first GroupNorm of
        # the next block then takes no statistics pass)
        return K.conv2d_ring(o.view(B, C, H, W), self._pk_out,
                             self.attn.out_proj.weight[:, :, None, None], self.attn.out_proj.bias,
                             res=x, out=out, out_scale=self._scale_f, emit_stats=True)


class ResidualBlock(nn.Module):
    def __init__(self, in_channels, out_channels, emb_channels, gn_num_groups=8, gn_eps=1e-6,
                 scale=1 / np.sqrt(2), dropout=0.0, ring=False):
        super().__init__()
        self.has_emb = emb_channels is not None
        self.norm1 = ops.GroupNorm(gn_num_groups, in_channels, gn_eps)
        self.silu1 = nn.SiLU()
        self.conv1 = ops.Conv2d(in_channels, out_channels, 3, 1, 1, ring=ring)
        if self.has_emb:
            self.norm2 = ops.AdaGN(emb_channels, out_channels, gn_num_groups, gn_eps)
        else:
            self.norm2 = ops.GroupNorm(gn_num_groups, out_channels, gn_eps)
        self.silu2 = nn.SiLU()
        self.drop2 = nn.Dropout(dropout)
        self.conv2 = ops.Conv2d(out_channels, out_channels, 3, 1, 1, ring=ring)
        self.conv2.apply(ops.zero_out)
        self.skip = (ops.Conv2d(in_channels, out_channels, 1, 1, 0)
                     if in_channels != out_channels else nn.Identity())
        self.register_buffer("scale", torch.tensor(scale).float())
        self._scale_f = float(scale)

    def forward(self, x, emb=None, scale_shift=None, out=None):
        # every conv whose output feeds a GroupNorm leaves octet statistics of what it stores
        if K.fuse_gn(self.conv1.out_channels):   # GN -> SiLU -> conv: (producer stats +) conv only
            h = self.conv1(x, gn_coeffs=self.norm1.coeffs(x), emit_stats=True)
            c2 = (self.norm2.coeffs(h, emb, scale_shift=scale_shift) if self.has_emb
                  else self.norm2.coeffs(h))
            if isinstance(self.skip, nn.Identity):
                sk = x
            else:
                sk = self.skip(x)
            return self.conv2(h, res=sk, out=out, out_scale=self._scale_f, gn_coeffs=c2,
                              emit_stats=True)
        # unfused GroupNorms (wide layers): the producing convs leave the statistics, the
        # GroupNorm is a single apply pass
        # (where the shape allows, the apply pass writes its result pre-split for the conv that
        #  consumes it -- hi / lo fp16 planes staged by LDS-DMA, ops.groupnorm `split_for`)
        a = self.norm1(x, act_silu=True, split_for=self.conv1._packed)
        h = self.conv1(a, emit_stats=True)
        if self.has_emb:
            a = self.norm2(h, emb, scale_shift=scale_shift, act_silu=True,
                           split_for=self.conv2._packed)
        else:
            a = self.norm2(h, act_silu=True, split_for=self.conv2._packed)
        sk = x if isinstance(self.skip, nn.Identity) else self.skip(x, out=h)
        return self.conv2(a, res=sk, out=out, out_scale=self._scale_f, emit_stats=True)


class Block(nn.Module):
    def __init__(self, in_channels, out_channels, num_residual_blocks, emb_channels,
                 gn_num_groups=8, gn_eps=1e-6, attn=False, attn_num_heads=8, up=1, down=1,
                 dropout=0.0, ring=False):
        super().__init__()
        self.downsample = (nn.Sequential(ops.Conv2d(in_channels, out_channels, 3, 1, 1, ring=ring),
                                         ops.Resample(down=down, ring=ring))
                           if down > 1 else nn.Identity())
        self.residual_blocks = ops.ConditionalSequential()
        for i in range(num_residual_blocks):
            self.residual_blocks.append(ResidualBlock(
                in_channels=out_channels if i != 0 or down > 1 else in_channels,
                out_channels=out_channels, emb_channels=emb_channels,
                gn_num_groups=gn_num_groups, gn_eps=gn_eps, dropout=dropout, ring=ring))
        self.self_attn_block = (SelfAttentionBlock(out_channels, attn_num_heads, gn_eps,
                                                   gn_num_groups) if attn else nn.Identity())
        self.upsample = (nn.Sequential(ops.Resample(up=up, ring=ring),
                                       ops.Conv2d(out_channels, out_channels, 3, 1, 1, ring=ring))
                         if up > 1 else nn.Identity())
        self._packed_up9 = K.PackedConv() if up > 1 else None     # the up-path fold's 1x1 projection (ops.conv_up2)
        self._up9_cache = None

    def forward(self, h, temb=None, scale_shifts=None, out=None):
        """`out`: optional destination view for the block's final tensor (concat-buffer slice).
        `scale_shifts`: per residual block (scale, shift) views, when precomputed by the UNet."""
        has_attn = not isinstance(self.self_attn_block, nn.Identity)
        has_up = not isinstance(self.upsample, nn.Identity)
        if not isinstance(self.downsample, nn.Identity):
            conv, rs = self.downsample
            B_, Ci_, H_, W_ = h.shape
            if rs.down == 2 and conv.ring and K.can_fold_down(Ci_, conv.out_channels, H_, W_):
                # conv at full resolution + FIR + decimation == stride-2 conv of the FIR-pre-filtered input: a quarter of the
                # conv's multiply-adds and output bytes, no resampling pass (ops.conv_down2, csrc/conv_f16x2_s2.hip);
                # it leaves the statistics entries the first residual block's GroupNorm can fold (octets, else quads)
                cpg = conv.out_channels // self.residual_blocks[0].norm1.num_groups
                unit = 8 if cpg % 8 == 0 else (4 if cpg % 4 == 0 else 0)
                h = K.conv_down2(h, conv._packed, conv.weight, conv.bias, emit_stats=unit)
            else:
                h = rs(conv(h))
        n = len(self.residual_blocks)
        for i, rb in enumerate(self.residual_blocks):
            last = (i == n - 1) and not has_attn and not has_up
            ss = scale_shifts[i] if scale_shifts is not None else None
            h = rb(h, temb, scale_shift=ss, out=out if last else None)
        if has_attn:
            h = self.self_attn_block(h, out=out if not has_up else None)
        if has_up:   # feeds the next block's first GroupNorm (through the concat buffer)
            rs, conv = self.upsample
            B_, C_, H_, W_ = h.shape
            if rs.up == 2 and conv.ring and h.is_cuda and h.dtype == torch.float32 and h.data_ptr() % 16 == 0 \
                    and (H_ * W_) % 4 == 0 and K.can_fold_up(C_, conv.out_channels, H_, W_):
                # conv3x3(up(h)) at the LOW resolution: h pre-split as it is, one 1x1 projection to the nine tap planes (a
                # quarter of the multiply-adds), the combine pass (ops.conv_up2, csrc/upfold.hip); per-channel statistics
                xs = K.split_act(h, self._packed_up9)
                h = K.conv_up2(xs, self._packed_up9, self._up9_weight(), conv.bias, out=out, emit_stats=True)
            else:
                h = conv(rs(h), out=out, emit_stats=True)
        return h

    def _up9_weight(self):
        """ops.up9_weight(upsample conv weight), rebuilt when the parameter changes (address / version)."""
        w = self.upsample[1].weight
        try:
            ver = w._version
        except RuntimeError:      # inference-mode tensors do not track one
            ver = -1
        key = (w.data_ptr(), ver, w.device)
        if self._up9_cache is None or self._up9_cache[0] != key:
            self._up9_cache = (key, K.up9_weight(w))
        return self._up9_cache[1]


class EfficientUNet(nn.Module):
    def __init__(self, in_channels: int, resolution, out_channels: int | None = None,
                 base_channels: int = 128, temb_channels: int | None = None,
                 channel_multiplier=(1, 2, 4, 8), num_residual_blocks=(3, 3, 3, 3),
                 gn_num_groups: int = 32 // 4, gn_eps: float = 1e-6, attn_num_heads: int = 8,
                 coords_encoding: Literal["spherical_harmonics", "polar_coordinates",
                                          "fourier_features", None] = "spherical_harmonics",
                 ring: bool = True):
        super().__init__()
        self.resolution = _n_tuple(resolution, 2)
        self.in_channels = in_channels
        self.out_channels = in_channels if out_channels is None else out_channels
        temb_channels = base_channels * 4 if temb_channels is None else temb_channels
        self.register_buffer("coords", encoding.generate_polar_coords(*self.resolution))
        self.coords_encoding = None
        if coords_encoding == "fourier_features":
            self.coords_encoding = encoding.FourierFeatures(self.resolution)
            in_channels += self.coords_encoding.extra_ch
        elif coords_encoding == "polar_coordinates":
            self.coords_encoding = nn.Identity()
            in_channels += self.coords.shape[1]
        elif coords_encoding is not None:
            raise NotImplementedError(
                f"coords_encoding={coords_encoding!r}: every shipped nuScenes config uses "
                "'fourier_features' (option_unet_nusc.py:18); others are out of scope")
        if not ring:
            raise NotImplementedError("ring=False is not on the path (all shipped configs: ring=True)")
        self.time_embedding = nn.Sequential(
            ops.SinusoidalPositionalEmbedding(base_channels),
            nn.Linear(base_channels, temb_channels), nn.SiLU(),
            nn.Linear(temb_channels, temb_channels))
        L = 4
        mult = _n_tuple(channel_multiplier, L)
        C = [base_channels] + [base_channels * m for m in mult]
        N = _n_tuple(num_residual_blocks, L)
        cfgs = dict(emb_channels=temb_channels, gn_num_groups=gn_num_groups, gn_eps=gn_eps,
                    attn_num_heads=attn_num_heads, dropout=0.0, ring=ring)
        self.in_conv = ops.Conv2d(in_channels, C[0], 3, 1, 1, ring=ring)
        self.d_block1 = Block(C[0], C[1], N[0], **cfgs)
        self.d_block2 = Block(C[1], C[2], N[1], down=2, **cfgs)
        self.d_block3 = Block(C[2], C[3], N[2], down=2, **cfgs)
        self.d_block4 = Block(C[3], C[4], N[3], down=2, attn=True, **cfgs)
        self.u_block4 = Block(C[4], C[3], N[3], up=2, attn=True, **cfgs)
        self.u_block3 = Block(C[3] + C[3], C[2], N[2], up=2, **cfgs)
        self.u_block2 = Block(C[2] + C[2], C[1], N[1], up=2, **cfgs)
        self.u_block1 = Block(C[1] + C[1], C[0], N[0], **cfgs)
        self.out_conv = ops.Conv2d(C[0], self.out_channels, 3, 1, 1, ring=ring)
        self.out_conv.apply(ops.zero_out)
        self._C = C
        self._ada_cache = None
        self._in_buf = None
        K.name_packed_convs(self)

    # ---- step-invariant / batched helpers ------------------------------------------------------
    _BLOCKS = ("d_block1", "d_block2", "d_block3", "d_block4",
               "u_block4", "u_block3", "u_block2", "u_block1")

    def _ada_modules(self):
        return [rb.norm2 for name in self._BLOCKS for rb in getattr(self, name).residual_blocks]

    def _ada_weights(self):
        mods = self._ada_modules()
        key = tuple((m.proj[1].weight.data_ptr(), m.proj[1].weight._version,
                     m.proj[1].bias._version) for m in mods)
        if self._ada_cache is None or self._ada_cache[0] != key:
            w = torch.cat([m.proj[1].weight.detach() for m in mods], 0).contiguous()
            b = torch.cat([m.proj[1].bias.detach() for m in mods], 0).contiguous()
            self._ada_cache = (key, w, b)
        return self._ada_cache[1], self._ada_cache[2]

    def time_features(self, log_snr: torch.Tensor):
        """log-SNR [M] -> (temb [M, T], all AdaGN (scale|shift) rows [M, sum 2C])."""
        te = self.time_embedding
        h = te[0](log_snr)
        h = K.linear(h, te[1].weight, te[1].bias, act_out=True)
        temb = K.linear(h, te[3].weight, te[3].bias)
        w, b = self._ada_weights()
        return temb, K.linear(temb, w, b, act_in=True)

    def _split_ss(self, ss):
        out, off = {}, 0
        for name in self._BLOCKS:
            lst = []
            for rb in getattr(self, name).residual_blocks:
                C = rb.norm2.num_channels
                lst.append((ss[:, off:off + C], ss[:, off + C:off + 2 * C]))
                off += 2 * C
            out[name] = lst
        return out

    def _input_buffer(self, B, x):
        """Persistent [B, in+enc, H, W] buffer whose encoding channels are filled once."""
        H, W = self.resolution
        cin = self.in_channels
        enc = None
        if isinstance(self.coords_encoding, encoding.FourierFeatures):
            enc = self.coords_encoding(self.coords)
        elif self.coords_encoding is not None:
            enc = self.coords.float()
        ce = 0 if enc is None else enc.shape[1]
        key = (B, x.device, None if enc is None else enc.data_ptr())
        if self._in_buf is None or self._in_buf[0] != key:
            buf = torch.empty((B, cin + ce, H, W), device=x.device, dtype=torch.float32)
            if enc is not None:
                K.copy_into(buf[:, cin:], enc.expand(B, -1, -1, -1) if B > 1 else enc)
            self._in_buf = (key, buf)
        return self._in_buf[1]

    @torch.compiler.disable
    @K.range_checked
    def forward(self, images: torch.Tensor, timesteps: torch.Tensor, time_features=None):
        """images [B, C, H, W], timesteps = log-SNR [B] (or 0-d) -> prediction [B, C_out, H, W].
        `time_features`: optional precomputed `self.time_features(log_snr)` (sampler hoists it).
        Range-checked: a standalone call polls the conv range records afterwards and recomputes if
        a layer's fp16 operands saturated (ops.range_checked); samplers defer that to the run's end."""
        if time_features is None and AG.training_active(self, images):
            # training (tools/train/train_lidm.py): the differentiable composition of the same layers
            return AG.efficient_unet_forward(self, images, timesteps.to(images))
        B, _, H, W = images.shape
        if time_features is None:
            if timesteps.dim() == 0:
                timesteps = timesteps[None].repeat_interleave(B, dim=0)
            time_features = self.time_features(timesteps.to(images))
        temb, ss = time_features
        ssd = self._split_ss(ss)
        C = self._C
        dev = images.device
        buf = self._input_buffer(B, images)
        if images.data_ptr() != buf.data_ptr():
            K.copy_into(buf[:, : self.in_channels], images)

        def cat_buf(c_total, h, w):
            return torch.empty((B, c_total, h, w), device=dev, dtype=torch.float32)

        # pre-concatenated skip buffers: [:, :Cx] decoder half, [:, Cx:] encoder skip
        cat1 = cat_buf(2 * C[1], H, W)
        cat2 = cat_buf(2 * C[2], H // 2, W // 2)
        cat3 = cat_buf(2 * C[3], H // 4, W // 4)
        h = self.in_conv(buf, emit_stats=True)      # feeds d_block1's first GroupNorm: no statistics pass
        h1 = self.d_block1(h, temb, ssd["d_block1"], out=K.chan_slice(cat1, C[1], 2 * C[1]))
        h2 = self.d_block2(h1, temb, ssd["d_block2"], out=K.chan_slice(cat2, C[2], 2 * C[2]))
        h3 = self.d_block3(h2, temb, ssd["d_block3"], out=K.chan_slice(cat3, C[3], 2 * C[3]))
        h4 = self.d_block4(h3, temb, ssd["d_block4"])
        self.u_block4(h4, temb, ssd["u_block4"], out=K.chan_slice(cat3, 0, C[3]))
        self.u_block3(cat3, temb, ssd["u_block3"], out=K.chan_slice(cat2, 0, C[2]))
        self.u_block2(cat2, temb, ssd["u_block2"], out=K.chan_slice(cat1, 0, C[1]))
        h = self.u_block1(cat1, temb, ssd["u_block1"])
        return self.out_conv(h)
