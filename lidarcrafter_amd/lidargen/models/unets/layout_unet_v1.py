"""Layout-conditioned range-image denoiser (LayoutUnetV1) on the gfx950 kernels.

API / state_dict mirror of the reference's lidargen/models/unets/layout_unet_v1.py
(TimestepEmbedSequential :63-78, ResBlock :143-249, ObjectAwareCrossAttention :347-532,
LayoutUnetV1 :599-902) for the configuration every shipped layout config uses
(`ObjectAwareCrossAttention`, `use_scale_shift_norm`, `resblock_updown`, fp32), restructured:

  * joint image/layout attention never materialises the [B*heads, L1, L1+13] score tensor nor the
    torch.cat of content|positional channels or image|layout keys: one flash-style launch
    (lc_attention_fwd) takes the parts as separate operands;
  * everything that depends only on the layout condition (positional embeddings of image patches
    and objects, layout keys/values of all 11 attention blocks) is computed once per batch in
    `prepare_condition`, not once per step as in the reference;
  * emb_layers of all ResBlocks are one dense launch; skip concatenations are free (producers
    write into slices of pre-concatenated buffers); GN->SiLU fused; skip + h in the conv epilogue.
"""
from __future__ import annotations

import math
import os

import torch
import torch as th
import torch.nn as nn

from lidarcrafter_amd import autograd as AG
from lidarcrafter_amd import ops as K

from . import encoding, ops
from .nn import SiLU, conv_nd, conv_nd_range, gn32_coeffs, linear, normalization, zero_module


class TimestepBlock(nn.Module):
    """Any module whose forward takes the timestep embedding as a second argument."""


PAIR_STATS = os.environ.get("LC_GN_PAIR_STATS", "1") != "0"   # developer switch (A/B of the pair entries)
QUAD_STATS = os.environ.get("LC_GN_QUAD_STATS", "1") != "0"   # ... of the pre-split kernel's quad entries (round 5)


PREPARE_GRAPH = os.environ.get("LC_PREPARE_GRAPH", "1") != "0"   # the condition operands of a run as one replayed graph


def _weights_fp(mods):
    """(address, version) of every parameter below `mods` (a plain walk: no dotted names are built)."""
    fp, stack, seen = [], list(mods), set()
    while stack:
        m = stack.pop()
        if id(m) in seen:
            continue
        seen.add(id(m))
        for t in m._parameters.values():
            if t is not None:
                fp.append((t.data_ptr(), _ver(t)))
        stack.extend(c for c in m._modules.values() if c is not None)
    return tuple(fp)


def _stats_unit(channels: int):
    """`emit_stats` of a conv whose output feeds a GroupNorm32: octet entries when the 32 groups are
    whole octets (>= 256 channels), quad entries at 128 channels (4 per group; the pre-split kernel writes quads, the
    fp32-input kernels pairs), pair entries at 64 (2 per group)."""
    cpg = channels // 32
    return True if cpg % 8 == 0 or not PAIR_STATS else (4 if cpg % 4 == 0 and QUAD_STATS else 2)


class ResBlock(TimestepBlock):
    def __init__(self, channels, emb_channels, dropout, out_channels=None, use_conv=False,
                 use_scale_shift_norm=False, dims=2, use_checkpoint=False, up=False, down=False):
        super().__init__()
        if not use_scale_shift_norm or dims != 2:
            raise NotImplementedError("HIP ResBlock: use_scale_shift_norm=True, dims=2 "
                                      "(all shipped layout configs)")
        self.channels, self.emb_channels, self.dropout = channels, emb_channels, dropout
        self.out_channels = out_channels or channels
        self.use_scale_shift_norm = True
        self.in_layers = nn.Sequential(
            normalization(channels), SiLU(),
            conv_nd_range(dims, channels, self.out_channels, 3, padding=1, ring=True))
        self.updown = up or down
        self._packed_up9 = K.PackedConv() if up else None      # the fold's 1x1 projection (ops.conv_up2)
        self._up9_cache = None
        self.fold_up_off = False
        if up:
            self.op = ops.Resample(up=2, ring=True)
        elif down:
            self.op = ops.Resample(down=2, ring=True)
        else:
            self.op = nn.Identity()
        self.emb_layers = nn.Sequential(SiLU(), linear(emb_channels, 2 * self.out_channels))
        self.out_layers = nn.Sequential(
            normalization(self.out_channels), SiLU(), nn.Dropout(p=dropout),
            zero_module(conv_nd_range(dims, self.out_channels, self.out_channels, 3, padding=1,
                                      ring=True)))
        if self.out_channels == channels:
            self.skip_connection = nn.Identity()
        elif use_conv:
            self.skip_connection = conv_nd_range(dims, channels, self.out_channels, 3, padding=1,
                                                 ring=True)
        else:
            self.skip_connection = conv_nd(dims, channels, self.out_channels, 1)

    def scale_shift(self, emb):
        ss = K.linear(emb, self.emb_layers[1].weight, self.emb_layers[1].bias, act_in=True)
        C = self.out_channels
        return ss[:, :C], ss[:, C:]

    # ---- up-sampling ResBlock folded (round 6): in_conv(op(act(norm(x)))) at the low resolution ------------------------
    def _up9_weight(self):
        """ops.up9_weight(in_conv.weight), rebuilt when the parameter changes (address / version)."""
        w = self.in_layers[2].weight
        key = (w.data_ptr(), _ver(w), w.device)
        if self._up9_cache is None or self._up9_cache[0] != key:
            self._up9_cache = (key, K.up9_weight(w))
        return self._up9_cache[1]

    def _fold_up_operand(self, x):
        """SiLU(GroupNorm(x)) pre-split for the fold's 1x1 projection, or None where the fold does not apply."""
        if not (x.is_cuda and x.dtype == th.float32 and x.dim() == 4):
            return None
        B, C, H, W = x.shape
        if not K.can_fold_up(C, self.out_channels, H, W) or self.fold_up_off:
            return None
        a = self.in_layers[0](x, act_silu=True, split_for=self._packed_up9)
        return a if isinstance(a, K.SplitAct) else None

    def forward(self, x, emb=None, scale_shift=None, out=None):
        scale, shift = scale_shift if scale_shift is not None else self.scale_shift(emb)
        fuse = K.fuse_gn(self.out_channels)
        su = _stats_unit(self.out_channels)
        if self.updown:      # GN -> SiLU -> resample -> conv: the norm cannot ride on the conv
            a = self._fold_up_operand(x) if self._packed_up9 is not None else None
            if a is not None:
                # conv3x3(up(a)) computed at the LOW resolution: one 1x1 projection to the nine tap planes (a quarter of the
                # multiply-adds) + the combine pass (ops.conv_up2, csrc/upfold.hip); it leaves per-channel statistics entries
                # (x's own up-sampling rides in the combine launch where the two tensors have the same channels)
                if K.FOLD_UP_X and x.shape[1] == self.out_channels:
                    h, x = K.conv_up2(a, self._packed_up9, self._up9_weight(), self.in_layers[2].bias, emit_stats=True,
                                      up_also=x)
                else:
                    h = K.conv_up2(a, self._packed_up9, self._up9_weight(), self.in_layers[2].bias, emit_stats=True)
                    x = self.op(x)
                a2 = self.out_layers[0](h, scale, shift, act_silu=True, split_for=self.out_layers[3]._packed)
                sk = x if isinstance(self.skip_connection, nn.Identity) else self.skip_connection(x, out=h)
                return self.out_layers[3](a2, res=sk, out=out, emit_stats=su)
            if K.RESAMPLE_PAIR and x.is_cuda and x.dtype == th.float32:
                n0 = self.in_layers[0]       # one pass over x for both resampled tensors
                a, x = K.groupnorm_resample_pair(x, n0.num_groups, n0.eps, n0.weight, n0.bias, up=self.op.up == 2)
            else:
                a = self.op(self.in_layers[0](x, act_silu=True))
                x = self.op(x)
            h = self.in_layers[2](a, emit_stats=su)
        elif fuse:
            a = None
            h = self.in_layers[2](x, gn_coeffs=gn32_coeffs(self.in_layers[0], x), emit_stats=su)
        else:
            a = self.in_layers[0](x, act_silu=True, split_for=self.in_layers[2]._packed)
            h = self.in_layers[2](a, emit_stats=su)   # statistics for out_layers[0]
            if isinstance(a, K.SplitAct):
                a = None
        if fuse:
            sk = x if isinstance(self.skip_connection, nn.Identity) else self.skip_connection(x)
            return self.out_layers[3](h, res=sk, out=out, emit_stats=su,
                                      gn_coeffs=gn32_coeffs(self.out_layers[0], h, scale, shift))
        reuse = a if a is not None and a.shape == h.shape and not K.can_presplit(
            self.out_channels, self.out_layers[0].num_groups) else None
        a2 = self.out_layers[0](h, scale, shift, act_silu=True, out=reuse,
                                split_for=self.out_layers[3]._packed)
        sk = x if isinstance(self.skip_connection, nn.Identity) else self.skip_connection(x, out=h)
        return self.out_layers[3](a2, res=sk, out=out, emit_stats=su)   # ... for the next block's norm


class ObjectAwareCrossAttention(nn.Module):
    def __init__(self, channels, num_heads=1, num_head_channels=-1, use_checkpoint=False,
                 encoder_channels=None, return_attention_embeddings=False, ds=None,
                 resolution=None, type=None, use_positional_embedding=True,
                 use_key_padding_mask=False, channels_scale_for_positional_embedding=1.0,
                 norm_first=False, norm_for_obj_embedding=False):
        super().__init__()
        assert use_positional_embedding and encoder_channels is not None
        self.type, self.ds, self.resolution, self.channels = type, ds, resolution, channels
        # the options no shipped configuration sets (layout_unet_v1.py:367-376, 402-412): normalisation before the
        # projectors, an extra norm of xf_out, positional channels = channels * scale, masked layout keys, and the
        # positional operands handed back to a direct caller of the layer
        self.norm_first = norm_first
        self.channels_scale_for_positional_embedding = channels_scale_for_positional_embedding
        self.use_key_padding_mask = use_key_padding_mask
        self.return_attention_embeddings = return_attention_embeddings
        if num_head_channels == -1:
            self.num_heads = num_heads
        else:
            assert channels % num_head_channels == 0
            self.num_heads = channels // num_head_channels
        self.encoder_channels = encoder_channels
        cpos = int(channels * channels_scale_for_positional_embedding)
        self.pos_channels = cpos
        if cpos % self.num_heads or (channels // self.num_heads) + cpos // self.num_heads > 64:
            raise NotImplementedError(
                f"HIP ObjectAwareCrossAttention: {cpos} positional channels over {self.num_heads} heads -- the attention "
                "kernels take content + positional channels per head <= 64 in whole heads")
        self.qkv_projector = conv_nd(1, channels, 3 * channels, 1)
        self.norm_for_qkv = normalization(channels)
        self.layout_content_embedding_projector = conv_nd(1, encoder_channels, channels * 2, 1)
        self.layout_position_embedding_projector = conv_nd(1, encoder_channels, cpos, 1)
        self.norm_for_obj_embedding = None
        if norm_first:
            if norm_for_obj_embedding:
                self.norm_for_obj_embedding = normalization(encoder_channels)
            self.norm_for_obj_class_embedding = normalization(encoder_channels)
            self.norm_for_layout_positional_embedding = normalization(encoder_channels)
            self.norm_for_image_patch_positional_embedding = normalization(encoder_channels)
        else:
            self.norm_for_obj_class_embedding = normalization(encoder_channels)
            self.norm_for_layout_positional_embedding = normalization(cpos)
            self.norm_for_image_patch_positional_embedding = normalization(cpos)
        self.proj_out = zero_module(conv_nd(1, channels, channels, 1))
        self._cond_cache = None

    def condition_operands(self, cond, refresh=False):
        """Step-invariant operands (reference recomputes them every step, :431-476).  `refresh`: recompute even when the
        cache key matches (a new condition may live at the addresses of the previous one)."""
        img = cond[f"image_patch_bbox_embedding_for_resolution{self.resolution}"]
        mask = cond["key_padding_mask"] if self.use_key_padding_mask else None
        key = (img.data_ptr(), cond["obj_bbox_embedding"].data_ptr(), cond["xf_out"].data_ptr(),
               cond["obj_class_embedding"].data_ptr(), _ver(cond["xf_out"]),
               self.layout_position_embedding_projector.weight._version,
               self.layout_content_embedding_projector.weight._version,
               None if mask is None else (mask.data_ptr(), _ver(mask)))
        if refresh or self._cond_cache is None or self._cond_cache[0] != key:
            C = self.channels
            proj = self.layout_position_embedding_projector
            nimg = self.norm_for_image_patch_positional_embedding
            # the image-side positional operand is a function of weights only when the encoder says so (its tag) and this
            # layer's own weights have not moved: kept across conditions
            itag = getattr(img, "_lc_weights_only", None)
            if itag is not None:
                itag = (itag, tuple(img.shape), proj.weight.data_ptr(), _ver(proj.weight), _ver(proj.bias),
                        _ver(nimg.weight), _ver(nimg.bias))
            old = self._cond_cache
            if itag is not None and old is not None and len(old) > 7 and old[7] == itag and old[1].is_inference() \
                    and torch.is_inference_mode_enabled():
                pos_img = old[1]
            elif self.norm_first:                                   # :431-433
                pos_img = proj(nimg(img))
            else:                                                   # :435-438
                pos_img = nimg(proj(img))
            if self.norm_first:                                     # :457-459
                pos_lay = proj(self.norm_for_layout_positional_embedding(cond["obj_bbox_embedding"]))
            else:                                                   # :461-462
                pos_lay = self.norm_for_layout_positional_embedding(proj(cond["obj_bbox_embedding"]))
            cls = self.norm_for_obj_class_embedding(cond["obj_class_embedding"])
            B, E, L2 = cls.shape
            xf = cond["xf_out"] if self.norm_for_obj_embedding is None else self.norm_for_obj_embedding(cond["xf_out"])
            content = K.add_scale(xf.reshape(B, E, 1, L2), cls.view(B, E, 1, L2), 0.5)
            kv = self.layout_content_embedding_projector(content.view(B, E, L2))
            k_lay, v_lay = kv[:, :C], kv[:, C:]
            per_sample = None
            if mask is not None:
                # masked_fill(-inf) before the softmax (:478-500) == those layout keys do not exist: every sample keeps
                # its own set of valid keys (compacted once per condition; the image keys are never masked)
                keep = (~mask.bool()).cpu()
                per_sample = []
                for b in range(B):
                    idx = torch.nonzero(keep[b]).flatten().to(kv.device)
                    per_sample.append(None if idx.numel() == 0 else tuple(
                        t[b:b + 1].index_select(2, idx).contiguous() for t in (k_lay, v_lay, pos_lay)))
            # the keyed tensors are kept alive by the cache: their addresses cannot be handed to
            # a different condition while it is valid (versions are not tracked in inference mode)
            hold = (img, cond["obj_bbox_embedding"], cond["xf_out"], cond["obj_class_embedding"], mask)
            ops4 = self._refill(self._cond_cache, (pos_img, pos_lay, k_lay, v_lay), per_sample)
            units = self._static_units(self._cond_cache, ops4, per_sample)
            self._cond_cache = (key,) + ops4 + (per_sample, hold, itag, units)
        return self._cond_cache[1:6]

    def _static_units(self, old, ops4, per_sample):
        """Keys / values in the attention kernel's unit form (ops.AttnUnits): the positional channels of every key and the
        layout keys / values are step-invariant -- split and packed here, once per condition; the content channels of the
        image keys and the image values are written per step by the qkv projection's epilogue (forward)."""
        pos_img, pos_lay, k_lay, v_lay = ops4
        B, _, L1 = pos_img.shape
        heads, C = self.num_heads, self.channels
        dqk, dpos, L2 = C // heads, self.pos_channels // heads, k_lay.shape[-1]
        if per_sample is not None or not pos_img.is_cuda or not K.AttnUnits.eligible(heads, L1, L2, dqk, dpos, dqk):
            return None
        u = old[8] if old is not None and len(old) > 8 else None
        if u is None or not u.fits(B, heads, L1, L2, dqk, dpos, dqk, pos_img.device) or not torch.is_inference_mode_enabled() \
                or not u.buf.is_inference():
            u = K.AttnUnits(B, heads, L1, L2, dqk, dpos, dqk, pos_img.device)
        K.attention_pack_units(u, pos_img, "k_pos")
        K.attention_pack_units(u, k_lay, "k", segment=1)
        K.attention_pack_units(u, pos_lay, "k_pos", segment=1)
        K.attention_pack_units(u, v_lay, "v", segment=1)
        return u

    def _refill(self, old, new4, per_sample):
        """The operands of a NEW condition go into the tensors of the previous one when they fit (same shapes, inside a
        sampling run): their addresses then stay what a captured denoising step reads, and the sampler can replay the
        graph of an earlier run (continuous_time.py::_graph_key) instead of capturing one per condition.  Never when
        the layer hands its operands out (`return_attention_embeddings`) or carries per-sample key sets."""
        if old is None or per_sample is not None or old[5] is not None or self.return_attention_embeddings \
                or not torch.is_inference_mode_enabled():
            return tuple(new4)
        prev = old[1:5]
        if not all(o.is_inference() and o.shape == n.shape and o.dtype == n.dtype and o.device == n.device
                   for o, n in zip(prev, new4)):
            return tuple(new4)
        for o, n in zip(prev, new4):
            if o is not n:                       # (the image-side operand may BE the previous one: weights only)
                o.copy_(n)
        return tuple(prev)

    def forward(self, x, cond_kwargs, out=None):
        B, C, H, W = x.shape
        L1 = H * W
        if K._dense_plane(x):   # token view of the same memory (a concat-buffer slice included)
            xs = K.alias(torch.as_strided(x, (B, C, L1), (x.stride(0), L1, 1)), x)
        else:
            xs = x.contiguous().view(B, C, L1)
        pos_img, pos_lay, k_lay, v_lay, per_sample = self.condition_operands(cond_kwargs)
        heads = self.num_heads
        # (q*s)(k*s) with s = (((1 + scale_pos) C) / heads)^-1/4, :489-492
        scale = 1.0 / math.sqrt(int((1 + self.channels_scale_for_positional_embedding) * C) // heads)
        units = self._cond_cache[8] if K.ATTN_UNITS and K.ATTN_PRECISION == "f16x2" else None
        if units is not None and units.B == B:
            # keys and values in unit form: split once per step (by the projection's own epilogue where its tile fits the
            # heads, else by one pass over k and one over v) instead of once per query block inside the attention kernel
            if K.presplit_1x1(C, 3 * C, self.norm_for_qkv.num_groups) and K.qkv_units_ok(C, heads, L1):
                xsplit = self.norm_for_qkv(xs, split_for=self.qkv_projector._packed)
                q = K.qkv_project_units(xsplit, self.qkv_projector._packed, self.qkv_projector.weight,
                                        self.qkv_projector.bias, units)
            else:
                if K.fuse_gn(3 * C):
                    qkv = self.qkv_projector(xs, gn_coeffs=gn32_coeffs(self.norm_for_qkv, xs))
                else:
                    qkv = self.qkv_projector(self.norm_for_qkv(xs))
                q = qkv[:, :C]
                K.attention_pack_units(units, qkv[:, C:2 * C], "k")
                K.attention_pack_units(units, qkv[:, 2 * C:], "v")
            a = K.attention_units(q, units, heads, scale, q_pos=pos_img)
            return self._project_out(x, xs, a, out, pos_img, pos_lay)
        if K.presplit_1x1(C, 3 * C, self.norm_for_qkv.num_groups):
            # many output channels: the norm writes hi / lo planes once, the projection stages them by LDS-DMA
            qkv = self.qkv_projector(self.norm_for_qkv(xs, split_for=self.qkv_projector._packed))
        elif K.fuse_gn(3 * C):
            qkv = self.qkv_projector(xs, gn_coeffs=gn32_coeffs(self.norm_for_qkv, xs))
        else:
            qkv = self.qkv_projector(self.norm_for_qkv(xs))
        if per_sample is None:
            a = K.attention_cm(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], heads, scale,
                               k2=k_lay, v2=v_lay, q_pos=pos_img, k_pos=pos_img, k2_pos=pos_lay)
        else:
            a = torch.empty((B, C, L1), device=x.device, dtype=torch.float32)
            for b in range(B):
                sl = slice(b, b + 1)
                kl, vl, pl = per_sample[b] if per_sample[b] is not None else (None, None, None)
                K.attention_cm(qkv[sl, :C], qkv[sl, C:2 * C], qkv[sl, 2 * C:], heads, scale, k2=kl, v2=vl,
                               q_pos=pos_img[sl], k_pos=pos_img[sl], k2_pos=pl, out=a[sl])
        return self._project_out(x, xs, a, out, pos_img, pos_lay)

    def _project_out(self, x, xs, a, out, pos_img, pos_lay):
        B, C, H, W = x.shape
        L1, heads = H * W, self.num_heads
        o3 = None if out is None else out
        # (statistics for the next block's GroupNorm: they follow the tensor through the token view)
        y = self.proj_out(a, res=xs, out=None if o3 is None else K.alias(_as3(o3), o3), emit_stats=True)
        extra = None
        if self.return_attention_embeddings:                        # :512-530
            extra = {"type": self.type, "ds": self.ds, "resolution": self.resolution, "num_heads": heads,
                     "num_channels": C, "image_query_embeddings": pos_img.detach().reshape(B, -1, L1),
                     "layout_key_embeddings": pos_lay.detach().reshape(B, -1, pos_lay.shape[-1])}
        return (K.alias(y.view(B, C, H, W), y) if out is None else out), extra


def _ver(t):
    """Version counter of a tensor (inference-mode tensors do not track one)."""
    try:
        return t._version
    except RuntimeError:
        return -1


def _as3(t4):
    B, C, H, W = t4.shape
    return torch.as_strided(t4, (B, C, H * W), (t4.stride(0), t4.stride(1), 1))


class TimestepEmbedSequential(nn.Sequential, TimestepBlock):
    def forward(self, x, emb, cond_kwargs=None, scale_shifts=None, out=None):
        """`scale_shifts`: iterator of precomputed (scale, shift) pairs for the ResBlocks inside;
        `out`: destination view for the LAST layer's result."""
        n = len(self)
        extra = None                                 # layout_unet_v1.py:70-78: the last attention layer's extra output
        for i, layer in enumerate(self):
            o = out if i == n - 1 else None
            if isinstance(layer, ResBlock):
                ss = next(scale_shifts) if scale_shifts is not None else None
                x = layer(x, emb, scale_shift=ss, out=o)
            elif isinstance(layer, ObjectAwareCrossAttention):
                x, extra = layer(x, cond_kwargs, out=o)
            elif isinstance(layer, ops.Conv2d):      # the input convolution feeds the first block's norm
                x = layer(x, out=o, emit_stats=_stats_unit(layer.out_channels))
            else:
                x = layer(x, out=o)
        return x, extra


class LayoutUnetV1(nn.Module):
    def __init__(self, in_channels, resolution, model_channels, out_channels, num_res_blocks,
                 attention_ds, encoder_channels=None, dropout=0, channel_mult=(1, 2, 4, 8),
                 conv_resample=True, dims=2, use_checkpoint=False, use_fp16=False, num_heads=1,
                 num_head_channels=-1, num_heads_upsample=-1, use_scale_shift_norm=False,
                 resblock_updown=False, use_positional_embedding_for_attention=False,
                 image_size=256, attention_block_type="GLIDE", num_attention_blocks=1,
                 use_key_padding_mask=False, channels_scale_for_positional_embedding=1.0,
                 norm_first=False, norm_for_obj_embedding=False,
                 coords_encoding="fourier_features", **kwargs):
        super().__init__()
        if attention_block_type != "ObjectAwareCrossAttention" or not resblock_updown or use_fp16 \
                or coords_encoding != "fourier_features":
            raise NotImplementedError(
                "HIP LayoutUnetV1 implements the shipped configuration: ObjectAwareCrossAttention, "
                "resblock_updown, fourier_features, fp32 (option_nusc_box_layout_v6.py:10-32)")
        self.in_channels = in_channels
        self.image_size = image_size
        self.resolution = tuple(resolution)
        self.register_buffer("coords", encoding.generate_polar_coords(*self.resolution))
        self.coords_encoding = encoding.FourierFeatures(self.resolution)
        cin = in_channels + self.coords_encoding.extra_ch
        self.encoder_channels, self.model_channels = encoder_channels, model_channels
        self.out_channels, self.num_res_blocks = out_channels, num_res_blocks
        self.attention_ds, self.dropout, self.channel_mult = attention_ds, dropout, channel_mult
        self.dtype = th.float32
        self.num_heads, self.num_head_channels = num_heads, num_head_channels
        self.num_heads_upsample = num_heads if num_heads_upsample == -1 else num_heads_upsample
        self.num_attention_blocks = num_attention_blocks
        ted = model_channels * 4
        self.time_embed = nn.Sequential(ops.SinusoidalPositionalEmbedding(model_channels),
                                        nn.Linear(model_channels, ted), nn.SiLU(),
                                        nn.Linear(ted, ted))

        def res(ch_in, ch_out=None, **kw):
            return ResBlock(ch_in, ted, dropout, out_channels=ch_out, dims=dims,
                            use_scale_shift_norm=use_scale_shift_norm, **kw)

        def attn(ch, ds, kind, heads):
            return ObjectAwareCrossAttention(
                ch, num_heads=heads, num_head_channels=num_head_channels,
                encoder_channels=encoder_channels, ds=ds, resolution=int(image_size // ds),
                type=kind, use_positional_embedding=use_positional_embedding_for_attention,
                use_key_padding_mask=use_key_padding_mask,
                channels_scale_for_positional_embedding=channels_scale_for_positional_embedding,
                norm_first=norm_first, norm_for_obj_embedding=norm_for_obj_embedding)

        ch = input_ch = int(channel_mult[0] * model_channels)
        self.input_blocks = nn.ModuleList(
            [TimestepEmbedSequential(conv_nd_range(dims, cin, ch, 3, padding=1, ring=True))])
        chans, dss, ds = [ch], [1], 1
        for level, mult in enumerate(channel_mult):
            for _ in range(num_res_blocks):
                layers = [res(ch, int(mult * model_channels))]
                ch = int(mult * model_channels)
                if ds in attention_ds:
                    layers += [attn(ch, ds, "input", num_heads) for _ in range(num_attention_blocks)]
                self.input_blocks.append(TimestepEmbedSequential(*layers))
                chans.append(ch), dss.append(ds)
            if level != len(channel_mult) - 1:
                self.input_blocks.append(TimestepEmbedSequential(res(ch, ch, down=True)))
                ds *= 2
                chans.append(ch), dss.append(ds)
        self.middle_block = TimestepEmbedSequential(res(ch), attn(ch, ds, "middle", num_heads),
                                                    res(ch))
        self._skip_chans, self._skip_ds = list(chans), list(dss)
        self.output_blocks = nn.ModuleList([])
        for level, mult in list(enumerate(channel_mult))[::-1]:
            for i in range(num_res_blocks + 1):
                ich = chans.pop()
                layers = [res(ch + ich, int(model_channels * mult))]
                ch = int(model_channels * mult)
                if ds in attention_ds:
                    layers += [attn(ch, ds, "output", self.num_heads_upsample)
                               for _ in range(num_attention_blocks)]
                if level and i == num_res_blocks:
                    layers.append(res(ch, ch, up=True))
                    ds //= 2
                self.output_blocks.append(TimestepEmbedSequential(*layers))
        self.out = nn.Sequential(normalization(ch), SiLU(),
                                 zero_module(conv_nd_range(dims, input_ch, out_channels, 3,
                                                           padding=1, ring=True)))
        self.use_fp16 = use_fp16
        self._emb_cache = None
        self._in_buf = None
        K.name_packed_convs(self)

    # ---- batched / step-invariant helpers -------------------------------------------------------
    def _res_blocks(self):
        seqs = list(self.input_blocks) + [self.middle_block] + list(self.output_blocks)
        return [m for s in seqs for m in s if isinstance(m, ResBlock)]

    def _emb_weights(self):
        mods = self._res_blocks()
        key = tuple((m.emb_layers[1].weight.data_ptr(), m.emb_layers[1].weight._version,
                     m.emb_layers[1].bias._version) for m in mods)
        if self._emb_cache is None or self._emb_cache[0] != key:
            w = torch.cat([m.emb_layers[1].weight.detach() for m in mods], 0).contiguous()
            b = torch.cat([m.emb_layers[1].bias.detach() for m in mods], 0).contiguous()
            self._emb_cache = (key, w, b)
        return self._emb_cache[1], self._emb_cache[2]

    def time_features(self, log_snr: torch.Tensor, other_condition: dict = None):
        """log-SNR [M] (M = S*B rows, step-major) + xf_proj [B, T] ->
        (emb [M, T], all ResBlock (scale|shift) rows [M, sum 2C])."""
        te = self.time_embed
        h = K.linear(te[0](log_snr), te[1].weight, te[1].bias, act_out=True)
        emb = K.linear(h, te[3].weight, te[3].bias)
        if other_condition is not None:
            xf = other_condition["xf_proj"].to(emb)
            emb = (emb.view(-1, xf.shape[0], emb.shape[1]) + xf[None]).view(emb.shape)
        w, b = self._emb_weights()
        return emb, K.linear(emb, w, b, act_in=True)

    def _ss_iter(self, ss):
        off = 0
        for m in self._res_blocks():
            C = m.out_channels
            yield ss[:, off:off + C], ss[:, off + C:off + 2 * C]
            off += 2 * C

    def _input_buffer(self, B, x):
        """Persistent [B, in_channels + 30, H, W] buffer: x | concat_cond | Fourier features."""
        H, W = self.resolution
        enc = self.coords_encoding(self.coords)
        key = (B, x.device, enc.data_ptr())
        if self._in_buf is None or self._in_buf[0] != key:
            buf = torch.empty((B, self.in_channels + enc.shape[1], H, W), device=x.device,
                              dtype=torch.float32)
            K.copy_into(buf[:, self.in_channels:], enc.expand(B, -1, -1, -1) if B > 1 else enc)
            self._in_buf = (key, buf, None)
        return self._in_buf[1]

    def prepare_condition(self, layout_outputs: dict):
        """Compute everything that depends only on the layout condition (once per batch)."""
        if self._in_buf is not None:          # a new condition: recopy its channels (now when the buffer of this
            self._in_buf = (self._in_buf[0], self._in_buf[1], None)   # batch exists, else in the next forward)
            cc = layout_outputs.get("concat_cond")
            buf = self._in_buf[1]
            if cc is not None and buf.shape[0] == cc.shape[0] and buf.device == cc.device:
                self._bind_concat(buf, cc, self.in_channels - cc.shape[1])
        layers = self._attention_layers()
        if not self._prepare_by_graph(layout_outputs, layers):
            for m in layers:
                m.condition_operands(layout_outputs, refresh=True)   # never trust a cache across conditions

    # The operands of a condition are ~12 small launches per attention layer (projections, norms, the unit form of the
    # static keys): 3.6 ms of host time per `sample()` call for the 11 layers of the shipped model, paid while the GPU
    # idles -- all of it a function of three small tensors of the layout encoder.  From the second condition of a shape
    # on they are copied into static inputs and the refresh is ONE replayed HIP graph (captured at that second condition;
    # the results land in the tensors of the first, which is also what lets the sampler replay its step graph).
    _PREP_KEYS = ("obj_bbox_embedding", "xf_out", "obj_class_embedding")

    def _prepare_by_graph(self, lay, layers) -> bool:
        if not (PREPARE_GRAPH and layers and torch.is_inference_mode_enabled() and K.PROFILE is None
                and all(k in lay and lay[k].is_cuda for k in self._PREP_KEYS)
                and not torch.cuda.is_current_stream_capturing()):
            return False
        if any(m.use_key_padding_mask or m.return_attention_embeddings for m in layers):
            return False
        tags = tuple(getattr(lay.get(f"image_patch_bbox_embedding_for_resolution{m.resolution}"), "_lc_weights_only", None)
                     for m in layers)
        if any(t is None for t in tags):          # the image-side operand must be the kept, weights-only one
            return False
        sig = (tuple((k, tuple(lay[k].shape), lay[k].dtype, lay[k].device) for k in self._PREP_KEYS), tags,
               _weights_fp(layers), K.route_signature())
        st = self.__dict__.get("_prep")
        if st is None or st["sig"] != sig:
            st = dict(sig=sig, inp={k: lay[k].clone() for k in self._PREP_KEYS}, graph=None, runs=0)
            self.__dict__["_prep"] = st
        else:
            for k in self._PREP_KEYS:
                st["inp"][k].copy_(lay[k])
        # the run reads the condition from the static inputs from here on: `forward` swaps THIS condition (recognised by the
        # addresses of its three tensors, which `src` keeps alive) for the static view -- the caller's dict is left alone, it
        # may be used again after other conditions have gone through the static inputs
        st["src"] = tuple(lay[k] for k in self._PREP_KEYS)
        st["src_id"] = tuple((t.data_ptr(), tuple(t.shape)) for t in st["src"])
        lay = dict(lay)
        lay.update(st["inp"])
        st["lay"] = lay
        if st["graph"] is None and st["runs"] >= 1:
            try:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    for m in layers:
                        m.condition_operands(lay, refresh=True)
                st["graph"] = g
            except Exception as e:                  # an optimisation only
                import warnings

                warnings.warn(f"HIP graph capture of the condition operands failed ({e!r}); staying eager")
                st["graph"] = False
        if st["graph"]:
            st["graph"].replay()
            return True
        for m in layers:
            m.condition_operands(lay, refresh=True)
        st["runs"] += 1
        return True

    def _static_condition(self, lay):
        """The static view of the condition `prepare_condition` last saw, when `lay` is that condition."""
        st = self.__dict__.get("_prep")
        if st is None or st.get("lay") is None or not all(k in lay for k in self._PREP_KEYS):
            return lay
        if tuple((lay[k].data_ptr(), tuple(lay[k].shape)) for k in self._PREP_KEYS) != st["src_id"]:
            return lay
        return st["lay"]

    def __deepcopy__(self, memo):
        # (a captured graph neither copies nor pickles: copy.deepcopy(ddpm) -- the reference trainers' EMA wrapper)
        import copy

        cls = self.__class__
        new = cls.__new__(cls)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            new.__dict__[k] = None if k == "_prep" else copy.deepcopy(v, memo)
        return new

    def __getstate__(self):
        d = dict(self.__dict__)
        d["_prep"] = None
        return d

    def _attention_layers(self):
        seqs = list(self.input_blocks) + [self.middle_block] + list(self.output_blocks)
        return [m for s in seqs for m in s if isinstance(m, ObjectAwareCrossAttention)]

    def graph_operands(self):
        """The condition tensors a captured denoising step reads by address (every attention layer's step-invariant
        operands), or None when a step cannot be replayed across conditions (per-sample key sets, nothing prepared)."""
        out = []
        for m in self._attention_layers():
            c = m._cond_cache
            if c is None or c[5] is not None:
                return None
            out.extend(c[1:5])
            if len(c) > 8 and c[8] is not None:
                out.append(c[8].buf)                 # keys / values in unit form (static parts + the per-step rows)
        return out

    def _bind_concat(self, buf, cc, cx):
        """Condition channels are step-invariant: copied into the resident input buffer once per condition."""
        ck = (cc.data_ptr(), _ver(cc), cc)       # holds cc: its address cannot be reused meanwhile
        if self._in_buf[2] is None or self._in_buf[2][:2] != ck[:2]:
            K.copy_into(buf[:, cx:cx + cc.shape[1]], cc.float().contiguous())
            self._in_buf = (self._in_buf[0], buf, ck)

    @torch.compiler.disable
    @K.range_checked
    def forward(self, x, cond_dict, time_features=None):
        lay = cond_dict["other_condition"]
        if time_features is None and AG.training_active(self, x, lay.get("xf_proj"), lay.get("xf_out")):
            # training: the autograd graph over the HIP kernels (lidarcrafter_amd/autograd.py)
            return AG.layout_unet_v1_forward(self, x, cond_dict)
        lay = self._static_condition(lay)
        B, cx, H, W = x.shape
        if time_features is None:
            t = cond_dict["time_condition"]
            if t.dim() == 0:
                t = t[None].repeat_interleave(B, dim=0)
            time_features = self.time_features(t.to(x), lay)
        emb, ss = time_features
        ssi = self._ss_iter(ss)
        buf = self._input_buffer(B, x)
        if x.data_ptr() != buf.data_ptr():
            K.copy_into(buf[:, :cx], x)
        if "concat_cond" in lay:
            self._bind_concat(buf, lay["concat_cond"], cx)
        dev = x.device
        # pre-concatenated buffers: output block j reads cat[h, skip_(n-1-j)]
        n_in = len(self.input_blocks)
        first = [next(m for m in blk if isinstance(m, ResBlock)).channels for blk in self.output_blocks]
        cats = []
        for j in range(len(self.output_blocks)):
            i = n_in - 1 - j
            d = self._skip_ds[i]
            cats.append(torch.empty((B, first[j], H // d, W // d), device=dev, dtype=torch.float32))
        h = buf
        for i, blk in enumerate(self.input_blocks):
            j = n_in - 1 - i
            dst = K.chan_slice(cats[j], first[j] - self._skip_chans[i], first[j])
            h, _ = blk(h, emb, lay, scale_shifts=ssi, out=dst)
        dst = K.chan_slice(cats[0], 0, first[0] - self._skip_chans[n_in - 1])
        self.middle_block(h, emb, lay, scale_shifts=ssi, out=dst)
        for j, blk in enumerate(self.output_blocks):
            if j + 1 < len(self.output_blocks):
                dst = K.chan_slice(cats[j + 1], 0, first[j + 1] - self._skip_chans[n_in - 2 - j])
            else:
                dst = None
            h, _ = blk(cats[j], emb, lay, scale_shifts=ssi, out=dst)
        if K.fuse_gn(self.out_channels):
            return self.out[2](h, gn_coeffs=gn32_coeffs(self.out[0], h))
        return self.out[2](self.out[0](h, act_silu=True))
