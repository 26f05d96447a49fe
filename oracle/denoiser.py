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


# ----------------------------------------------------------------------------- layout-conditioned
# Restates /root/reference/lidargen/models/unets/layout_encoder.py:61-137 (Transformer),
# :237-303 (LayoutTransformerEncoder.forward) and layout_unet_v1.py:143-249 (ResBlock),
# :347-532 (ObjectAwareCrossAttention), :866-902 (LayoutUnetV1.forward); nn.py:17-19 GroupNorm32.
def gn32(sd, pre, x):
    return F.group_norm(x.float(), 32, sd[pre + ".weight"], sd[pre + ".bias"], 1e-5)


def _ln(sd, pre, x):
    return F.layer_norm(x, x.shape[-1:], sd[pre + ".weight"], sd[pre + ".bias"], 1e-5)


def _xf_attention(qkv, heads, key_padding_mask=None):
    """layout_encoder.py:67-84 (key_padding_mask [B, T] bool: True = padded key, :77-81)."""
    B, T, width = qkv.shape
    ch = width // heads // 3
    scale = 1 / math.sqrt(math.sqrt(ch))
    q, k, v = qkv.view(B, T, heads, -1).split(ch, dim=-1)
    w = torch.einsum("bthc,bshc->bhts", q * scale, k * scale)
    if key_padding_mask is not None:
        w = w.masked_fill(key_padding_mask[:, None, None, :], float("-inf"))
    w = w.softmax(-1)
    return torch.einsum("bhts,bshc->bthc", w, v).reshape(B, T, -1)


@torch.no_grad()
def layout_encoder_forward(sd, batch, *, feature_map_size, resolution_to_attention, num_heads=4,
                           prefix="", use_key_padding_mask=False):
    """-> the condition dict the denoiser consumes (layout_encoder.py:237-303, configuration
    used_condition_types=[obj_class,obj_bbox,is_valid_obj], no positional embedding, final LN)."""
    p = prefix
    boxes = batch["scaled_gt_boxes"]
    cls_emb = F.embedding(boxes[..., -1].long(), sd[p + "obj_class_embedding.weight"])
    box3 = F.linear(boxes[..., :8].float(), sd[p + "obj_bbox_embedding.weight"],
                    sd[p + "obj_bbox_embedding.bias"])
    box2 = F.linear(batch["gt_boxes_2d"].float(), sd[p + "obj_bbox_2d_embedding.weight"],
                    sd[p + "obj_bbox_2d_embedding.bias"])
    out = {"obj_class_embedding": cls_emb.permute(0, 2, 1),
           "obj_bbox_embedding": box2.permute(0, 2, 1)}
    B = boxes.shape[0]
    for r in resolution_to_attention:
        nh, nw = int(feature_map_size[0] / r), int(feature_map_size[1] / r)
        di, dj = 1.0 / (feature_map_size[0] / r), 1.0 / (feature_map_size[1] / r)
        cells = torch.tensor([(dj * j, di * i, dj * (j + 1), di * (i + 1))
                              for i in range(nh) for j in range(nw)], dtype=torch.float32)
        e = F.linear(cells, sd[p + "obj_bbox_2d_embedding.weight"],
                     sd[p + "obj_bbox_2d_embedding.bias"])
        out[f"image_patch_bbox_embedding_for_resolution{nh}"] = \
            e[None].repeat_interleave(B, 0).permute(0, 2, 1)
    out["key_padding_mask"] = (1 - batch["is_valid_obj"]).bool()
    x = cls_emb + box3 + box2
    i = 0
    while f"{p}transform.resblocks.{i}.ln_1.weight" in sd:
        q = f"{p}transform.resblocks.{i}"
        a = F.linear(_ln(sd, q + ".ln_1", x), sd[q + ".attn.c_qkv.weight"], sd[q + ".attn.c_qkv.bias"])
        a = F.linear(_xf_attention(a, num_heads, out["key_padding_mask"] if use_key_padding_mask else None),
                     sd[q + ".attn.c_proj.weight"],
                     sd[q + ".attn.c_proj.bias"])
        x = x + a
        m = F.linear(_ln(sd, q + ".ln_2", x), sd[q + ".mlp.c_fc.weight"], sd[q + ".mlp.c_fc.bias"])
        x = x + F.linear(F.gelu(m), sd[q + ".mlp.c_proj.weight"], sd[q + ".mlp.c_proj.bias"])
        i += 1
    x = _ln(sd, p + "final_ln", x)
    out["xf_proj"] = F.linear(x[:, 0], sd[p + "transformer_proj.weight"],
                              sd[p + "transformer_proj.bias"])
    out["xf_out"] = x.permute(0, 2, 1)
    if "concat_cond" in batch:
        cc = batch["concat_cond"]
        if "autoregressive_cond" in batch:
            cc = torch.cat([cc, batch["autoregressive_cond"]], dim=1)
        out["concat_cond"] = cc
    return out


def _conv1d(sd, pre, x):
    return F.conv1d(x, sd[pre + ".weight"], sd[pre + ".bias"])


def object_aware_attention(sd, pre, x, cond, resolution, opts=None):
    """layout_unet_v1.py:416-532.  opts: the constructor options no shipped configuration sets (:367-376) --
    {"norm_first": bool, "use_key_padding_mask": bool, "channels_scale_for_positional_embedding": float};
    norm_for_obj_embedding is discovered from its keys."""
    o = opts or {}
    norm_first, use_mask = bool(o.get("norm_first")), bool(o.get("use_key_padding_mask"))
    s_pos = float(o.get("channels_scale_for_positional_embedding", 1.0))
    B, C, H, W = x.shape
    heads = C // 32
    d = C // heads
    L1 = H * W
    xs = x.reshape(B, C, L1)
    qkv = _conv1d(sd, pre + ".qkv_projector", gn32(sd, pre + ".norm_for_qkv", xs))
    img = cond[f"image_patch_bbox_embedding_for_resolution{resolution}"]
    if norm_first:                                                       # :431-433, :457-459
        pos_img = _conv1d(sd, pre + ".layout_position_embedding_projector",
                          gn32(sd, pre + ".norm_for_image_patch_positional_embedding", img))
        pos_lay = _conv1d(sd, pre + ".layout_position_embedding_projector",
                          gn32(sd, pre + ".norm_for_layout_positional_embedding", cond["obj_bbox_embedding"]))
    else:                                                                # :435-438, :461-462
        pos_img = gn32(sd, pre + ".norm_for_image_patch_positional_embedding",
                       _conv1d(sd, pre + ".layout_position_embedding_projector", img))
        pos_lay = gn32(sd, pre + ".norm_for_layout_positional_embedding",
                       _conv1d(sd, pre + ".layout_position_embedding_projector",
                               cond["obj_bbox_embedding"]))
    xf = cond["xf_out"]
    if (pre + ".norm_for_obj_embedding.weight") in sd:                   # :466-467
        xf = gn32(sd, pre + ".norm_for_obj_embedding", xf)
    content = (xf + gn32(sd, pre + ".norm_for_obj_class_embedding", cond["obj_class_embedding"])) / 2
    k_lay, v_lay = _conv1d(sd, pre + ".layout_content_embedding_projector", content).split(C, 1)
    hv = lambda t: t.reshape(B * heads, -1, t.shape[-1])
    q, k, v = [hv(t) for t in qkv.split(C, dim=1)]
    pi, pl = hv(pos_img), hv(pos_lay)
    qm = torch.cat([q, pi], 1)
    km = torch.cat([torch.cat([k, pi], 1), torch.cat([hv(k_lay), pl], 1)], 2)
    vm = torch.cat([v, hv(v_lay)], 2)
    scale = 1 / math.sqrt(math.sqrt(int((1 + s_pos) * C) // heads))      # :489
    w = torch.einsum("bct,bcs->bts", qm * scale, km * scale)
    if use_mask:                                                         # :478-500
        L2 = pl.shape[-1]
        kpm = torch.cat([torch.zeros((B, L1), dtype=torch.bool), cond["key_padding_mask"].bool()], 1)
        w = w.view(B, heads, L1, L1 + L2).masked_fill(kpm[:, None, None, :], float("-inf")).view(B * heads, L1, L1 + L2)
    w = w.float().softmax(-1)
    a = torch.einsum("bts,bcs->bct", w, vm).reshape(B, C, L1)
    return (xs + _conv1d(sd, pre + ".proj_out", a)).reshape(B, C, H, W)


def res_block_v1(sd, pre, x, emb):
    """layout_unet_v1.py:143-249 with use_scale_shift_norm=True; up/down discovered from keys."""
    h = silu(gn32(sd, pre + ".in_layers.0", x))
    if (pre + ".op.kernel") in sd:
        up = float(sd[pre + ".op.kernel"].sum()) > 1.5  # up kernels are scaled by 2
        rs = resample_up2 if up else resample_down2
        h, x = rs(h), rs(x)
    h = conv_sd(sd, pre + ".in_layers.2", h)
    e = F.linear(silu(emb), sd[pre + ".emb_layers.1.weight"], sd[pre + ".emb_layers.1.bias"])
    scale, shift = e[:, :, None, None].chunk(2, dim=1)
    h = gn32(sd, pre + ".out_layers.0", h) * (1 + scale) + shift
    h = conv_sd(sd, pre + ".out_layers.3", silu(h))
    if (pre + ".skip_connection.weight") in sd:
        x = F.conv2d(x, sd[pre + ".skip_connection.weight"], sd[pre + ".skip_connection.bias"])
    return x + h


def _run_sequential(sd, pre, h, emb, cond, image_size, ds, attn_opts=None):
    """TimestepEmbedSequential (layout_unet_v1.py:63-78): children discovered from the keys."""
    i = 0
    while True:
        q = f"{pre}.{i}"
        if (q + ".in_layers.0.weight") in sd:
            h = res_block_v1(sd, q, h, emb)
            if (q + ".op.kernel") in sd:
                ds = ds * 2 if float(sd[q + ".op.kernel"].sum()) < 1.5 else ds // 2
        elif (q + ".qkv_projector.weight") in sd:
            h = object_aware_attention(sd, q, h, cond, image_size // ds, attn_opts)
        elif (q + ".weight") in sd and sd[q + ".weight"].ndim == 4:
            h = conv_sd(sd, q, h)
        else:
            break
        i += 1
    return h, ds


@torch.no_grad()
def layout_unet_v1_forward(sd, x, log_snr, cond, *, image_size, model_channels=64, prefix="", attn_opts=None):
    """layout_unet_v1.py:866-902."""
    p = prefix
    B, _, H, W = x.shape
    emb = time_mlp(sd, p + "time_embed", log_snr.float(), model_channels) + cond["xf_proj"]
    h = x.float()
    if "concat_cond" in cond:
        h = torch.cat([h, cond["concat_cond"]], dim=1)
    h = torch.cat([h, fourier_features(sd[p + "coords"], H, W).expand(B, -1, -1, -1)], dim=1)
    hs, ds, i = [], 1, 0
    while f"{p}input_blocks.{i}.0.weight" in sd or f"{p}input_blocks.{i}.0.in_layers.0.weight" in sd:
        h, ds = _run_sequential(sd, f"{p}input_blocks.{i}", h, emb, cond, image_size, ds, attn_opts)
        hs.append(h)
        i += 1
    h, ds = _run_sequential(sd, p + "middle_block", h, emb, cond, image_size, ds, attn_opts)
    i = 0
    while f"{p}output_blocks.{i}.0.in_layers.0.weight" in sd:
        h = torch.cat([h, hs.pop()], dim=1)
        h, ds = _run_sequential(sd, f"{p}output_blocks.{i}", h, emb, cond, image_size, ds, attn_opts)
        i += 1
    h = silu(gn32(sd, p + "out.0", h))
    return conv_sd(sd, p + "out.2", h)
