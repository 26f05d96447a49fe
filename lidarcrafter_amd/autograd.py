"""Training path of the denoiser: autograd through the HIP kernels.

The reference trains by running PyTorch autograd through the nn.Modules of the denoiser
(tools/train/train_lidm.py:214-265, train_lidm_cond.py:259-322: `loss = ddpm(x_0)`,
`accelerator.backward(loss)`, AdamW, EMA).  The inference forward of this build is not an autograd
graph (fused GroupNorm, statistics hand-overs, pre-split activations, in-place concat buffers), so
training takes a second, plain composition of the same layers in which every heavy op is a
`torch.autograd.Function` over the C ABI:

  ConvRing      forward  lc_conv2d_ring_*_fwd
                backward dX: the same ring convolution of dY with the transposed, 180-degree rotated
                         kernel (forward kernel; its weight packed straight from the forward weight,
                         lc_pack_conv_weight_f16x2_dx);  dW, db: lc_conv2d_ring_wgrad[_f16x2] (MFMA implicit
                         GEMM over pixels, deterministic two-stage reduction)
  GroupNormAct  forward  lc_groupnorm_stats + lc_groupnorm_apply_train (+ AdaGN scale/shift, + SiLU; leaves (mean,
                         rstd) for the backward and the partial maxima of |y| for the range record of the conv
                         that consumes y)
                backward lc_groupnorm_bwd_train (rows + dx, the parameter gradients, partial maxima of |dx|)
  Resample2x    forward  lc_resample2x_fwd;  backward: the adjoint FIR = the opposite resampling
                         (down^T = up / 4, up^T = 4 down: same window, ring / zero padding)

  FlashAttention forward lc_attention_train_fwd (flash kernel + log2-sum-exp); backward lc_attention_bwd[_f16x2]

The per-step glue that is tiny next to those -- time MLP and AdaGN projections ([B, 256] dense layers), the
13-token layout operands, residual adds, channel concatenation, dropout -- runs as differentiable torch ops ON
THE DEVICE; nothing runs on the CPU.  Gradient parity against autograd of the CPU
oracle: tests/test_training.py.  Data-parallel training: the parameters are ordinary
nn.Parameters, so torch DistributedDataParallel (RCCL bucketed all-reduce overlapped with backward)
wraps `ddpm` unchanged, exactly like `accelerator.prepare` does in the reference.
"""
from __future__ import annotations

import os

import torch
import torch.nn.functional as F

from . import ops as K
from ._lib import check, lib

# arithmetic of the forward / dX convolutions inside the training graph: "f16x2" = the split kernels
# (fp32-class accuracy; the pre-scale of every operand -- activation or gradient -- comes from max|.| of that very
# tensor, measured on the device right before its conv (K.range_from_tensor) or left behind by the GroupNorm pass that
# wrote it (K.range_from_amax)), "f32" = exact-fp32 MFMA kernels.
TRAIN_CONV_PRECISION = os.environ.get("LC_TRAIN_CONV_PRECISION", "f16x2")
# ... and of the weight gradient: "f16x2" = lc_conv2d_ring_wgrad_f16x2 (same split, same records; used
# when the forward / dX convs are split too and the shape has whole 2 x 32 tiles), "f32" = exact fp32.
TRAIN_WGRAD_PRECISION = os.environ.get("LC_TRAIN_WGRAD_PRECISION", "f16x2")


# Attention of the training graph: "hip" = flash forward (+ log2-sum-exp) and flash backward that recomputes P, each in
# the f16x2 split (default) or exact-fp32 MFMA (csrc/attention.hip, attention_bwd_h.hip, attention_bwd.hip: no score
# matrix in HBM, deterministic); "torch" = einsum / softmax over materialised scores (the round-2/3 route; kept for
# A/B and for heads wider than 64 channels).
TRAIN_ATTENTION = os.environ.get("LC_TRAIN_ATTENTION", "hip")
TRAIN_ATTN_FWD_PRECISION = os.environ.get("LC_TRAIN_ATTN_FWD_PRECISION", "f16x2")
TRAIN_ATTN_BWD_PRECISION = os.environ.get("LC_TRAIN_ATTN_BWD_PRECISION", "f16x2")   # "f32": exact-fp32 MFMA backward


def training_active(module: torch.nn.Module, *tensors) -> bool:
    """True when the caller expects an autograd graph: grad mode on and something requires grad."""
    if not torch.is_grad_enabled():
        return False
    # (a pure predicate: the "this forward took the training graph" mark of ops.range_checked is set where the graph
    #  is ENTERED -- begin_training_forward -- not here: a caller that merely asks, e.g. `training_active(...) and
    #  hasattr(...)`, must not switch the fp16 saturation check off for a forward that runs the inference kernels)
    return any(isinstance(t, torch.Tensor) and t.requires_grad for t in tensors) or \
        any(p.requires_grad for p in module.parameters())


# max|.| left behind by the pass that WROTE a tensor (GroupNormAct forward / backward: one partial maximum per block, plain
# stores), so that the conv consuming the tensor sets its range record from a few hundred floats instead of reading the
# tensor again (lc_range_from_amax).  The tensor object carries (partials, version, data_ptr, bound_mult) and a consumer
# trusts the tag only while version and address still match -- the autograd engine may accumulate a second gradient into
# the same tensor in place, which bumps the version.
PRODUCER_AMAX = os.environ.get("LC_TRAIN_PRODUCER_AMAX", "1") != "0"
# every conv weight of the training graph packed in three launches per optimizer step (ops.TrainWeightPlan) instead of
# per layer; "0" keeps the per-layer packing
MULTI_WEIGHT_PACK = os.environ.get("LC_TRAIN_MULTI_WEIGHT_PACK", "1") != "0"


def begin_training_forward(device) -> None:
    """Start of a forward through the training graph: weights that an optimizer step moved are packed now, all at once.
    Marks the forward for ops.range_checked: every operand's pre-scale is measured on the device right before its conv
    (range_from_tensor / producer maxima), so the post-forward poll and retry of the inference path are skipped."""
    K._range_defer.autograd_route = True
    if MULTI_WEIGHT_PACK and TRAIN_CONV_PRECISION == "f16x2":
        K.train_weight_plan(device).refresh()


def _amax_slot(device, B, C, H, W, G, backward: bool) -> torch.Tensor:
    n = int(lib().lc_groupnorm_amax_partials(B, C, H, W, G, int(backward)))
    return torch.empty(n, device=device, dtype=torch.float32)


def _tag_amax(t: torch.Tensor, slot: torch.Tensor, mult: float = 1.0) -> None:
    t._lc_amax = (slot, t._version, t.data_ptr(), mult)


def _amax_of(t: torch.Tensor):
    """(partials, bound_mult) if `t` still is the tensor its producer measured, else None."""
    tag = getattr(t, "_lc_amax", None)
    if tag is None or tag[1] != t._version or tag[2] != t.data_ptr():
        return None
    return tag[0], tag[3]


def _carry_amax(src: torch.Tensor, dst: torch.Tensor, mult: float = 1.0) -> torch.Tensor:
    """`dst` holds the values of `src` (a view / reshape) or an elementwise contraction of them by at most `mult`."""
    m = _amax_of(src) if PRODUCER_AMAX else None
    if m is not None:
        _tag_amax(dst, m[0], m[1] * mult)
    return dst


def _tag_amax(t: torch.Tensor, slot: torch.Tensor, mult: float = 1.0) -> None:
    t._lc_amax = (slot, t._version, t.data_ptr(), mult)


def _amax_of(t: torch.Tensor):
    """(slot, bound_mult) if `t` still is the tensor its producer measured, else None."""
    tag = getattr(t, "_lc_amax", None)
    if tag is None or tag[1] != t._version or tag[2] != t.data_ptr():
        return None
    return tag[0], tag[3]


def dropout(h: torch.Tensor, p: float, training: bool) -> torch.Tensor:
    """F.dropout that keeps the producer's max|h| usable: the kept elements are h / (1 - p), so max|h| / (1 - p) bounds
    the result (an upper bound is all a range record needs)."""
    if not training or p <= 0.0:
        return h
    y = F.dropout(h, p, training=True)
    return _carry_amax(h, y, 1.0 / (1.0 - p)) if p < 1.0 else y


def _c4(t):
    """A [B,C,H,W] tensor the kernels can take as it is: inner [C,H,W] block contiguous, ANY batch
    stride (the gradient of a channel concatenation arrives as a slice of the concatenated
    gradient -- copying it would cost a pass over the tensor per conv); anything else is copied."""
    if t.dim() == 4:
        B, C, H, W = t.shape
        st = t.stride()
        if (W == 1 or st[3] == 1) and (H == 1 or st[2] == W) and (C == 1 or st[1] == H * W) and \
                (B == 1 or st[0] >= C * H * W):
            return t
    return t.contiguous()


def _bs(t):
    """Batch stride in elements of a tensor accepted by _c4."""
    return t.stride(0) if t.shape[0] > 1 else t.shape[1] * t.shape[2] * t.shape[3]


class ConvRing(torch.autograd.Function):
    """y = (conv_ring(x, W) + b [+ res]) * out_scale (3x3: W circular / H zero padding, 1x1: plain).  The residual add
    and the 1/sqrt(2) of a block's exit run in the conv's epilogue, as in the inference forward; the backward
    differentiates g = dy * out_scale through the conv and hands g to `res`."""

    @staticmethod
    def forward(ctx, x, weight, bias, holder, amax=None, res=None, out_scale=1.0):
        x = _c4(x)
        x_rec = None
        if TRAIN_CONV_PRECISION == "f16x2":
            if amax is not None:
                K.range_from_amax(amax[0], holder["fwd"], x.device, amax[1])
            else:
                K.range_from_tensor(x, holder["fwd"])
            # the record x was measured with, kept with the saved activation: the same module may run forward again
            # (shared weights, two forwards feeding one loss) before this backward and re-measure the live record
            x_rec = holder["fwd"].range_snapshot(x.device)
        y = K.conv2d_ring(x, holder["fwd"], weight, bias, res=None if res is None else _c4(res),
                          out_scale=float(out_scale), precision=TRAIN_CONV_PRECISION)
        ctx.save_for_backward(x, weight)
        ctx.holder, ctx.has_bias, ctx.x_rec = holder, bias is not None, x_rec
        ctx.out_scale, ctx.has_res = float(out_scale), res is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy_amax = _amax_of(dy) if PRODUCER_AMAX else None      # of the tensor as its producer handed it over
        if ctx.out_scale != 1.0:
            dy = dy * ctx.out_scale                             # the gradient of everything inside the parentheses
            if dy_amax is not None:
                dy_amax = (dy_amax[0], dy_amax[1] * max(1.0, abs(ctx.out_scale)))     # a bound, never below the truth
        d_res = dy if ctx.has_res and ctx.needs_input_grad[5] else None
        dy = _c4(dy)
        B, Ci, H, W = x.shape
        Co, ks = weight.shape[0], weight.shape[-1]
        dx = dw = db = None
        need_w = ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2])
        # the split weight gradient needs whole 2 x 32 pixel tiles and 16-byte aligned rows
        w_split = (need_w and TRAIN_WGRAD_PRECISION == "f16x2" and TRAIN_CONV_PRECISION == "f16x2" and
                   H % 2 == 0 and W % 32 == 0 and _bs(x) % 4 == 0 and _bs(dy) % 4 == 0 and
                   x.data_ptr() % 16 == 0 and dy.data_ptr() % 16 == 0)
        if TRAIN_CONV_PRECISION == "f16x2" and (ctx.needs_input_grad[0] or w_split):
            if dy_amax is not None:                             # one measurement serves dX and dW
                K.range_from_amax(dy_amax[0], ctx.holder["bwd"], dy.device, dy_amax[1])
            else:
                K.range_from_tensor(dy, ctx.holder["bwd"])
        if ctx.needs_input_grad[0]:
            if TRAIN_CONV_PRECISION == "f16x2":
                # the transposed, rotated kernel is packed straight from the forward weight
                dx = K.conv2d_ring(dy, ctx.holder["bwd"], weight.detach(), None, precision="f16x2",
                                   weight_is_fwd=True, dx_of=ctx.holder["fwd"])
            else:
                wt = weight.detach().flip(2, 3).transpose(0, 1).contiguous()       # [Ci, Co, ks, ks]
                dx = K.conv2d_ring(dy, ctx.holder["bwd"], wt, None, precision=TRAIN_CONV_PRECISION)
        if need_w:
            dw = torch.empty_like(weight)
            db = torch.empty(Co, device=x.device, dtype=torch.float32) if ctx.has_bias else None
            n = int(lib().lc_conv2d_ring_wgrad_scratch_elems(B, Ci, Co, H, W, ks))
            scratch = torch.empty(n, device=x.device, dtype=torch.float32)
            with torch.cuda.device(x.device):
                st = torch.cuda.current_stream().cuda_stream
                if w_split:
                    # x's record = the snapshot taken when its forward conv measured it
                    check(lib().lc_conv2d_ring_wgrad_f16x2(
                        x.data_ptr(), _bs(x), dy.data_ptr(), _bs(dy),
                        ctx.x_rec.data_ptr(), ctx.holder["bwd"].range_ptr(x.device),
                        scratch.data_ptr(), dw.data_ptr(), None if db is None else db.data_ptr(), B, Ci,
                        Co, H, W, ks, 0, st), "lc_conv2d_ring_wgrad_f16x2")
                else:
                    check(lib().lc_conv2d_ring_wgrad(x.data_ptr(), _bs(x), dy.data_ptr(), _bs(dy),
                                                     scratch.data_ptr(), dw.data_ptr(),
                                                     None if db is None else db.data_ptr(), B, Ci, Co, H,
                                                     W, ks, 0, st), "lc_conv2d_ring_wgrad")
        return dx, dw, db, None, None, d_res, None


def conv(module, x, res=None, out_scale=1.0):
    """Differentiable call of an ops.Conv2d / PointwiseConv1d-like module (weight [Co,Ci,k,k]):
    (conv(x) + bias [+ res]) * out_scale."""
    holder = module.__dict__.get("_train_packed")
    if holder is None:
        holder = {"fwd": K.PackedConv("train.fwd"), "bwd": K.PackedConv("train.bwd")}
        module.__dict__["_train_packed"] = holder
    w = module.weight if module.weight.dim() == 4 else module.weight[:, :, :, None]
    if MULTI_WEIGHT_PACK and TRAIN_CONV_PRECISION == "f16x2" and w.requires_grad:
        K.train_weight_plan(w.device).register(w, holder)
    return ConvRing.apply(x, w, module.bias, holder, _amax_of(x) if PRODUCER_AMAX else None, res, out_scale)


class FlashAttention(torch.autograd.Function):
    """o[b,h,c,t] = sum_s softmax_s(scale * sum_c' q[b,h,c',t] k[b,h,c',s]) v[b,h,c,s] for channel-major operands
    q [B,h,dqk,Lq], k [B,h,dqk,Lk], v [B,h,dv,Lk] (dqk, dv <= 64).  nn.MultiheadAttention of SelfAttentionBlock
    (efficient_unet.py:28-58) and ObjectAwareCrossAttention.forward (layout_unet_v1.py:489-506) with the content /
    positional channels and the image / layout keys concatenated by the caller."""

    @staticmethod
    def forward(ctx, q, k, v, scale):
        q, k, v = q.contiguous().float(), k.contiguous().float(), v.contiguous().float()
        B, h, dqk, Lq = q.shape
        Lk, dv = k.shape[-1], v.shape[2]
        o = torch.empty((B, h, dv, Lq), device=q.device, dtype=torch.float32)
        lse = torch.empty((B * h, Lq), device=q.device, dtype=torch.float32)
        # max |q|, |k|, |v| measured on the device by the forward entry; both passes derive the pre-scales of their fp16
        # operand splits from these three words (csrc/attention_pre.h): no host synchronisation, any operand magnitude
        amax = torch.empty(3, device=q.device, dtype=torch.float32) if TRAIN_ATTN_FWD_PRECISION == "f16x2" else None
        with torch.cuda.device(q.device):
            check(lib().lc_attention_train_fwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), lse.data_ptr(),
                                               B * h, Lq, Lk, dqk, dv, float(scale),
                                               1 if TRAIN_ATTN_FWD_PRECISION == "f16x2" else 0,
                                               None if amax is None else amax.data_ptr(),
                                               torch.cuda.current_stream().cuda_stream), "lc_attention_train_fwd")
        ctx.save_for_backward(q, k, v, o, lse)
        ctx.qkv_amax = amax
        ctx.scale = float(scale)
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, o, lse = ctx.saved_tensors
        do = do.contiguous().float()
        B, h, dqk, Lq = q.shape
        Lk, dv = k.shape[-1], v.shape[2]
        dq, dk, dvv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        scratch = torch.empty(B * h * Lq + 1, device=q.device, dtype=torch.float32)
        with torch.cuda.device(q.device):
            st = torch.cuda.current_stream().cuda_stream
            args = (q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), do.data_ptr(), lse.data_ptr(),
                    scratch.data_ptr(), dq.data_ptr(), dk.data_ptr(), dvv.data_ptr(), B * h, Lq, Lk, dqk, dv, ctx.scale)
            if TRAIN_ATTN_BWD_PRECISION == "f16x2":
                amax = ctx.qkv_amax
                if amax is None:           # (exact-fp32 forward + split backward: measure here, same rule)
                    amax = torch.stack([q.abs().amax(), k.abs().amax(), v.abs().amax()]).float()
                check(lib().lc_attention_bwd_f16x2(*args, amax.data_ptr(), st), "lc_attention_bwd_f16x2")
            else:
                check(lib().lc_attention_bwd(*args, st), "lc_attention_bwd")
        return dq, dk, dvv, None


def flash_attention(q, k, v, scale):
    """The training graph's attention core (see FlashAttention); torch ops when switched off or out of range."""
    if TRAIN_ATTENTION == "hip" and q.is_cuda and q.shape[2] <= 64 and v.shape[2] <= 64:
        return FlashAttention.apply(q, k, v, scale)
    w = (torch.einsum("bhct,bhcs->bhts", q, k) * scale).softmax(-1)
    return torch.einsum("bhts,bhcs->bhct", w, v)


class GroupNormAct(torch.autograd.Function):
    """y = silu?( GN(x) * gamma + beta ) * (1 + scale) + shift ) -- any of gamma/beta, scale/shift None."""

    @staticmethod
    def forward(ctx, x, gamma, beta, scale, shift, G, eps, act):
        x = _c4(x)
        B, C, H, W = x.shape
        dev = x.device
        sc = None if scale is None else scale.contiguous()
        sf = None if shift is None else shift.contiguous()
        n = int(lib().lc_groupnorm_partials_elems(B, C, H, W, G))
        part = torch.empty(n, device=dev, dtype=torch.float64)
        mr = torch.empty((B, G, 2), device=dev, dtype=torch.float32)
        y = torch.empty(x.shape, device=dev, dtype=torch.float32)
        p = lambda t: None if t is None else t.data_ptr()
        with torch.cuda.device(dev):
            st = torch.cuda.current_stream().cuda_stream
            check(lib().lc_groupnorm_stats(x.data_ptr(), _bs(x), part.data_ptr(), B, C, H, W, G, st),
                  "lc_groupnorm_stats")
            slot = _amax_slot(dev, B, C, H, W, G, False) if PRODUCER_AMAX and TRAIN_CONV_PRECISION == "f16x2" else None
            # apply pass; it also leaves the (mean, rstd) it used for the backward and the partial maxima of |y|
            check(lib().lc_groupnorm_apply_train(x.data_ptr(), _bs(x), part.data_ptr(), p(gamma), p(beta),
                                                 p(sc), p(sf), C, y.data_ptr(), C * H * W, B, C, H, W, G,
                                                 float(eps), int(act), mr.data_ptr(), p(slot), st),
                  "lc_groupnorm_apply_train")
        ctx.save_for_backward(x, mr, gamma, beta, sc, sf)
        ctx.G, ctx.act = G, act
        ctx.amax_slot = slot             # the tag itself is set on the tensor apply() returns (group_norm below)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mr, gamma, beta, sc, sf = ctx.saved_tensors
        dy = _c4(dy)
        B, C, H, W = x.shape
        rows = torch.empty((B, C, 2), device=x.device, dtype=torch.float64)
        dx = torch.empty(x.shape, device=x.device, dtype=torch.float32)
        p = lambda t: None if t is None else t.data_ptr()
        with torch.cuda.device(x.device):
            slot = _amax_slot(x.device, B, C, H, W, ctx.G, True) \
                if PRODUCER_AMAX and TRAIN_CONV_PRECISION == "f16x2" else None
            # rows + dx, and in the same launches: the small parameter gradients from the rows (dshift = r1,
            # dscale = g r3 + be r1, dbeta = sum_b (1 + sc) r1, dgamma = sum_b (1 + sc) r3; fp64 arithmetic) and the
            # partial maxima of |dx|
            dgamma = dbeta = dscale = dshift = None
            if gamma is not None:
                dgamma = torch.empty(C, device=x.device, dtype=torch.float32)
                dbeta = torch.empty(C, device=x.device, dtype=torch.float32)
            if sc is not None:
                dscale = torch.empty((B, C), device=x.device, dtype=torch.float32)
                dshift = torch.empty((B, C), device=x.device, dtype=torch.float32)
            check(lib().lc_groupnorm_bwd_train(x.data_ptr(), _bs(x), dy.data_ptr(), _bs(dy), mr.data_ptr(),
                                               p(gamma), p(beta), p(sc), p(sf), C, rows.data_ptr(),
                                               dx.data_ptr(), C * H * W, p(dgamma), p(dbeta), p(dscale), p(dshift),
                                               B, C, H, W, ctx.G, int(ctx.act), p(slot),
                                               torch.cuda.current_stream().cuda_stream), "lc_groupnorm_bwd_train")
            if slot is not None:
                _tag_amax(dx, slot)      # dx is the dY of the conv that produced x (when nothing else consumed x)
            if beta is None:
                dbeta = None
            if sf is None:
                dshift = None
        return dx, dgamma, dbeta, dscale, dshift, None, None, None


def group_norm(module, x, act=False, scale=None, shift=None):
    gamma = getattr(module, "weight", None)
    beta = getattr(module, "bias", None)
    y = GroupNormAct.apply(x, gamma, beta, scale, shift, module.num_groups, module.eps, act)
    fn = y.grad_fn
    slot = getattr(fn, "amax_slot", None) if fn is not None else None
    if slot is not None:
        _tag_amax(y, slot)
    return y


class Resample2x(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, up):
        ctx.up = up
        return K.resample2x(_c4(x), up=up)

    @staticmethod
    def backward(ctx, dy):
        # adjoint of the separable [1,3,3,1] FIR: the opposite direction with the same window;
        # up carries a gain of 2 per axis, so down^T = up / 4 and up^T = 4 * down
        g = K.resample2x(_c4(dy), up=not ctx.up)
        return (g * 4.0 if ctx.up else g * 0.25), None


def resample(x, up: bool):
    # |resampled| <= max|x|: every output sample is a convex combination of inputs (down: [1,3,3,1]/8 per axis; up:
    # gain 2 on the zero-stuffed signal = the phases [1,3]/4 and [3,1]/4 per axis)
    return _carry_amax(x, Resample2x.apply(x, up))


# ------------------------------------------------------------------------------------------------
def efficient_unet_forward(m, images: torch.Tensor, log_snr: torch.Tensor) -> torch.Tensor:
    """Differentiable forward of lidargen.models.unets.EfficientUNet (reference
    efficient_unet.py:274-300) on the Functions above; same parameters, same arithmetic order as the
    reference modules (GN -> SiLU -> conv -> AdaGN -> SiLU -> conv, (skip + h) / sqrt(2), ...)."""
    B = images.shape[0]
    begin_training_forward(images.device)
    if log_snr.dim() == 0:
        log_snr = log_snr[None].repeat_interleave(B, dim=0)
    te = m.time_embedding
    h = te[0](log_snr.float())                                   # sinusoid of the log-SNR (no parameters)
    temb = F.linear(F.silu(F.linear(h, te[1].weight, te[1].bias)), te[3].weight, te[3].bias)
    x = images
    if m.coords_encoding is not None:
        with torch.no_grad():
            enc = m.coords_encoding(m.coords) if not isinstance(m.coords_encoding, torch.nn.Identity) \
                else m.coords.float()
        x = torch.cat([images, enc.expand(B, -1, -1, -1)], dim=1)

    def res_block(rb, x):
        h = conv(rb.conv1, group_norm(rb.norm1, x, act=True))
        if rb.has_emb:
            ss = F.linear(F.silu(temb), rb.norm2.proj[1].weight, rb.norm2.proj[1].bias)
            C = rb.norm2.num_channels
            h = group_norm(rb.norm2, h, act=True, scale=ss[:, :C], shift=ss[:, C:])
        else:
            h = group_norm(rb.norm2, h, act=True)
        sk = x if isinstance(rb.skip, torch.nn.Identity) else conv(rb.skip, x)
        return conv(rb.conv2, h, res=sk, out_scale=rb._scale_f)          # (skip + h) / sqrt(2) in the epilogue

    def attn_block(sa, x):
        B_, C, H, W = x.shape
        heads = sa.attn.num_heads
        q = conv(_LinearAsConv(sa.attn.in_proj_weight, sa.attn.in_proj_bias, sa, "in"),
                 group_norm(sa.norm, x))
        q = q.view(B_, 3, heads, C // heads, H * W)
        o = flash_attention(q[:, 0], q[:, 1], q[:, 2], (C // heads) ** -0.5).reshape(B_, C, H, W)
        return conv(_LinearAsConv(sa.attn.out_proj.weight, sa.attn.out_proj.bias, sa, "out"), o, res=x,
                    out_scale=sa._scale_f)

    def block(blk, h):
        if not isinstance(blk.downsample, torch.nn.Identity):
            h = resample(conv(blk.downsample[0], h), up=False)
        for rb in blk.residual_blocks:
            h = res_block(rb, h)
        if not isinstance(blk.self_attn_block, torch.nn.Identity):
            h = attn_block(blk.self_attn_block, h)
        if not isinstance(blk.upsample, torch.nn.Identity):
            h = conv(blk.upsample[1], resample(h, up=True))
        return h

    h = conv(m.in_conv, x)
    h1 = block(m.d_block1, h)
    h2 = block(m.d_block2, h1)
    h3 = block(m.d_block3, h2)
    h4 = block(m.d_block4, h3)
    h = block(m.u_block4, h4)
    h = block(m.u_block3, torch.cat([h, h3], dim=1))
    h = block(m.u_block2, torch.cat([h, h2], dim=1))
    h = block(m.u_block1, torch.cat([h, h1], dim=1))
    return conv(m.out_conv, h)


class _LinearAsConv:
    """A [N, K] dense weight seen as a 1x1 conv of channel-major tokens (the MHA projections)."""

    def __init__(self, weight, bias, owner, tag):
        self.weight, self.bias = weight[:, :, None, None], bias
        self.__dict__["_train_packed"] = owner.__dict__.setdefault(
            "_train_packed_" + tag, {"fwd": K.PackedConv("train.fwd"), "bwd": K.PackedConv("train.bwd")})


# ------------------------------------------------------------------------------------------------
def _tok(x3):
    """[B, C, L] tokens as the [B, C, 1, L] image the conv / GroupNorm Functions take."""
    return _carry_amax(x3, x3.unsqueeze(2))


def _gn_tok(module, x3, **kw):
    y = group_norm(module, _tok(x3), **kw)
    return _carry_amax(y, y.squeeze(2))


def _conv_tok(module, x3):
    """1x1 projection of channel-major tokens: the MFMA conv for image-sized token sets, one dense
    product for the 13-token layout operands (a few kFLOP; NOT F.conv1d -- MIOpen answers that shape with
    a Winograd forward and a naive backward kernel of 0.36 ms per call, 6 ms of a C3 training step)."""
    if x3.shape[-1] < 64:
        y = torch.matmul(module.weight[:, :, 0], x3)
        return y if module.bias is None else y + module.bias[:, None]
    return conv(module, _tok(x3)).squeeze(2)


def layout_unet_v1_forward(m, x: torch.Tensor, cond_dict: dict) -> torch.Tensor:
    """Differentiable forward of lidargen.models.unets.LayoutUnetV1 (reference layout_unet_v1.py:
    ResBlock :143-249, ObjectAwareCrossAttention :416-532, forward :866-902) -- the graph
    `accelerator.backward(loss)` of tools/train/train_lidm_cond.py:259-322 differentiates, including
    the path into the layout encoder's outputs (xf_proj, xf_out, class / box embeddings).  Ring
    convolutions, GroupNorm(+scale/shift)(+SiLU), FIR resampling and the image-token projections are
    the HIP Functions above; the joint image|layout attention core and the 13-token layout
    operands are differentiable torch ops on the device."""
    from lidarcrafter_amd.lidargen.models.unets import layout_unet_v1 as L

    for mod in m.modules():
        if isinstance(mod, L.ObjectAwareCrossAttention) and (
                mod.norm_first or mod.use_key_padding_mask or mod.norm_for_obj_embedding is not None or
                mod.channels_scale_for_positional_embedding != 1.0):
            raise NotImplementedError("the training graph covers the shipped ObjectAwareCrossAttention configuration "
                                      "(norm_first=False, no key padding mask, positional channels = channels); "
                                      "the inference forward implements the other constructor options")
    lay = cond_dict["other_condition"]
    B, _, H, W = x.shape
    begin_training_forward(x.device)
    t = cond_dict["time_condition"]
    if t.dim() == 0:
        t = t[None].repeat_interleave(B, dim=0)
    te = m.time_embed
    emb = F.linear(F.silu(F.linear(te[0](t.float()), te[1].weight, te[1].bias)), te[3].weight,
                   te[3].bias) + lay["xf_proj"].float()
    with torch.no_grad():
        enc = m.coords_encoding(m.coords)
    parts = [x.float()]
    if "concat_cond" in lay:
        parts.append(lay["concat_cond"].float())
    parts.append(enc.expand(B, -1, -1, -1))
    h = torch.cat(parts, dim=1)

    def res_block(rb, x):
        a = group_norm(rb.in_layers[0], x, act=True)
        if rb.updown:
            up = rb.op.up == 2
            a, x = resample(a, up), resample(x, up)
        h = conv(rb.in_layers[2], a)
        e = F.linear(F.silu(emb), rb.emb_layers[1].weight, rb.emb_layers[1].bias)
        C = rb.out_channels
        h = group_norm(rb.out_layers[0], h, act=True, scale=e[:, :C], shift=e[:, C:])
        h = dropout(h, rb.dropout, rb.training)
        sk = x if isinstance(rb.skip_connection, torch.nn.Identity) else conv(rb.skip_connection, x)
        return conv(rb.out_layers[3], h, res=sk)

    def attention(at, x):
        B_, C, H_, W_ = x.shape
        L1 = H_ * W_
        heads = at.num_heads
        d = C // heads
        xs = x.reshape(B_, C, L1)
        qkv = _conv_tok(at.qkv_projector, _gn_tok(at.norm_for_qkv, xs))
        img = lay[f"image_patch_bbox_embedding_for_resolution{at.resolution}"].float()
        pos_img = _gn_tok(at.norm_for_image_patch_positional_embedding,
                          _conv_tok(at.layout_position_embedding_projector, img.contiguous()))
        pos_lay = _gn_tok(at.norm_for_layout_positional_embedding,
                          _conv_tok(at.layout_position_embedding_projector, lay["obj_bbox_embedding"].float()))
        content = (lay["xf_out"].float() + _gn_tok(at.norm_for_obj_class_embedding,
                                                   lay["obj_class_embedding"].float())) * 0.5
        kv = _conv_tok(at.layout_content_embedding_projector, content)
        hv = lambda t_: t_.reshape(B_, heads, d, -1)
        q, k, v = (hv(t_) for t_ in qkv.split(C, dim=1))
        pi, pl, kl, vl = hv(pos_img), hv(pos_lay), hv(kv[:, :C]), hv(kv[:, C:])
        s2 = 1.0 / (2 * d) ** 0.5                     # (q s)(k s) with s = (2d)^-1/4
        # the concatenated operands of layout_unet_v1.py:453-454,476-480: channels = content ++ positional,
        # keys = image tokens ++ the 13 layout tokens
        a = flash_attention(torch.cat([q, pi], dim=2),
                            torch.cat([torch.cat([k, pi], dim=2), torch.cat([kl, pl], dim=2)], dim=3),
                            torch.cat([v, vl], dim=3), s2)
        # x + proj_out(attention): the residual add in the projection's epilogue
        return conv(at.proj_out, a.reshape(B_, C, H_, W_), res=x)

    def seq(blk, h):
        for layer in blk:
            if isinstance(layer, L.ResBlock):
                h = res_block(layer, h)
            elif isinstance(layer, L.ObjectAwareCrossAttention):
                h = attention(layer, h)
            else:
                h = conv(layer, h)
        return h

    hs = []
    for blk in m.input_blocks:
        h = seq(blk, h)
        hs.append(h)
    h = seq(m.middle_block, h)
    for blk in m.output_blocks:
        h = seq(blk, torch.cat([h, hs.pop()], dim=1))
    return conv(m.out[2], group_norm(m.out[0], h, act=True))
