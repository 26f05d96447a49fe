"""Tensor-level wrappers over the C ABI (include/lidarcrafter_hip.h).

PyTorch is used here only as the owner of device memory and streams: every function below takes
torch CUDA tensors, checks layout, and hands raw device pointers + sizes + the current HIP stream
to liblidarcrafter_hip.so.  No function has a CPU / eager-PyTorch fallback: a non-CUDA tensor or a
missing library raises.
"""
from __future__ import annotations

from typing import Optional

import torch

from ._lib import check, lib

_F32 = torch.float32
_HALF = (torch.float16, torch.bfloat16)


def _stream() -> int:
    # the current stream of the CURRENT device: every public op runs under `_entry`, which makes
    # the device of its tensors current for the duration of the call
    return torch.cuda.current_stream().cuda_stream


def _req(t: torch.Tensor, name: str) -> None:
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"lidarcrafter_amd.ops: `{name}` must be a CUDA(HIP) tensor -- the hot "
                           "path has no CPU fallback (oracle/ holds the CPU restatement for tests)")
    if t.dtype != _F32:
        raise TypeError(f"`{name}` must be float32, got {t.dtype} (float16 / bfloat16 INPUTS are "
                        "up-cast at the op boundary, e.g. under torch.autocast; `out=` tensors and "
                        "other dtypes are not)")


def _entry(fn):
    """Boundary of every public op:
      * device: the kernels are launched on the current stream of the device the tensors live on
        (the reference's modules work on any device index); when that is not the current device it
        is made current for the call (torch.cuda.device), so `setup_model(device='cuda:1')` works
        without torch.cuda.set_device;
      * dtype: the kernels compute in fp32.  float16 / bfloat16 CUDA tensor inputs -- what
        torch.autocast makes of the layout encoder's nn.Linear outputs in the reference's bulk
        harness (tools/evaluation/sample_and_save_cond.py:64,145) -- are up-cast here; the result is
        float32 (an `out=` tensor must be float32)."""
    import functools

    @functools.wraps(fn)
    def wrapper(*args, **kw):
        dev = None
        cast = False
        for a in args:
            if isinstance(a, torch.Tensor) and a.is_cuda:
                if dev is None:
                    dev = a.device
                if a.dtype in _HALF:
                    cast = True
        for k, a in kw.items():
            if isinstance(a, torch.Tensor) and a.is_cuda:
                if dev is None:
                    dev = a.device
                if a.dtype in _HALF and k != "out":
                    cast = True
        if cast:
            args = tuple(a.float() if isinstance(a, torch.Tensor) and a.is_cuda and a.dtype in _HALF
                         else a for a in args)
            kw = {k: (a.float() if isinstance(a, torch.Tensor) and a.is_cuda and a.dtype in _HALF
                      and k != "out" else a) for k, a in kw.items()}
        if dev is None or dev.index is None or dev.index == torch.cuda.current_device():
            return fn(*args, **kw)
        with torch.cuda.device(dev):
            return fn(*args, **kw)

    wrapper._lc_entry = True
    return wrapper


def _bs4(t: torch.Tensor, name: str) -> int:
    """Batch stride (elements) of a [B,C,H,W] tensor whose inner [C,H,W] block is contiguous."""
    _req(t, name)
    if t.dim() != 4:
        raise ValueError(f"`{name}` must be [B,C,H,W], got {tuple(t.shape)}")
    B, C, H, W = t.shape
    st = t.stride()
    inner_ok = (W == 1 or st[3] == 1) and (H == 1 or st[2] == W) and (C == 1 or st[1] == H * W)
    if not inner_ok:
        raise ValueError(f"`{name}`: inner [C,H,W] block must be contiguous, strides={st}")
    return st[0] if B > 1 else C * H * W


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


# Optional per-launch timing with HIP events on the launch stream (bench.py's roofline leg sets
# PROFILE to a list; entries are (kernel_family, algorithmic_work, start_event, end_event, bytes_read, bytes_written)).
PROFILE = None


class _Timed:
    __slots__ = ("name", "work", "e0", "rd", "wr", "executed")

    def __init__(self, name, work, rd=0.0, wr=0.0, executed=None):
        # work: ALGORITHMIC flops -- what the reference's operator needs for this output (bench.py's roofline contract);
        # executed: the flops the launch really issues where that is less (the folded down-sampling conv: a quarter);
        # rd / wr: algorithmic bytes the launch must read / write once (inputs + packed weights + residual; output)
        self.name, self.work, self.e0, self.rd, self.wr = name, work, None, rd, wr
        self.executed = work if executed is None else executed

    def __enter__(self):
        if PROFILE is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if self.e0 is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            PROFILE.append((self.name, self.work, self.e0, e1, self.rd, self.wr, self.executed))
        return False


# ------------------------------------------------------------------------------------ conv
import os as _os

# Arithmetic of the convolution kernels:
#   "f32"   : v_mfma_f32_32x32x2_f32, exact fp32 (== fmaf chain), 157 TFLOP/s peak
#   "f16x2" : hi/lo fp16 split, 3 x v_mfma_f32_32x32x16_f16 per block, ~5e-7 relative per product
CONV_PRECISION = _os.environ.get("LC_CONV_PRECISION", "f16x2")


# Fuse GroupNorm(+AdaGN)+SiLU into the consuming convolution's staging pass (f16x2 kernels):
# the normalised/activated tensor is never written to HBM (saves one read + one write per GN).
FUSE_GN = _os.environ.get("LC_FUSE_GN", "1") != "0"


def fuse_gn(out_channels: int = 0) -> bool:
    """Fuse only where it pays (devtools/gn_fuse_sweep.py on MI355X): every 64-output-channel
    block of the conv re-applies the activation to its input tile, so the fused form wins for
    Co <= 64 (-4...-9 %) and loses for wider layers (+2...+16 %)."""
    return FUSE_GN and CONV_PRECISION == "f16x2" and out_channels <= FUSE_GN_MAX_CO


FUSE_GN_MAX_CO = int(_os.environ.get("LC_FUSE_GN_MAX_CO", "64"))


# Products per multiply of the f16x2 convolutions: 3 = the split (fp32-class accuracy, the product path and
# every parity claim); 1 = ONE fp16 product with fp32 accumulation (operands rounded to 11 bits) -- what
# fp16 autocast computes in the reference.  Opt-in only: set_conv_products(1), or LC_AUTOCAST_SINGLE_PRODUCT=1
# to take it while (and only while) the caller runs under torch.autocast(dtype=float16), as the reference's
# bulk harness does (tools/evaluation/sample_and_save_cond.py:64,145).  Frames of a 50-step run differ from the
# fp32 reference by ~1.4e-2 in this mode (profiles/r02_passes_error.json): it is NOT the benchmarked path.
_CONV_PRODUCTS = 1 if _os.environ.get("LC_CONV_PRODUCTS") == "1" else 3
AUTOCAST_SINGLE_PRODUCT = _os.environ.get("LC_AUTOCAST_SINGLE_PRODUCT", "0") == "1"


def set_conv_products(n: int) -> int:
    global _CONV_PRODUCTS
    if n not in (1, 3):
        raise ValueError("conv products: 1 or 3")
    old, _CONV_PRODUCTS = _CONV_PRODUCTS, n
    return old


def conv_products() -> int:
    if _CONV_PRODUCTS == 1:
        return 1
    if AUTOCAST_SINGLE_PRODUCT and torch.is_autocast_enabled() and \
            torch.get_autocast_gpu_dtype() == torch.float16:
        return 1
    return 3


def _conv_lib():
    if conv_products() == 1:
        from ._lib import lib_p1
        return lib_p1()
    return lib()


def set_conv_precision(mode: str) -> str:
    global CONV_PRECISION
    if mode not in ("f32", "f16x2"):
        raise ValueError(mode)
    old, CONV_PRECISION = CONV_PRECISION, mode
    return old


# What a captured denoising step depends on besides its tensors: every routing switch of this module (the upper-case
# scalars: precision, fusion thresholds, statistics routes, fold / split-K switches ...), the effective product count
# (autocast included) and an epoch that moves whenever a packed weight is rebuilt, a range slot is (re)assigned or a
# pre-scale is moved by the poll.  A sampler may replay a graph captured by an EARLIER run only under an equal signature
# (lidargen/models/diffusion/continuous_time.py::_graph_key).
_EPOCH = 0


def bump_epoch() -> None:
    global _EPOCH
    _EPOCH += 1


def route_signature() -> tuple:
    g = globals()
    flags = tuple((k, g[k]) for k in sorted(g) if k.isupper() and k != "_EPOCH"
                  and isinstance(g[k], (bool, int, float, str, tuple, type(None))))
    return flags + (conv_products(), _EPOCH)


# ------------------------------------------------------------------------------------ GN stats
# Producer-side GroupNorm statistics: the pipelined f16x2 conv can emit, per octet of output
# channels and wave tile, the shifted sums of what it stores (lc_conv2d_ring_f16x2_fwd
# gn_ostats_out); a following `groupnorm(...)` then needs no statistics pass over the tensor.  The
# handle travels as an attribute of the OUTPUT TENSOR OBJECT (for a channel slice of a concat
# buffer: of the base buffer, keyed by channel range), so a different tensor that later reuses the
# same memory can never pick up stale statistics, and every wrapper that writes into an `out=`
# tensor drops the statistics of what it overwrites.
# Default ON again since round 4.  (End of round 3 it was switched off: the entry was ONE 128-bit buffer store with
# an SGPR soffset, and ~3 entries in 10^7 arrived with a foreign upper dword under the store pressure of the deferred
# epilogue.  The entry is now ONE 32-bit store instruction in which the four lanes below the reducing lane carry one field
# each (DefEpi::store_entry_lanes: the sum reaches lane R-1 by a DPP row_shl, pivot and count are uniform over lanes
# R-3 .. R); devtools/entry_stress.py recomputes every entry of 10^8 per entry unit: 0 off, against 17 in 5e7 for the
# old form on the same box -- profiles/r04_entry_store.txt.  Regression guard in the GPU suite:
# tests/test_hip_parity.py::test_conv_statistics_entries_stress.)
# LC_GN_PRODUCER_STATS=0 selects the statistics-pass route (one lc_groupnorm_stats launch per GroupNorm).
PRODUCER_GN_STATS = _os.environ.get("LC_GN_PRODUCER_STATS", "1") != "0"
# ... and the x2 down-sampler's per-channel entries (round 5); "0": a statistics pass behind every Resample(down=2)
RESAMPLE_STATS = _os.environ.get("LC_RESAMPLE_STATS", "1") != "0"


class _OctStatsHandle:
    __slots__ = ("buf", "channels", "slots", "shape", "unit")

    def __init__(self, buf, channels, slots, shape, unit=8):
        self.buf, self.channels, self.slots, self.shape, self.unit = buf, channels, slots, shape, unit


def _dense_plane(t: torch.Tensor) -> bool:
    H, W = t.shape[2], t.shape[3]
    st = t.stride()
    return (W == 1 or st[3] == 1) and (H == 1 or st[2] == W) and st[1] == H * W


def alias(view: torch.Tensor, src: torch.Tensor, c0: int = 0) -> torch.Tensor:
    """Declare `view` an alias of channels [c0, ...) of `src` (same memory; the plane may be re-indexed,
    e.g. [B,C,H,W] <-> [B,C,1,H*W] <-> [B,C,H*W]) so that producer statistics follow it.  Needed
    because torch.inference_mode (the samplers) does not record `_base` on views; a no-op for
    correctness otherwise.  Returns `view`."""
    own, b0 = _stats_owner(src)
    if own is not None:
        view._lc_owner = (own, b0 + c0)
    return view


def chan_slice(buf: torch.Tensor, c0: int, c1: int) -> torch.Tensor:
    """buf[:, c0:c1] of a [B,C,H,W] buffer, statistics bookkept on `buf` in every grad mode."""
    return alias(buf[:, c0:c1], buf, c0)


def _stats_owner(t: torch.Tensor):
    """(owner tensor object, first channel of `t` inside it), or (None, 0) when `t` is not a plain
    channel slice of its base."""
    o = getattr(t, "_lc_owner", None)
    if o is not None:
        return o
    if t.dim() != 4:
        return None, 0
    base = t._base
    if base is None:
        return t, 0
    # a channel slice, possibly with the plane re-indexed ([B,C,H,W] <-> [B,C,1,H*W] token views of the
    # attention blocks): the statistics are sums over the pixels of a channel octet, whatever the
    # plane's shape -- what must agree are the batch / channel strides and the (dense) plane size
    if base.dim() != 4 or t.dim() != 4 or base.shape[0] != t.shape[0] or \
            base.stride()[:2] != t.stride()[:2] or \
            base.shape[2] * base.shape[3] != t.shape[2] * t.shape[3] or \
            not _dense_plane(base) or not _dense_plane(t):
        return None, 0
    off = t.storage_offset() - base.storage_offset()
    if off < 0 or off % base.stride(1):
        return None, 0
    return base, off // base.stride(1)


def _drop_stats(t: Optional[torch.Tensor]) -> None:
    """`t` is about to be (over)written: forget the statistics of that channel range."""
    if t is None:
        return
    own, c0 = _stats_owner(t)
    d = getattr(t, "_lc_gnstats", None)
    if d and own is not t:
        d.clear()
    d = getattr(own, "_lc_gnstats", None) if own is not None else None
    if d:
        C = t.shape[1] if t.dim() == 4 else 1 << 30
        for k in [k for k in d if k[0] < c0 + C and c0 < k[0] + k[1]]:
            del d[k]


def _attach_stats(out: torch.Tensor, h: _OctStatsHandle) -> None:
    own, c0 = _stats_owner(out)
    if own is None:
        return
    d = getattr(own, "_lc_gnstats", None)
    if d is None:
        d = {}
        own._lc_gnstats = d
    d[(c0, out.shape[1])] = h


GN_TRACE = None   # developer aid: a collections.Counter of (shape, G, found) per statistics lookup


def _find_stats(x: torch.Tensor, G: int, octet_groups: bool = False):
    """Every consumer folds entries of any unit (8 / 4 / 2 / 1 channels per entry: round 5) as long as a group is a
    whole number of entries.  octet_groups: the consumer is the pre-split apply pass (one block = one channel octet):
    groups of whole octets, or 2 / 4 channels per group (one group per wave)."""
    if octet_groups and (x.dim() != 4 or ((x.shape[1] // G) % 8 and x.shape[1] // G not in (2, 4))):
        r = None
    else:
        r = _find_stats_impl(x, G)
        if octet_groups and r is not None and len(r) == 2 and r[0].channels % 8:
            r = None         # (the pre-split pass needs the segment boundary on a channel octet: lc_groupnorm_apply_os_split)
    if GN_TRACE is not None:
        why = ""
        if r is None:        # what the tensor does carry (developer trace only)
            own, c0 = _stats_owner(x) if x.dim() == 4 else (None, 0)
            d = getattr(own, "_lc_gnstats", None) if own is not None else None
            why = "none" if not d else ",".join(f"{k0 - c0}+{kc}/u{h.unit}" for (k0, kc), h in sorted(d.items()))
        GN_TRACE[(tuple(x.shape), G, r is not None, why)] += 1
    return r


def _find_stats_impl(x: torch.Tensor, G: int):
    """Handles covering all channels of x with at most two segments (each a whole number of
    groups, groups whole entries), or None."""
    if x.dim() != 4:
        return None
    own, c0 = _stats_owner(x)
    if own is None:
        return None
    d = getattr(own, "_lc_gnstats", None)
    if not d:
        return None
    C = x.shape[1]
    if C % G:
        return None
    cpg = C // G

    def usable(h):
        return h.shape == shape and cpg % h.unit == 0

    shape = (x.shape[0], x.shape[2] * x.shape[3])
    h = d.get((c0, C))
    if h is not None and usable(h):
        return (h,)
    for (k0, kc), h0 in d.items():
        if k0 == c0 and kc < C and kc % cpg == 0:
            h1 = d.get((c0 + kc, C - kc))
            if h1 is not None and usable(h0) and usable(h1):
                return (h0, h1)
    return None


# ------------------------------------------------------------------------------------ range safety
# The f16x2 kernels multiply every conv input by a per-layer power of two (x_scale, default 16)
# before the fp16 hi/lo split; fp16 saturates at 65504.  Each layer owns a 16-byte `lc_conv_range`
# record in device memory (one arena per device, so a poll is ONE device->host copy): the kernels
# read x_scale from it and atomically publish max |x * x_scale| of everything they staged.
#   range_poll(device)  reads all records (synchronises), resets the maxima and re-derives x_scale
#                       of every layer whose scaled maximum left the safe window; it returns the
#                       layers whose results since the last poll are INVALID (operands saturated,
#                       or so small that the lo halves were lost) -- the caller must recompute.
# The model forwards (`range_checked`) and the samplers poll after each run and recompute until
# clean: a result computed from clipped operands is never returned.  Inside a sampling loop /
# graph capture the check is deferred to the end of the run (`defer_range_checks`).
RANGE_SLOTS = 4096
X_SCALE_DEFAULT = 16.0
_A_TARGET_LOG2 = 12            # recalibration puts max|x| * x_scale into [2^12, 2^13)
_A_INVALID_HI = 65504.0 * (1.0 - 2.0 ** -11)   # hi half saturates
_A_RECAL_HI = 2.0 ** 15        # within 2x of saturation: move the scale (results still exact)
_A_INVALID_LO = 2.0 ** -3      # error relative to max|x| exceeds the split's own 2^-22
_A_RECAL_LO = 2.0 ** 2
_range_arenas = {}
import threading as _threading


class _Defer(_threading.local):
    depth = 0
    autograd_route = False      # set by autograd.begin_training_forward when a forward ENTERS the training graph


_range_defer = _Defer()          # per thread: a sampler thread's deferral does not silence another's


class ConvRangeError(ArithmeticError):
    """The f16x2 convolution could not find a pre-scale under which its input fits fp16 (e.g. inf /
    NaN activations).  Use LC_CONV_PRECISION=f32 (exact fp32 MFMA kernel) for this model."""


class _RangeArena:
    def __init__(self, device):
        self.device = device
        # (a NORMAL tensor even when the first conv runs under torch.inference_mode(): the poll
        #  updates it in place from any mode)
        with torch.inference_mode(False):
            init = torch.zeros((RANGE_SLOTS, 4), dtype=_F32)
            init[:, 0], init[:, 1] = X_SCALE_DEFAULT, 1.0 / X_SCALE_DEFAULT
            self.buf = init.to(device)
        self.scale = [X_SCALE_DEFAULT] * RANGE_SLOTS      # host mirror of x_scale
        # slots whose x_scale is (re)written ON THE DEVICE (lc_range_from_tensor, the training
        # graph): the host mirror is meaningless for them and the poll leaves them alone
        self.device_managed = set()
        self.free = list(range(RANGE_SLOTS - 1, -1, -1))
        self.owner = {}                                    # slot -> weakref to the owning PackedConv
        with torch.inference_mode(False):
            self.host = torch.empty((RANGE_SLOTS, 4), dtype=_F32)
            if device.type == "cuda":
                self.host = self.host.pin_memory()

    def alloc(self, owner) -> int:
        import weakref

        if not self.free:
            raise RuntimeError("range arena exhausted (more than 4096 live conv layers)")
        slot = self.free.pop()
        bump_epoch()
        self.owner[slot] = weakref.ref(owner)
        self.device_managed.discard(slot)
        # a recycled slot starts clean: a running maximum left by its previous owner would hide
        # the new layer's (smaller) values from the poll
        self.scale[slot] = X_SCALE_DEFAULT
        with torch.cuda.device(self.device), torch.inference_mode(False):
            self.buf[slot] = torch.tensor([X_SCALE_DEFAULT, 1.0 / X_SCALE_DEFAULT, 0.0, 0.0], dtype=_F32)
        return slot

    def release(self, slot: int) -> None:
        self.owner.pop(slot, None)
        self.free.append(slot)                   # (re-initialised by the next alloc)

    def _set_scale(self, slot: int, scale: float) -> None:
        bump_epoch()
        self.scale[slot] = scale
        self.buf[slot, :2] = torch.tensor([scale, 1.0 / scale], dtype=_F32)

    def ptr(self, slot: int) -> int:
        return self.buf.data_ptr() + 16 * slot


def _norm_dev(device) -> torch.device:
    device = torch.device("cuda") if device is None else torch.device(device)
    if device.type == "cuda" and device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    return device


def _arena(device) -> _RangeArena:
    device = _norm_dev(device)
    a = _range_arenas.get(device)
    if a is None:
        a = _range_arenas[device] = _RangeArena(device)
    return a


class RangeEvent:
    __slots__ = ("layer", "amax", "old_scale", "new_scale", "invalid")

    def __init__(self, layer, amax, old_scale, new_scale, invalid):
        self.layer, self.amax, self.old_scale, self.new_scale, self.invalid = \
            layer, amax, old_scale, new_scale, invalid

    def __repr__(self):
        return (f"RangeEvent({self.layer!r}: max|x|={self.amax:.4g}, x_scale {self.old_scale:g} -> "
                f"{self.new_scale:g}, {'INVALID' if self.invalid else 'recalibrated'})")


def _scale_for(amax: float) -> float:
    import math

    if not (amax > 0.0) or not math.isfinite(amax):
        return X_SCALE_DEFAULT
    m, e = math.frexp(amax)                 # amax = m * 2^e, m in [0.5, 1)
    k = max(-100, min(100, _A_TARGET_LOG2 + 1 - e))
    return math.ldexp(1.0, k)               # amax * scale in [2^12, 2^13)


def range_poll(device=None, quiet: bool = True):
    """Read the range records of every f16x2 conv layer on `device` (one device->host copy; this
    SYNCHRONISES the current stream), reset the running maxima and move the pre-scale of every layer
    whose scaled maximum left the safe window.  Returns the list of RangeEvents with
    `invalid=True` -- layers whose outputs since the previous poll were computed from saturated
    (or vanishing) fp16 operands and must be recomputed; [] means everything handed out is good."""
    import math
    import warnings

    if not torch.cuda.is_available():
        return []
    a = _range_arenas.get(_norm_dev(device))
    if a is None:
        return []
    with torch.cuda.device(a.device):
        a.host.copy_(a.buf, non_blocking=False)
        a.buf[:, 2].zero_()
    am = a.host[:, 2]
    bad = []
    hot = torch.nonzero((am >= _A_RECAL_HI) | ((am > 0) & (am < _A_RECAL_LO)) | ~torch.isfinite(am))
    for slot in hot.flatten().tolist():
        ref = a.owner.get(slot)
        owner = ref() if ref is not None else None
        if owner is None or slot in a.device_managed:
            continue
        A = float(am[slot])
        old = a.scale[slot]
        amax = A / old if math.isfinite(A) else float("inf")
        invalid = (not math.isfinite(A)) or A >= _A_INVALID_HI or A < _A_INVALID_LO
        new = _scale_for(amax)
        if new != old:
            with torch.cuda.device(a.device):
                a._set_scale(slot, new)
        ev = RangeEvent(owner.name or f"conv#{slot}", amax, old, new, invalid)
        if invalid:
            bad.append(ev)
        if not quiet or invalid:
            warnings.warn(f"lidarcrafter_amd f16x2 conv range: {ev}", RuntimeWarning, stacklevel=2)
    return bad


def name_packed_convs(model) -> None:
    """Give every PackedConv below `model` its module path (range events name the layer)."""
    for mname, m in model.named_modules():
        for attr, v in vars(m).items():
            if isinstance(v, PackedConv) and not v.name:
                v.name = f"{mname}.{attr.lstrip('_')}" if mname else attr.lstrip("_")


def rng_snapshot(rng, device):
    """State of what `GaussianDiffusion.randn` would draw from (None / Generator / list)."""
    if rng is None:
        dev = _norm_dev(device) if torch.device(device).type == "cuda" else None
        return ("global", torch.get_rng_state(),
                torch.cuda.get_rng_state(dev) if dev is not None else None, dev)
    if isinstance(rng, torch.Generator):
        return ("one", rng.get_state())
    return ("list", [g.get_state() for g in rng])


def rng_restore(rng, snap) -> None:
    if snap[0] == "global":
        torch.set_rng_state(snap[1])
        if snap[2] is not None:
            torch.cuda.set_rng_state(snap[2], snap[3])
    elif snap[0] == "one":
        rng.set_state(snap[1])
    else:
        for g, st in zip(rng, snap[1]):
            g.set_state(st)


@torch.compiler.disable
def run_range_safe(run, rng, device, what: str = "sampling run"):
    """(Opaque to Dynamo: `torch.compile(ddpm.sample)`, as the reference's bulk harness wraps it,
    traces the trivial method body and calls this eagerly -- the loop inside already replays one
    captured HIP graph per step.)
    `run()` under deferred range checks; poll afterwards; if a layer computed from saturated
    operands, its pre-scale has been moved: restore the generators and run again (bit-identical
    draws), so the caller never sees a clipped trajectory."""
    if CONV_PRECISION != "f16x2" or torch.device(device).type != "cuda":
        return run()
    range_poll(device)                        # forget what earlier, unrelated work left behind
    bad = None
    for _ in range(5):
        snap = rng_snapshot(rng, device)
        with defer_range_checks():
            out = run()
        bad = range_poll(device)
        if not bad:
            return out
        rng_restore(rng, snap)
    raise ConvRangeError(f"{what}: f16x2 conv pre-scales did not converge: {bad}")


class defer_range_checks:
    """Context: model forwards inside do not poll (sampling loops / graph capture poll once at the
    end of the run instead)."""

    def __enter__(self):
        _range_defer.depth += 1

    def __exit__(self, *exc):
        _range_defer.depth -= 1
        return False


def range_checked(forward):
    """Decorator of a denoiser's `forward`: run, poll the conv range records, recompute while any
    layer reports saturated operands (its pre-scale has been moved by the poll).  No-op for the
    exact fp32 kernels, inside `defer_range_checks()` and during stream capture."""
    import functools

    @functools.wraps(forward)
    def wrapper(self, *args, **kw):
        x = args[0] if args else None
        if _range_defer.depth > 0 or CONV_PRECISION != "f16x2" or not isinstance(x, torch.Tensor) or \
                not x.is_cuda or torch.cuda.is_current_stream_capturing():
            return forward(self, *args, **kw)
        for _ in range(5):
            _range_defer.autograd_route = False
            out = forward(self, *args, **kw)
            if _range_defer.autograd_route:
                # the forward took the training graph (autograd.begin_training_forward ran): every operand's
                # pre-scale was measured on the device right before its conv
                # (range_from_tensor), nothing can be invalid -- and a poll would be a blocking device->host copy per
                # step, a retry a second graph with new dropout masks.  A grad-mode call that ran the INFERENCE
                # kernels (precomputed time_features) is polled like any other.
                return out
            bad = range_poll(x.device)
            if not bad:
                return out
        raise ConvRangeError(f"f16x2 conv pre-scales did not converge: {bad}")

    return wrapper


# Pre-split activations (conv_f16x2_ps_kernel): the GroupNorm apply pass in front of a 3x3 conv
# writes fp16 hi / lo planes [B][2][C/8][H][W][8] (already multiplied by the consumer's x_scale)
# and the conv stages them with LDS-DMA.  LC_PRESPLIT=0 keeps the fp32 route everywhere.
PRESPLIT = _os.environ.get("LC_PRESPLIT", "1") != "0"


class SplitAct:
    """A [B, C, H, W] activation in the pre-split form, produced for ONE consumer conv (`packed`:
    the layer whose x_scale it carries).  Not a tensor: only `conv2d_ring` consumes it."""

    __slots__ = ("buf", "shape", "packed")

    def __init__(self, buf, shape, packed):
        self.buf, self.shape, self.packed = buf, tuple(shape), packed

    @property
    def device(self):
        return self.buf.device

    @property
    def is_cuda(self):
        return self.buf.is_cuda


def can_presplit(C: int, G: int) -> bool:
    return PRESPLIT and CONV_PRECISION == "f16x2" and C % 16 == 0 and G > 0 and C % G == 0


# 1x1 projections with many output channels take a pre-split input too (lc_conv1x1_f16x2_ps_fwd): the fp32-input 1x1
# kernel splits the same input tile once per 64-channel output block.  Measured ahead from 512 output channels
# (GroupNorm + projection 256 -> 768 @ 8 x 256 at batch 8: 74.9 -> 54.4 us; 512 -> 1536 @ 4 x 128: 64.8 -> 39.9 us -- the
# candidate's own harness, round 5, recorded in DESIGN.md section 9.4); LC_PS1X1_MIN_CO=0 turns the route off.
PS1X1_MIN_CO = int(_os.environ.get("LC_PS1X1_MIN_CO", "512"))


def presplit_1x1(Ci: int, Co: int, G: int) -> bool:
    """Should the GroupNorm in front of a Ci -> Co 1x1 projection write the pre-split form for it?"""
    return PS1X1_MIN_CO > 0 and Co >= PS1X1_MIN_CO and Ci % 32 == 0 and can_presplit(Ci, G)


class PackedConv:
    """Packed copies of an OIHW conv weight for the MFMA kernels (fp32 wp[tap][Ci^8][Co^64] and/or
    the f16x2 hi/lo planes + their device-derived pre-scale), rebuilt when the parameter changes;
    also owns the layer's input range record (lc_conv_range) in the device's arena."""

    __slots__ = ("wp", "wh", "wl", "wmeta", "Co", "Ci", "ks", "_key", "_w4", "name", "_slot",
                 "_arena", "__weakref__")

    def __init__(self, name: str = ""):
        self.wp = self.wh = self.wl = self.wmeta = None
        self._key = None
        self.name = name
        self._slot = self._arena = None

    # A copy (copy.deepcopy(model) -- what ema_pytorch's EMA(ddpm) in the reference trainers does --
    # or pickling, torch.save(model)) is a FRESH PackedConv: it owns no arena slot and no packed
    # caches, so it lazily allocates its own slot in the REGISTERED arena of its device and its
    # amax is polled like any other layer's.  (Copying the fields would clone the whole arena into a
    # private object `range_poll` never reads and share the slot index with the original.)
    def __deepcopy__(self, memo):
        new = PackedConv(self.name)
        memo[id(self)] = new
        return new

    def __copy__(self):
        return PackedConv(self.name)

    def __reduce__(self):
        return (PackedConv, (self.name,))

    def range_snapshot(self, device) -> torch.Tensor:
        """A private 16-byte copy of this layer's range record as it is NOW on the stream (training: the record the
        forward conv of a saved activation split with -- a second forward of the same module before the backward
        re-measures the live record, and the weight-gradient kernel must split the saved x with the scale that was
        measured for IT)."""
        self.range_ptr(device)
        return self._arena.buf[self._slot].clone()

    def range_ptr(self, device) -> int:
        if self._arena is None or self._arena.device != device:
            if self._arena is not None:
                self._arena.release(self._slot)
            self._arena = _arena(device)
            self._slot = self._arena.alloc(self)
        return self._arena.ptr(self._slot)

    @property
    def x_scale(self) -> float:
        if self._arena is None:
            return X_SCALE_DEFAULT
        if self._slot in self._arena.device_managed:      # the truth lives on the device
            return float(self._arena.buf[self._slot, 0])
        return self._arena.scale[self._slot]

    def __del__(self):
        try:
            if self._arena is not None:
                self._arena.release(self._slot)
        except Exception:
            pass

    def _refresh(self, weight: torch.Tensor):
        if isinstance(weight, torch.Tensor) and weight.dtype in _HALF:
            weight = weight.float()
        _req(weight, "weight")
        key = (weight.data_ptr(), weight._version, tuple(weight.shape))
        if key != self._key:
            w = weight.detach()
            if w.dim() == 3:  # Conv1d 1x1 [Co, Ci, 1]
                w = w.unsqueeze(-1)
            Co, Ci, kh, kw = w.shape
            if kh != kw or kh not in (1, 3):
                raise ValueError(f"only 1x1 / 3x3 kernels, got {tuple(w.shape)}")
            self.Co, self.Ci, self.ks, self._key = Co, Ci, kh, key
            self._w4 = w.contiguous()
            self.wp = self.wh = self.wl = None
            bump_epoch()

    def get(self, weight: torch.Tensor) -> torch.Tensor:
        self._refresh(weight)
        if self.wp is None:
            n = lib().lc_packed_conv_weight_elems(self.Co, self.Ci, self.ks)
            self.wp = torch.empty(n, device=weight.device, dtype=_F32)
            check(lib().lc_pack_conv_weight(self._w4.data_ptr(), self.wp.data_ptr(), self.Co,
                                            self.Ci, self.ks, _stream()), "lc_pack_conv_weight")
        return self.wp

    def get_f16x2(self, weight: torch.Tensor):
        self._refresh(weight)
        if self.wh is None:
            n = lib().lc_packed_conv_weight_f16x2_elems(self.Co, self.Ci, self.ks)
            both = torch.empty(2 * n, device=weight.device, dtype=torch.float16)
            self.wh, self.wl = both[:n], both[n:]      # ONE allocation: the LDS-DMA kernel addresses
            self.wmeta = torch.empty(4, device=weight.device, dtype=_F32)
            check(lib().lc_pack_conv_weight_f16x2(self._w4.data_ptr(), self.wh.data_ptr(),
                                                  self.wl.data_ptr(), self.Co, self.Ci, self.ks,
                                                  self.wmeta.data_ptr(), _stream()),
                  "lc_pack_conv_weight_f16x2")
        return self.wh, self.wl

    def get_f16x2_dx(self, w_fwd: torch.Tensor, fwd: Optional["PackedConv"] = None):
        """The packed weight of the INPUT-GRADIENT conv of the layer whose forward weight is `w_fwd` ([Cf_o, Cf_i, k, k]):
        this record becomes a conv with Co = Cf_i, Ci = Cf_o (lc_pack_conv_weight_f16x2_dx: the rotated / transposed
        weight is never materialised).  `fwd` = the layer's forward PackedConv: when it holds the pack of this very
        weight version, its max|w| is reused."""
        if isinstance(w_fwd, torch.Tensor) and w_fwd.dtype in _HALF:
            w_fwd = w_fwd.float()
        _req(w_fwd, "weight")
        w = w_fwd.detach()
        if w.dim() == 3:
            w = w.unsqueeze(-1)
        key = ("dx", w_fwd.data_ptr(), w_fwd._version, tuple(w.shape))
        if key != self._key or self.wh is None:
            Cfo, Cfi, kh, kw = w.shape
            if kh != kw or kh not in (1, 3):
                raise ValueError(f"only 1x1 / 3x3 kernels, got {tuple(w.shape)}")
            self.Co, self.Ci, self.ks, self._key = Cfi, Cfo, kh, key
            self._w4 = w.contiguous()
            self.wp = None
            n = lib().lc_packed_conv_weight_f16x2_elems(self.Co, self.Ci, self.ks)
            both = torch.empty(2 * n, device=w.device, dtype=torch.float16)
            self.wh, self.wl = both[:n], both[n:]
            self.wmeta = torch.empty(4, device=w.device, dtype=_F32)
            src = None
            if fwd is not None and fwd.wh is not None and fwd.wmeta is not None and \
                    fwd._key == (w_fwd.data_ptr(), w_fwd._version, tuple(w_fwd.shape)):
                src = fwd.wmeta.data_ptr()
            check(lib().lc_pack_conv_weight_f16x2_dx(self._w4.data_ptr(), self.wh.data_ptr(), self.wl.data_ptr(),
                                                     self.Co, self.Ci, self.ks, self.wmeta.data_ptr(), src,
                                                     _stream()), "lc_pack_conv_weight_f16x2_dx")
        return self.wh, self.wl


class TrainWeightPlan:
    """All conv weights of the training graph packed in THREE launches per optimizer step (lc_pack_conv_weights_f16x2_multi)
    instead of ~5 per layer: every layer registers (forward weight, its forward and input-gradient PackedConv) the first
    time it runs; from the next step on `refresh()` -- called at the start of a training forward -- notices that weights
    moved on, packs all of them into per-layer buffers that stay put, and primes the PackedConv caches, so the
    per-layer `get_f16x2` / `get_f16x2_dx` calls of that step find their packs valid.

    "Moved on" is read from the tensors' autograd VERSION COUNTERS (what optimizer steps, `copy_`, `load_state_dict` and
    every in-place op on the parameter bump).  An update that bypasses the counter -- in-place arithmetic on `p.data`,
    a raw-pointer write -- is invisible to it: call `invalidate()` (or `ops.train_weight_plan(dev).invalidate()`) after
    such an update, otherwise the forward / input-gradient packs stay stale while the weight gradient reads the live
    weights."""

    def __init__(self, device):
        self.device = device
        self.entries = {}            # id(holder) -> entry
        self.table = None            # device array of lc_weight_pack_job
        self.order = []

    @staticmethod
    def _base(w: torch.Tensor) -> torch.Tensor:
        return w._base if w._base is not None else w

    def register(self, w4: torch.Tensor, holder: dict) -> None:
        if id(holder) in self.entries or not w4.is_cuda or w4.dtype != _F32:
            return
        base = self._base(w4)
        if not base.is_contiguous() or base.data_ptr() != w4.data_ptr() or base.numel() != w4.numel():
            return                                   # a strided / partial view: the per-layer route packs it
        import weakref
        Co, Ci, ks, _ = w4.shape
        nf = int(lib().lc_packed_conv_weight_f16x2_elems(Co, Ci, ks))
        nd = int(lib().lc_packed_conv_weight_f16x2_elems(Ci, Co, ks))
        buf = torch.empty(2 * nf + 2 * nd, device=w4.device, dtype=torch.float16)
        meta = torch.zeros(8, device=w4.device, dtype=_F32)
        self.entries[id(holder)] = dict(
            base=weakref.ref(base), fwd=holder["fwd"], bwd=holder["bwd"],
            shape=(Co, Ci, ks, ks), ptr=base.data_ptr(), fh=buf[:nf], fl=buf[nf:2 * nf], dh=buf[2 * nf:2 * nf + nd],
            dl=buf[2 * nf + nd:], fmeta=meta[:4], dmeta=meta[4:], buf=buf, meta=meta, packed_version=None)
        self.table = None

    def invalidate(self) -> None:
        """Forget every pack's version: the next refresh() packs all registered weights again."""
        for e in self.entries.values():
            e["packed_version"] = None

    def refresh(self) -> int:
        """Pack every registered weight if any of them changed since its last pack; returns the number packed."""
        dead = [k for k, e in self.entries.items() if e["base"]() is None or e["base"]().data_ptr() != e["ptr"]]
        for k in dead:
            del self.entries[k]
            self.table = None
        if not self.entries:
            return 0
        if all(e["base"]()._version == e["packed_version"] for e in self.entries.values()):
            return 0
        if self.table is None:
            import struct
            self.order = list(self.entries.values())
            raw = b"".join(struct.pack("<7Q4i", e["ptr"], e["fh"].data_ptr(), e["fl"].data_ptr(), e["fmeta"].data_ptr(),
                                       e["dh"].data_ptr(), e["dl"].data_ptr(), e["dmeta"].data_ptr(),
                                       e["shape"][0], e["shape"][1], e["shape"][2], 0) for e in self.order)
            self.table = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(self.device)
        with torch.cuda.device(self.device):
            check(lib().lc_pack_conv_weights_f16x2_multi(self.table.data_ptr(), len(self.order), 1, _stream()),
                  "lc_pack_conv_weights_f16x2_multi")
        for e in self.order:
            base = e["base"]()
            Co, Ci, ks, _ = e["shape"]
            ver = base._version
            e["packed_version"] = ver
            f, b = e["fwd"], e["bwd"]
            f.Co, f.Ci, f.ks, f._key = Co, Ci, ks, (e["ptr"], ver, e["shape"])
            f._w4, f.wp, f.wh, f.wl, f.wmeta = base.detach().view(e["shape"]), None, e["fh"], e["fl"], e["fmeta"]
            b.Co, b.Ci, b.ks, b._key = Ci, Co, ks, ("dx", e["ptr"], ver, e["shape"])
            b._w4, b.wp, b.wh, b.wl, b.wmeta = f._w4, None, e["dh"], e["dl"], e["dmeta"]
        return len(self.order)


_train_weight_plans = {}


def train_weight_plan(device) -> TrainWeightPlan:
    plan = _train_weight_plans.get(device)
    if plan is None:
        plan = _train_weight_plans[device] = TrainWeightPlan(device)
    return plan


def prepare_model(module: torch.nn.Module) -> int:
    """Pay the one-time costs of `module`'s conv layers NOW instead of inside the first sampling step: pack every conv
    weight that already lives on a GPU (the packs are cached per weight version, so the first forward finds them) and
    load the library's code objects for that device (HIP loads a translation unit's kernels at its first launch).
    Returns the number of weights packed.  lidargen.utils.inference.setup_model calls it for a model set up on a GPU."""
    n = 0
    dev = None
    for m in module.modules():
        d = m.__dict__
        pairs = []
        if isinstance(d.get("_packed"), PackedConv) and isinstance(getattr(m, "weight", None), torch.Tensor):
            pairs.append((d["_packed"], m.weight))
        if isinstance(d.get("_pk_in"), PackedConv):                      # SelfAttentionBlock: MHA projections as 1x1 convs
            att = getattr(m, "attn", None)
            if att is not None and getattr(att, "in_proj_weight", None) is not None:
                pairs.append((d["_pk_in"], att.in_proj_weight[:, :, None, None]))
                pairs.append((d["_pk_out"], att.out_proj.weight[:, :, None, None]))
        for pk, w in pairs:
            if not w.is_cuda or w.dim() not in (3, 4) or w.shape[-1] not in (1, 3):
                continue
            dev = w.device
            with torch.cuda.device(dev):
                if CONV_PRECISION == "f16x2":
                    pk.get_f16x2(w)
                else:
                    pk.get(w)
            n += 1
    if dev is not None:
        # best effort: this only moves the code-object loads out of the first sampling step.  A failure here (a positive
        # return is the hipError_t of hipFuncGetAttributes, e.g. on a device the library was not built for) must not
        # abort model setup with a code that points at the warm-up -- the first real launch reports the real problem.
        with torch.cuda.device(dev):
            rc = int(lib().lc_load_code_objects())
        if rc != 0:
            import warnings

            warnings.warn(f"lc_load_code_objects: hipError {rc} while pre-loading the kernels' code objects; continuing "
                          "(they load at their first launch)")
    return n


def range_from_amax(amax: torch.Tensor, packed: PackedConv, device, bound_mult: float = 1.0) -> None:
    """`packed`'s input range record from the partial maxima of |x| the producer of x left (a contiguous fp32 tensor;
    see lc_range_from_amax) -- the training graph's GroupNorm passes do (autograd.GroupNormAct)."""
    ptr = packed.range_ptr(device)
    packed._arena.device_managed.add(packed._slot)
    check(lib().lc_range_from_amax(amax.data_ptr(), amax.numel(), float(bound_mult), ptr, _stream()),
          "lc_range_from_amax")


def range_from_tensor(x: torch.Tensor, packed: PackedConv) -> None:
    """Set `packed`'s input range record from max|x| ON THE DEVICE (lc_range_from_tensor): the next
    f16x2 conv of `x` through `packed` splits with the exact scale of this very tensor.  Used by the
    training graph, where activations / gradients change every step and a backward pass cannot be
    repeated after a host-side poll."""
    _req(x, "x")
    B = x.shape[0]
    n = x.numel() // B
    if x.dim() == 4:
        x_bs = _bs4(x, "x")                       # inner block contiguous, any batch stride
    else:
        x = x if x.is_contiguous() else x.contiguous()
        x_bs = n
    ptr = packed.range_ptr(x.device)
    packed._arena.device_managed.add(packed._slot)
    check(lib().lc_range_from_tensor(x.data_ptr(), x_bs, B, n, ptr, _stream()), "lc_range_from_tensor")


def conv2d_ring(x: torch.Tensor, packed: PackedConv, weight: torch.Tensor,
                bias: Optional[torch.Tensor] = None, res: Optional[torch.Tensor] = None,
                out: Optional[torch.Tensor] = None, out_scale: float = 1.0,
                tile_cfg: int = 0, precision: Optional[str] = None,
                gn_coeffs: Optional[torch.Tensor] = None, gn_silu: bool = True,
                emit_stats: bool = False, dx_of: Optional[PackedConv] = None,
                weight_is_fwd: bool = False) -> torch.Tensor:
    """y = (conv_ring(x', W) + bias [+ res]) * out_scale with x' = x, or -- when `gn_coeffs`
    (from `groupnorm_coeffs`) is given -- x' = silu?(GroupNorm(x)) applied on the fly while the
    input tile is staged (f16x2 kernels only).  ops.py:149-173 of the reference.

    emit_stats: True / 8 = the conv also leaves per-octet GroupNorm statistics of what it stores (2 =
    per channel pair, for a consumer GroupNorm with 2 / 4 / 6 channels per group; 4 = per channel quad, for
    4 / 12 per group: the pre-split kernel writes quads, the fp32-input kernels the finer pairs; a 1x1 launch of
    the fp32-input kernels writes octet entries only -- asked for pairs / quads it leaves NO statistics and the
    consumer runs its own statistics pass), attached to the output tensor object; a following `groupnorm` / `groupnorm_stats` of that tensor (or of a
    concat buffer whose halves both carry them) then skips its statistics pass.  Every wrapper of
    this module that writes into an `out=` tensor forgets the statistics of what it overwrites;
    a caller who modifies such a tensor with a torch in-place op must not request them (inference
    tensors carry no version counter that could catch it).

    weight_is_fwd (f16x2 only): `weight` is the FORWARD weight of the layer whose input gradient this call computes
    (x = dY); the conv runs with its transposed, rotated kernel, packed in place by `PackedConv.get_f16x2_dx`
    (dx_of = that layer's forward PackedConv, for its max|w|)."""
    if isinstance(x, SplitAct):
        return _conv2d_ring_presplit(x, packed, weight, bias, res, out, out_scale, tile_cfg,
                                     emit_stats)
    x_bs = _bs4(x, "x")
    prec = precision or CONV_PRECISION
    if gn_coeffs is not None and prec != "f16x2":
        raise ValueError("fused input GroupNorm exists for the f16x2 conv kernels only")
    if weight_is_fwd and prec != "f16x2":
        raise ValueError("weight_is_fwd exists for the f16x2 kernels only")
    if prec == "f16x2":
        wh, wl = packed.get_f16x2_dx(weight, dx_of) if weight_is_fwd else packed.get_f16x2(weight)
    else:
        wp = packed.get(weight)
    B, Ci, H, W = x.shape
    if Ci != packed.Ci:
        raise ValueError(f"conv: input has {Ci} channels, weight expects {packed.Ci}")
    Co = packed.Co
    if out is None:
        out = torch.empty((B, Co, H, W), device=x.device, dtype=_F32)
    y_bs = _bs4(out, "out")
    if tuple(out.shape) != (B, Co, H, W):
        raise ValueError(f"conv: out shape {tuple(out.shape)} != {(B, Co, H, W)}")
    r_bs = 0
    if res is not None:
        r_bs = _bs4(res, "res")
        if tuple(res.shape) != (B, Co, H, W):
            raise ValueError("conv: residual shape mismatch")
    if bias is not None:
        _req(bias, "bias")
    ks = packed.ks
    _drop_stats(out)
    with _Timed("conv3x3" if ks == 3 else "conv1x1", 2.0 * B * H * W * Co * Ci * ks * ks,
                rd=4.0 * (B * Ci * H * W + Co * Ci * ks * ks + (B * Co * H * W if res is not None else 0)),
                wr=4.0 * B * Co * H * W):
        if prec == "f16x2":
            cpad = 0
            gs_ref = None
            if isinstance(gn_coeffs, GnStats):
                if gn_coeffs.shape != (B, Ci, H, W):
                    raise ValueError("conv: GroupNorm statistics belong to a different tensor")
                gs_ref, gn_coeffs = gn_coeffs.byref(), None
            if gn_coeffs is not None:
                _req(gn_coeffs, "gn_coeffs")
                if gn_coeffs.dim() != 3 or gn_coeffs.shape[0] != B or gn_coeffs.shape[2] != 4 or \
                        not gn_coeffs.is_contiguous():
                    raise ValueError("gn_coeffs must be contiguous [B, Cpad, 4]")
                cpad = gn_coeffs.shape[1]
            sbuf, slots = None, 0
            # (4 = quad entries: the pre-split kernel's; the deferred epilogue of these kernels writes the finer pairs)
            unit = 2 if (emit_stats is not True and int(emit_stats) in (2, 4)) else 8
            if emit_stats and PRODUCER_GN_STATS and (unit == 8 or ks == 3):   # pair entries: 3x3 kernels only
                slots = int(lib().lc_conv2d_ring_f16x2_stats_slots(B, Ci, Co, H, W, ks, int(tile_cfg)))
                if slots > 0:
                    sbuf = torch.empty((B, Co // unit, slots, 4), device=x.device, dtype=_F32)
            check(_conv_lib().lc_conv2d_ring_f16x2_fwd(x.data_ptr(), x_bs, wh.data_ptr(), wl.data_ptr(),
                                                 _p(bias), _p(res), r_bs, out.data_ptr(), y_bs, B,
                                                 Ci, Co, H, W, ks, float(out_scale),
                                                 int(tile_cfg), _p(gn_coeffs), cpad, int(gn_silu),
                                                 gs_ref, _p(sbuf), unit, packed.wmeta.data_ptr(),
                                                 packed.range_ptr(x.device), _stream()),
                  "lc_conv2d_ring_f16x2_fwd")
            if sbuf is not None:
                _attach_stats(out, _OctStatsHandle(sbuf, Co, slots, (B, H * W), unit))
        else:
            check(lib().lc_conv2d_ring_fwd(x.data_ptr(), x_bs, wp.data_ptr(), _p(bias), _p(res),
                                           r_bs, out.data_ptr(), y_bs, B, Ci, Co, H, W, ks,
                                           float(out_scale), int(tile_cfg), _stream()),
                  "lc_conv2d_ring_fwd")
    return out


def _conv2d_ring_presplit(xs: SplitAct, packed: PackedConv, weight, bias, res, out, out_scale,
                          tile_cfg, emit_stats) -> torch.Tensor:
    """3x3 ring conv of a pre-split activation (lc_conv2d_ring_f16x2_ps_fwd, LDS-DMA staging)."""
    if xs.packed is not packed:
        raise ValueError("conv: the pre-split activation was produced for another layer "
                         "(its x_scale belongs to that layer's range record)")
    wh, wl = packed.get_f16x2(weight)
    B, Ci, H, W = xs.shape
    if Ci != packed.Ci or packed.ks not in (1, 3):
        raise ValueError("conv: pre-split input needs a 1x1 / 3x3 kernel with matching channels")
    Co = packed.Co
    dev = xs.buf.device
    if out is None:
        out = torch.empty((B, Co, H, W), device=dev, dtype=_F32)
    y_bs = _bs4(out, "out")
    if tuple(out.shape) != (B, Co, H, W):
        raise ValueError(f"conv: out shape {tuple(out.shape)} != {(B, Co, H, W)}")
    r_bs = 0
    if res is not None:
        r_bs = _bs4(res, "res")
        if tuple(res.shape) != (B, Co, H, W):
            raise ValueError("conv: residual shape mismatch")
    if bias is not None:
        _req(bias, "bias")
    _drop_stats(out)
    if packed.ks == 1:                       # (no statistics output: a consumer GroupNorm takes its own pass)
        with _Timed("conv1x1", 2.0 * B * H * W * Co * Ci):
            check(_conv_lib().lc_conv1x1_f16x2_ps_fwd(xs.buf.data_ptr(), wh.data_ptr(), wl.data_ptr(), _p(bias), _p(res),
                                                      r_bs, out.data_ptr(), y_bs, B, Ci, Co, H, W, float(out_scale),
                                                      packed.wmeta.data_ptr(), packed.range_ptr(dev), _stream()),
                  "lc_conv1x1_f16x2_ps_fwd")
        return out
    ks = splitk_factor(B, Ci, Co, H, W) if tile_cfg == 0 else 0
    sk_cfg = 0
    if _SPLITK_FORCE and tile_cfg == 0 and Ci >= _SPLITK_FORCE[2]:       # developer switch: "ks:cfg[:min_ci]"
        ks, sk_cfg = _SPLITK_FORCE[0], _SPLITK_FORCE[1]
    with _Timed("conv3x3", 2.0 * B * H * W * Co * Ci * 9,     # (pre-split x: 2 fp16 planes = 4 bytes per element too)
                rd=4.0 * (B * Ci * H * W + Co * Ci * 9 + (B * Co * H * W if res is not None else 0)),
                wr=4.0 * B * Co * H * W):
        sbuf, slots = None, 0
        # octet or quad entries (pair entries -- emit_stats == 2 -- come from the fp32-input kernel's deferred epilogue
        # only; quads: this kernel's epilogue, not the split-K reduction)
        unit = 8 if emit_stats is True else int(emit_stats or 0)
        want_stats = unit in (8, 4) and PRODUCER_GN_STATS and Co % 8 == 0
        if ks >= 2:
            # small grid: ksplit blocks per tile over disjoint K ranges + one deterministic reduce
            part = torch.empty((ks, B, Co, H, W), device=dev, dtype=_F32)
            check(_conv_lib().lc_conv2d_ring_f16x2_ps_fwd(xs.buf.data_ptr(), wh.data_ptr(), wl.data_ptr(),
                                                    None, None, 0, None, 0, B, Ci, Co, H, W, 1.0, sk_cfg,
                                                    None, 8, part.data_ptr(), ks,
                                                    packed.wmeta.data_ptr(), packed.range_ptr(dev),
                                                    _stream()), "lc_conv2d_ring_f16x2_ps_fwd")
            if want_stats and unit == 8:
                slots = int(lib().lc_splitk_stats_slots(H, W))
                sbuf = torch.empty((B, Co // 8, slots, 4), device=dev, dtype=_F32)
            check(lib().lc_splitk_reduce(part.data_ptr(), ks, _p(bias), _p(res), r_bs, out.data_ptr(),
                                         y_bs, B, Co, H, W, float(out_scale), _p(sbuf), _stream()),
                  "lc_splitk_reduce")
        else:
            if want_stats:
                slots = int(lib().lc_conv2d_ring_f16x2_stats_slots(B, max(Ci, 24), Co, H, W, 3,
                                                                   int(tile_cfg)))
                if slots > 0:
                    sbuf = torch.empty((B, Co // unit, slots, 4), device=dev, dtype=_F32)
            check(_conv_lib().lc_conv2d_ring_f16x2_ps_fwd(xs.buf.data_ptr(), wh.data_ptr(), wl.data_ptr(),
                                                    _p(bias), _p(res), r_bs, out.data_ptr(), y_bs, B,
                                                    Ci, Co, H, W, float(out_scale), int(tile_cfg),
                                                    _p(sbuf), unit if sbuf is not None else 8, None, 0,
                                                    packed.wmeta.data_ptr(),
                                                    packed.range_ptr(dev), _stream()),
                  "lc_conv2d_ring_f16x2_ps_fwd")
        if sbuf is not None:
            _attach_stats(out, _OctStatsHandle(sbuf, Co, slots, (B, H * W), unit))
    return out


# ---- Block.downsample folded (round 6): Conv2d(3x3, ring) -> Resample(down=2) as ONE stride-2 conv behind a FIR pre-filter.
# csrc/conv_f16x2_s2.hip has the algebra (both operators are linear; away from the H border they commute, and the two border
# cases are handled by variant rows of the pre-filtered tensor + the bias factor 7/8).  LC_FOLD_DOWN=0: the reference's
# order (conv at full resolution, then the resampling pass).
FOLD_DOWN = _os.environ.get("LC_FOLD_DOWN", "1") != "0"


def can_fold_down(Ci: int, Co: int, H: int, W: int) -> bool:
    return (FOLD_DOWN and PRESPLIT and CONV_PRECISION == "f16x2" and conv_products() == 3 and Ci % 16 == 0 and
            Co % 8 == 0 and H % 2 == 0 and H >= 4 and W % 128 == 0)


def conv_down2(x: torch.Tensor, packed: PackedConv, weight, bias, out: Optional[torch.Tensor] = None,
               emit_stats=False) -> torch.Tensor:
    """y = Resample(down=2)(Conv2d_ring3x3(x) + bias) of the reference (efficient_unet.py:132-135, ops.py:52-173) for
    x [B, Ci, H, W] -> [B, Co, H/2, W/2], computed as stride-2 conv(FIR-pre-filter(x)): lc_fir_down2_prefilter_split +
    lc_conv2d_ring_s2_f16x2_ps_fwd.  emit_stats (True / 8 / 4): octet / quad GroupNorm statistics of the result."""
    x_bs = _bs4(x, "x")
    B, Ci, H, W = x.shape
    wh, wl = packed.get_f16x2(weight)
    Co = packed.Co
    if packed.ks != 3 or Ci != packed.Ci or not can_fold_down(Ci, Co, H, W):
        raise ValueError("conv_down2: needs a 3x3 layer with Ci % 16 == 0, H even >= 4, W % 128 == 0 (ops.can_fold_down)")
    dev = x.device
    Ho, Wo = H // 2, W // 2
    if out is None:
        out = torch.empty((B, Co, Ho, Wo), device=dev, dtype=_F32)
    y_bs = _bs4(out, "out")
    if tuple(out.shape) != (B, Co, Ho, Wo):
        raise ValueError(f"conv_down2: out shape {tuple(out.shape)} != {(B, Co, Ho, Wo)}")
    if bias is not None:
        _req(bias, "bias")
    _drop_stats(out)
    units = int(lib().lc_fir_down2_split_units(B, Ci, H, W))
    buf = torch.empty((units, 8), device=dev, dtype=torch.float16)
    rng_ptr = packed.range_ptr(dev)
    with _Timed("resample", 8.0 * B * Ci * H * W):
        check(lib().lc_fir_down2_prefilter_split(x.data_ptr(), x_bs, buf.data_ptr(), B, Ci, H, W, rng_ptr, _stream()),
              "lc_fir_down2_prefilter_split")
    unit = 8 if emit_stats is True else int(emit_stats or 0)
    sbuf, slots = None, 0
    if unit in (8, 4) and PRODUCER_GN_STATS:
        slots = int(lib().lc_conv2d_ring_s2_stats_slots(Ho, Wo))
        if slots > 0:
            sbuf = torch.empty((B, Co // unit, slots, 4), device=dev, dtype=_F32)
    # (algorithmic work = the reference's 3x3 conv at FULL resolution, which this launch replaces; it executes a quarter)
    with _Timed("conv3x3", 2.0 * B * H * W * Co * Ci * 9, executed=2.0 * B * Ho * Wo * Co * Ci * 9,
                rd=4.0 * (B * Ci * (H + 3) * W + Co * Ci * 9), wr=4.0 * B * Co * Ho * Wo):
        check(lib().lc_conv2d_ring_s2_f16x2_ps_fwd(buf.data_ptr(), wh.data_ptr(), wl.data_ptr(), _p(bias), out.data_ptr(),
                                                   y_bs, B, Ci, Co, Ho, Wo, 1.0, _p(sbuf), unit if sbuf is not None else 8,
                                                   packed.wmeta.data_ptr(), rng_ptr, _stream()),
              "lc_conv2d_ring_s2_f16x2_ps_fwd")
    if sbuf is not None:
        _attach_stats(out, _OctStatsHandle(sbuf, Co, slots, (B, Ho * Wo), unit))
    return out


# ---- Conv2d(3x3, ring) BEHIND Resample(up=2) folded (round 6, third part; csrc/upfold.hip has the algebra): the nine tap
# planes W[:, :, ky, kx] . a are ONE 1x1 projection Ci -> 9 Co of the LOW-resolution operand (a quarter of the 3x3 conv's
# multiply-adds at the high resolution), lc_up2_combine9_fwd up-samples, shifts and sums them in one pass.  LC_FOLD_UP=0: the
# reference's order (resampling pass, then the conv at the high resolution).  Below LC_FOLD_UP_MIN_CI input channels the
# launches are bound by the bytes of the 9-plane intermediate (2.25x the conv's output), not by the matrix pipes: the order
# of the reference stays (profiles/r06_fold_up.txt: 64 -> 64 @ 8 x 16 x 512 folded 93 us against 89).
FOLD_UP = _os.environ.get("LC_FOLD_UP", "1") != "0"
FOLD_UP_MIN_CI = int(_os.environ.get("LC_FOLD_UP_MIN_CI", "128"))
# the skip path's Resample(up=2)(x) of an up-sampling ResBlock in the combine launch (lc_up2_combine9_xup_fwd); "0": its own launch
FOLD_UP_X = _os.environ.get("LC_FOLD_UP_X", "1") != "0"
# working set (nine planes + output) of one projection + combine pair, MB; a batch above it runs in slabs of samples; 0: one slab
FOLD_UP_SLAB_MB = int(_os.environ.get("LC_FOLD_UP_SLAB_MB", "224"))


def can_fold_up(Ci: int, Co: int, H: int, W: int) -> bool:
    """(H, W) = the LOW resolution.  Needs the pre-split 1x1 kernel (Ci % 32) and whole 128-column segments."""
    return (FOLD_UP and PRESPLIT and CONV_PRECISION == "f16x2" and Ci % 32 == 0 and Ci >= FOLD_UP_MIN_CI
            and W % 128 == 0 and H >= 1 and 9 * Co * H * W * 4 < (1 << 31))


def up9_weight(weight: torch.Tensor) -> torch.Tensor:
    """[Co, Ci, 3, 3] -> the 1x1 weight [9 Co, Ci, 1, 1] of the fold: row t Co + co = W[co, :, ky, kx], t = 3 ky + kx."""
    Co, Ci = weight.shape[:2]
    # (outside inference mode: the samplers run under torch.inference_mode, and PackedConv keys its caches on the version
    #  counter, which inference tensors do not carry)
    with torch.inference_mode(False), torch.no_grad():
        return weight.detach().permute(2, 3, 0, 1).reshape(9 * Co, Ci, 1, 1).contiguous()


def split_act(x: torch.Tensor, packed: PackedConv) -> SplitAct:
    """x as it is (no normalisation) in the pre-split form of `packed`'s input (lc_split_act_fwd)."""
    x_bs = _bs4(x, "x")
    B, C, H, W = x.shape
    units = int(lib().lc_split_act_units(B, C, H, W))
    buf = torch.empty((units, 8), device=x.device, dtype=torch.float16)
    with _Timed("groupnorm", 8.0 * B * C * H * W):
        check(lib().lc_split_act_fwd(x.data_ptr(), x_bs, buf.data_ptr(), B, C, H, W, packed.range_ptr(x.device), _stream()),
              "lc_split_act_fwd")
    return SplitAct(buf, (B, C, H, W), packed)


def conv_up2(xs, packed9: PackedConv, weight9: torch.Tensor, bias: Optional[torch.Tensor] = None,
             out: Optional[torch.Tensor] = None, emit_stats=False, up_also: Optional[torch.Tensor] = None):
    """y = Conv2d_ring3x3(Resample(up=2)(a)) + bias of the reference for a [B, Ci, H, W] -> [B, Co, 2H, 2W], computed at the
    LOW resolution.  xs: `a` pre-split for `packed9` (a SplitAct from `groupnorm(..., split_for=packed9)` / `split_act`),
    weight9 = `up9_weight(conv.weight)` (the caller caches it per weight version), packed9 its PackedConv.
    emit_stats: per-channel GroupNorm statistics entries of the result (any consumer folds them).
    up_also: a second low-resolution tensor [B, Co, H, W] whose plain Resample(up=2) is written by the same combine launch
    (the skip path of LayoutUnetV1's up-sampling ResBlock; bit-identical to `resample2x(up_also, up=True)`): the call then
    returns (y, resample2x(up_also))."""
    if not isinstance(xs, SplitAct) or xs.packed is not packed9:
        raise ValueError("conv_up2: needs the operand pre-split for this layer (groupnorm(split_for=packed9) / split_act)")
    B, Ci, H, W = xs.shape
    if weight9.dim() != 4 or weight9.shape[1] != Ci or weight9.shape[0] % 9 or tuple(weight9.shape[2:]) != (1, 1):
        raise ValueError("conv_up2: weight9 must be up9_weight(conv.weight) = [9 Co, Ci, 1, 1]")
    Co = weight9.shape[0] // 9
    if not can_fold_up(Ci, Co, H, W):
        raise ValueError("conv_up2: needs Ci % 32 == 0, Ci >= LC_FOLD_UP_MIN_CI, W % 128 == 0 (ops.can_fold_up)")
    dev = xs.buf.device
    if out is None:
        out = torch.empty((B, Co, 2 * H, 2 * W), device=dev, dtype=_F32)
    y_bs = _bs4(out, "out")
    if tuple(out.shape) != (B, Co, 2 * H, 2 * W):
        raise ValueError(f"conv_up2: out shape {tuple(out.shape)} != {(B, Co, 2 * H, 2 * W)}")
    if bias is not None:
        _req(bias, "bias")
    _drop_stats(out)
    x2, y2, x2_bs, y2_bs = None, None, 0, 0
    if up_also is not None:
        x2 = up_also
        x2_bs = _bs4(x2, "up_also")
        if tuple(x2.shape) != (B, Co, H, W):
            raise ValueError(f"conv_up2: up_also must be {(B, Co, H, W)}, got {tuple(x2.shape)}")
        if x2.data_ptr() % 8 or x2_bs % 2:
            x2 = x2.contiguous()
            x2_bs = Co * H * W
        y2 = torch.empty((B, Co, 2 * H, 2 * W), device=dev, dtype=_F32)
        y2_bs = Co * 4 * H * W
    wh, wl = packed9.get_f16x2(weight9)
    if packed9.ks != 1 or packed9.Ci != Ci:
        raise ValueError("conv_up2: packed9 does not hold weight9")
    sbuf, slots = None, 0
    if emit_stats and PRODUCER_GN_STATS:
        slots = int(lib().lc_up2_combine9_stats_slots(H, W))
        if slots > 0:
            sbuf = torch.empty((B, Co, slots, 4), device=dev, dtype=_F32)
    # Slabs of samples: the nine planes of a slab + its output should stay inside the 256 MB Infinity Cache between the two
    # launches (the combine pass runs at ~6 TB/s on a 218 MB working set and at 3.4 TB/s on 436 MB, profiles/r06_fold_up.txt)
    per_sample = 4.0 * (9 + 4) * Co * H * W
    nb = B if FOLD_UP_SLAB_MB <= 0 else max(1, min(B, int(FOLD_UP_SLAB_MB * 2.0 ** 20 // per_sample)))
    while B % nb:
        nb -= 1
    p9 = torch.empty((nb, 9 * Co, H, W), device=dev, dtype=_F32)
    xs_bs = xs.buf.numel() // B * 2          # bytes per sample of the pre-split operand (fp16 elements)
    st = _stream()
    for b0 in range(0, B, nb):
        # (algorithmic work = the reference's 3x3 conv at the HIGH resolution, which this launch replaces; it executes a quarter)
        with _Timed("conv3x3", 2.0 * nb * 4 * H * W * Co * Ci * 9, executed=2.0 * nb * H * W * 9 * Co * Ci,
                    rd=4.0 * (nb * Ci * H * W + Co * Ci * 9), wr=4.0 * nb * 9 * Co * H * W):
            check(_conv_lib().lc_conv1x1_f16x2_ps_fwd(xs.buf.data_ptr() + b0 * xs_bs, wh.data_ptr(), wl.data_ptr(), None, None,
                                                      0, p9.data_ptr(), 9 * Co * H * W, nb, Ci, 9 * Co, H, W, 1.0,
                                                      packed9.wmeta.data_ptr(), packed9.range_ptr(dev), st),
                  "lc_conv1x1_f16x2_ps_fwd")
        sp = None if sbuf is None else sbuf.data_ptr() + 16 * b0 * Co * slots
        with _Timed("resample", 4.0 * nb * Co * H * W * (13.0 if x2 is None else 18.0)):
            if x2 is None:
                check(lib().lc_up2_combine9_fwd(p9.data_ptr(), 9 * Co * H * W, _p(bias), out.data_ptr() + 4 * b0 * y_bs, y_bs,
                                                nb, Co, H, W, sp, st), "lc_up2_combine9_fwd")
            else:
                check(lib().lc_up2_combine9_xup_fwd(p9.data_ptr(), 9 * Co * H * W, _p(bias), out.data_ptr() + 4 * b0 * y_bs,
                                                    y_bs, x2.data_ptr() + 4 * b0 * x2_bs, x2_bs,
                                                    y2.data_ptr() + 4 * b0 * y2_bs, y2_bs, nb, Co, H, W, sp, st),
                      "lc_up2_combine9_xup_fwd")
    if sbuf is not None:
        _attach_stats(out, _OctStatsHandle(sbuf, Co, slots, (B, 4 * H * W), 1))
    return out if up_also is None else (out, y2)


# Split-K (pre-split conv): when a 3x3 conv has fewer than SPLITK_MAX_BLOCKS output tiles (batch
# 1-2 at the deep levels), its K range is divided over several blocks per tile.
SPLITK = _os.environ.get("LC_SPLITK", "1") != "0"
SPLITK_MAX_BLOCKS = int(_os.environ.get("LC_SPLITK_MAX_BLOCKS", "64"))
_SPLITK_FORCE = tuple(int(v) for v in (_os.environ["LC_SPLITK_FORCE"] + ":0").split(":")[:3]) \
    if _os.environ.get("LC_SPLITK_FORCE") else None


def splitk_factor(B: int, Ci: int, Co: int, H: int, W: int) -> int:
    """Blocks per output tile for the pre-split conv (0 / 1 = no split)."""
    if not SPLITK or Ci % 16:
        return 0
    nchunk = Ci // 16
    # the tile shapes the heuristic would pick: 64 co x 256 / 128 / 64 px
    px = B * H * W
    co_blocks = (Co + 63) // 64
    for tile in (256, 128, 64):
        blocks = co_blocks * ((px + tile - 1) // tile)
        if blocks >= 256 or tile == 64:
            break
    if blocks > SPLITK_MAX_BLOCKS or nchunk < 4:
        return 0
    ks = min(8, nchunk // 2, max(1, 256 // blocks))
    return ks if ks >= 2 else 0


# ------------------------------------------------------------------------------------ norm
_gn_scratch = {}


def _partials(dev, n: int) -> torch.Tensor:
    key = (dev, torch.cuda.current_stream().cuda_stream)
    buf = _gn_scratch.get(key)
    if buf is None or buf.numel() < n:
        buf = torch.empty(max(n, 1 << 16), device=dev, dtype=torch.float64)
        _gn_scratch[key] = buf
    return buf


def groupnorm(x: torch.Tensor, G: int, eps: float, gamma=None, beta=None, scale=None, shift=None,
              act_silu: bool = False, out: Optional[torch.Tensor] = None,
              split_for: Optional["PackedConv"] = None):
    """GroupNorm (+affine) (+ (1+scale)*h+shift) (+SiLU).  scale/shift: [B, C] views (row stride
    may exceed C, e.g. the two halves of one [B, 2C] AdaGN projection).
    split_for: the PackedConv of the 3x3 conv that consumes the result -- when the shape allows
    (`can_presplit`) the result is written PRE-SPLIT for that layer and returned as a `SplitAct`
    (same bytes, no fp32 copy exists); otherwise a plain fp32 tensor comes back."""
    x_bs = _bs4(x, "x")
    B, C, H, W = x.shape
    if C % G:
        raise ValueError(f"groupnorm: C={C} not divisible by G={G}")
    if split_for is not None and out is None and can_presplit(C, G):
        return _groupnorm_split(x, x_bs, G, eps, gamma, beta, scale, shift, act_silu, split_for)
    if out is None:
        out = torch.empty((B, C, H, W), device=x.device, dtype=_F32)
    y_bs = _bs4(out, "out")
    ss_bs = 0
    if scale is not None:
        _req(scale, "scale"), _req(shift, "shift")
        if scale.shape != (B, C) or shift.shape != (B, C) or scale.stride(1) != 1 or \
                shift.stride(1) != 1 or scale.stride(0) != shift.stride(0):
            raise ValueError("groupnorm: scale/shift must be [B,C] with unit inner stride")
        ss_bs = scale.stride(0)
    st = _stream()
    hs = _find_stats(x, G)
    _drop_stats(out)
    if hs is not None:   # the producer conv left the statistics: one pass over the tensor
        import ctypes as C_
        from ._lib import OctStats

        keep = [OctStats(h.buf.data_ptr(), h.channels, h.slots, h.unit) for h in hs]
        with _Timed("groupnorm", 8.0 * B * C * H * W):
            check(lib().lc_groupnorm_apply_os(x.data_ptr(), x_bs, C_.byref(keep[0]),
                                              C_.byref(keep[1]) if len(keep) > 1 else None, _p(gamma),
                                              _p(beta), _p(scale), _p(shift), ss_bs, out.data_ptr(),
                                              y_bs, B, C, H, W, G, float(eps), int(act_silu), st),
                  "lc_groupnorm_apply_os")
        return out
    n = lib().lc_groupnorm_partials_elems(B, C, H, W, G)
    part = _partials(x.device, n)
    with _Timed("groupnorm", 12.0 * B * C * H * W):  # bytes: stats read + apply read + write
        check(lib().lc_groupnorm_stats(x.data_ptr(), x_bs, part.data_ptr(), B, C, H, W, G, st),
              "lc_groupnorm_stats")
        check(lib().lc_groupnorm_apply(x.data_ptr(), x_bs, part.data_ptr(), _p(gamma), _p(beta),
                                       _p(scale), _p(shift), ss_bs, out.data_ptr(), y_bs, B, C, H,
                                       W, G, float(eps), int(act_silu), st), "lc_groupnorm_apply")
    return out


def _groupnorm_split(x, x_bs, G, eps, gamma, beta, scale, shift, act_silu, packed) -> SplitAct:
    import ctypes as C_

    from ._lib import OctStats

    B, C, H, W = x.shape
    ss_bs = 0
    if scale is not None:
        _req(scale, "scale"), _req(shift, "shift")
        if scale.shape != (B, C) or shift.shape != (B, C) or scale.stride(1) != 1 or \
                shift.stride(1) != 1 or scale.stride(0) != shift.stride(0):
            raise ValueError("groupnorm: scale/shift must be [B,C] with unit inner stride")
        ss_bs = scale.stride(0)
    units = int(lib().lc_split_act_units(B, C, H, W))
    buf = torch.empty((units, 8), device=x.device, dtype=torch.float16)
    rng_ptr = packed.range_ptr(x.device)
    st = _stream()
    hs = _find_stats(x, G, octet_groups=True)
    if hs is not None:
        keep = [OctStats(h.buf.data_ptr(), h.channels, h.slots, h.unit) for h in hs]
        with _Timed("groupnorm", 8.0 * B * C * H * W):
            check(lib().lc_groupnorm_apply_os_split(
                x.data_ptr(), x_bs, C_.byref(keep[0]), C_.byref(keep[1]) if len(keep) > 1 else None,
                _p(gamma), _p(beta), _p(scale), _p(shift), ss_bs, buf.data_ptr(), B, C, H, W, G,
                float(eps), int(act_silu), rng_ptr, st), "lc_groupnorm_apply_os_split")
    else:
        n = lib().lc_groupnorm_partials_elems(B, C, H, W, G)
        part = _partials(x.device, n)
        with _Timed("groupnorm", 12.0 * B * C * H * W):
            check(lib().lc_groupnorm_stats(x.data_ptr(), x_bs, part.data_ptr(), B, C, H, W, G, st),
                  "lc_groupnorm_stats")
            check(lib().lc_groupnorm_apply_split(
                x.data_ptr(), x_bs, part.data_ptr(), _p(gamma), _p(beta), _p(scale), _p(shift), ss_bs,
                buf.data_ptr(), B, C, H, W, G, float(eps), int(act_silu), rng_ptr, st),
                "lc_groupnorm_apply_split")
    return SplitAct(buf, (B, C, H, W), packed)


class GnStats:
    """Statistics of ONE tensor + the GroupNorm / AdaGN parameters, for a conv that normalises its
    input on the fly (`conv2d_ring(..., gn_coeffs=<GnStats>)`): the conv blocks derive the
    per-channel rows themselves, no lc_groupnorm_coeffs launch.  Keeps everything it points to alive."""

    __slots__ = ("shape", "_struct", "_keep")

    def __init__(self, shape, part, G, nch, eps, gamma, beta, scale, shift, ss_bs, oct_handles=None):
        import ctypes as C

        from ._lib import GnStatsInput, OctStats

        self.shape = shape
        segs = [OctStats(h.buf.data_ptr(), h.channels, h.slots, h.unit) for h in (oct_handles or ())]
        self._keep = (part, gamma, beta, scale, shift, oct_handles, segs)
        self._struct = GnStatsInput(_p(part), G, nch, float(eps), _p(gamma), _p(beta),
                                    _p(scale), _p(shift), ss_bs,
                                    C.pointer(segs[0]) if segs else None,
                                    C.pointer(segs[1]) if len(segs) > 1 else None)

    def byref(self):
        import ctypes as C

        return C.byref(self._struct)


def groupnorm_stats(x: torch.Tensor, G: int, eps: float, gamma=None, beta=None, scale=None,
                    shift=None) -> GnStats:
    """Statistics for a conv that normalises its input on the fly: the producer's octet statistics
    when `x` carries them (no launch at all), else one statistics pass; the consumer conv turns
    either into rows in its prologue."""
    x_bs = _bs4(x, "x")
    B, C, H, W = x.shape
    if C % G:
        raise ValueError(f"groupnorm: C={C} not divisible by G={G}")
    ss_bs = 0
    if scale is not None:
        _req(scale, "scale"), _req(shift, "shift")
        if scale.shape != (B, C) or shift.shape != (B, C) or scale.stride(1) != 1 or \
                shift.stride(1) != 1 or scale.stride(0) != shift.stride(0):
            raise ValueError("groupnorm: scale/shift must be [B,C] with unit inner stride")
        ss_bs = scale.stride(0)
    for n_, t_ in (("gamma", gamma), ("beta", beta)):
        if t_ is not None:
            _req(t_, n_)
    hs = _find_stats(x, G) if G <= 128 else None
    if hs is not None:
        return GnStats((B, C, H, W), None, G, 0, eps, gamma, beta, scale, shift, ss_bs, hs)
    n = lib().lc_groupnorm_partials_elems(B, C, H, W, G)
    part = torch.empty((n,), device=x.device, dtype=torch.float64)   # owned by the handle
    with _Timed("groupnorm", 4.0 * B * C * H * W):
        check(lib().lc_groupnorm_stats(x.data_ptr(), x_bs, part.data_ptr(), B, C, H, W, G, _stream()),
              "lc_groupnorm_stats")
    return GnStats((B, C, H, W), part, G, n // (2 * B * G), eps, gamma, beta, scale, shift, ss_bs)


def groupnorm_coeffs(x: torch.Tensor, G: int, eps: float, gamma=None, beta=None, scale=None,
                     shift=None) -> torch.Tensor:
    """Statistics pass only: returns the per-(b, channel) rows (mu, A, Bc, 0) [B, C^16, 4] that
    `conv2d_ring(..., gn_coeffs=...)` applies while staging its input."""
    x_bs = _bs4(x, "x")
    B, C, H, W = x.shape
    if C % G:
        raise ValueError(f"groupnorm: C={C} not divisible by G={G}")
    ss_bs = 0
    if scale is not None:
        _req(scale, "scale"), _req(shift, "shift")
        if scale.shape != (B, C) or shift.shape != (B, C) or scale.stride(1) != 1 or \
                shift.stride(1) != 1 or scale.stride(0) != shift.stride(0):
            raise ValueError("groupnorm: scale/shift must be [B,C] with unit inner stride")
        ss_bs = scale.stride(0)
    cpad = (C + 15) // 16 * 16
    out = torch.empty((B, cpad, 4), device=x.device, dtype=_F32)
    n = lib().lc_groupnorm_partials_elems(B, C, H, W, G)
    part = _partials(x.device, n)
    st = _stream()
    with _Timed("groupnorm", 4.0 * B * C * H * W):
        check(lib().lc_groupnorm_stats(x.data_ptr(), x_bs, part.data_ptr(), B, C, H, W, G, st),
              "lc_groupnorm_stats")
        check(lib().lc_groupnorm_coeffs(x.data_ptr(), x_bs, part.data_ptr(), _p(gamma), _p(beta),
                                        _p(scale), _p(shift), ss_bs, out.data_ptr(), B, C, cpad,
                                        H, W, G, float(eps), st), "lc_groupnorm_coeffs")
    return out


# ------------------------------------------------------------------------------------ resample
def resample2x(x: torch.Tensor, up: bool, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    x_bs = _bs4(x, "x")
    B, C, H, W = x.shape
    shape = (B, C, 2 * H, 2 * W) if up else (B, C, H // 2, W // 2)
    if out is None:
        out = torch.empty(shape, device=x.device, dtype=_F32)
    y_bs = _bs4(out, "out")
    if tuple(out.shape) != shape:
        raise ValueError("resample: out shape mismatch")
    _drop_stats(out)
    # the down-sampler leaves per-channel GroupNorm statistics of what it stores where its vector kernel runs (W % 256 == 0,
    # aligned rows): the GroupNorm behind it then takes no statistics pass
    slots = 0
    if not up and PRODUCER_GN_STATS and RESAMPLE_STATS and x.data_ptr() % 16 == 0 and x_bs % 4 == 0 and y_bs % 2 == 0 \
            and out.data_ptr() % 8 == 0:
        slots = int(lib().lc_resample2x_stats_slots(H, W, -1))
    with _Timed("resample", 4.0 * B * C * H * W * (5.0 if up else 1.25)):
        if slots > 0:
            sbuf = torch.empty((B, C, slots, 4), device=x.device, dtype=_F32)
            check(lib().lc_resample2x_stats_fwd(x.data_ptr(), x_bs, out.data_ptr(), y_bs, B, C, H, W, -1,
                                                sbuf.data_ptr(), _stream()), "lc_resample2x_stats_fwd")
            _attach_stats(out, _OctStatsHandle(sbuf, C, slots, (B, (H // 2) * (W // 2)), 1))
        else:
            check(lib().lc_resample2x_fwd(x.data_ptr(), x_bs, out.data_ptr(), y_bs, B, C, H, W,
                                          1 if up else -1, _stream()), "lc_resample2x_fwd")
    return out


# A resampling ResBlock of the layout model needs op(SiLU(GroupNorm(x))) AND op(x): one pass over x instead of three
# (GroupNorm apply, resample of its result, resample of x).  LC_RESAMPLE_PAIR=0: the three passes.
RESAMPLE_PAIR = _os.environ.get("LC_RESAMPLE_PAIR", "1") != "0"


def groupnorm_resample_pair(x: torch.Tensor, G: int, eps: float, gamma, beta, up: bool):
    """-> (resample2x(silu(groupnorm(x))), resample2x(x)).  The second is bit-identical to `resample2x(x, up)`; the first
    applies the normalisation in its per-channel affine form (mu, A, Bc) while filtering."""
    x_bs = _bs4(x, "x")
    B, C, H, W = x.shape
    if C % G:
        raise ValueError(f"groupnorm: C={C} not divisible by G={G}")
    cpad = (C + 15) // 16 * 16
    hs = _find_stats(x, G)
    st = _stream()
    if hs is not None:
        import ctypes as C_
        from ._lib import OctStats

        coef = torch.empty((B, cpad, 4), device=x.device, dtype=_F32)
        keep = [OctStats(h.buf.data_ptr(), h.channels, h.slots, h.unit) for h in hs]
        check(lib().lc_groupnorm_coeffs_os(C_.byref(keep[0]), C_.byref(keep[1]) if len(keep) > 1 else None, _p(gamma),
                                           _p(beta), None, None, 0, coef.data_ptr(), B, C, cpad, G, float(eps), st),
              "lc_groupnorm_coeffs_os")
    else:
        coef = groupnorm_coeffs(x, G, eps, gamma, beta)
    shape = (B, C, 2 * H, 2 * W) if up else (B, C, H // 2, W // 2)
    y = torch.empty(shape, device=x.device, dtype=_F32)
    ya = torch.empty(shape, device=x.device, dtype=_F32)
    with _Timed("resample", 4.0 * B * C * H * W * (9.0 if up else 1.5)):
        check(lib().lc_resample2x_pair_fwd(x.data_ptr(), x_bs, coef.data_ptr(), cpad, y.data_ptr(), _bs4(y, "y"), ya.data_ptr(),
                                           _bs4(ya, "ya"), B, C, H, W, 1 if up else -1, st), "lc_resample2x_pair_fwd")
    return ya, y


# ------------------------------------------------------------------------------------ dense
def linear(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor] = None,
           act_in: bool = False, act_out: bool = False) -> torch.Tensor:
    _req(x, "x"), _req(w, "w")
    if x.dim() != 2 or not x.is_contiguous() or not w.is_contiguous():
        raise ValueError("linear: x must be contiguous [M,K], w contiguous [N,K]")
    M, K = x.shape
    N = w.shape[0]
    y = torch.empty((M, N), device=x.device, dtype=_F32)
    check(lib().lc_linear_fwd(x.data_ptr(), w.data_ptr(), _p(b), y.data_ptr(), M, K, N,
                              int(act_in), int(act_out), _stream()), "lc_linear_fwd")
    return y


def sinusoid(t: torch.Tensor, channels: int, max_period: float = 10_000.0) -> torch.Tensor:
    _req(t, "t")
    t = t.contiguous()
    y = torch.empty((t.shape[0], channels), device=t.device, dtype=_F32)
    check(lib().lc_sinusoid_fwd(t.data_ptr(), y.data_ptr(), t.shape[0], channels,
                                float(max_period), _stream()), "lc_sinusoid_fwd")
    return y


# ------------------------------------------------------------------------------------ attention
# "f16x2" (default): hi/lo-split fp16 MFMA with fp32 accumulation (fp32-class accuracy, the same
# scheme as the conv); "f32": the fp32-MFMA kernel.
ATTN_PRECISION = _os.environ.get("LC_ATTN_PRECISION", "f16x2")


def attention_cm(q, k, v, heads: int, scale: float, k2=None, v2=None, q_pos=None, k_pos=None,
                 k2_pos=None, out: Optional[torch.Tensor] = None,
                 precision: Optional[str] = None) -> torch.Tensor:
    """Channel-major attention.  q:[B, heads*dqk, Lq]  k:[B, heads*dqk, Lk0]  v:[B, heads*dv, Lk0]
    (views with arbitrary batch/channel strides, unit token stride).  Optional positional parts
    q_pos/k_pos/k2_pos [B, heads*dpos, L] are concatenated to each head's q/k channels inside the
    kernel; optional second key/value segment k2/v2 with Lk1 tokens (the 13 layout tokens of
    ObjectAwareCrossAttention).  Batch stride 0 (expand) is allowed for step-invariant operands."""
    from ._lib import CmOperand
    import ctypes as C

    def chk(n_, t_):
        _req(t_, n_)
        if t_.dim() != 3 or (t_.shape[2] > 1 and t_.stride(2) != 1):
            raise ValueError(f"attention: `{n_}` must be [B,C,L] with unit token stride")

    for n_, t_ in (("q", q), ("k", k), ("v", v)):
        chk(n_, t_)
    B, Cq, Lq = q.shape
    dqk, dv = Cq // heads, v.shape[1] // heads
    Lk0 = k.shape[2]
    Lk1 = 0 if k2 is None else k2.shape[2]
    dpos = 0 if q_pos is None else q_pos.shape[1] // heads
    if out is None:
        out = torch.empty((B, heads * dv, Lq), device=q.device, dtype=_F32)

    def op(t, d):
        if t is None:
            return None
        chk("operand", t)
        return C.byref(CmOperand(t.data_ptr(), t.stride(0) if t.shape[0] > 1 else 0,
                                 d * t.stride(1), t.stride(1)))

    prec = precision or ATTN_PRECISION
    if prec not in ("f32", "f16x2"):
        raise ValueError(prec)
    fn = lib().lc_attention_f16x2_fwd if prec == "f16x2" else lib().lc_attention_fwd
    with _Timed("attention", 2.0 * B * heads * Lq * (Lk0 + Lk1) * (dqk + dpos + dv)):
        check(fn(op(q, dqk), op(q_pos, dpos), op(k, dqk), op(k_pos, dpos), op(v, dv), op(k2, dqk),
                 op(k2_pos, dpos), op(v2, dv), out.data_ptr(), out.stride(0), dv * out.stride(1),
                 out.stride(1), B, heads, Lq, Lk0, Lk1, dqk, dpos, dv, float(scale), _stream()),
              "lc_attention_fwd")
    return out


# Keys / values in unit form (csrc/attention_units.hip): the fp16 hi / lo split of the attention operands made once per
# step -- or once per CONDITION for the step-invariant parts -- instead of once per query block inside the kernel.
# LC_ATTN_UNITS=0 keeps ObjectAwareCrossAttention on lc_attention_f16x2_fwd.
ATTN_UNITS = _os.environ.get("LC_ATTN_UNITS", "1") != "0"


class AttnUnits:
    """K / V of one attention layer in unit form: `buf` [B, heads, tiles, 768 * 8] halves (zero where a head has no
    channel or key).  Image keys first (Lk0, a multiple of 32), then one tile of up to 32 further keys (Lk1)."""

    __slots__ = ("buf", "B", "heads", "Lk0", "Lk1", "dqk", "dpos", "dv")

    def __init__(self, B, heads, Lk0, Lk1, dqk, dpos, dv, device):
        n = lib().lc_attention_units_elems(B, heads, Lk0, Lk1)
        if n < 0 or dqk % 8 or dpos % 8 or dqk + dpos > 64 or dv > 32:
            raise ValueError("attention units: Lk0 % 32 == 0, Lk1 <= 32, dqk / dpos multiples of 8 with dqk + dpos <= 64, dv <= 32")
        self.buf = torch.zeros((B, heads, Lk0 // 32 + (1 if Lk1 > 0 else 0), 768 * 8), device=device, dtype=torch.float16)
        self.B, self.heads, self.Lk0, self.Lk1, self.dqk, self.dpos, self.dv = B, heads, Lk0, Lk1, dqk, dpos, dv

    @staticmethod
    def eligible(heads, Lk0, Lk1, dqk, dpos, dv) -> bool:
        return (ATTN_UNITS and ATTN_PRECISION == "f16x2" and Lk0 % 32 == 0 and 0 <= Lk1 <= 32 and dqk % 8 == 0
                and dpos % 8 == 0 and 32 < dqk + dpos <= 64 and dv <= 32)

    def fits(self, B, heads, Lk0, Lk1, dqk, dpos, dv, device) -> bool:
        return (self.B, self.heads, self.Lk0, self.Lk1, self.dqk, self.dpos, self.dv, self.buf.device) == \
            (B, heads, Lk0, Lk1, dqk, dpos, dv, device)


def _cm_operand(t, d):
    from ._lib import CmOperand
    import ctypes as C

    _req(t, "operand")
    if t.dim() != 3 or (t.shape[2] > 1 and t.stride(2) != 1):
        raise ValueError("attention: operands must be [B,C,L] with unit token stride")
    return C.byref(CmOperand(t.data_ptr(), t.stride(0) if t.shape[0] > 1 else 0, d * t.stride(1), t.stride(1)))


def attention_pack_units(units: AttnUnits, src: torch.Tensor, what: str, segment: int = 0) -> None:
    """Write one operand into the unit form.  what: "k" (content channels of the keys), "k_pos" (their positional
    channels), "v"; segment 0 = the image keys [0, Lk0), 1 = the extra tile (layout keys).  src: [B, heads * d, L] fp32
    (batch stride 0 allowed), split with the constants of lc_attention_f16x2_fwd."""
    u = units
    d = {"k": u.dqk, "k_pos": u.dpos, "v": u.dv}[what]
    L = u.Lk0 if segment == 0 else u.Lk1
    if src.shape[1] != u.heads * d or src.shape[2] != L or src.shape[0] not in (1, u.B):
        raise ValueError(f"attention_pack_units: `{what}` of segment {segment} must be [{u.B}, {u.heads * d}, {L}], got {tuple(src.shape)}")
    check(lib().lc_attention_pack_units(_cm_operand(src, d), u.buf.data_ptr(), u.B, u.heads, L, u.Lk0, u.Lk1, d,
                                        u.dqk // 8 if what == "k_pos" else 0, 0 if segment == 0 else u.Lk0,
                                        1 if what == "v" else 0, _stream()), "lc_attention_pack_units")


def qkv_project_units(xs: SplitAct, packed: PackedConv, weight: torch.Tensor, bias, units: AttnUnits) -> torch.Tensor:
    """The qkv projection (1x1, C -> 3 C) of a pre-split token tensor with its keys and values written straight into
    `units` (image-key segment): returns q [B, C, L] fp32; k and v never exist as fp32 tensors
    (lc_conv1x1_f16x2_ps_qkv_fwd).  32 channels per head, C % 128 == 0."""
    if xs.packed is not packed:
        raise ValueError("conv: the pre-split activation was produced for another layer")
    wh, wl = packed.get_f16x2(weight)
    B, Ci, H, W = xs.shape
    C, u = packed.Co // 3, units
    if packed.ks != 1 or packed.Co != 3 * C or Ci != packed.Ci or (u.B, u.heads * 32, u.Lk0, u.dqk, u.dv) != (B, C, H * W, 32, 32):
        raise ValueError("qkv_project_units: a 1x1 projection C -> 3 C onto units of C / 32 heads over H * W keys")
    if bias is not None:
        _req(bias, "bias")
    q = torch.empty((B, C, H * W), device=xs.buf.device, dtype=_F32)
    with _Timed("conv1x1", 2.0 * B * H * W * 3 * C * Ci):
        check(_conv_lib().lc_conv1x1_f16x2_ps_qkv_fwd(xs.buf.data_ptr(), wh.data_ptr(), wl.data_ptr(), _p(bias), q.data_ptr(),
                                                      q.stride(0), u.buf.data_ptr(), B, Ci, C, H, W, u.Lk1,
                                                      packed.wmeta.data_ptr(), packed.range_ptr(xs.buf.device), _stream()),
              "lc_conv1x1_f16x2_ps_qkv_fwd")
    return q


def qkv_units_ok(C: int, heads: int, L: int) -> bool:
    """Can the qkv projection write the unit form itself?  (else: lc_attention_pack_units after a plain projection)"""
    return C % 128 == 0 and C == 32 * heads and L % 32 == 0


def attention_units(q, units: AttnUnits, heads: int, scale: float, q_pos=None, out: Optional[torch.Tensor] = None):
    """attention_cm(q, k, v, ..., k2, v2, q_pos, k_pos, k2_pos) with the keys and values taken from `units`: the same
    arithmetic, bit for bit (the split happens in the writers of the units instead of the attention kernel)."""
    u = units
    B, Cq, Lq = q.shape
    dpos = 0 if q_pos is None else q_pos.shape[1] // heads
    if heads != u.heads or B != u.B or Cq // heads != u.dqk or dpos != u.dpos:
        raise ValueError("attention_units: q does not match the units")
    if out is None:
        out = torch.empty((B, heads * u.dv, Lq), device=q.device, dtype=_F32)
    with _Timed("attention", 2.0 * B * heads * Lq * (u.Lk0 + u.Lk1) * (u.dqk + dpos + u.dv)):
        check(lib().lc_attention_units_fwd(_cm_operand(q, u.dqk), None if q_pos is None else _cm_operand(q_pos, dpos),
                                           u.buf.data_ptr(), out.data_ptr(), out.stride(0), u.dv * out.stride(1),
                                           out.stride(1), B, heads, Lq, u.Lk0, u.Lk1, u.dqk, dpos, u.dv, float(scale),
                                           _stream()), "lc_attention_units_fwd")
    return out


# ------------------------------------------------------------------------------------ sampler
def pstep(x_t, pred, noise, coef, objective: int, mode: int, out=None) -> torch.Tensor:
    xb, pb = _bs4(x_t, "x_t"), _bs4(pred, "pred")
    B, C, H, W = x_t.shape
    if out is None:
        out = torch.empty((B, C, H, W), device=x_t.device, dtype=_F32)
    ob = _bs4(out, "out")
    _drop_stats(out)
    nb = _bs4(noise, "noise") if noise is not None else 0
    _req(coef, "coef")
    if tuple(coef.shape) != (B, 8) or not coef.is_contiguous():
        raise ValueError("pstep: coef must be contiguous [B,8]")
    check(lib().lc_pstep_fwd(x_t.data_ptr(), xb, pred.data_ptr(), pb, _p(noise), nb,
                             coef.data_ptr(), out.data_ptr(), ob, B, C * H * W, objective, mode,
                             _stream()), "lc_pstep_fwd")
    return out


def gate_bias_act(x: torch.Tensor, gate_logit: torch.Tensor, bias: torch.Tensor, leaky: bool,
                  res: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """PCNet epilogue (reference point_unet.py:22-26) on channel-major rows x [B, C, N]:
    y = act(x * sigmoid(gate_logit[b, c]) + bias[b, c]) [+ res]; gate_logit / bias: [B, C] views of
    one wider projection (equal row stride, unit inner stride)."""
    _req(x, "x"), _req(gate_logit, "gate_logit"), _req(bias, "bias")
    if x.dim() != 3 or not x.is_contiguous():
        raise ValueError("gate_bias_act: x must be contiguous [B, C, N]")
    B, C, N = x.shape
    if gate_logit.shape != (B, C) or bias.shape != (B, C) or gate_logit.stride(1) != 1 or \
            bias.stride(1) != 1 or gate_logit.stride(0) != bias.stride(0):
        raise ValueError("gate_bias_act: gate_logit / bias must be [B, C] with equal row strides")
    if res is not None:
        _req(res, "res")
        if res.shape != x.shape or not res.is_contiguous():
            raise ValueError("gate_bias_act: res must be contiguous and shaped like x")
    if out is None:
        out = torch.empty_like(x)
    check(lib().lc_gate_bias_act(x.data_ptr(), C * N, gate_logit.data_ptr(), bias.data_ptr(),
                                 gate_logit.stride(0), _p(res), C * N, out.data_ptr(), C * N, B, C,
                                 N, 1 if leaky else 0, _stream()), "lc_gate_bias_act")
    return out


def copy_into(dst: torch.Tensor, src: torch.Tensor) -> torch.Tensor:
    db, sb = _bs4(dst, "dst"), _bs4(src, "src")
    if dst.shape != src.shape:
        raise ValueError("copy_into: shape mismatch")
    B, C, H, W = src.shape
    _drop_stats(dst)
    check(lib().lc_copy_strided(src.data_ptr(), sb, dst.data_ptr(), db, B, C * H * W, _stream()),
          "lc_copy_strided")
    return dst


def add_scale(a: torch.Tensor, b: torch.Tensor, scale: float, out=None) -> torch.Tensor:
    ab, bb = _bs4(a, "a"), _bs4(b, "b")
    B, C, H, W = a.shape
    if out is None:
        out = torch.empty((B, C, H, W), device=a.device, dtype=_F32)
    ob = _bs4(out, "out")
    _drop_stats(out)
    check(lib().lc_add_scale(a.data_ptr(), ab, b.data_ptr(), bb, out.data_ptr(), ob, B, C * H * W,
                             float(scale), _stream()), "lc_add_scale")
    return out


# ------------------------------------------------------------------------------------ geometry
# Arithmetic of the elevation -> image row step: "native" = float64 on the float32 asin, what the
# reference computes under numpy >= 2 (and what the committed reference fixtures hold); "f32" = the
# all-float32 flow of the reference's pinned numpy 1.23.5 (environment.yml:233).
PROJECTION_DTYPE = _os.environ.get("LC_PROJECTION_DTYPE", "native")


import threading as _threading

_proj_ws = {}          # (device, stream, H * W, thread) -> self-emptying z-buffer
_PROJ_WS_MAX = 16


def project_points(points: torch.Tensor, H: int, W: int, fov_up: float, fov_down: float,
                   min_depth: float, max_depth: float, return_cells: bool = False,
                   dtype_mode: Optional[str] = None):
    """[N,4] (x,y,z,intensity) -> image [H,W,6], winner int32 [H,W] (+ cells int32 [N,2]).
    dtype_mode: "native" (default) | "f32", see PROJECTION_DTYPE."""
    mode = dtype_mode or PROJECTION_DTYPE
    if mode not in ("native", "f32"):
        raise ValueError(f"projection dtype mode {mode!r}")
    if isinstance(points, torch.Tensor) and points.is_cuda and points.dtype == torch.float64:
        # float64 point sets of the temporal glue: the reference projects them in float64
        if return_cells:
            raise ValueError("project_points: return_cells is not available for float64 points")
        if points.dim() != 2 or points.shape[1] != 4 or not points.is_contiguous():
            raise ValueError("project_points: points must be contiguous [N,4]")
        N, dev = points.shape[0], points.device
        zbuf = torch.empty(H * W, device=dev, dtype=torch.int64)
        img = torch.empty((H, W, 6), device=dev, dtype=_F32)
        win = torch.empty((H, W), device=dev, dtype=torch.int32)
        check(lib().lc_project_points_f64(points.data_ptr(), N, H, W, float(fov_up), float(fov_down),
                                          float(min_depth), float(max_depth), zbuf.data_ptr(),
                                          win.data_ptr(), img.data_ptr(), _stream()),
              "lc_project_points_f64")
        return img, win
    _req(points, "points")
    if points.dim() != 2 or points.shape[1] != 4 or not points.is_contiguous():
        raise ValueError("project_points: points must be contiguous [N,4]")
    N = points.shape[0]
    dev = points.device
    # z-buffer: one per (device, stream, H * W), emptied once; every projection hands it back empty
    # (lc_project_points_ws): two launches and no scratch allocation per call
    # The cached buffer is shared state: it is keyed by the calling THREAD as well (two host threads on one stream would
    # otherwise interleave scatter A, scatter B, gather A, gather B on it), never used while the stream is being
    # captured (a replayed graph would race with eager calls on the same buffer: a capture gets a private one, owned by
    # the graph's pool), and the cache holds at most _PROJ_WS_MAX entries (streams come and go; oldest evicted).
    st = _stream()
    capturing = torch.cuda.is_current_stream_capturing()
    key = (dev, st, H * W, _threading.get_ident())
    zbuf = None if capturing else _proj_ws.get(key)
    if zbuf is None:
        with torch.inference_mode(False):
            zbuf = torch.empty(H * W, device=dev, dtype=torch.int64)
        check(lib().lc_project_workspace_init(zbuf.data_ptr(), H * W, st), "lc_project_workspace_init")
        if not capturing:
            while len(_proj_ws) >= _PROJ_WS_MAX:
                _proj_ws.pop(next(iter(_proj_ws)))
            _proj_ws[key] = zbuf
    img = torch.empty((H, W, 6), device=dev, dtype=_F32)
    win = torch.empty((H, W), device=dev, dtype=torch.int32)
    cells = torch.empty((N, 2), device=dev, dtype=torch.int32) if return_cells else None
    check(lib().lc_project_points_ws(points.data_ptr(), N, H, W, float(fov_up), float(fov_down),
                                     float(min_depth), float(max_depth), zbuf.data_ptr(),
                                     img.data_ptr(), win.data_ptr(), _p(cells),
                                     1 if mode == "native" else 0, st),
          "lc_project_points_ws")
    return (img, win, cells) if return_cells else (img, win)


PIB_MAX_BOXES = 1250   # csrc/geometry.hip: 12 floats per box in LDS (60 000 bytes); larger sets run in slabs


def points_in_boxes_mask(points: torch.Tensor, boxes: torch.Tensor, margin: float) -> torch.Tensor:
    _req(points, "points"), _req(boxes, "boxes")
    points, boxes = points.contiguous(), boxes.contiguous()
    K = boxes.shape[0]
    out = torch.empty((K, points.shape[0]), device=points.device, dtype=torch.int32)
    for k0 in range(0, K, PIB_MAX_BOXES):                  # rows of `out` are per box: slabs are independent
        k1 = min(K, k0 + PIB_MAX_BOXES)
        check(lib().lc_points_in_boxes_mask(boxes[k0:k1].data_ptr(), k1 - k0, points.data_ptr(),
                                            points.shape[0], float(margin), out[k0:k1].data_ptr(), _stream()),
              "lc_points_in_boxes_mask")
    return out


def points_in_boxes_index(points: torch.Tensor, boxes: torch.Tensor, margin: float) -> torch.Tensor:
    _req(points, "points"), _req(boxes, "boxes")
    points, boxes = points.contiguous(), boxes.contiguous()
    B, M, _ = points.shape
    K = boxes.shape[1]
    out = torch.empty((B, M), device=points.device, dtype=torch.int32)
    if K <= PIB_MAX_BOXES:
        check(lib().lc_points_in_boxes_index(boxes.data_ptr(), points.data_ptr(), B, K, M,
                                             float(margin), out.data_ptr(), _stream()),
              "lc_points_in_boxes_index")
        return out
    out.fill_(-1)                                          # first containing box wins (roiaware_pool3d_kernel.cu:313-336)
    part = torch.empty_like(out)
    for k0 in range(0, K, PIB_MAX_BOXES):
        slab = boxes[:, k0:k0 + PIB_MAX_BOXES].contiguous()
        check(lib().lc_points_in_boxes_index(slab.data_ptr(), points.data_ptr(), B, slab.shape[1], M,
                                             float(margin), part.data_ptr(), _stream()),
              "lc_points_in_boxes_index")
        torch.where((out < 0) & (part >= 0), part + k0, out, out=out)
    return out


# ------------------------------------------------------------------------------------ temporal glue
def _pts4(t: torch.Tensor, name: str) -> int:
    _req(t, name)
    if t.dim() != 2 or t.shape[1] != 4 or not t.is_contiguous():
        raise ValueError(f"{name}: expected contiguous [N,4] points, got {tuple(t.shape)}")
    return t.shape[0]


def transform_points(points: torch.Tensor, T, out: Optional[torch.Tensor] = None,
                     f64: Optional[str] = None) -> torch.Tensor:
    """[N,4] rows (x,y,z,intensity) -> (T[:3,:3] p + T[:3,3], intensity); T: 4x4 host matrix
    (numpy / nested list / CPU tensor), product in fp64 rounded once (pipe_related.py:245-249).
    f64 = "keep": float64 rows out, not rounded (the reference's float64 `Ts @ homo`);
    f64 = "rot32": float64 rows (double)(float)(R p) + t (float32 rotation + float64 centre,
    pipe_related.py:263-266)."""
    import ctypes as C
    import numpy as np

    N = _pts4(points, "points")
    T = np.ascontiguousarray(np.asarray(T, dtype=np.float64))
    if T.shape != (4, 4):
        raise ValueError("transform_points: T must be 4x4")
    if f64 is not None:
        if f64 not in ("keep", "rot32") or out is not None:
            raise ValueError("transform_points: f64 must be 'keep' or 'rot32' (no out=)")
        o64 = torch.empty((N, 4), device=points.device, dtype=torch.float64)
        check(lib().lc_transform_points_f64(points.data_ptr(), N, T.ctypes.data_as(C.POINTER(C.c_double)),
                                            1 if f64 == "rot32" else 0, o64.data_ptr(), _stream()),
              "lc_transform_points_f64")
        return o64
    if out is None:
        out = torch.empty_like(points)
    elif _pts4(out, "out") != N:
        raise ValueError("transform_points: out shape mismatch")
    check(lib().lc_transform_points(points.data_ptr(), N, T.ctypes.data_as(C.POINTER(C.c_double)),
                                    out.data_ptr(), _stream()), "lc_transform_points")
    return out


def image_to_points(xyz: torch.Tensor, refl: Optional[torch.Tensor] = None,
                    cond: Optional[torch.Tensor] = None, refl_scale: float = 1.0,
                    min_norm: float = -1.0, ego_radius: float = 0.0):
    """xyz [3,H,W] (+ refl [H,W] / [1,H,W]) -> (rows [H*W,4], keep int32 [H*W]); `cond` [H,W]:
    pixels with cond > 0 are zeroed (foreground removed).  See lc_image_to_points."""
    _req(xyz, "xyz")
    if xyz.dim() != 3 or xyz.shape[0] != 3 or xyz.stride(2) != 1 or xyz.stride(1) != xyz.shape[2]:
        raise ValueError("image_to_points: xyz must be [3,H,W] with contiguous planes")
    H, W = xyz.shape[1:]

    def plane(t, name):
        if t is None:
            return None
        _req(t, name)
        t = t.reshape(H, W)
        if not t.is_contiguous():
            raise ValueError(f"image_to_points: {name} must be a contiguous [H,W] plane")
        return t

    refl, cond = plane(refl, "refl"), plane(cond, "cond")
    pts = torch.empty((H * W, 4), device=xyz.device, dtype=_F32)
    keep = torch.empty((H * W,), device=xyz.device, dtype=torch.int32)
    check(lib().lc_image_to_points(xyz.data_ptr(), xyz.stride(0), _p(refl), _p(cond), H, W,
                                   float(refl_scale), float(min_norm), float(ego_radius),
                                   pts.data_ptr(), keep.data_ptr(), _stream()), "lc_image_to_points")
    return pts, keep


def points_in_boxes_mask4(points: torch.Tensor, boxes: torch.Tensor, margin: float = 1e-2,
                          want_mask: bool = True, want_count: bool = True):
    """points_in_boxes_cpu semantics on [N,4] rows: (mask int32 [K,N] | None, count int32 [N] | None)."""
    N = _pts4(points, "points")
    _req(boxes, "boxes")
    boxes = boxes.contiguous()
    K = boxes.shape[0]
    mask = torch.empty((K, N), device=points.device, dtype=torch.int32) if want_mask else None
    cnt = torch.empty((N,), device=points.device, dtype=torch.int32) if want_count else None
    if N > 0 and 0 < K <= PIB_MAX_BOXES:
        check(lib().lc_points_in_boxes_mask4(boxes.data_ptr(), K, points.data_ptr(), N, float(margin),
                                             _p(mask), _p(cnt), _stream()), "lc_points_in_boxes_mask4")
    elif N > 0 and K > 0:                                  # slabs of boxes: mask rows are per box, the counts add up
        if cnt is not None:
            cnt.zero_()
            part = torch.empty_like(cnt)
        for k0 in range(0, K, PIB_MAX_BOXES):
            k1 = min(K, k0 + PIB_MAX_BOXES)
            check(lib().lc_points_in_boxes_mask4(boxes[k0:k1].data_ptr(), k1 - k0, points.data_ptr(), N, float(margin),
                                                 _p(mask[k0:k1]) if mask is not None else None,
                                                 _p(part) if cnt is not None else None, _stream()),
                  "lc_points_in_boxes_mask4")
            if cnt is not None:
                cnt += part
    elif cnt is not None:
        cnt.zero_()
    return mask, cnt


def compact_points(rows: torch.Tensor, keep: torch.Tensor, keep_if_zero: bool = False,
                   return_index: bool = False):
    """rows[keep != 0] (or == 0) in input order, like numpy boolean indexing.  One 4-byte
    device->host read (the number of kept rows) per call."""
    N = _pts4(rows, "rows")
    _req_i32 = keep.is_cuda and keep.dtype == torch.int32 and keep.is_contiguous() and keep.numel() == N
    if not _req_i32:
        raise ValueError("compact_points: keep must be a contiguous int32 device tensor of N flags")
    out = torch.empty_like(rows)
    idx = torch.empty((max(N, 1),), device=rows.device, dtype=torch.int32) if return_index else None
    count = torch.empty((1,), device=rows.device, dtype=torch.int32)
    scratch = torch.empty((lib().lc_compact_scratch_elems(N),), device=rows.device, dtype=torch.int32)
    check(lib().lc_compact_points(rows.data_ptr(), keep.data_ptr(), N, int(keep_if_zero),
                                  out.data_ptr(), _p(idx), count.data_ptr(), scratch.data_ptr(),
                                  _stream()), "lc_compact_points")
    n = int(count.item())
    return (out[:n], idx[:n]) if return_index else out[:n]


# ------------------------------------------------------------------------------------ BEV metrics
_hist_edges = {}


def bev_histogram(points: torch.Tensor, field_size: float = 160.0, bins: int = 100,
                  min_depth: float = 3.0, max_depth: float = 70.0) -> torch.Tensor:
    """[N, >=3] device points -> float32 [bins, bins] counts of (x, y), torch.histogramdd rule
    (lidargen/metrics/bev.py:5-24).  The bin edges are the ones torch itself computes."""
    _req(points, "points")
    if points.dim() != 2 or points.shape[1] < 3 or not points.is_contiguous():
        raise ValueError("bev_histogram: points must be contiguous [N, >=3]")
    bound = field_size / 2
    key = (float(bound), int(bins), points.device)
    e = _hist_edges.get(key)
    if e is None:
        e = torch.histogramdd(torch.empty(0, 2), bins=bins,
                              range=[-bound, bound, -bound, bound]).bin_edges[0]
        e = e.float().contiguous().to(points.device)
        _hist_edges[key] = e
    hist = torch.empty((bins, bins), device=points.device, dtype=_F32)
    scratch = torch.empty((bins * bins,), device=points.device, dtype=torch.int32)
    check(lib().lc_bev_histogram(points.data_ptr(), points.shape[1], points.shape[0], e.data_ptr(),
                                 bins, float(min_depth), float(max_depth), hist.data_ptr(),
                                 scratch.data_ptr(), _stream()), "lc_bev_histogram")
    return hist


class BevOccupancy:
    """Accumulator of pcd2bev_sum (reference metric_utils.py:233-258): `add(points)` scatters one
    sweep ([N, >=2] CUDA float32 rows, x / y first) into the [nx, ny] grid -- every voxel a sweep
    touches counts once -- with lc_bev_occupancy_accumulate; `grid` is the float32 volume sum."""

    def __init__(self, x_range, y_range, voxel_size: float, device):
        import math

        self.x_range, self.y_range, self.voxel = x_range, y_range, float(voxel_size)
        self.nx = math.ceil((x_range[1] - x_range[0]) / voxel_size)
        self.ny = math.ceil((y_range[1] - y_range[0]) / voxel_size)
        self.min_bound = (math.ceil(x_range[0] / voxel_size), math.ceil(y_range[0] / voxel_size))
        self.grid = torch.zeros((self.nx, self.ny), device=device, dtype=_F32)
        self._stamps = torch.zeros(self.nx * self.ny, device=device, dtype=torch.int32)
        self._n = 0

    def add(self, points: torch.Tensor) -> None:
        _req(points, "points")
        if points.dim() != 2 or points.shape[1] < 2 or not points.is_contiguous():
            raise ValueError("BevOccupancy.add: points must be contiguous [N, >=2]")
        self._n += 1
        with torch.cuda.device(points.device):
            check(lib().lc_bev_occupancy_accumulate(
                points.data_ptr(), points.shape[1], points.shape[0], float(self.x_range[0]),
                float(self.x_range[1]), float(self.y_range[0]), float(self.y_range[1]), self.voxel,
                self.min_bound[0], self.min_bound[1], self.nx, self.ny, self._n,
                self._stamps.data_ptr(), self.grid.data_ptr(), _stream()),
                "lc_bev_occupancy_accumulate")


def sparse_quantize(coords: torch.Tensor, voxel_size=1.0, return_index: bool = False,
                    return_inverse: bool = False):
    """Device `sparse_quantize` (reference metric_utils.py:43-66): coords [N, 2|3] float32 ->
    unique int32 voxels in ravel-hash order (+ first-occurrence indices, + inverse map, int64 like
    np.unique).  One 8-byte device->host read for the number of unique voxels."""
    _req(coords, "coords")
    if coords.dim() != 2 or coords.shape[1] not in (2, 3) or not coords.is_contiguous():
        raise ValueError("sparse_quantize: coords must be contiguous [N, 2] or [N, 3]")
    N, D = coords.shape
    vs = (float(voxel_size),) * D if isinstance(voxel_size, (int, float)) else tuple(map(float, voxel_size))
    if len(vs) != D:
        raise ValueError("sparse_quantize: voxel_size must have one entry per coordinate")
    dev = coords.device
    if N == 0:
        outs = [torch.empty((0, D), device=dev, dtype=torch.int32)]
        outs += [torch.empty(0, device=dev, dtype=torch.int64)] * (int(return_index) + int(return_inverse))
        return outs[0] if len(outs) == 1 else outs
    scratch = torch.empty(int(lib().lc_sparse_quantize_scratch_bytes(N, D)), device=dev, dtype=torch.uint8)
    oc = torch.empty((N, D), device=dev, dtype=torch.int32)
    oi = torch.empty(N, device=dev, dtype=torch.int64) if return_index else None
    ov = torch.empty(N, device=dev, dtype=torch.int64) if return_inverse else None
    cnt = torch.zeros(1, device=dev, dtype=torch.int64)
    check(lib().lc_sparse_quantize(coords.data_ptr(), N, D, vs[0], vs[1], vs[2] if D == 3 else 1.0,
                                   scratch.data_ptr(), oc.data_ptr(), _p(oi), _p(ov), cnt.data_ptr(),
                                   _stream()), "lc_sparse_quantize")
    n = int(cnt.item())
    outs = [oc[:n]]
    if return_index:
        outs.append(oi[:n])
    if return_inverse:
        outs.append(ov)
    return outs[0] if len(outs) == 1 else outs


def rbf_kernel_mean(p: torch.Tensor, q: torch.Tensor, sigma: float = 0.5) -> torch.Tensor:
    """mean_ij exp(-|p_i - q_j|^2 / (2 sigma^2)) as a 0-d float64 device tensor (bev.py:27-34)."""
    _req(p, "p"), _req(q, "q")
    if p.dim() != 2 or q.dim() != 2 or p.shape[1] != q.shape[1] or not p.is_contiguous() or \
            not q.is_contiguous():
        raise ValueError("rbf_kernel_mean: p [M,D] and q [Mq,D] contiguous")
    M, Mq, D = p.shape[0], q.shape[0], p.shape[1]
    part = torch.empty((lib().lc_rbf_partials_elems(M, Mq),), device=p.device, dtype=torch.float64)
    check(lib().lc_rbf_kernel_sum(p.data_ptr(), q.data_ptr(), M, Mq, D, 1.0 / (2.0 * sigma ** 2),
                                  part.data_ptr(), _stream()), "lc_rbf_kernel_sum")
    return part.sum() / (M * Mq)


def chamfer3d(xyz1: torch.Tensor, xyz2: torch.Tensor):
    """xyz1 [B,N,3], xyz2 [B,M,3] -> (dist1 [B,N], dist2 [B,M], idx1 int32 [B,N], idx2 int32 [B,M]):
    squared nearest-neighbour distances both ways (dist_chamfer_3D.py:27-49)."""
    _req(xyz1, "xyz1"), _req(xyz2, "xyz2")
    if xyz1.dim() != 3 or xyz2.dim() != 3 or xyz1.shape[2] != 3 or xyz2.shape[2] != 3 or \
            xyz1.shape[0] != xyz2.shape[0]:
        raise ValueError("chamfer3d: expected [B,N,3] and [B,M,3]")
    xyz1, xyz2 = xyz1.contiguous(), xyz2.contiguous()
    B, N, _ = xyz1.shape
    M = xyz2.shape[1]
    dev = xyz1.device
    d1 = torch.empty((B, N), device=dev, dtype=_F32)
    d2 = torch.empty((B, M), device=dev, dtype=_F32)
    i1 = torch.empty((B, N), device=dev, dtype=torch.int32)
    i2 = torch.empty((B, M), device=dev, dtype=torch.int32)
    check(lib().lc_chamfer3d_fwd(xyz1.data_ptr(), xyz2.data_ptr(), B, N, M, d1.data_ptr(),
                                 i1.data_ptr(), d2.data_ptr(), i2.data_ptr(), _stream()),
          "lc_chamfer3d_fwd")
    return d1, d2, i1, i2


def roiaware_pool3d_forward(rois, pts, pts_feature, out_size, max_pts_each_voxel: int, method: int):
    """-> (pooled [N,X,Y,Z,C], pts_idx_of_voxels int32 [N,X,Y,Z,max_pts], argmax int32 [N,X,Y,Z,C])."""
    for n_, t_ in (("rois", rois), ("pts", pts), ("pts_feature", pts_feature)):
        _req(t_, n_)
    rois, pts, pts_feature = rois.contiguous(), pts.contiguous(), pts_feature.contiguous()
    ox, oy, oz = out_size
    N, P, C = rois.shape[0], pts.shape[0], pts_feature.shape[1]
    dev = rois.device
    pooled = torch.zeros((N, ox, oy, oz, C), device=dev, dtype=_F32)
    argmax = torch.zeros((N, ox, oy, oz, C), device=dev, dtype=torch.int32)
    vox = torch.zeros((N, ox, oy, oz, max_pts_each_voxel), device=dev, dtype=torch.int32)
    scratch = torch.empty((N, P), device=dev, dtype=torch.int32)
    check(lib().lc_roiaware_pool3d_fwd(rois.data_ptr(), pts.data_ptr(), pts_feature.data_ptr(), N,
                                       P, C, ox, oy, oz, max_pts_each_voxel, method,
                                       scratch.data_ptr(), vox.data_ptr(), argmax.data_ptr(),
                                       pooled.data_ptr(), _stream()), "lc_roiaware_pool3d_fwd")
    return pooled, vox, argmax


def roiaware_pool3d_backward(vox, argmax, grad_out, num_pts: int, method: int) -> torch.Tensor:
    _req(grad_out, "grad_out")
    grad_out = grad_out.contiguous()
    N, ox, oy, oz, C = grad_out.shape
    grad_in = torch.zeros((num_pts, C), device=grad_out.device, dtype=_F32)
    check(lib().lc_roiaware_pool3d_bwd(vox.data_ptr(), argmax.data_ptr(), grad_out.data_ptr(),
                                       grad_in.data_ptr(), N, C, ox, oy, oz, vox.shape[-1], method,
                                       _stream()), "lc_roiaware_pool3d_bwd")
    return grad_in


_DEPTH_FMT = {"log_depth": 0, "inverse_depth": 1, "depth": 2}


def range_postprocess(sample: torch.Tensor, ray_angles: torch.Tensor, depth_format: str,
                      min_depth: float, max_depth: float) -> torch.Tensor:
    """[B,2,H,W] normalised (depth, reflectance) -> [B,5,H,W] (metric depth, x, y, z, reflectance)."""
    sb = _bs4(sample, "sample")
    _req(ray_angles, "ray_angles")
    B, C, H, W = sample.shape
    if C != 2 or tuple(ray_angles.shape) != (1, 2, H, W) or not ray_angles.is_contiguous():
        raise ValueError("range_postprocess: sample [B,2,H,W], ray_angles contiguous [1,2,H,W]")
    out = torch.empty((B, 5, H, W), device=sample.device, dtype=_F32)
    check(lib().lc_range_postprocess(sample.data_ptr(), sb, ray_angles.data_ptr(), out.data_ptr(),
                                     B, H, W, _DEPTH_FMT[depth_format], float(min_depth),
                                     float(max_depth), _stream()), "lc_range_postprocess")
    return out


def condition_preprocess(condition_mask: torch.Tensor, num_classes: int, depth_format: str,
                         min_depth: float, max_depth: float, out=None) -> torch.Tensor:
    """[B,2,H,W] (class id, metric depth) -> [B,num_classes+1,H,W] one-hot ++ normalised depth."""
    cb = _bs4(condition_mask, "condition_mask")
    B, C, H, W = condition_mask.shape
    if C != 2:
        raise ValueError("condition_preprocess: condition_mask must be [B,2,H,W]")
    if out is None:
        out = torch.empty((B, num_classes + 1, H, W), device=condition_mask.device, dtype=_F32)
    ob = _bs4(out, "out")
    check(lib().lc_condition_preprocess(condition_mask.data_ptr(), cb, out.data_ptr(), ob, B, H, W,
                                        num_classes, _DEPTH_FMT[depth_format], float(min_depth),
                                        float(max_depth), _stream()), "lc_condition_preprocess")
    return out


def layout_condition(boxes: torch.Tensor, n_valid: torch.Tensor, H: int, W: int, fov_up: float,
                     fov_down: float, with_weight_map: bool = False):
    """boxes [B,T,>=8] (x,y,z,l,w,h,yaw,class), n_valid int32 [B] -> (corners_2d [B,T,4],
    condition_mask [B,2,H,W][, loss_weight_map [B,H,W]]).  float32 or float64 boxes: the kernel
    follows the dtype flow numpy gives the reference for each (lc_layout_condition)."""
    f64 = isinstance(boxes, torch.Tensor) and boxes.is_cuda and boxes.dtype == torch.float64
    if not f64:
        _req(boxes, "boxes")
    if boxes.dim() != 3 or boxes.shape[2] < 8 or not boxes.is_contiguous():
        raise ValueError("layout_condition: boxes must be contiguous [B,T,>=8]")
    if n_valid.dtype != torch.int32 or not n_valid.is_cuda:
        raise TypeError("layout_condition: n_valid must be a CUDA int32 tensor")
    B, T, S = boxes.shape
    dev = boxes.device
    scratch = torch.empty(lib().lc_layout_scratch_bytes(B, T), device=dev, dtype=torch.uint8)
    c2d = torch.empty((B, T, 4), device=dev, dtype=_F32)
    mask = torch.empty((B, 2, H, W), device=dev, dtype=_F32)
    wmap = torch.empty((B, H, W), device=dev, dtype=_F32) if with_weight_map else None
    check(lib().lc_layout_condition(boxes.data_ptr(), int(f64), S, n_valid.data_ptr(), B, T, H, W,
                                    float(fov_up), float(fov_down), scratch.data_ptr(),
                                    c2d.data_ptr(), mask.data_ptr(), _p(wmap), _stream()),
          "lc_layout_condition")
    return (c2d, mask, wmap) if with_weight_map else (c2d, mask)


# ------------------------------------------------------------------------------------ boundary
# every public op goes through `_entry` (device guard + up-cast of half-precision inputs)
def _wrap_public_ops():
    import types

    g = globals()
    skip = {"fuse_gn", "set_conv_precision", "range_poll", "range_checked", "defer_range_checks",
            "name_packed_convs", "rng_snapshot", "rng_restore", "run_range_safe", "can_presplit", "bump_epoch",
            "route_signature"}
    for name, obj in list(g.items()):
        if isinstance(obj, types.FunctionType) and not name.startswith("_") and name not in skip \
                and obj.__module__ == __name__ and not getattr(obj, "_lc_entry", False):
            g[name] = _entry(obj)


_wrap_public_ops()
