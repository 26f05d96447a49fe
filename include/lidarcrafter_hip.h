/*
 * lidarcrafter_hip.h -- C ABI of the MI355X (gfx950) hot path of LiDARCrafter's range-image
 * diffusion denoiser.  Built as lidarcrafter_amd/liblidarcrafter_hip.so by hipcc.
 *
 * Conventions (all entry points):
 *   - extern "C", return int: 0 = ok, >0 = hipError_t of the launch, <0 = LC_E* argument error.
 *   - raw DEVICE pointers + sizes + a hipStream_t passed as void* (NULL = default stream).
 *   - no allocation, no synchronisation, no host<->device copies inside: every call is legal
 *     under hipStreamBeginCapture (HIP graphs).
 *   - tensors are fp32, [B, C, H, W] with the inner [C, H, W] block contiguous; `*_bs` is the
 *     batch stride in ELEMENTS (>= C*H*W).  A channel slice of a wider buffer is therefore a
 *     valid tensor, which is how torch.cat([h, skip], 1) of the reference
 *     (efficient_unet.py:293-295, layout_unet_v1.py:893) is made free: producers write into
 *     slices of one pre-concatenated buffer.
 *
 * Each function cites the reference code it replaces (paths relative to /root/reference/).
 * The reference reaches these through ATen ops of its nn.Modules, not through an FFI of its
 * own; INTEGRATION.md shows the binding a maintainer of the reference would add.
 */
#ifndef LIDARCRAFTER_HIP_H
#define LIDARCRAFTER_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LC_OK 0
#define LC_EINVAL (-1)  /* bad size / null pointer */
#define LC_EUNSUP (-2)  /* shape outside what the kernels are instantiated for */

typedef void* lc_stream_t;

/* Library / device probe.  Returns the ABI version (this header = 5: the up-path fold's entry points, lc_up2_combine9_fwd / lc_split_act_fwd; 4: the
 * unit-form attention; 3: round 6 -- lc_conv2d_ring_f16x2_ps_fwd carries
 * gn_ostats_unit, the stride-2 / calibration entry points exist; 2: round 3 -- field-of-view arguments are doubles,
 * lc_layout_condition takes float32 or float64 boxes, float64 point sets).  Bumped whenever an exported signature
 * changes; lidarcrafter_amd/_lib.py refuses a library of another version. */
int lc_abi_version(void);
/* Writes gcnArchName of the current device into buf (NUL terminated). */
int lc_device_arch(char* buf, int buflen);
/* Loads the code objects of every translation unit of the library for the CURRENT device now (HIP defers that to a
 * unit's first kernel launch, i.e. into the first sampling step of a process).  Idempotent, no device work.
 * Returns LC_OK or -- like every launcher of this header when the HIP runtime refuses a launch -- the POSITIVE hipError_t
 * of the failing runtime call (LC_E* codes are negative); callers treat it as a warm-up that did not happen. */
int lc_load_code_objects(void);

/* ---------------------------------------------------------------------------------------------
 * Convolution: lidargen/models/unets/ops.py:149-173 (ops.Conv2d) with ops.py:32-49 (Pad,
 * ring=True: W circular, H zeros) for 3x3, and plain 1x1 (ring=False, padding 0).
 * Used by every conv of EfficientUNet (efficient_unet.py:79,89,94,141,179,260,271) and
 * LayoutUnetV1 (nn.py:34-44 conv_nd_range; Conv1d 1x1 projections layout_unet_v1.py:386-399).
 *
 * Weights are consumed in a PACKED layout wp[tap][Cip][Cop] (tap = ky*ks+kx, Cip = Ci rounded
 * up to 8, Cop = Co rounded up to 64, zero filled) produced once per weight version by
 * lc_pack_conv_weight from the checkpoint layout OIHW [Co][Ci][ks][ks].
 * Arithmetic: fp32 inputs, fp32 MFMA (v_mfma_f32_32x32x2_f32) accumulate == an fmaf chain.
 *   y = (conv(x) + bias [+ res]) * out_scale          (ResidualBlock efficient_unet.py:111-115)
 * ------------------------------------------------------------------------------------------- */
int64_t lc_packed_conv_weight_elems(int Co, int Ci, int ks);
int lc_pack_conv_weight(const float* w_oihw, float* wp, int Co, int Ci, int ks, lc_stream_t s);
int lc_conv2d_ring_fwd(const float* x, int64_t x_bs, const float* wp, const float* bias,
                       const float* res, int64_t res_bs, float* y, int64_t y_bs,
                       int B, int Ci, int Co, int H, int W, int ks, float out_scale,
                       int tile_cfg /* 0 = auto */, lc_stream_t s);

/* ---------------------------------------------------------------------------------------------
 * Same convolution on the f16 matrix cores with fp32-class accuracy ("f16x2 split"): operands are
 * split into hi+lo fp16 halves (activations on the fly, weights at pack time) and
 * xh*wh + xh*wl + xl*wh is accumulated in fp32 by three v_mfma_f32_32x32x16_f16 per block --
 * ~5e-7 relative error per product, 5.3x the fp32 matrix peak.  Packed weights: two planes
 * (hi, lo) of lc_packed_conv_weight_f16x2_elems() fp16 values each, layout [tap][Ci^16/8][Co^64][8].
 * Arguments otherwise identical to lc_conv2d_ring_fwd.
 * ------------------------------------------------------------------------------------------- */
int64_t lc_packed_conv_weight_f16x2_elems(int Co, int Ci, int ks);
/* wmeta: 4 device floats written here, read by lc_conv2d_ring_f16x2_fwd:
 *   {w_scale, 1 / w_scale, max |w|, 0}.  w_scale is the power of two the weights are multiplied by
 * before the hi/lo split: 256 while max|w| * 256 lies in [2^5, 2^15] (every trained layer seen so
 * far), otherwise the power of two that puts max|w| * w_scale into [2^12, 2^13) -- so weights of
 * any magnitude (1e-30 ... 1e30) keep 22 significant bits and none saturates fp16.  The scale is
 * derived on the device (no host synchronisation): an all-zero / non-finite tensor keeps 256. */
int lc_pack_conv_weight_f16x2(const float* w_oihw, void* wp_hi, void* wp_lo, int Co, int Ci, int ks,
                              float* wmeta, lc_stream_t s);
/* The packed weight of a layer's INPUT-GRADIENT conv (dX = the same ring conv of dY with the transposed, 180-degree
 * rotated kernel) straight from the layer's forward weight w_fwd [Cf_o][Cf_i][ks][ks] -- the rotation and transposition
 * happen in the pack kernel's addressing, no rotated copy exists.  Co / Ci are those of the dX conv (Co = Cf_i,
 * Ci = Cf_o).  wmeta_fwd (may be NULL) = the wmeta lc_pack_conv_weight_f16x2 wrote for the SAME version of w_fwd: its
 * max|w| is reused (one launch instead of memset + reduction + pack).  Training packs every weight twice per step
 * (lidarcrafter_amd/autograd.py ConvRing.backward). */
int lc_pack_conv_weight_f16x2_dx(const float* w_fwd, void* wp_hi, void* wp_lo, int Co, int Ci, int ks,
                                 float* wmeta, const float* wmeta_fwd, lc_stream_t s);
/* All conv weights of a model at once (training: every weight changes at every optimizer step -- two packs per layer
 * and step): `jobs` is a DEVICE array of n records; each names a forward weight w [Co][Ci][ks][ks] (contiguous), the
 * buffers of its own pack (as lc_pack_conv_weight_f16x2: planes of lc_packed_conv_weight_f16x2_elems(Co, Ci, ks)
 * halves, fwd_meta 4 floats) and, when with_dx != 0, of its input-gradient pack (as lc_pack_conv_weight_f16x2_dx:
 * planes of lc_packed_conv_weight_f16x2_elems(Ci, Co, ks) halves, dx_meta 4 floats).  Three launches in total, same
 * bytes as the per-layer functions. */
typedef struct lc_weight_pack_job {
    const float* w;
    void *fwd_hi, *fwd_lo;
    float* fwd_meta;
    void *dx_hi, *dx_lo;
    float* dx_meta;
    int Co, Ci, ks, reserved;
} lc_weight_pack_job;
int lc_pack_conv_weights_f16x2_multi(const lc_weight_pack_job* jobs, int n, int with_dx, lc_stream_t s);
/* Range state of ONE conv layer's input (device memory, 16 bytes, owned by the caller, initialise
 * with {16, 1/16, 0, 0}).  The kernel multiplies x by x_scale before the fp16 hi/lo split and
 * publishes the largest |x * x_scale| it staged (after the fused input GroupNorm, if any) into
 * amax_scaled with atomicMax -- fp16 saturates at 65504, so the caller must treat a launch whose
 * amax_scaled reached 2^15 as INVALID: pick x_scale = a power of two with max|x| * x_scale ~ 2^12,
 * reset amax_scaled and run the layer (and everything downstream) again.  lidarcrafter_amd.ops
 * does exactly that after every forward / sampling run (`range_poll`); nothing saturated is ever
 * returned silently.  amax_scaled accumulates over launches until the caller zeroes it. */
typedef struct lc_conv_range {
    float x_scale, x_unscale, amax_scaled, reserved;
} lc_conv_range;
/* The record of a tensor measured on the device: max |x| over B samples of n floats (sample stride
 * x_bs) -> x_scale = the power of two with max|x| * x_scale in [2^12, 2^13), amax_scaled = 0.  Two
 * tiny launches, no host synchronisation; `reserved` is the reduction's scratch word (0 between
 * calls).  Training uses it right before every f16x2 conv (activations in the forward, gradients in
 * the backward: lidarcrafter_amd/autograd.py), where a poll-and-repeat protocol is not available. */
int lc_range_from_tensor(const float* x, int64_t x_bs, int B, int64_t n, lc_conv_range* range,
                         lc_stream_t s);
/* The same record from the partial maxima the PRODUCER of x left while writing it (amax: n device floats whose maximum is
 * max|x| -- lc_groupnorm_apply_train / lc_groupnorm_bwd_train store one per block, no atomics) -- one single-block launch,
 * x is not read again.  bound_mult >= 1 (LC_EINVAL otherwise): the tensor the conv will read is an elementwise rescale of
 * the measured one by at most this factor (dropout: 1 / (1 - p)); max|x| * bound_mult is then an upper bound, which is
 * all the record needs. */
int lc_range_from_amax(const float* amax, int64_t n, float bound_mult, lc_conv_range* range, lc_stream_t s);
/* Producer-side GroupNorm statistics of one channel segment: the entries a conv wrote through
 * gn_ostats_out (below), consumed by lc_groupnorm_apply_os or by the next conv's fused input norm. */
typedef struct lc_oct_stats {
    const float* p;     /* [B, channels/unit, slots, 4] */
    int channels, slots;
    int unit;           /* channels per entry: 8 (octets: the conv epilogues), 2 (pairs: lc_conv2d_ring_f16x2_fwd with
                           gn_ostats_unit = 2, for a GroupNorm with 2 / 4 / 6 channels per group), 1 (single
                           channels: lc_resample2x_stats_fwd), 4.  Every consumer -- the conv's fused input norm,
                           lc_groupnorm_apply_os, lc_groupnorm_apply_os_split -- folds any of them as long as a
                           group is a whole number of entries (round 5; before: octets only in the apply passes) */
} lc_oct_stats;
/* Input normalisation straight from the statistics (no lc_groupnorm_coeffs launch): the partials
 * of lc_groupnorm_stats over the conv's input plus the GroupNorm / AdaGN parameters; every block
 * derives the rows of its sample in its prologue with the arithmetic of lc_groupnorm_coeffs.
 * Alternatively (partials == NULL, os0 != NULL) from the octet statistics the input's producer(s)
 * emitted -- then no statistics pass over the input exists at all; os0 covers channels
 * [0, os0->channels), os1 (may be NULL) the rest, as in lc_groupnorm_apply_os; needs
 * (Ci/G) % 8 == 0, every group inside one segment and G <= 128 (LC_EUNSUP otherwise). */
typedef struct lc_gn_stats_input {
    const double* partials;   /* lc_groupnorm_stats(x, ...) of THIS conv's input, or NULL */
    int G, nch;               /* groups; chunks per group = partials_elems / (2 * B * G) */
    float eps;
    const float *gamma, *beta, *scale, *shift;   /* each may be NULL, as in lc_groupnorm_apply */
    int64_t ss_bs;
    const lc_oct_stats *os0, *os1;               /* used when partials == NULL */
} lc_gn_stats_input;

/* Layout constraints of the pipelined tile configurations (12-28, 33; the configurations the heuristic picks for every
 * 3x3 conv with Ci >= 24): they fetch both weight planes by LDS-DMA through ONE buffer descriptor, so wp_lo must lie
 * ABOVE wp_hi at a distance below 2 GiB -- in practice one allocation holding both planes, which is what
 * lc_pack_conv_weight_f16x2's callers in lidarcrafter_amd.ops make (LC_EINVAL otherwise); and a sample of the input,
 * of the output and of the residual is addressed with 32-bit byte offsets: Ci*H*W*4, Co*H*W*4 < 2^31 (LC_EUNSUP).
 * tile_cfg 33 = the ping-pong kernel (csrc/conv_f16x2_pp.h): 3x3, Ci % 16 == 0, 64 <= Ci <= 512, Co % 64 == 0,
 * H % 4 == 0, W % 64 == 0 (LC_EUNSUP otherwise); the heuristic (tile_cfg 0) picks it only with LC_PP_MIN_STRIPS set.
 * tile_cfg 27 (round 5) = the tall kernel (csrc/conv_f16x2_tall.hip: halo rows kept in LDS across the block's walk down H,
 * 64-bit input loads): 3x3, Ci 32 or 64, H % 4 == 0, W % 64 == 0; any other shape given this id runs on the pipelined
 * 64 co x (4 x 64 px) tile (23), whose statistics partition it shares.  The heuristic picks 27 wherever it would pick 23 and
 * the shape qualifies (LC_TALL=0: never). */
int lc_conv2d_ring_f16x2_fwd(const float* x, int64_t x_bs, const void* wp_hi, const void* wp_lo,
                             const float* bias, const float* res, int64_t res_bs, float* y,
                             int64_t y_bs, int B, int Ci, int Co, int H, int W, int ks,
                             float out_scale, int tile_cfg /* 0 = auto */,
                             const float* gn_coeffs /* NULL or [B, gn_cpad, 4] */, int gn_cpad,
                             int gn_silu, const lc_gn_stats_input* gn_stats /* NULL, or instead of
                             gn_coeffs */,
                             float* gn_ostats_out /* NULL or [B, Co/unit, slots, 4], see below */,
                             int gn_ostats_unit /* 8 (octet entries) or 2 (pair entries, lc_oct_stats) */,
                             const float* wmeta /* from lc_pack_conv_weight_f16x2 */,
                             lc_conv_range* range /* this layer's input range state */,
                             lc_stream_t s);
/* Output statistics for the NEXT GroupNorm (every GroupNorm input of the denoiser is a conv
 * output, efficient_unet.py:101-108): with gn_ostats_out != NULL every wave of the pipelined kernel
 * also writes, per octet of 8 consecutive output channels and wave tile, the entry
 * (pivot, n, sum(y - pivot), sum((y - pivot)^2)) of the values it stored; lc_groupnorm_apply_os
 * folds them, so no statistics pass re-reads the tensor.  slots = entries per (sample, octet) for
 * this problem and tile configuration, 0 when the chosen kernel emits none (Co % 8 != 0, or the
 * 2-blocks/CU kernel): ask before allocating. */
int64_t lc_conv2d_ring_f16x2_stats_slots(int B, int Ci, int Co, int H, int W, int ks, int tile_cfg);
/* PRE-SPLIT activations.  The f16x2 kernels need every conv input as fp16 hi + lo halves; instead of
 * splitting the fp32 tile inside the conv's K loop (once per 64-output-channel block), the PRODUCER
 * of the tensor -- the GroupNorm apply pass in front of almost every conv of the denoisers
 * (efficient_unet.py:101-108, layout_unet_v1.py:171-175) -- can write the split form directly:
 *   y_split: 16-byte units (8 fp16 channels of one pixel), [B][plane = hi, lo][C/8][H][W]
 *            = lc_split_act_units(B, C, H, W) units, the same bytes as fp32 NCHW (needs C % 16 == 0;
 *            lc_groupnorm_apply_os_split also needs groups of whole octets -- LC_EUNSUP otherwise,
 *            use lc_groupnorm_apply_split or the fp32 forms then),
 * already multiplied by the CONSUMER layer's x_scale (`range`, whose amax_scaled it maintains: the
 * range-safety contract of lc_conv_range moves to the producer).  lc_conv2d_ring_f16x2_ps_fwd then
 * stages its tiles with LDS-DMA (`buffer_load_dwordx4 ... lds`): no VGPRs, no VALU, no ds_write in
 * the K loop.  3x3 ring convolution, pipelined tile shapes (tile_cfg 0 = auto, 12/13/15/22/23/25/28);
 * wp_lo must be wp_hi + one plane (ONE allocation holding both planes of
 * lc_pack_conv_weight_f16x2).  Everything else as lc_conv2d_ring_f16x2_fwd. */
int64_t lc_split_act_units(int B, int C, int H, int W);
int lc_groupnorm_apply_split(const float* x, int64_t x_bs, const double* partials, const float* gamma,
                             const float* beta, const float* scale, const float* shift, int64_t ss_bs,
                             void* y_split, int B, int C, int H, int W, int G, float eps, int act_silu,
                             lc_conv_range* range, lc_stream_t s);
int lc_groupnorm_apply_os_split(const float* x, int64_t x_bs, const lc_oct_stats* s0,
                                const lc_oct_stats* s1, const float* gamma, const float* beta,
                                const float* scale, const float* shift, int64_t ss_bs, void* y_split,
                                int B, int C, int H, int W, int G, float eps, int act_silu,
                                lc_conv_range* range, lc_stream_t s);
int lc_conv2d_ring_f16x2_ps_fwd(const void* x_split, const void* wp_hi, const void* wp_lo,
                                const float* bias, const float* res, int64_t res_bs, float* y,
                                int64_t y_bs, int B, int Ci, int Co, int H, int W, float out_scale,
                                int tile_cfg, float* gn_ostats_out /* NULL, or [B, Co / unit, slots, 4] */,
                                int gn_ostats_unit /* 8 = octet entries, 4 = quad entries (else LC_EUNSUP) */,
                                float* splitk_part /* NULL, or [ksplit, B, Co, H, W] */, int ksplit,
                                const float* wmeta, lc_conv_range* range, lc_stream_t s);
/* 1x1 convolution of a pre-split activation (the same planes; H * W is one pixel axis): LDS-DMA staging, 128 output
 * channels x 256 pixels per block -- for projections with many output channels, where the fp32-input kernel splits the
 * same input tile once per 64-channel block (qkv projections of the layout model).  Needs Ci % 32 == 0 (LC_EUNSUP
 * otherwise); weights: the ks = 1 pack, lo plane directly behind the hi plane; no statistics output, no split-K. */
int lc_conv1x1_f16x2_ps_fwd(const void* x_split, const void* wp_hi, const void* wp_lo, const float* bias,
                            const float* res, int64_t res_bs, float* y, int64_t y_bs, int B, int Ci, int Co,
                            int H, int W, float out_scale, const float* wmeta, lc_conv_range* range,
                            lc_stream_t s);
/* SPLIT-K for small grids (batch 1-2 at the deep levels: 16-64 blocks on 256 CUs): with splitk_part
 * != NULL the conv launches ksplit (2 ... Ci/16) blocks per tile, each over a contiguous range of
 * the K chunks, and stores raw partial sums only (bias / res / out_scale / statistics are ignored);
 * lc_splitk_reduce sums the planes in index order (deterministic) and applies the epilogue, with
 * optional GroupNorm statistics of the result in the conv epilogue's entry format
 * (lc_splitk_stats_slots(H, W) entries per (sample, octet)). */
int64_t lc_splitk_stats_slots(int H, int W);
int lc_splitk_reduce(const float* part, int ksplit, const float* bias, const float* res,
                     int64_t res_bs, float* y, int64_t y_bs, int B, int Co, int H, int W,
                     float out_scale, float* gn_ostats_out, lc_stream_t s);

/* Fused input normalisation: with gn_coeffs != NULL the kernel applies
 *   x <- silu?( (x - mu) * A + Bc )       rows (mu, A, Bc, 0) from lc_groupnorm_coeffs
 * while staging the input tile (the GN -> SiLU -> Conv chain of efficient_unet.py:101-108 and
 * layout_unet_v1.py:171-175 becomes stats + conv; the normalised tensor never reaches HBM).
 * gn_cpad = channels per sample in the table, >= Ci rounded up to 16. */
int lc_groupnorm_coeffs(const float* x, int64_t x_bs, const double* partials, const float* gamma,
                        const float* beta, const float* scale, const float* shift, int64_t ss_bs,
                        float* coeffs, int B, int C, int Cpad, int H, int W, int G, float eps,
                        lc_stream_t s);

/* ---------------------------------------------------------------------------------------------
 * GroupNorm (+ affine | + AdaGN scale/shift) (+ SiLU):
 *   nn.GroupNorm(8, C, 1e-6) efficient_unet.py:37,77; ops.AdaGN ops.py:176-200;
 *   GroupNorm32 nn.py:17-19 + scale-shift norm layout_unet_v1.py:243-245; nn.SiLU.
 * Two launches: lc_groupnorm_stats writes per-(b,g,chunk) fp64 partial (sum, sumsq) into
 * `partials` (>= lc_groupnorm_partials_elems doubles); lc_groupnorm_apply folds them in a fixed
 * order (deterministic), then
 *   y = ((x-mean)*rstd * gamma[c] + beta[c]) * (1 + scale[b,c]) + shift[b,c] ; y = silu(y) if act
 * gamma/beta may be NULL (affine=False); scale/shift may be NULL; ss_bs = batch stride of
 * scale/shift rows in elements.
 * ------------------------------------------------------------------------------------------- */
int64_t lc_groupnorm_partials_elems(int B, int C, int H, int W, int G);
int lc_groupnorm_stats(const float* x, int64_t x_bs, double* partials, int B, int C, int H, int W,
                       int G, lc_stream_t s);
int lc_groupnorm_apply(const float* x, int64_t x_bs, const double* partials, const float* gamma,
                       const float* beta, const float* scale, const float* shift, int64_t ss_bs,
                       float* y, int64_t y_bs, int B, int C, int H, int W, int G, float eps,
                       int act_silu, lc_stream_t s);
/* The apply pass as the training graph runs it: additionally writes the (mean, rstd) it normalised with into
 * mean_rstd_out [B, G, 2] (may be NULL; what lc_groupnorm_bwd* reads -- no lc_groupnorm_meanrstd launch) and stores max|y|
 * of every block of the pass into amax_out[0 .. lc_groupnorm_amax_partials(B, C, H, W, G, 0)) (may be NULL; every
 * element is written, nothing to initialise) for lc_range_from_amax of the conv that consumes y. */
int64_t lc_groupnorm_amax_partials(int B, int C, int H, int W, int G, int backward);
int lc_groupnorm_apply_train(const float* x, int64_t x_bs, const double* partials, const float* gamma,
                             const float* beta, const float* scale, const float* shift, int64_t ss_bs,
                             float* y, int64_t y_bs, int B, int C, int H, int W, int G, float eps,
                             int act_silu, float* mean_rstd_out, float* amax_out, lc_stream_t s);

/* The same normalisation from the PRODUCER's octet statistics (lc_conv2d_ring_f16x2_fwd
 * gn_ostats_out): one launch, the tensor is read once.  A tensor may be the channel concatenation of
 * two producers' outputs (torch.cat([h, skip]) efficient_unet.py:293-295): s0 covers channels
 * [0, s0->channels), s1 (may be NULL) the rest.  Needs (C/G) % 8 == 0 and every group inside one
 * segment (LC_EUNSUP otherwise: use stats + apply).  The fold is fp64 around one common pivot per
 * group in a fixed order (deterministic). */
int lc_groupnorm_apply_os(const float* x, int64_t x_bs, const lc_oct_stats* s0, const lc_oct_stats* s1,
                          const float* gamma, const float* beta, const float* scale,
                          const float* shift, int64_t ss_bs, float* y, int64_t y_bs, int B, int C,
                          int H, int W, int G, float eps, int act_silu, lc_stream_t s);

/* ---------------------------------------------------------------------------------------------
 * FIR resampling x2, window [1,3,3,1], ring=True: ops.Resample ops.py:52-146
 * (closed forms SURVEY.md §8a-10).  dir = +1 upsample (H,W -> 2H,2W), -1 downsample.
 * ------------------------------------------------------------------------------------------- */
int lc_resample2x_fwd(const float* x, int64_t x_bs, float* y, int64_t y_bs, int B, int C, int H,
                      int W, int dir, lc_stream_t s);
/* Round 6: a resampling ResBlock of the layout model (layout_unet_v1.py:81-150: h = op(SiLU(GroupNorm(x))), x = op(x)) in ONE
 * pass over x.  lc_groupnorm_coeffs_os: the rows (mu, A, Bc, 0) [B][Cpad] of lc_groupnorm_coeffs, derived from the producer's
 * statistics entries (no statistics pass); lc_resample2x_pair_fwd: y = resample(x) (bit-identical to lc_resample2x_fwd) and
 * y_act = resample(SiLU((x - mu) * A + Bc)). */
int lc_groupnorm_coeffs_os(const lc_oct_stats* s0, const lc_oct_stats* s1, const float* gamma, const float* beta,
                           const float* scale, const float* shift, int64_t ss_bs, float* coeffs, int B, int C, int Cpad,
                           int G, float eps, lc_stream_t s);
int lc_resample2x_pair_fwd(const float* x, int64_t x_bs, const float* coeffs, int Cpad, float* y, int64_t y_bs,
                           float* y_act, int64_t ya_bs, int B, int C, int H, int W, int dir, lc_stream_t s);
/* Down-sampling that also leaves GroupNorm statistics of its OUTPUT (round 5): one entry per (sample, channel, slot) in
 * the producer-statistics format with unit = 1 -- ostats[B, C, slots, 4], slots = lc_resample2x_stats_slots(H, W, -1)
 * (0: the shape takes the scalar kernel, which leaves none; lc_resample2x_stats_fwd then returns LC_EUNSUP).  The
 * GroupNorm behind a Resample(down=2) (efficient_unet.py:141-143 -> :79) folds them instead of taking a statistics pass. */
int64_t lc_resample2x_stats_slots(int H, int W, int dir);
int lc_resample2x_stats_fwd(const float* x, int64_t x_bs, float* y, int64_t y_bs, int B, int C, int H,
                            int W, int dir, float* ostats, lc_stream_t s);

/* Round 6 (third part): Conv2d(3x3, ring) BEHIND Resample(up=2) folded -- layout_unet_v1.py:219-235 (ResBlock up=True:
 * in_rest -> op(h) -> in_conv) and efficient_unet.py:143-145 (Block.upsample), ops.py:52-173.  Both operators are linear:
 *   conv3x3(U(a))[r][s] = bias + sum_{ky,kx} U(P_{ky,kx})[r + ky - 1][s + kx - 1 (ring)],   P_{ky,kx} = W[:, :, ky, kx] . a
 * (rows of U(.) outside [0, 2H) are the conv's zero padding).  The nine planes are ONE 1x1 projection Ci -> 9 Co of the
 * LOW-resolution operand (lc_conv1x1_f16x2_ps_fwd with the weight rows ordered t Co + co, t = 3 ky + kx: a quarter of the
 * 3x3 conv's multiply-adds), lc_up2_combine9_fwd reads them once and writes y [B][Co][2H][2W] once, with the bias and --
 * ostats != NULL -- one GroupNorm statistics entry per (sample, channel, slot) of what it stores (lc_oct_stats, unit = 1,
 * ostats[B][Co][slots][4], slots = lc_up2_combine9_stats_slots(H, W); 0: shape unsupported).  Needs W % 128 == 0, p9
 * 8-byte and y 16-byte aligned (LC_EUNSUP otherwise).
 * lc_split_act_fwd: the plain fp32 -> pre-split pass (x * range->x_scale as fp16 hi / lo planes in the layout of
 * lc_groupnorm_apply_split, lc_split_act_units(B, C, H, W) units; publishes max |x * x_scale|) for an operand that no
 * GroupNorm apply pass writes; C % 16 == 0, H * W % 4 == 0, x 16-byte aligned. */
int lc_split_act_fwd(const float* x, int64_t x_bs, void* y_split, int B, int C, int H, int W, lc_conv_range* range,
                     lc_stream_t s);
int64_t lc_up2_combine9_stats_slots(int H, int W);
int lc_up2_combine9_fwd(const float* p9, int64_t p_bs, const float* bias, float* y, int64_t y_bs, int B, int Co, int H,
                        int W, float* ostats, lc_stream_t s);
/* ... the same, and y2 [B][Co][2H][2W] = Resample(up=2)(x) of a second low-resolution tensor x [B][Co][H][W] in the same launch
 * (the skip path of LayoutUnetV1's up-sampling ResBlock, layout_unet_v1.py:232; bit-identical to lc_resample2x_fwd). */
int lc_up2_combine9_xup_fwd(const float* p9, int64_t p_bs, const float* bias, float* y, int64_t y_bs, const float* x,
                            int64_t x_bs, float* y2, int64_t y2_bs, int B, int Co, int H, int W, float* ostats, lc_stream_t s);

/* ---------------------------------------------------------------------------------------------
 * Small dense layer  y[m, n] = sum_k act(x[m,k]) * w[n,k] + b[n]   (w in nn.Linear layout).
 * Time-embedding MLP efficient_unet.py:237-242 / layout_unet_v1.py:683-688, AdaGN projection
 * ops.py:190-194, ResBlock.emb_layers layout_unet_v1.py:196-202.  act_in: 0 none, 1 SiLU.
 * lc_sinusoid: ops.SinusoidalPositionalEmbedding ops.py:14-29 (input = log-SNR).
 * ------------------------------------------------------------------------------------------- */
int lc_linear_fwd(const float* x, const float* w, const float* b, float* y, int M, int K, int N,
                  int act_in, int act_out, lc_stream_t s);
int lc_sinusoid_fwd(const float* t, float* y, int M, int channels, float max_period,
                    lc_stream_t s);

/* ---------------------------------------------------------------------------------------------
 * Attention over range-image tokens, channel-major operands (a [d, L] matrix per head with L
 * contiguous), flash-style (no score matrix in HBM), fp32 MFMA for QK^T and PV:
 *   o[c, t] = sum_s softmax_s(scale * sum_c' q[c',t] k[c',s]) v[c, s]
 * nn.MultiheadAttention in SelfAttentionBlock efficient_unet.py:28-58;
 * QKVAttentionLegacy layout_unet_v1.py:555-596; ObjectAwareCrossAttention.forward
 * layout_unet_v1.py:489-506 (image keys ++ 13 layout keys, d_qk = 2*d_v).
 * An operand is described by lc_cm_operand: head (b,h) starts at p + b*bs + h*hs, channel
 * stride cs, unit token stride.  The q/k channels of a head are the concatenation of a content
 * part (dqk channels, operands q / k / k2) and an optional positional part (dpos channels,
 * operands q_pos / k_pos / k2_pos; NULL when dpos == 0) -- the torch.cat([content, positional])
 * of layout_unet_v1.py:453-454,476 is never materialised.  Keys/values come in two token
 * segments: (k, k_pos, v) with Lk0 tokens and optionally (k2, k2_pos, v2) with Lk1 tokens
 * (the 13 layout tokens, layout_unet_v1.py:479-480).  dqk+dpos <= 64, dv <= 64.
 * ------------------------------------------------------------------------------------------- */
typedef struct lc_cm_operand {
    const float* p;
    int64_t bs, hs, cs;
} lc_cm_operand;

int lc_attention_fwd(const lc_cm_operand* q, const lc_cm_operand* q_pos,
                     const lc_cm_operand* k, const lc_cm_operand* k_pos, const lc_cm_operand* v,
                     const lc_cm_operand* k2, const lc_cm_operand* k2_pos, const lc_cm_operand* v2,
                     float* o, int64_t o_bs, int64_t o_hs, int64_t o_cs,
                     int B, int heads, int Lq, int Lk0, int Lk1, int dqk, int dpos, int dv,
                     float scale, lc_stream_t s);
/* Same contract, f16x2-split arithmetic (the default of the Python layer): q, k, p and v are split
 * into fp16 hi/lo parts of the pre-scaled fp32 values, every product is accumulated in fp32 as
 * ah*bh + ah*bl + al*bh with v_mfma_f32_32x32x16_f16 (per-product error ~5e-7, same tolerance as
 * lc_attention_fwd in tests/test_hip_parity.py). */
int lc_attention_f16x2_fwd(const lc_cm_operand* q, const lc_cm_operand* q_pos,
                           const lc_cm_operand* k, const lc_cm_operand* k_pos,
                           const lc_cm_operand* v, const lc_cm_operand* k2,
                           const lc_cm_operand* k2_pos, const lc_cm_operand* v2, float* o,
                           int64_t o_bs, int64_t o_hs, int64_t o_cs, int B, int heads, int Lq,
                           int Lk0, int Lk1, int dqk, int dpos, int dv, float scale,
                           lc_stream_t s);
/* Keys and values in UNIT FORM (round 6; csrc/attention_units.hip): the fp16 hi / lo split of lc_attention_f16x2_fwd
 * made once per step (or once per condition, for the step-invariant positional channels and the layout keys of
 * ObjectAwareCrossAttention, layout_unet_v1.py:431-476) instead of once per query block inside the attention kernel.
 * kv: [B][heads][tiles = Lk0 / 32 + (Lk1 > 0)][768] units of 8 halves -- per 32-key tile K hi, K lo ([8 channels-octets][32
 * keys]) and V hi, V lo ([2][2][32 channels], keys in MFMA accumulator order); zero where a head has no channel / key.
 * lc_attention_units_elems: halves to allocate (zero-filled by the caller), -1 for unsupported sizes (Lk0 % 32, Lk1 > 32).
 * lc_attention_pack_units: which = 0 writes keys (src: [B][heads * d][L] fp32 channel-major; the d channels land in
 * octets cb0 ... of a head's 64 q/k channels), which = 1 values (d <= 32); key0 (multiple of 32) = first destination key.
 * lc_attention_units_fwd: same result, bit for bit, as lc_attention_f16x2_fwd on the operands the units were packed
 * from (dqk + dpos <= 64, both multiples of 8, dv <= 32). */
/* lc_conv1x1_f16x2_ps_qkv_fwd: the qkv projection of ObjectAwareCrossAttention (conv_nd(1, C, 3 C, 1),
 * layout_unet_v1.py:381,416-430) on a pre-split input (lc_groupnorm_apply*_split) whose key and value rows are written
 * in unit form by the kernel's own epilogue -- q [B][C][H * W] fp32 (batch stride q_bs); 32 channels per head,
 * C % 128 == 0, (H * W) % 32 == 0; kv sized by lc_attention_units_elems(B, C / 32, H * W, Lk1). */
int lc_conv1x1_f16x2_ps_qkv_fwd(const void* x_split, const void* wp_hi, const void* wp_lo, const float* bias, float* q,
                                int64_t q_bs, void* kv, int B, int Ci, int C, int H, int W, int Lk1, const float* wmeta,
                                lc_conv_range* range, lc_stream_t s);
int64_t lc_attention_units_elems(int B, int heads, int Lk0, int Lk1);
int lc_attention_pack_units(const lc_cm_operand* src, void* kv, int B, int heads, int L, int Lk0, int Lk1, int d,
                            int cb0, int key0, int which, lc_stream_t s);
int lc_attention_units_fwd(const lc_cm_operand* q, const lc_cm_operand* q_pos, const void* kv, float* o,
                           int64_t o_bs, int64_t o_hs, int64_t o_cs, int B, int heads, int Lq, int Lk0, int Lk1,
                           int dqk, int dpos, int dv, float scale, lc_stream_t s);
/* ---------------------------------------------------------------------------------------------
 * Training: the same attention with a backward pass, so that loss.backward() never materialises
 * the [B*heads, Lq, Lk] scores or their gradient (autograd of nn.MultiheadAttention efficient_unet.py:28-58 and
 * of ObjectAwareCrossAttention.forward layout_unet_v1.py:489-506 as tools/train/train_lidm_cond.py:259-322 runs
 * them; softmax in fp32, layout_unet_v1.py:502).  Plain operands: heads back to back, per head a [d, L] matrix
 * with L contiguous -- q [BH, dqk, Lq], k [BH, dqk, Lk], v [BH, dv, Lk], o / do [BH, dv, Lq]; the caller
 * concatenates content / positional channels and image / layout keys (lidarcrafter_amd/autograd.py).
 * lc_attention_train_fwd = lc_attention_fwd (f16x2 = 0) or lc_attention_f16x2_fwd (1) + lse [BH, Lq]: the
 * log2-sum-exp of the scaled scores per query.  lc_attention_bwd recomputes P = exp2(S - lse) tile by tile in
 * exact-fp32 MFMA (v_mfma_f32_32x32x2_f32) and writes dq, dk, dv (same shapes as q, k, v); deterministic (one
 * kernel with the queries as the outer dimension for dq, one with the keys for dk / dv; no atomics).
 * dsum_scratch: float [BH, Lq].  dqk, dv <= 64.
 * qkv_amax (round 6; f16x2 = 1 only, may be NULL): 3 floats of scratch the caller keeps alive until the backward pass has
 * run.  The forward measures max |q|, max |k|, max |v| into them on the device and both passes derive the powers of two
 * their fp16 hi / lo operand splits use from those words (16 while max |x| * 16 lies in [2^2, 2^15): the results of
 * rounds 1-5; otherwise the power of two that fits -- csrc/attention_pre.h).  NULL: the constant 16, which saturates
 * |x| >= 4094 silently.
 * ------------------------------------------------------------------------------------------- */
int lc_attention_train_fwd(const float* q, const float* k, const float* v, float* o, float* lse, int BH, int Lq,
                           int Lk, int dqk, int dv, float scale, int f16x2, float* qkv_amax, lc_stream_t s);
int lc_attention_bwd(const float* q, const float* k, const float* v, const float* o, const float* dout,
                     const float* lse, float* dsum_scratch, float* dq, float* dk, float* dv, int BH, int Lq, int Lk,
                     int dqk, int dv_ch, float scale, lc_stream_t s);
/* The same backward pass with the f16x2-split arithmetic of lc_attention_f16x2_fwd (three v_mfma_f32_32x32x16_f16 per
 * product, fp32 accumulation; the default of lidarcrafter_amd.autograd).  The gradient dO may have any magnitude: its
 * maximum is measured next to D and the power of two that normalises it is carried exactly through dP, dS and the
 * outputs.  dsum_scratch: float [BH * Lq + 1] (the extra word holds max |dO|).  qkv_amax: the words
 * lc_attention_train_fwd measured for the same q, k, v (or NULL: constant pre-scale 16). */
int lc_attention_bwd_f16x2(const float* q, const float* k, const float* v, const float* o, const float* dout,
                           const float* lse, float* dsum_scratch, float* dq, float* dk, float* dv, int BH, int Lq,
                           int Lk, int dqk, int dv_ch, float scale, const float* qkv_amax, lc_stream_t s);

/* ---------------------------------------------------------------------------------------------
 * Reverse-diffusion update, one fused elementwise pass:
 *   ContinuousTimeGaussianDiffusion.p_step continuous_time.py:209-231 (also
 *   continuous_time_cond.py:229-252).  coef[b*8 + ...] = {alpha_t, sigma_t, alpha_s, sigma_s,
 *   k0, k1, clip_range (<=0: no clip), unused}: ddpm k0 = c = -expm1(l_t-l_s), k1 = sigma_s*sqrt(c);
 *   ddim k0 = c1, k1 = c2.  objective: 0 eps, 1 v, 2 x_0.  mode: 0 ddpm, 1 ddim.
 *   noise may be NULL when its coefficient is 0 (ddim eta=0).  n = C*H*W per sample.
 *   DiscreteTimeGaussianDiffusion.p_step discrete_time.py:126-180 (round 3): mode 2 ddpm, 3 ddim with
 *   coef = {A, Bc, c2, c3, c4, c5, clip, c7}: x0 = A x_t - Bc pred (eps: A = rsqrt(ab), Bc =
 *   sqrt(1/ab - 1); v: sqrt(ab), sqrt(1 - ab); x_0: x0 = pred); ddpm (c2 x0 + c3 x_t) + c4 noise with
 *   c2 = sqrt(ab_prev) beta / (1 - ab), c3 = (1 - ab_prev) sqrt(alpha) / (1 - ab), c4 = sigma (0 at
 *   step 0); ddim c4 x0 + c5 (x_t - c2 x0) / c3 [+ c7 noise] with c2 = sqrt(ab), c3 = sqrt(1 - ab),
 *   c4 = sqrt(ab_prev), c5 = sqrt(1 - ab_prev - sd^2), c7 = sd.
 * ------------------------------------------------------------------------------------------- */
int lc_pstep_fwd(const float* x_t, int64_t xt_bs, const float* pred, int64_t pred_bs,
                 const float* noise, int64_t noise_bs, const float* coef, float* x_s,
                 int64_t xs_bs, int B, int64_t n, int objective, int mode, lc_stream_t s);

/* Point Condition Network epilogue of the foreground-object denoiser
 * (lidargen/models/unets/point_unet.py:14-26, PCNet): channel-major rows [B, C, N],
 *   y = act( x * sigmoid(gate_logit[b,c]) + bias[b,c] ) [+ res],  act: 0 none, 1 leaky_relu(0.01)
 * x = fea_layer(fea) from lc_conv2d_ring_*_fwd (1x1); gate_logit / bias rows [B, >=C] with row
 * stride gb_bs (slices of ONE batched cond_gate | cond_bias projection); res may be NULL. */
int lc_gate_bias_act(const float* x, int64_t x_bs, const float* gate_logit, const float* bias,
                     int64_t gb_bs, const float* res, int64_t res_bs, float* y, int64_t y_bs,
                     int B, int C, int N, int act, lc_stream_t s);

/* ---------------------------------------------------------------------------------------------
 * Block.downsample of EfficientUNet folded: `ops.Conv2d(3x3, ring)` followed by `ops.Resample(down=2)`
 * (lidargen/models/unets/efficient_unet.py:132-135; ops.py:52-146 FIR [1 3 3 1] / 8, ring along W, zero padding of the
 * conv's OUTPUT along H; ops.py:149-173) evaluated as ONE stride-2 3x3 convolution of the FIR-pre-filtered input -- a quarter
 * of the reference conv's multiply-adds and output bytes, no resampling pass (csrc/conv_f16x2_s2.hip has the algebra).
 *
 * lc_fir_down2_prefilter_split: x [B, C, H, W] fp32 (batch stride x_bs floats, channel stride H*W; C % 16 == 0, H even >= 4,
 *   W % 128 == 0) -> y_split: the pre-filtered tensor F in the pre-split form of lc_groupnorm_apply_split (fp16 hi / lo
 *   planes, 16-byte units of 8 channels, already multiplied by range->x_scale; max |F * x_scale| is published into *range),
 *   per sample [2 planes][C / 8][H + 3 rows][W units]: rows 0 .. H = F[-1 .. H-1], row H + 1 / H + 2 = the top / bottom
 *   boundary variants; inside a row the odd input columns first (unit i = column 2 i - 1, ring), then the even ones.
 *   lc_fir_down2_split_units gives the number of 16-byte units of y_split.
 * lc_conv2d_ring_s2_f16x2_ps_fwd: y [B, Co, Ho, Wo] = (stride-2 conv of F with the layer's packed 3x3 weights + bias) *
 *   out_scale; Ho = H / 2, Wo = W / 2 (Wo % 64 == 0).  wp_hi / wp_lo / wmeta / range: as lc_conv2d_ring_f16x2_ps_fwd.
 *   gn_ostats_out (optional): octet (unit 8) or quad (unit 4) GroupNorm statistics entries of what it stores,
 *   [B][Co / unit][lc_conv2d_ring_s2_stats_slots(Ho, Wo)][4] floats. */
int64_t lc_fir_down2_split_units(int B, int C, int H, int W);
int lc_fir_down2_prefilter_split(const float* x, int64_t x_bs, void* y_split, int B, int C, int H, int W,
                                 lc_conv_range* range, lc_stream_t s);
int64_t lc_conv2d_ring_s2_stats_slots(int Ho, int Wo);
int lc_conv2d_ring_s2_f16x2_ps_fwd(const void* x_split, const void* wp_hi, const void* wp_lo, const float* bias, float* y,
                                   int64_t y_bs, int B, int Ci, int Co, int Ho, int Wo, float out_scale,
                                   float* gn_ostats_out, int gn_ostats_unit, const float* wmeta, lc_conv_range* range,
                                   lc_stream_t s);

/* Box calibration for bench.py (`box_calibration`): what THIS box's matrix pipes and memory sustain, so that numbers of
 * different boxes of the pool can be compared.  No counterpart in the reference.
 * lc_calibrate_mfma_f16: `blocks` blocks of 8 waves run `iters` x 16 v_mfma_f32_32x32x16_f16 per wave on the caller's
 * operands (blocks * 512 * 4 half8 vectors = blocks * 32 KiB of fp16; random data: the sustained rate depends on the
 * operand bits), one float per thread into sink[blocks * 512].  Returns the FLOPs of the launch (> 0) or a negative code.
 * lc_calibrate_stream_copy: dst[i] = src[i], n floats (multiple of 4, 16-byte aligned), float4 grid-stride. */
int64_t lc_calibrate_mfma_f16(const void* operands, int blocks, int iters, float* sink, lc_stream_t s);
int lc_calibrate_stream_copy(const float* src, float* dst, int64_t n, lc_stream_t s);

/* Strided copy of [B, C*H*W] blocks (fills a channel slice of a concat buffer). */
int lc_copy_strided(const float* x, int64_t x_bs, float* y, int64_t y_bs, int B, int64_t n,
                    lc_stream_t s);
/* y = (a + b) * scale  (SelfAttentionBlock residual efficient_unet.py:55-57). */
int lc_add_scale(const float* a, int64_t a_bs, const float* b, int64_t b_bs, float* y,
                 int64_t y_bs, int B, int64_t n, float scale, lc_stream_t s);

/* ---------------------------------------------------------------------------------------------
 * BACKWARD kernels (training: tools/train/train_lidm[_cond].py:259-322 run autograd through the
 * denoiser).  Plain fp32 NCHW tensors with batch strides as in the forward.
 *
 * Ring convolution.  The gradient with respect to the input is the SAME ring convolution applied to
 * dY with the transposed, 180-degree rotated kernel (run lc_conv2d_ring_*_fwd on it); the gradient
 * with respect to the weights and bias is lc_conv2d_ring_wgrad:
 *   dW[co][ci][ky][kx] (+)= sum_{b,h,w} dY[b,co,h,w] * Xpad[b,ci,h+ky-1,w+kx-1],  db[co] (+)= sum dY
 * fp32 matrix cores (exact fp32 products, fp32 accumulation), deterministic two-stage reduction;
 * scratch: lc_conv2d_ring_wgrad_scratch_elems floats (8-byte aligned); dbias may be NULL; accumulate != 0 adds to
 * dw / dbias (gradient accumulation), 0 overwrites.
 *
 * GroupNorm (+affine) (+AdaGN scale/shift) (+SiLU), forward y = silu?(((x-mu) rstd g + be)(1+sc) + sf):
 *   lc_groupnorm_meanrstd: (mean, rstd) per (sample, group) [B, G, 2] from the lc_groupnorm_stats partials;
 *   lc_groupnorm_bwd: rows[b][c] = (sum_hw dt2, sum_hw dt2 * xhat) as doubles [B, C, 2] with
 *   dt2 = dy * silu'(.), and dx = rstd (g (1+sc) dt2 - mean_g(.) - xhat mean_g(. xhat)).  The small
 *   parameter gradients follow from `rows`: dshift = r1, dscale = g r3 + be r1,
 *   dbeta = sum_b (1+sc) r1, dgamma = sum_b (1+sc) r3.
 * ------------------------------------------------------------------------------------------- */
int64_t lc_conv2d_ring_wgrad_scratch_elems(int B, int Ci, int Co, int H, int W, int ks);
int lc_conv2d_ring_wgrad(const float* x, int64_t x_bs, const float* dy, int64_t dy_bs, float* scratch,
                         float* dw /* [Co,Ci,ks,ks] */, float* dbias /* [Co] or NULL */, int B, int Ci,
                         int Co, int H, int W, int ks, int accumulate, lc_stream_t s);
/* The same gradient on the f16 matrix cores with the operand split of the f16x2 forward (3 MFMAs per
 * product, per-product error ~2^-22, fp32 accumulation; contraction over pixels, which NCHW stores
 * contiguously -- no transposition): x and dy are pre-scaled by the power-of-two x_scale of
 * `x_range` / `dy_range` (records of exactly these tensors: lc_range_from_tensor right before the
 * forward conv of x and the input-gradient conv of dy, lidarcrafter_amd/autograd.py) and the result
 * is unscaled by both.  Needs H even, W % 32 == 0 and 16-byte aligned rows (LC_EUNSUP otherwise: use
 * lc_conv2d_ring_wgrad); scratch and everything else as lc_conv2d_ring_wgrad. */
int lc_conv2d_ring_wgrad_f16x2(const float* x, int64_t x_bs, const float* dy, int64_t dy_bs,
                               const lc_conv_range* x_range, const lc_conv_range* dy_range,
                               float* scratch, float* dw, float* dbias, int B, int Ci, int Co, int H,
                               int W, int ks, int accumulate, lc_stream_t s);
int lc_groupnorm_meanrstd(const float* x, int64_t x_bs, const double* partials, float* mean_rstd,
                          int B, int C, int H, int W, int G, float eps, lc_stream_t s);
int lc_groupnorm_bwd(const float* x, int64_t x_bs, const float* dy, int64_t dy_bs,
                     const float* mean_rstd, const float* gamma, const float* beta, const float* scale,
                     const float* shift, int64_t ss_bs, double* rows, float* dx, int64_t dx_bs, int B,
                     int C, int H, int W, int G, int act_silu, lc_stream_t s);
/* lc_groupnorm_bwd plus, in the same two launches: the parameter gradients from `rows` (fp64 arithmetic, fp32 results;
 * any of them may be NULL): dgamma, dbeta [C]; dscale, dshift [B, C] contiguous -- and the partial maxima of |dx| into
 * amax_out[0 .. lc_groupnorm_amax_partials(..., 1)) (may be NULL; as lc_groupnorm_apply_train): dx is the dY of the conv
 * that produced x. */
int lc_groupnorm_bwd_train(const float* x, int64_t x_bs, const float* dy, int64_t dy_bs,
                           const float* mean_rstd, const float* gamma, const float* beta, const float* scale,
                           const float* shift, int64_t ss_bs, double* rows, float* dx, int64_t dx_bs,
                           float* dgamma, float* dbeta, float* dscale, float* dshift, int B, int C, int H, int W,
                           int G, int act_silu, float* amax_out, lc_stream_t s);

/* ---------------------------------------------------------------------------------------------
 * Voxel scatter of the weight-free metrics (lidargen/metrics/metric_utils.py).
 * lc_bev_occupancy_accumulate: ONE sweep of pcd2bev_sum (:233-258): points with x in (x0, x1) and
 *   y in (y0, y1) are binned to ix = floor(x / voxel) - min_bound_x (float32 division, as numpy),
 *   every voxel the sweep touches gets +1 in `grid` [nx, ny] float32 -- exactly once per sweep:
 *   `stamps` [nx*ny] int32 (zero-initialised by the caller, one per grid) holds the id of the last
 *   sweep that touched the voxel; pass a `stamp` != 0 that differs from sweep to sweep, and run
 *   the sweeps of one grid in stream order.  pt_stride = floats per point row (>= 2).
 * lc_sparse_quantize (:28-66): floor(coords / voxel) -> int32 [N, D] (D = 2 or 3), unique rows in
 *   ravel-hash (= lexicographic) order -> out_coords (first *out_count rows valid), out_index =
 *   index of each unique row's FIRST occurrence (may be NULL), out_inverse [N] (may be NULL) --
 *   np.unique(ravel_hash(q), return_index, return_inverse).  out_count: device uint64.
 *   scratch: lc_sparse_quantize_scratch_bytes(N, D) bytes of device memory.
 * ------------------------------------------------------------------------------------------- */
int lc_bev_occupancy_accumulate(const float* pts, int pt_stride, int N, float x0, float x1, float y0,
                                float y1, float voxel, int min_bound_x, int min_bound_y, int nx,
                                int ny, int stamp, int32_t* stamps, float* grid, lc_stream_t s);
int64_t lc_sparse_quantize_scratch_bytes(int N, int D);
int lc_sparse_quantize(const float* coords, int N, int D, float vx, float vy, float vz, void* scratch,
                       int32_t* out_coords, int64_t* out_index, int64_t* out_inverse,
                       uint64_t* out_count, lc_stream_t s);

/* ---------------------------------------------------------------------------------------------
 * Point cloud -> range image, lidargen/dataset/transforms_3d/common.py:26-91 with
 * scan_unfolding=False: spherical cell per point, nearest point per cell wins (z-buffer via
 * 64-bit atomicMin on (depth_bits<<32 | point_idx); equal depths: lowest index wins), mask
 * channel = depth in [min_depth, max_depth] applied by the caller like the reference.
 * points [N,4] (x,y,z,intensity); zbuf u64[H*W] scratch; image [H,W,6]; winner int32[H*W] (-1
 * empty) may be NULL; cells int32[N,2] (grid_h, grid_w) may be NULL.  All float32 operations
 * correctly rounded (asin/atan2 evaluated in fp64, rounded once).  elev_f64 = 1: the elevation ->
 * row arithmetic runs in float64 on the float32 asin, as the reference computes under numpy >= 2
 * (oracle/lidar.py "native_cr" -- the mode the committed reference fixtures are reproduced in);
 * 0: all-float32, the reference under its pinned numpy 1.23.5 (oracle "f32").
 * ------------------------------------------------------------------------------------------- */
int lc_project_points(const float* points, int N, int H, int W, double fov_up_deg,
                      double fov_down_deg, float min_depth, float max_depth, uint64_t* zbuf,
                      float* image, int32_t* winner, int32_t* cells, int elev_f64, lc_stream_t s);
/* Workspace variant (round 4): `zbuf` is a caller-owned u64[H*W] that holds ~0 in every cell on entry --
 * lc_project_workspace_init leaves it so -- and is handed back in that state (the gather pass re-empties each cell
 * it reads), so a projection is two launches instead of three and the caller allocates nothing per call: 18 -> 8 us
 * at the 34 720 points of a nuScenes sweep, where the three-launch form is latency-bound.  One projection at a time
 * per workspace.  Same results as lc_project_points. */
int lc_project_workspace_init(uint64_t* zbuf, int n_cells, lc_stream_t s);
int lc_project_points_ws(const float* points, int N, int H, int W, double fov_up_deg,
                         double fov_down_deg, float min_depth, float max_depth, uint64_t* zbuf,
                         float* image, int32_t* winner, int32_t* cells, int elev_f64, lc_stream_t s);
/* The same projection of FLOAT64 points [N,4] -- what the temporal glue hands over
 * (tools/vis_tools/utils/pipe_related.py:245-258: the float64 product `Ts @ homo` and the float64
 * concatenation [background | re-posed objects]): every line of common.py:41-84 runs in float64
 * and the image is rounded to float32 once (:87-91).  winner int32[H*W] is REQUIRED (pass 2 of the
 * z-buffer: lowest index among the points at the cell's minimum depth).  Pinned on the reference's
 * own CustomDataset item / refine_next_frame_points (tests/golden/pipe_next.npz). */
int lc_project_points_f64(const double* points, int N, int H, int W, double fov_up_deg,
                          double fov_down_deg, double min_depth, double max_depth, uint64_t* zbuf,
                          int32_t* winner, float* image, lc_stream_t s);

/* ---------------------------------------------------------------------------------------------
 * Points in rotated boxes: lidargen/ops/roiaware_pool3d/src/roiaware_pool3d.cpp:128-168
 * (points_in_boxes_cpu: int32 [N_box, M] 0/1, MARGIN 1e-2) and
 * src/roiaware_pool3d_kernel.cu:23-36,313-336 (points_in_boxes_gpu: int32 [B, M] index of the
 * first containing box or -1, MARGIN 1e-5).  boxes [.., 7] = x,y,z,dx,dy,dz,heading.
 * The boxes of a block sit in LDS with their per-box constants (rotation, half extents): at most
 * 1250 boxes per call / per sample (LC_EUNSUP beyond; the reference's scenes hold tens).
 * ------------------------------------------------------------------------------------------- */
int lc_points_in_boxes_mask(const float* boxes, int n_boxes, const float* pts, int n_pts,
                            float margin, int32_t* out_mask, lc_stream_t s);
int lc_points_in_boxes_index(const float* boxes, const float* pts, int B, int n_boxes, int n_pts,
                             float margin, int32_t* out_idx, lc_stream_t s);

/* ---------------------------------------------------------------------------------------------
 * Range-image pre/post-processing fused into single passes.
 * lc_range_postprocess: sample [B,2,H,W] in [-1,1] (depth, reflectance) -> out [B,5,H,W] =
 *   (metric depth, x, y, z, reflectance): LiDARUtility.denormalize / revert_depth / to_xyz
 *   (lidargen/utils/lidar.py:61-82,109-128) as chained in tools/evaluation/
 *   sample_and_save_cond.py:119-124.  ray_angles [1,2,H,W] (elevation, azimuth) radians.
 * lc_condition_preprocess: condition_mask [B,2,H,W] (class id, metric depth) -> out
 *   [B,num_classes+1,H,W] = one_hot(class) ++ LiDARUtility.convert_depth(depth)
 *   (sample_and_save_cond.py:106-117, utils/lidar.py:84-107).
 * depth_format: 0 log_depth, 1 inverse_depth, 2 depth.
 * ------------------------------------------------------------------------------------------- */
int lc_range_postprocess(const float* sample, int64_t sample_bs, const float* ray_angles,
                         float* out, int B, int H, int W, int depth_format, float min_depth,
                         float max_depth, lc_stream_t s);
int lc_condition_preprocess(const float* condition_mask, int64_t cm_bs, float* out, int64_t out_bs,
                            int B, int H, int W, int num_classes, int depth_format,
                            float min_depth, float max_depth, lc_stream_t s);

/* ---------------------------------------------------------------------------------------------
 * Layout condition rasteriser (the step immediately before the path):
 * lidargen/dataset/transforms_3d/common.py:99-215 (convert_boxes_to_2d + convert_points_to_2d).
 * boxes [B,T,box_stride>=8] = (x,y,z,l,w,h,yaw,class,...), n_valid int32 [B] (first n rows used),
 * -> corners_2d [B,T,4] (x1,y1,x2,y2 normalised; zeros for padding rows; may be NULL),
 *    condition_mask [B,2,H,W] (class id, centre depth; later boxes overwrite earlier ones),
 *    loss_weight_map [B,H,W] (may be NULL).  scratch: lc_layout_scratch_bytes(B,T) bytes.
 * ------------------------------------------------------------------------------------------- */
int64_t lc_layout_scratch_bytes(int B, int T);
/* boxes_f64 = 0: float32 boxes (yaw cos/sin and centre depth in float32, like numpy on float32
 * input); 1: float64 boxes -- what NuscDataset.pre_process hands over (nuscenes_dataset.py:384-397)
 * -- everything in float64, the centre depth rounded to float32 when painted. */
int lc_layout_condition(const void* boxes, int boxes_f64, int box_stride, const int32_t* n_valid,
                        int B, int T, int H, int W, double fov_up_deg, double fov_down_deg,
                        void* scratch, float* corners_2d, float* condition_mask,
                        float* loss_weight_map, lc_stream_t s);

/* ---------------------------------------------------------------------------------------------
 * RoI-aware voxel pooling of point features: lidargen/ops/roiaware_pool3d/
 *   roiaware_pool3d_utils.py:55-107 (RoIAwarePool3dFunction) -> src/roiaware_pool3d.cpp:30-117
 *   (pybind forward / backward :173-174) -> src/roiaware_pool3d_kernel.cu:39-310.
 * rois [N,7], pts [P,3], pts_feature [P,C] -> pooled [N,X,Y,Z,C] (must be zero-filled by the
 * caller like the reference's new_zeros), pts_idx_of_voxels int32 [N,X,Y,Z,max_pts] (slot 0 =
 * count), argmax int32 [N,X,Y,Z,C] (max pooling).  pts_mask_scratch: int32 [N,P].
 * pool_method: 0 max, 1 avg.  Backward accumulates into grad_in [P,C] (zero-filled by the caller)
 * with atomicAdd, like the reference.
 * ------------------------------------------------------------------------------------------- */
int lc_roiaware_pool3d_fwd(const float* rois, const float* pts, const float* pts_feature,
                           int n_boxes, int n_pts, int channels, int out_x, int out_y, int out_z,
                           int max_pts_each_voxel, int pool_method, int32_t* pts_mask_scratch,
                           int32_t* pts_idx_of_voxels, int32_t* argmax, float* pooled,
                           lc_stream_t s);
int lc_roiaware_pool3d_bwd(const int32_t* pts_idx_of_voxels, const int32_t* argmax,
                           const float* grad_out, float* grad_in, int n_boxes, int channels,
                           int out_x, int out_y, int out_z, int max_pts_each_voxel,
                           int pool_method, lc_stream_t s);

/* ---------------------------------------------------------------------------------------------
 * Point-set glue of the autoregressive (temporal) loop, SURVEY.md section 8(f)-2 -- keeps
 * tools/evaluation/sample_and_save_temporal.py:262-331 on the device between frames.
 * Points are [N,4] float32 rows (x, y, z, intensity), 16-byte aligned.
 *  lc_transform_points: out.xyz = T[0:3,0:3] p + T[0:3,3] (T row-major 4x4 of doubles on the HOST,
 *    product in fp64, rounded once), intensity copied: `(Ts @ homo_pts.T).T`
 *    tools/vis_tools/utils/pipe_related.py:245-249; warp_lidar_future common.py:59-112; the
 *    object <-> box-frame moves pipe_related.py:56-66,263-268 are the same op with another T.
 *  lc_image_to_points: xyz [3,H,W] (+reflectance [H,W]) -> rows, times the background mask
 *    !(cond[h,w] > 0) when cond != NULL; keep[i] = 0 for |xyz| <= min_norm (min_norm < 0: off)
 *    and for |x|,|y| < ego_radius (<= 0: off): pipe_related.py:68-75,274-283 and :11-13.
 *  lc_points_in_boxes_mask4: points_in_boxes_cpu on [N,4] rows; out_mask [n_boxes, N] and / or
 *    out_count [N] = number of boxes containing the point (delete_fg_points :266-272).
 *  lc_compact_points: out = rows[keep != 0] (or keep == 0 when keep_if_zero) in input order --
 *    numpy boolean indexing; count[0] = rows kept; src_index (may be NULL) = source row of each
 *    kept row; scratch = lc_compact_scratch_elems(N) int32.  Three launches, no host sync.
 * ------------------------------------------------------------------------------------------- */
int lc_transform_points(const float* pts, int N, const double* T16_host, float* out, lc_stream_t s);
/* float64 rows out [N,4] (32-byte aligned), NOT rounded: the reference keeps `(Ts @ homo_pts.T).T`
 * in float64 and projects it in float64 (pipe_related.py:245-257).  rot_f32 = 1: out.xyz =
 * (double)(float)(R p) + t, i.e. `rotate_points_along_z(p, yaw)` (float32 matmul,
 * lidargen/dataset/utils.py:37-59) `+ np.array([x, y, z])` (float64), pipe_related.py:263-266. */
int lc_transform_points_f64(const float* pts, int N, const double* T16_host, int rot_f32,
                            double* out, lc_stream_t s);
int lc_image_to_points(const float* xyz, int64_t plane_stride, const float* refl, const float* cond,
                       int H, int W, float refl_scale, float min_norm, float ego_radius, float* pts,
                       int32_t* keep, lc_stream_t s);
int lc_points_in_boxes_mask4(const float* boxes, int n_boxes, const float* pts4, int n_pts,
                             float margin, int32_t* out_mask, int32_t* out_count, lc_stream_t s);
int64_t lc_compact_scratch_elems(int N);
int lc_compact_points(const float* rows, const int32_t* keep, int N, int keep_if_zero, float* out,
                      int32_t* src_index, int32_t* count, int32_t* scratch, lc_stream_t s);

/* ---------------------------------------------------------------------------------------------
 * BEV metrics front-end, lidargen/metrics/bev.py (SURVEY.md section 8f-3 ii):
 *  lc_bev_histogram: point_cloud_to_histogram :5-24 -- rows with min_depth < |xyz| < max_depth are
 *    binned by (x, y) into hist[bx*bins + by] with torch.histogramdd's rule (edges [bins+1] as
 *    torch computes them, e[b] <= v < e[b+1], last bin right-inclusive); scratch = bins*bins int32.
 *  lc_rbf_kernel_sum: partials[] (lc_rbf_partials_elems doubles) whose sum is
 *    sum_ij exp(-gamma |p_i - q_j|^2), p [M,D], q [Mq,D] -- cdist_rbf(...).mean() * M * Mq of
 *    compute_mmd_2d :47-55.
 * ------------------------------------------------------------------------------------------- */
int lc_bev_histogram(const float* pts, int pt_stride, int N, const float* edges, int bins,
                     float min_depth, float max_depth, float* hist, int32_t* scratch, lc_stream_t s);
int64_t lc_rbf_partials_elems(int M, int Mq);
int lc_rbf_kernel_sum(const float* p, const float* q, int M, int Mq, int D, float gamma,
                      double* partials, lc_stream_t s);
/* Chamfer distance forward (lidargen/metrics/modules/chamfer3D/chamfer3D.cu:12-155 via
 * dist_chamfer_3D.py:27-49): xyz1 [B,N,3], xyz2 [B,M,3] -> squared distance to and index of the
 * nearest point of the other set, both directions; first minimum wins. */
int lc_chamfer3d_fwd(const float* xyz1, const float* xyz2, int B, int N, int M, float* dist1,
                     int32_t* idx1, float* dist2, int32_t* idx2, lc_stream_t s);

#ifdef __cplusplus
}
#endif
#endif /* LIDARCRAFTER_HIP_H */
