// Conv2d(3x3, ring) BEHIND Resample(up=2), folded (round 6, third part): both operators are linear, so
//
//     conv3x3(U(a))[r][s] = b + sum_{ky,kx} U(P_{ky,kx})[r + ky - 1][s + kx - 1 (ring)],     P_{ky,kx} = W[:, :, ky, kx] . a
//
// where U is the FIR x2 up-sampler (per axis y[2i] = .25 x[i-1] + .75 x[i], y[2i+1] = .75 x[i] + .25 x[i+1]; rows of x
// outside the image are 0, columns wrap) and rows of U(.) outside [0, 2H) are 0 (the conv's own zero padding).  The nine
// P planes are ONE 1x1 projection Ci -> 9 Co of the LOW-resolution tensor (lc_conv1x1_f16x2_ps_fwd on a pre-split
// operand: a QUARTER of the multiply-adds of the 3x3 conv at the high resolution), and this file's pass combines them:
// reads [B][9 Co][H][W] once, writes [B][Co][2H][2W] once (+ bias, + the GroupNorm statistics entries of what it stores).
// Reference: layout_unet_v1.py:219-235 (ResBlock with up=True: in_rest -> op(h) -> in_conv), efficient_unet.py:143-145
// (Block.upsample = Resample(up) -> Conv2d), models/unets/ops.py:52-173.
//
// The H border is the only place where the two orders differ in form: the conv's tap ky = 0 at output row 0 and ky = 2 at
// output row 2H - 1 fall on the conv's zero padding, NOT on an up-sampled row -- those two (row, ky) terms are dropped.
//
// Also here: the plain fp32 -> pre-split pass (x * x_scale as fp16 hi / lo planes, no normalisation) for a low-resolution
// operand that no GroupNorm apply pass writes (EfficientUNet's Block.upsample takes the block's output as it is).
#include "common.h"

namespace {

typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef float float2_t __attribute__((ext_vector_type(2)));

// ---- fp32 [B][C][HW] -> ysp[b][plane][c/8][p][8] (the layout of lc_groupnorm_apply*_split), split rule of conv_f16x2.hip:
// s = x * x_scale, hi = 11 significant bits truncated (packed toward zero), lo = fp16(s - hi); max |s| -> the range record.
__global__ __launch_bounds__(256) void split_plain_kernel(const float* __restrict__ x, long long x_bs,
                                                         half8_t* __restrict__ ysp, long long ysp_bs, int C,
                                                         long long HW, lc_conv_range* range) {
    const int oct = blockIdx.y, b = blockIdx.z;
    const float* xp = x + b * x_bs + (long long)oct * 8 * HW;
    const float xs = range->x_scale;
    const float seen = range->amax_scaled;
    half8_t* yh = ysp + b * ysp_bs + (long long)oct * HW;
    half8_t* yl = yh + (long long)(C >> 3) * HW;
    float am = 0.0f;
    for (long long p = (blockIdx.x * 256ll + threadIdx.x) * 4; p < HW; p += (long long)gridDim.x * 1024) {
        f32x4 c4[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) c4[k] = *reinterpret_cast<const f32x4*>(xp + (long long)k * HW + p);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            half8_t h8, l8;
#pragma unroll
            for (int k = 0; k < 8; k += 2) {
                const float s0 = c4[k][q] * xs, s1 = c4[k + 1][q] * xs;
                am = fmaxf(am, fmaxf(fabsf(s0), fabsf(s1)));
                const float h0 = __uint_as_float(__float_as_uint(s0) & 0xFFFFE000u);
                const float h1 = __uint_as_float(__float_as_uint(s1) & 0xFFFFE000u);
                const half2_t ph = __builtin_bit_cast(half2_t, __builtin_amdgcn_cvt_pkrtz(h0, h1));
                float2_t r; r.x = s0 - h0; r.y = s1 - h1;
                const half2_t pl = __builtin_convertvector(r, half2_t);
                h8[k] = ph.x; h8[k + 1] = ph.y; l8[k] = pl.x; l8[k + 1] = pl.y;
            }
            yh[p + q] = h8;
            yl[p + q] = l8;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) am = fmaxf(am, __shfl_xor(am, o, 64));
    if ((threadIdx.x & 63) == 0 && am > seen)
        atomicMax(reinterpret_cast<unsigned*>(&range->amax_scaled), __float_as_uint(am));
}

// ---- the combine pass.  One wave = (sample, output channel, 128-column segment of the low-resolution plane); lane l owns
// the low-resolution columns j0 = 128 sg + 2 l and j0 + 1 (4 output columns) and walks down the rows with a sliding window
// of three rows of horizontally combined values per ky, the next row's 27 loads in flight under the current row's
// arithmetic and stores.  Per low-resolution row: 9 planes x (8-byte own pair + left + right neighbour) loads, two 16-byte
// write-through stores, one statistics entry (pivot, n = 512, sum (v - pivot), sum (v - pivot)^2) of the 512 values the
// wave stores -- the producer-statistics format with ONE channel per entry, as lc_resample2x_stats_fwd writes it:
// ostats[(b Co + co) slots + i segs + sg], slots = H segs.
__device__ __forceinline__ float hz0(float p0m, float p0, float p1m, float p1, float p2, float p2p) {
    // output column 2j:  kx = 0 reads U column 2j - 1, kx = 1 column 2j, kx = 2 column 2j + 1
    float s = 0.75f * p0m;
    s = fmaf(0.25f, p0, s);
    s = fmaf(0.25f, p1m, s);
    s = fmaf(0.75f, p1, s);
    s = fmaf(0.75f, p2, s);
    return fmaf(0.25f, p2p, s);
}
__device__ __forceinline__ float hz1(float p0m, float p0, float p1, float p1p, float p2, float p2p) {
    // output column 2j + 1:  kx = 0 reads U column 2j, kx = 1 column 2j + 1, kx = 2 column 2j + 2
    float s = 0.25f * p0m;
    s = fmaf(0.75f, p0, s);
    s = fmaf(0.75f, p1, s);
    s = fmaf(0.25f, p1p, s);
    s = fmaf(0.25f, p2, s);
    return fmaf(0.75f, p2p, s);
}

// XUP: the same launch also writes Resample(up=2)(x) of a second low-resolution tensor x [B][Co][H][W] into y2 -- the skip path
// of LayoutUnetV1's up-sampling ResBlock (layout_unet_v1.py:232: x = self.op(x)), which shares the (sample, channel, segment,
// row) walk of this kernel: one launch and one pass over the row window instead of a second kernel.  Arithmetic and order of
// resample.hip's up2_kernel (no contraction): bit-identical to lc_resample2x_fwd.
struct XupArgs { const float* x; long long x_bs; float* y2; long long y2_bs; };
#pragma clang fp contract(off)
__device__ __forceinline__ void xup_row(float xm, float xa, float xb, float xp, float (&o)[4]) {
    o[0] = 0.25f * xm + 0.75f * xa;      // ev(j0)
    o[1] = 0.75f * xa + 0.25f * xb;      // od(j0)
    o[2] = 0.25f * xa + 0.75f * xb;      // ev(j0 + 1)
    o[3] = 0.75f * xb + 0.25f * xp;      // od(j0 + 1)
}
__device__ __forceinline__ float xup_top(float m, float c) { return 0.25f * m + 0.75f * c; }
__device__ __forceinline__ float xup_bot(float c, float p) { return 0.75f * c + 0.25f * p; }
#pragma clang fp contract(fast)

template <bool STATS, bool XUP>
__global__ __launch_bounds__(256) void up2_combine9_kernel(const float* __restrict__ p9, long long p_bs,
                                                          const float* __restrict__ bias, float* __restrict__ y,
                                                          long long y_bs, int Co, int H, int W,
                                                          f32x4* __restrict__ ostats, XupArgs xa) {
    const int segs = W >> 7;
    // (readfirstlane: the wave index is uniform, but only this tells hipcc -- otherwise channel / segment live in VGPRs, the output
    //  descriptors with them, and every 16-byte store sits in a waterfall loop)
    const int item = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (item >= Co * segs) return;                                   // (wave-uniform)
    const int lane = threadIdx.x & 63, b = blockIdx.y;
    const int co = item / segs, sg = item - co * segs;
    const int j0 = sg * 128 + 2 * lane;
    const int jm = j0 == 0 ? W - 1 : j0 - 1, jp = j0 + 2 == W ? 0 : j0 + 2;
    const long long HW = (long long)H * W, ts = (long long)Co * HW;  // ts: distance between tap planes
    const float* pb = p9 + b * p_bs + (long long)co * HW;
    const float bs = bias ? bias[co] : 0.0f;
    float* yc = y + b * y_bs + (long long)co * 4 * HW;
    const __amdgpu_buffer_rsrc_t rs_y = lc_wt_buf(yc);
    const unsigned col_off = (unsigned)(2 * j0) * 4u, row_bytes = (unsigned)(2 * W) * 4u;

    float L[9], A[9], Bv[9], R[9];
    float xl = 0.f, xc0 = 0.f, xc1 = 0.f, xr = 0.f;                  // XUP: the raw row of x
    const float* xb_ = XUP ? xa.x + b * xa.x_bs + (long long)co * HW : nullptr;
    auto load_row = [&](int i) {
        const float* row = pb + (long long)i * W;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const float* q = row + t * ts;
            const float2 v = *reinterpret_cast<const float2*>(q + j0);
            L[t] = q[jm]; A[t] = v.x; Bv[t] = v.y; R[t] = q[jp];
        }
        if (XUP) {
            const float* q = xb_ + (long long)i * W;
            const float2 v = *reinterpret_cast<const float2*>(q + j0);
            xl = q[jm]; xc0 = v.x; xc1 = v.y; xr = q[jp];
        }
    };
    float Xm[4] = {0.f, 0.f, 0.f, 0.f}, X0[4] = {0.f, 0.f, 0.f, 0.f}, Xp[4] = {0.f, 0.f, 0.f, 0.f};
    auto xrow = [&](float (&o)[4], bool inside) {                    // rows outside the image are zeros (not 0 * x: NaN-safe)
        xup_row(xl, xc0, xc1, xr, o);
        if (!inside) { o[0] = 0.f; o[1] = 0.f; o[2] = 0.f; o[3] = 0.f; }
    };
    float* y2c = XUP ? xa.y2 + b * xa.y2_bs + (long long)co * 4 * HW : nullptr;
    const __amdgpu_buffer_rsrc_t rs_y2 = lc_wt_buf(y2c);
    auto horiz = [&](float (&Hh)[3][4], float f) {
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int t0 = 3 * ky, t1 = t0 + 1, t2 = t0 + 2;
            Hh[ky][0] = f * hz0(L[t0], A[t0], L[t1], A[t1], A[t2], Bv[t2]);
            Hh[ky][1] = f * hz1(L[t0], A[t0], A[t1], Bv[t1], A[t2], Bv[t2]);
            Hh[ky][2] = f * hz0(A[t0], Bv[t0], A[t1], Bv[t1], Bv[t2], R[t2]);
            Hh[ky][3] = f * hz1(A[t0], Bv[t0], Bv[t1], R[t1], Bv[t2], R[t2]);
        }
    };
    float Hm[3][4], H0[3][4], Hp[3][4];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int c = 0; c < 4; ++c) Hm[ky][c] = 0.0f;
    load_row(0);
    horiz(H0, 1.0f);
    if (XUP) xrow(X0, true);
    load_row(H > 1 ? 1 : 0);
    horiz(Hp, H > 1 ? 1.0f : 0.0f);
    if (XUP) xrow(Xp, H > 1);
    for (int i = 0; i < H; ++i) {
        // rows past the image are read as the last row and multiplied by 0 (no branch around the loads)
        load_row(i + 2 < H ? i + 2 : H - 1);
        const float ft = i > 0 ? 1.0f : 0.0f, fb = i + 1 < H ? 1.0f : 0.0f;
        f32x4 top, bot;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            // output row 2i:     ky = 0 -> U row 2i - 1 (dropped at i = 0), ky = 1 -> U row 2i, ky = 2 -> U row 2i + 1
            float t = ft * fmaf(0.25f, H0[0][c], 0.75f * Hm[0][c]);
            t = fmaf(0.25f, Hm[1][c], t);
            t = fmaf(0.75f, H0[1][c], t);
            t = fmaf(0.75f, H0[2][c], t);
            t = fmaf(0.25f, Hp[2][c], t);
            top[c] = t + bs;
            // output row 2i + 1: ky = 0 -> U row 2i, ky = 1 -> U row 2i + 1, ky = 2 -> U row 2i + 2 (dropped at i = H - 1)
            float u = fb * fmaf(0.75f, Hp[2][c], 0.25f * H0[2][c]);
            u = fmaf(0.25f, Hm[0][c], u);
            u = fmaf(0.75f, H0[0][c], u);
            u = fmaf(0.75f, H0[1][c], u);
            u = fmaf(0.25f, Hp[1][c], u);
            bot[c] = u + bs;
        }
        const unsigned off = (unsigned)(2 * i) * row_bytes + col_off;
        lc_st4(rs_y, off, top);
        lc_st4(rs_y, off + row_bytes, bot);
        if (XUP) {
            f32x4 xt, xb2;
#pragma unroll
            for (int c = 0; c < 4; ++c) { xt[c] = xup_top(Xm[c], X0[c]); xb2[c] = xup_bot(X0[c], Xp[c]); }
            lc_st4(rs_y2, off, xt);
            lc_st4(rs_y2, off + row_bytes, xb2);
        }
        if (STATS) {
            const float piv = __builtin_amdgcn_readfirstlane(top[0]);
            float s_ = 0.0f, q_ = 0.0f;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float d0 = top[c] - piv, d1 = bot[c] - piv;
                s_ += d0 + d1;
                q_ = fmaf(d0, d0, q_);
                q_ = fmaf(d1, d1, q_);
            }
#pragma unroll
            for (int sh = 32; sh > 0; sh >>= 1) { s_ += __shfl_xor(s_, sh, 64); q_ += __shfl_xor(q_, sh, 64); }
            if (lane == 0)
                ostats[((long long)b * Co + co) * ((long long)H * segs) + ((long long)i * segs + sg)] = f32x4{piv, 512.0f, s_, q_};
        }
        float Hn[3][4];
        horiz(Hn, i + 2 < H ? 1.0f : 0.0f);
        if (XUP) {
            float Xn[4];
            xrow(Xn, i + 2 < H);
#pragma unroll
            for (int c = 0; c < 4; ++c) { Xm[c] = X0[c]; X0[c] = Xp[c]; Xp[c] = Xn[c]; }
        }
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int c = 0; c < 4; ++c) { Hm[ky][c] = H0[ky][c]; H0[ky][c] = Hp[ky][c]; Hp[ky][c] = Hn[ky][c]; }
    }
}

}  // namespace

// x [B][C][H][W] fp32 (batch stride x_bs) -> y_split: lc_split_act_units(B, C, H, W) 16-byte units in the layout of
// lc_groupnorm_apply_split, multiplied by range->x_scale; publishes max |x * x_scale| like that pass.
extern "C" int lc_split_act_fwd(const float* x, int64_t x_bs, void* y_split, int B, int C, int H, int W,
                                lc_conv_range* range, lc_stream_t s) {
    if (!x || !y_split || !range || B <= 0 || C <= 0 || H <= 0 || W <= 0) return LC_EINVAL;
    const long long HW = (long long)H * W;
    if (C % 16 || HW % 4 || (x_bs & 3) || (reinterpret_cast<uintptr_t>(x) & 15)) return LC_EUNSUP;
    if (2ll * (C / 8) * HW * 16 >= (1ll << 31)) return LC_EUNSUP;
    long long gx = (HW + 1023) / 1024;
    while (gx > 1 && gx * (C / 8) * B > 8192) gx = (gx + 1) / 2;
    hipLaunchKernelGGL(split_plain_kernel, dim3((unsigned)gx, C / 8, B), dim3(256), 0, lc_s(s), x, (long long)x_bs,
                       reinterpret_cast<half8_t*>(y_split), (long long)2 * (C / 8) * HW, C, HW, range);
    return lc_launch_status();
}

// statistics entries per channel lc_up2_combine9_fwd leaves (0: the shape is not supported)
extern "C" int64_t lc_up2_combine9_stats_slots(int H, int W) {
    if (H <= 0 || W <= 0 || W % 128) return 0;
    return (int64_t)H * (W / 128);
}

// p9: [B][9 Co][H][W] fp32, channel t Co + co = tap t = 3 ky + kx of output channel co (the 1x1 projection of the
// low-resolution operand by W[:, :, ky, kx]); y: [B][Co][2H][2W] = conv3x3_ring(Resample(up=2)(a)) + bias.
// ostats: NULL or [B][Co][slots][4] (lc_oct_stats with unit = 1, slots = lc_up2_combine9_stats_slots(H, W)).
// W % 128 == 0 (LC_EUNSUP otherwise); p9 8-byte, y 16-byte aligned with even / 4-multiple batch strides.
static int up2_combine9(const float* p9, int64_t p_bs, const float* bias, float* y, int64_t y_bs, int B, int Co, int H, int W,
                        float* ostats, const float* x, int64_t x_bs, float* y2, int64_t y2_bs, lc_stream_t s) {
    if (!p9 || !y || B <= 0 || Co <= 0 || H <= 0 || W <= 0) return LC_EINVAL;
    if (W % 128 || (p_bs & 1) || (y_bs & 3) || (reinterpret_cast<uintptr_t>(p9) & 7) ||
        (reinterpret_cast<uintptr_t>(y) & 15))
        return LC_EUNSUP;
    if (x && ((x_bs & 1) || (y2_bs & 3) || (reinterpret_cast<uintptr_t>(x) & 7) || (reinterpret_cast<uintptr_t>(y2) & 15)))
        return LC_EUNSUP;
    if (16ll * H * W >= (1ll << 32)) return LC_EUNSUP;              // 32-bit byte offsets inside one output plane
    const long long items = (long long)Co * (W / 128);
    const dim3 grid((unsigned)((items + 3) / 4), B);
    const XupArgs xa{x, (long long)x_bs, y2, (long long)y2_bs};
    f32x4* os = reinterpret_cast<f32x4*>(ostats);
#define LC_UPC(ST, XU) hipLaunchKernelGGL((up2_combine9_kernel<ST, XU>), grid, dim3(256), 0, lc_s(s), p9, (long long)p_bs, \
                                          bias, y, (long long)y_bs, Co, H, W, os, xa)
    if (ostats) { if (x) LC_UPC(true, true); else LC_UPC(true, false); }
    else { if (x) LC_UPC(false, true); else LC_UPC(false, false); }
#undef LC_UPC
    return lc_launch_status();
}

extern "C" int lc_up2_combine9_fwd(const float* p9, int64_t p_bs, const float* bias, float* y, int64_t y_bs, int B,
                                   int Co, int H, int W, float* ostats, lc_stream_t s) {
    return up2_combine9(p9, p_bs, bias, y, y_bs, B, Co, H, W, ostats, nullptr, 0, nullptr, 0, s);
}

// ... and y2 = Resample(up=2)(x) for x [B][Co][H][W] in the same launch (bit-identical to lc_resample2x_fwd): the skip path of
// LayoutUnetV1's up-sampling ResBlock.  x 8-byte, y2 16-byte aligned, even / 4-multiple batch strides (LC_EUNSUP otherwise).
extern "C" int lc_up2_combine9_xup_fwd(const float* p9, int64_t p_bs, const float* bias, float* y, int64_t y_bs,
                                       const float* x, int64_t x_bs, float* y2, int64_t y2_bs, int B, int Co, int H, int W,
                                       float* ostats, lc_stream_t s) {
    if (!x || !y2) return LC_EINVAL;
    return up2_combine9(p9, p_bs, bias, y, y_bs, B, Co, H, W, ostats, x, x_bs, y2, y2_bs, s);
}

LC_TOUCH_TU(upfold, up2_combine9_kernel<true, false>)
