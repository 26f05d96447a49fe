// FIR x2 resampling, window [1,3,3,1], ring in W / zeros in H -- ops.Resample ops.py:52-146.
// The reference builds it from pad + zero-insertion view + two depthwise convs + strided slice
// (~6 memory passes); closed forms, horizontal pass first like the reference:
//   down: y[i,j] = sum_a k[a] * ( sum_b k[b] x[2i+a-1, 2j+b-1] ),  k = [1,3,3,1]/8
//   up:   per axis y[2i] = .25 x[i-1] + .75 x[i],  y[2i+1] = .75 x[i] + .25 x[i+1]
// HBM-bound: one read of x (neighbours hit L1/L2), one write of y.
#include "common.h"

namespace {

#pragma clang fp contract(off)

// Scalar form (any even H, W): 16 strided loads per output.
// PAIR (round 6: a resampling ResBlock of the layout model, layout_unet_v1.py:81-150, needs BOTH resample(x) and
// resample(SiLU(GroupNorm(x)))): the same pass also filters act(v) = SiLU((v - mu) * A + Bc) with the per-(sample, channel)
// rows (mu, A, Bc, 0) of lc_groupnorm_coeffs / lc_groupnorm_coeffs_os into a second output -- x is read once instead of
// three times (GroupNorm apply, resample of its result, resample of x).  The plain output is bit-identical to the unpaired kernel's.
struct PairArgs {
    const f32x4* coef; int Cpad; float* y2; long long y2_bs;
};
__device__ __forceinline__ float pair_act(float v, const f32x4& r) { return lc_silu((v - r.x) * r.y + r.z); }

template <bool PAIR>
__global__ __launch_bounds__(256) void down2_kernel(const float* __restrict__ x, long long x_bs,
                                                   float* __restrict__ y, long long y_bs, int C,
                                                   int H, int W, PairArgs pa) {
    const int Ho = H / 2, Wo = W / 2;
    const long long total = (long long)C * Ho * Wo;
    const int b = blockIdx.y;
    const float* xb = x + b * x_bs;
    float* yb = y + b * y_bs;
    float* y2b = PAIR ? pa.y2 + b * pa.y2_bs : nullptr;
    for (long long e = blockIdx.x * 256ll + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int j = e % Wo;
        const long long r = e / Wo;
        const int i = r % Ho;
        const int c = r / Ho;
        const float* xc = xb + (long long)c * H * W;
        int cols[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            int w = 2 * j + q - 1;
            cols[q] = w < 0 ? w + W : (w >= W ? w - W : w);
        }
        float acc = 0.f, acc2 = 0.f;
        const float k[4] = {0.125f, 0.375f, 0.375f, 0.125f};
        f32x4 cr = {0.f, 0.f, 0.f, 0.f};
        if (PAIR) cr = pa.coef[(long long)b * pa.Cpad + c];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int h = 2 * i + a - 1;
            float hz = 0.f, hz2 = 0.f;
            if (h >= 0 && h < H) {
                const float* row = xc + (long long)h * W;
                const float v0 = row[cols[0]], v1 = row[cols[1]], v2 = row[cols[2]], v3 = row[cols[3]];
                hz = ((k[0] * v0 + k[1] * v1) + k[2] * v2) + k[3] * v3;
                if (PAIR)
                    hz2 = ((k[0] * pair_act(v0, cr) + k[1] * pair_act(v1, cr)) + k[2] * pair_act(v2, cr)) + k[3] * pair_act(v3, cr);
            }
            acc += k[a] * hz;
            acc2 += k[a] * hz2;
        }
        lc_st(yb + e, acc);
        if (PAIR) lc_st(y2b + e, acc2);
    }
}

// Vector form (W % 256 == 0, 16-byte aligned rows): a wave covers 256 consecutive input columns
// of one (channel, output row); lane l loads the aligned float4 x[4l .. 4l+3] of each of the 4
// input rows and takes x[4l-1] / x[4l+4] from its neighbours with two DPP-class shuffles (the
// ring wrap is a scalar load only at the two ends of the 256-column segment) -> 2 outputs per
// lane, 4 vector loads instead of 32 scalar ones.  Same arithmetic order as the scalar form.
// ostats != NULL (round 5): every item also leaves one GroupNorm statistics entry (pivot, n = 128, sum (v - pivot),
// sum (v - pivot)^2) of the 128 values it stores, in the producer-statistics format with ONE channel per entry:
// ostats[(b * C + c) * slots + (i * segs + sg)], slots = Ho * segs -- the GroupNorm behind a down-sampler then needs no
// statistics pass (3 per C2 step, 12 per C3 step before).
template <bool PAIR>
__global__ __launch_bounds__(256) void down2_vec_kernel(const float* __restrict__ x, long long x_bs,
                                                       float* __restrict__ y, long long y_bs, int C,
                                                       int H, int W, f32x4* __restrict__ ostats, PairArgs pa) {
    const int Ho = H / 2, Wo = W / 2;
    const int segs = W / 256;                                  // 256-column segments per row
    const long long n_items = (long long)C * Ho * segs;        // one wave each
    const int b = blockIdx.y;
    const float* xb = x + b * x_bs;
    float* yb = y + b * y_bs;
    float* y2b = PAIR ? pa.y2 + b * pa.y2_bs : nullptr;
    const int lane = threadIdx.x & 63;
    const float k[4] = {0.125f, 0.375f, 0.375f, 0.125f};
    for (long long it = blockIdx.x * 4ll + (threadIdx.x >> 6); it < n_items; it += (long long)gridDim.x * 4) {
        const int sg = it % segs;
        const long long r = it / segs;
        const int i = r % Ho;
        const int c = r / Ho;
        const float* xc = xb + (long long)c * H * W;
        const int w0 = sg * 256 + 4 * lane;
        float acc0 = 0.f, acc1 = 0.f, bcc0 = 0.f, bcc1 = 0.f;
        f32x4 cr = {0.f, 0.f, 0.f, 0.f};
        if (PAIR) cr = pa.coef[(long long)b * pa.Cpad + c];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int h = 2 * i + a - 1;
            float hz0 = 0.f, hz1 = 0.f, gz0 = 0.f, gz1 = 0.f;
            if (h >= 0 && h < H) {                             // uniform per wave
                const float* row = xc + (long long)h * W;
                const f32x4 v = *reinterpret_cast<const f32x4*>(row + w0);
                float left = __shfl_up(v.w, 1, 64), right = __shfl_down(v.x, 1, 64);
                if (lane == 0) left = row[w0 == 0 ? W - 1 : w0 - 1];
                if (lane == 63) right = row[w0 + 4 == W ? 0 : w0 + 4];
                hz0 = ((k[0] * left + k[1] * v.x) + k[2] * v.y) + k[3] * v.z;
                hz1 = ((k[0] * v.y + k[1] * v.z) + k[2] * v.w) + k[3] * right;
                if (PAIR) {
                    const float al = pair_act(left, cr), ax = pair_act(v.x, cr), ay = pair_act(v.y, cr), az = pair_act(v.z, cr),
                                aw = pair_act(v.w, cr), ar = pair_act(right, cr);
                    gz0 = ((k[0] * al + k[1] * ax) + k[2] * ay) + k[3] * az;
                    gz1 = ((k[0] * ay + k[1] * az) + k[2] * aw) + k[3] * ar;
                }
            }
            acc0 += k[a] * hz0;
            acc1 += k[a] * hz1;
            bcc0 += k[a] * gz0;
            bcc1 += k[a] * gz1;
        }
        float2 o; o.x = acc0; o.y = acc1;
        lc_st2(yb + ((long long)c * Ho + i) * Wo + w0 / 2, o);
        if (PAIR) {
            float2 o2; o2.x = bcc0; o2.y = bcc1;
            lc_st2(y2b + ((long long)c * Ho + i) * Wo + w0 / 2, o2);
        }
        if (ostats) {                                          // (uniform)
            const float piv = __builtin_amdgcn_readfirstlane(acc0);
            const float d0 = acc0 - piv, d1 = acc1 - piv;
            float s_ = d0 + d1, q_ = fmaf(d1, d1, d0 * d0);
#pragma unroll
            for (int sh = 32; sh > 0; sh >>= 1) { s_ += __shfl_xor(s_, sh, 64); q_ += __shfl_xor(q_, sh, 64); }
            if (lane == 0)
                ostats[((long long)b * C + c) * ((long long)Ho * segs) + ((long long)i * segs + sg)] = f32x4{piv, 128.0f, s_, q_};
        }
    }
}

// one thread per INPUT pixel -> 2x2 outputs.  (A vector form -- a wave per 256 input columns of a row, float4 loads, neighbours by
// shuffles, 16-byte stores -- was written and measured in round 5: bit-equal, and TWICE as slow, 29.6 vs ~19 us on the two C2
// shapes it took, 67.8 vs 29 us on C3's; this kernel writes 4 bytes for every byte it reads and already runs at the rate the
// write-through path absorbs them.  profiles/r05_level0.txt section 7.)
template <bool PAIR>
__global__ __launch_bounds__(256) void up2_kernel(const float* __restrict__ x, long long x_bs,
                                                 float* __restrict__ y, long long y_bs, int C,
                                                 int H, int W, PairArgs pa) {
    const long long total = (long long)C * H * W;
    const int b = blockIdx.y;
    const float* xb = x + b * x_bs;
    float* yb = y + b * y_bs;
    float* y2b = PAIR ? pa.y2 + b * pa.y2_bs : nullptr;
    const int W2 = 2 * W;
    for (long long e = blockIdx.x * 256ll + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int j = e % W;
        const long long r = e / W;
        const int i = r % H;
        const int c = r / H;
        const float* xc = xb + (long long)c * H * W;
        const int jm = j == 0 ? W - 1 : j - 1, jp = j == W - 1 ? 0 : j + 1;
        float ev[3], od[3];  // horizontal results for rows i-1, i, i+1
        float ev2[3], od2[3];
        f32x4 cr = {0.f, 0.f, 0.f, 0.f};
        if (PAIR) cr = pa.coef[(long long)b * pa.Cpad + c];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const int h = i + d - 1;
            ev2[d] = 0.f; od2[d] = 0.f;
            if (h >= 0 && h < H) {
                const float* row = xc + (long long)h * W;
                const float xm = row[jm], x0 = row[j], xp = row[jp];
                ev[d] = 0.25f * xm + 0.75f * x0;
                od[d] = 0.75f * x0 + 0.25f * xp;
                if (PAIR) {
                    const float am = pair_act(xm, cr), a0 = pair_act(x0, cr), ap = pair_act(xp, cr);
                    ev2[d] = 0.25f * am + 0.75f * a0;
                    od2[d] = 0.75f * a0 + 0.25f * ap;
                }
            } else {
                ev[d] = 0.f; od[d] = 0.f;
            }
        }
        float* yc = yb + (long long)c * 4 * H * W + (long long)(2 * i) * W2 + 2 * j;
        float2 top, bot;
        top.x = 0.25f * ev[0] + 0.75f * ev[1];
        top.y = 0.25f * od[0] + 0.75f * od[1];
        bot.x = 0.75f * ev[1] + 0.25f * ev[2];
        bot.y = 0.75f * od[1] + 0.25f * od[2];
        lc_st2(yc, top);
        lc_st2(yc + W2, bot);
        if (PAIR) {
            float* yc2 = y2b + (long long)c * 4 * H * W + (long long)(2 * i) * W2 + 2 * j;
            float2 t2, b2;
            t2.x = 0.25f * ev2[0] + 0.75f * ev2[1];
            t2.y = 0.25f * od2[0] + 0.75f * od2[1];
            b2.x = 0.75f * ev2[1] + 0.25f * ev2[2];
            b2.y = 0.75f * od2[1] + 0.25f * od2[2];
            lc_st2(yc2, t2);
            lc_st2(yc2 + W2, b2);
        }
    }
}

}  // namespace

// statistics entries per channel the down-sampler leaves (0: this shape takes the scalar kernel, which leaves none)
extern "C" int64_t lc_resample2x_stats_slots(int H, int W, int dir) {
    if (dir >= 0 || H <= 0 || W <= 0 || (H & 1) || W % 256) return 0;
    return (int64_t)(H / 2) * (W / 256);
}

static int resample2x(const float* x, int64_t x_bs, float* y, int64_t y_bs, int B, int C, int H, int W, int dir,
                      float* ostats, lc_stream_t s, const float* coef = nullptr, int Cpad = 0, float* y2 = nullptr,
                      int64_t y2_bs = 0);

// y = resample(x) and y_act = resample(SiLU((x - mu) * A + Bc)) in ONE pass over x; coeffs: [B][Cpad] rows (mu, A, Bc, 0) as
// lc_groupnorm_coeffs / lc_groupnorm_coeffs_os write them.  y is bit-identical to lc_resample2x_fwd's.
extern "C" int lc_resample2x_pair_fwd(const float* x, int64_t x_bs, const float* coeffs, int Cpad, float* y, int64_t y_bs,
                                      float* y_act, int64_t ya_bs, int B, int C, int H, int W, int dir, lc_stream_t s) {
    if (!coeffs || !y_act || Cpad < C) return LC_EINVAL;
    return resample2x(x, x_bs, y, y_bs, B, C, H, W, dir, nullptr, s, coeffs, Cpad, y_act, ya_bs);
}

extern "C" int lc_resample2x_fwd(const float* x, int64_t x_bs, float* y, int64_t y_bs, int B, int C,
                                 int H, int W, int dir, lc_stream_t s) {
    return resample2x(x, x_bs, y, y_bs, B, C, H, W, dir, nullptr, s);
}

// ... the same, and (dir < 0, lc_resample2x_stats_slots(H, W, dir) > 0) one statistics entry per (sample, channel, slot) of
// the OUTPUT into ostats[B, C, slots, 4] (lc_oct_stats with unit = 1).  LC_EUNSUP when the shape / alignment takes the
// scalar kernel: the caller asks lc_resample2x_stats_slots first and falls back to lc_resample2x_fwd + a statistics pass.
extern "C" int lc_resample2x_stats_fwd(const float* x, int64_t x_bs, float* y, int64_t y_bs, int B, int C,
                                       int H, int W, int dir, float* ostats, lc_stream_t s) {
    if (!ostats) return LC_EINVAL;
    return resample2x(x, x_bs, y, y_bs, B, C, H, W, dir, ostats, s);
}

static int resample2x(const float* x, int64_t x_bs, float* y, int64_t y_bs, int B, int C, int H, int W, int dir,
                      float* ostats, lc_stream_t s, const float* coef, int Cpad, float* y2, int64_t y2_bs) {
    if (!x || !y || B <= 0 || C <= 0 || H <= 0 || W <= 0) return LC_EINVAL;
    if (ostats && dir >= 0) return LC_EUNSUP;
    const bool pair = y2 != nullptr;
    const PairArgs pa{reinterpret_cast<const f32x4*>(coef), Cpad, y2, (long long)y2_bs};
    if (dir < 0) {
        if ((H & 1) || (W & 1)) return LC_EUNSUP;
        const long long total = (long long)C * (H / 2) * (W / 2);
        const bool vec = W % 256 == 0 && (x_bs & 3) == 0 && (y_bs & 1) == 0 &&
                         (reinterpret_cast<uintptr_t>(x) & 15) == 0 &&
                         (reinterpret_cast<uintptr_t>(y) & 7) == 0 &&
                         (!pair || ((y2_bs & 1) == 0 && (reinterpret_cast<uintptr_t>(y2) & 7) == 0));
        if (vec) {
            const long long items = (long long)C * (H / 2) * (W / 256);
            int blocks = (int)((items + 3) / 4 > 16384 ? 16384 : (items + 3) / 4);
            if (pair)
                hipLaunchKernelGGL(down2_vec_kernel<true>, dim3(blocks, B), dim3(256), 0, lc_s(s), x,
                                   (long long)x_bs, y, (long long)y_bs, C, H, W, (f32x4*)nullptr, pa);
            else
                hipLaunchKernelGGL(down2_vec_kernel<false>, dim3(blocks, B), dim3(256), 0, lc_s(s), x,
                                   (long long)x_bs, y, (long long)y_bs, C, H, W, reinterpret_cast<f32x4*>(ostats), pa);
            return lc_launch_status();
        }
        if (ostats) return LC_EUNSUP;
        int blocks = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
        if (pair)
            hipLaunchKernelGGL(down2_kernel<true>, dim3(blocks, B), dim3(256), 0, lc_s(s), x, (long long)x_bs, y,
                               (long long)y_bs, C, H, W, pa);
        else
            hipLaunchKernelGGL(down2_kernel<false>, dim3(blocks, B), dim3(256), 0, lc_s(s), x, (long long)x_bs, y,
                               (long long)y_bs, C, H, W, pa);
    } else {
        if (((y_bs | (int64_t)(2 * W)) & 1) || (reinterpret_cast<uintptr_t>(y) & 7)) return LC_EUNSUP;
        if (pair && ((y2_bs & 1) || (reinterpret_cast<uintptr_t>(y2) & 7))) return LC_EUNSUP;
        const long long total = (long long)C * H * W;
        int blocks = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
        if (pair)
            hipLaunchKernelGGL(up2_kernel<true>, dim3(blocks, B), dim3(256), 0, lc_s(s), x, (long long)x_bs, y,
                               (long long)y_bs, C, H, W, pa);
        else
            hipLaunchKernelGGL(up2_kernel<false>, dim3(blocks, B), dim3(256), 0, lc_s(s), x, (long long)x_bs, y,
                               (long long)y_bs, C, H, W, pa);
    }
    return lc_launch_status();
}

LC_TOUCH_TU(resample, down2_kernel<false>)
