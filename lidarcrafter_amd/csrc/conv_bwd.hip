// Backward of the ring convolution with respect to its WEIGHTS (training, SURVEY.md section 8f-4;
// reference: autograd of ops.Conv2d + ops.Pad, lidargen/models/unets/ops.py:32-49,149-173):
//   dW[co][ci][ky][kx] = sum_{b,h,w} dY[b,co,h,w] * Xpad[b,ci,h+ky-1,w+kx-1]      (W circular, H zeros)
//   db[co]             = sum_{b,h,w} dY[b,co,h,w]
// (the gradient with respect to the INPUT is the same ring convolution of dY with the transposed,
//  180-degree rotated kernel -- it runs on the forward kernels, lidarcrafter_amd/autograd.py).
//
// Implicit GEMM on the fp32 matrix cores (v_mfma_f32_32x32x2_f32, exact fp32): M = output channels,
// N = input channels, K = pixels, one accumulator per tap.  A block of 4 waves owns a 64 co x 64 ci
// tile for all KS*KS taps and walks over pixel tiles of one image row x 64 columns staged in LDS
// (dY [64][64], X [64][KS rows][64 + 2 halo], row strides odd so that the 32 lanes of an operand
// read hit 32 banks); the pixel tiles of the whole batch are dealt round-robin to `nsplit` blocks per
// channel tile, each writing its partial sums, and a second kernel adds the partials in index
// order (deterministic, no atomics).
#include "common.h"

namespace {

template <int KS>
__global__ __launch_bounds__(256, 2) void conv_wgrad_kernel(const float* __restrict__ x, long long x_bs,
                                                           const float* __restrict__ dy, long long dy_bs,
                                                           float* __restrict__ part, int B, int Ci, int Co,
                                                           int H, int W, int ncib, int nsplit) {
    constexpr int HALO = KS / 2, NTAP = KS * KS, TW = 64;
    constexpr int DS = TW + 1;                       // dY row stride (floats), odd
    constexpr int XC = TW + 2 * HALO;                // staged columns of X
    constexpr int XRS = XC | 1;                      // odd row stride
    constexpr int XCS = KS * XRS + (((KS * XRS) & 1) ? 0 : 1);   // odd channel stride
    // 17 KB + 52 KB (3x3): two blocks per CU, so that one block's staging (global loads, exposed
    // latency) runs under the other block's MFMAs -- the first version (128-pixel tiles, one block
    // per CU, scalar loads with a modulo per element) spent more than half its time staging.
    __shared__ float dyt[64 * DS];
    __shared__ float xt[64 * XCS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wco = wave >> 1, wci = wave & 1;
    const int cob = blockIdx.x / ncib, cib = blockIdx.x - cob * ncib;
    const int co0 = cob * 64, ci0 = cib * 64;
    const int split = blockIdx.y;
    const int tiles_w = (W + TW - 1) / TW;
    const int ntiles = B * H * tiles_w;
    f32x16 acc[NTAP];
#pragma unroll
    for (int t = 0; t < NTAP; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
    const int l31 = lane & 31, kk = lane >> 5;
    const long long HW = (long long)H * W;
    // float4 staging when rows are 16-byte aligned and tiles are whole
    const bool vec = (W % TW) == 0 && (HW & 3) == 0 && (x_bs & 3) == 0 && (dy_bs & 3) == 0 &&
                     (reinterpret_cast<unsigned long long>(x) & 15) == 0 &&
                     (reinterpret_cast<unsigned long long>(dy) & 15) == 0;
    for (int tile = split; tile < ntiles; tile += nsplit) {
        const int tw = tile % tiles_w;
        const int h = (tile / tiles_w) % H, b = tile / (tiles_w * H);
        const int w0 = tw * TW;
        const float* dyb = dy + b * dy_bs + (long long)h * W + w0;
        const float* xb = x + b * x_bs;
        __syncthreads();                              // previous tile's operands are consumed
        if (vec) {
            // dY: 64 channels x 16 float4; X interior: 64 channels x KS rows x 16 float4; halo
            // columns (ring) as scalars
            f32x4 dv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int e = tid + i * 256, c = e >> 4, q = e & 15;
                dv[i] = co0 + c < Co ? *reinterpret_cast<const f32x4*>(dyb + (long long)(co0 + c) * HW + 4 * q)
                                     : f32x4{0.f, 0.f, 0.f, 0.f};
            }
            float hv[(64 * KS * 2 * HALO + 255) / 256 + 1];
#pragma unroll
            for (int i = 0; i * 256 < 64 * KS * 2 * HALO; ++i) {
                const int e = tid + i * 256;
                const int cr = e / (2 * HALO > 0 ? 2 * HALO : 1), side = e - cr * (2 * HALO > 0 ? 2 * HALO : 1);
                const int c = cr / KS, r = cr - c * KS;
                const int gh = h - HALO + r;
                int gw = side < HALO ? w0 - HALO + side : w0 + TW + (side - HALO);
                gw = gw < 0 ? gw + W : (gw >= W ? gw - W : gw);
                hv[i] = (e < 64 * KS * 2 * HALO && ci0 + c < Ci && gh >= 0 && gh < H)
                            ? xb[(long long)(ci0 + c) * HW + (long long)gh * W + gw] : 0.0f;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int e = tid + i * 256, c = e >> 4, q = e & 15;
                float* d = dyt + c * DS + 4 * q;
                d[0] = dv[i].x; d[1] = dv[i].y; d[2] = dv[i].z; d[3] = dv[i].w;
            }
            // X interior in batches of 4 float4 per thread (the 144 accumulator registers leave
            // room for no more; the other resident block covers the latency)
#pragma unroll
            for (int i0 = 0; i0 < KS * 4; i0 += 4) {
                f32x4 xv[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int e = tid + (i0 + i) * 256, cr = e >> 4, q = e & 15;
                    const int c = cr / KS, r = cr - c * KS;
                    const int gh = h - HALO + r;
                    xv[i] = (ci0 + c < Ci && gh >= 0 && gh < H)
                                ? *reinterpret_cast<const f32x4*>(xb + (long long)(ci0 + c) * HW + (long long)gh * W + w0 + 4 * q)
                                : f32x4{0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int e = tid + (i0 + i) * 256, cr = e >> 4, q = e & 15;
                    const int c = cr / KS, r = cr - c * KS;
                    float* d = xt + c * XCS + r * XRS + HALO + 4 * q;
                    d[0] = xv[i].x; d[1] = xv[i].y; d[2] = xv[i].z; d[3] = xv[i].w;
                }
            }
#pragma unroll
            for (int i = 0; i * 256 < 64 * KS * 2 * HALO; ++i) {
                const int e = tid + i * 256;
                if (e < 64 * KS * 2 * HALO) {
                    const int cr = e / (2 * HALO > 0 ? 2 * HALO : 1), side = e - cr * (2 * HALO > 0 ? 2 * HALO : 1);
                    const int c = cr / KS, r = cr - c * KS;
                    xt[c * XCS + r * XRS + (side < HALO ? side : TW + side)] = hv[i];
                }
            }
        } else {
            for (int e = tid; e < 64 * TW; e += 256) {
                const int c = e / TW, p = e - c * TW;
                const int co = co0 + c, w = w0 + p;
                dyt[c * DS + p] = (co < Co && w < W) ? dy[b * dy_bs + (long long)co * HW + (long long)h * W + w] : 0.0f;
            }
            for (int e = tid; e < 64 * KS * XC; e += 256) {
                const int c = e / (KS * XC), rem = e - c * (KS * XC);
                const int r = rem / XC, q = rem - r * XC;
                const int ci = ci0 + c, gh = h - HALO + r;
                int gw = w0 - HALO + q;
                gw %= W; if (gw < 0) gw += W;
                xt[c * XCS + r * XRS + q] =
                    (ci < Ci && gh >= 0 && gh < H) ? xb[(long long)ci * HW + (long long)gh * W + gw] : 0.0f;
            }
        }
        __syncthreads();
        const float* ap = dyt + (wco * 32 + l31) * DS + kk;
        const float* bp = xt + (wci * 32 + l31) * XCS + kk;
#pragma unroll 4
        for (int p = 0; p < TW; p += 2) {
            const float a = ap[p];
#pragma unroll
            for (int t = 0; t < NTAP; ++t) {
                const int ky = t / KS, kx = t - ky * KS;
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bp[ky * XRS + p + kx], acc[t], 0, 0, 0);
            }
        }
    }
    // partial sums of this split: part[split][co][ci][tap]
    float* pp = part + (long long)split * Co * Ci * NTAP;
#pragma unroll
    for (int t = 0; t < NTAP; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + wco * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
            const int ci = ci0 + wci * 32 + l31;
            if (co < Co && ci < Ci) pp[((long long)co * Ci + ci) * NTAP + t] = acc[t][r];
        }
}

// ---------------------------------------------------------------------------------------------
// f16x2-split weight gradient (round 3).  The same implicit GEMM on v_mfma_f32_32x32x16_f16 with the
// operand split of conv_f16x2.hip: dY and X are pre-scaled by the power-of-two scales of their range
// records (measured on the device right before the backward / forward conv of the same tensors),
// split into fp16 hi (11 significant bits, exact) + lo, and every product is evaluated as
// ah*bh + ah*bl + al*bh with fp32 accumulation (per-product error ~2^-22): 3 MFMAs of 32 cycles per
// 16 pixels instead of 8 fp32 MFMAs of 64 cycles.
// The contraction index of this GEMM is the PIXEL, and an MFMA operand register holds 8 consecutive
// k -- 8 consecutive pixels of one channel row, exactly how NCHW lies in memory: no transposition
// anywhere.  The kx = 0 / 2 taps need the same 8 pixels shifted by one: each (row, plane) is read
// once as an aligned 16-byte fragment plus the two neighbouring dwords, and the shifted fragments
// are funnelled out of them with five v_alignbit_b32 (three of them shared by both shifts).
// Tile = 2 image rows x 32 columns (K = 64 pixels, 4 MFMA k-steps), block = 4 waves on a 64 co x 64 ci
// channel tile with one accumulator per tap; LDS 68 KB (3x3) -> two blocks per CU, and the next
// tile's global loads are in flight (registers) while the current one is contracted.
typedef _Float16 half8_w __attribute__((ext_vector_type(8)));
typedef _Float16 half4_w __attribute__((ext_vector_type(4)));
typedef _Float16 half2_w __attribute__((ext_vector_type(2)));
typedef unsigned u32x4_w __attribute__((ext_vector_type(4)));
typedef float f32x2_w __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void split4(const f32x4 v, float sc, half4_w& hi, half4_w& lo) {
    const float s0 = v.x * sc, s1 = v.y * sc, s2 = v.z * sc, s3 = v.w * sc;
    const float h0 = __uint_as_float(__float_as_uint(s0) & 0xFFFFE000u);
    const float h1 = __uint_as_float(__float_as_uint(s1) & 0xFFFFE000u);
    const float h2 = __uint_as_float(__float_as_uint(s2) & 0xFFFFE000u);
    const float h3 = __uint_as_float(__float_as_uint(s3) & 0xFFFFE000u);
    const half2_w a = __builtin_bit_cast(half2_w, __builtin_amdgcn_cvt_pkrtz(h0, h1));
    const half2_w b = __builtin_bit_cast(half2_w, __builtin_amdgcn_cvt_pkrtz(h2, h3));
    f32x2_w r0 = {s0 - h0, s1 - h1}, r1 = {s2 - h2, s3 - h3};
    const half2_w c = __builtin_convertvector(r0, half2_w), d = __builtin_convertvector(r1, half2_w);
    hi = half4_w{a.x, a.y, b.x, b.y};
    lo = half4_w{c.x, c.y, d.x, d.y};
}

template <int KS>
__global__ __launch_bounds__(256, 2) void conv_wgrad_h_kernel(
    const float* __restrict__ x, long long x_bs, const float* __restrict__ dy, long long dy_bs,
    float* __restrict__ part, const lc_conv_range* __restrict__ rx, const lc_conv_range* __restrict__ rdy,
    int B, int Ci, int Co, int H, int W, int ncib, int nsplit) {
    constexpr int HALO = KS / 2, NTAP = KS * KS, TR = 2, TWX = 32;
    constexpr int XR = TR + 2 * HALO;                  // staged X rows
    constexpr int XOFF = HALO ? 8 : 0;                 // halfs in front of column w0 (16-byte aligned interior)
    constexpr int RS = TWX + 2 * XOFF;                 // halfs per staged X row
    constexpr int CS = XR * RS + 8;                    // halfs per X channel: 200 / 72 -> 100 / 36 dwords, = 4 * odd
    constexpr int DS = TR * TWX + 8;                   // halfs per dY channel (72)   (mod 64: 16 lanes, 16 bank quads)
    __shared__ __attribute__((aligned(16))) _Float16 xs_h[64 * CS];
    __shared__ __attribute__((aligned(16))) _Float16 xs_l[64 * CS];
    __shared__ __attribute__((aligned(16))) _Float16 ds_h[64 * DS];
    __shared__ __attribute__((aligned(16))) _Float16 ds_l[64 * DS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wco = wave >> 1, wci = wave & 1;
    const int cob = blockIdx.x / ncib, cib = blockIdx.x - cob * ncib;
    const int co0 = cob * 64, ci0 = cib * 64;
    const int split = blockIdx.y;
    const int tiles_w = W / TWX, tiles_h = H / TR;
    const int ntiles = B * tiles_h * tiles_w;
    const float sx = rx->x_scale, sdy = rdy->x_scale;
    f32x16 acc[NTAP];
#pragma unroll
    for (int t = 0; t < NTAP; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
    const int l31 = lane & 31, kk = lane >> 5;
    const long long HW = (long long)H * W;
    constexpr int NXV = XR * 2;                        // float4 of X per thread: 64 ch x XR rows x 8 / 256
    constexpr int NHV = HALO ? 2 : 0;                  // halo scalars per thread: 64 ch x XR rows x 2 / 256
    f32x4 dv[4], xv[NXV];
    float hv[NHV > 0 ? NHV : 1];
    auto load_tile = [&](int tile) {
        const int tw = tile % tiles_w;
        const int th = (tile / tiles_w) % tiles_h, b = tile / (tiles_w * tiles_h);
        const int w0 = tw * TWX, h0 = th * TR;
        const float* dyb = dy + b * dy_bs + (long long)h0 * W + w0;
        const float* xb = x + b * x_bs;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + i * 256, c = e >> 4, r = (e >> 3) & 1, q = e & 7;
            dv[i] = co0 + c < Co ? *reinterpret_cast<const f32x4*>(dyb + (long long)(co0 + c) * HW + r * W + 4 * q)
                                 : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int i = 0; i < NXV; ++i) {
            const int e = tid + i * 256, c = e / (XR * 8), rem = e - c * (XR * 8), r = rem >> 3, q = rem & 7;
            const int gh = h0 - HALO + r;
            xv[i] = (ci0 + c < Ci && gh >= 0 && gh < H)
                        ? *reinterpret_cast<const f32x4*>(xb + (long long)(ci0 + c) * HW + (long long)gh * W + w0 + 4 * q)
                        : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int i = 0; i < NHV; ++i) {
            const int e = tid + i * 256, c = e / (XR * 2), rem = e - c * (XR * 2), r = rem >> 1, side = rem & 1;
            const int gh = h0 - HALO + r;
            int gw = side ? w0 + TWX : w0 - 1;
            gw = gw < 0 ? gw + W : (gw >= W ? gw - W : gw);
            hv[i] = (ci0 + c < Ci && gh >= 0 && gh < H) ? xb[(long long)(ci0 + c) * HW + (long long)gh * W + gw] : 0.0f;
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + i * 256, c = e >> 4, r = (e >> 3) & 1, q = e & 7;
            half4_w hi, lo;
            split4(dv[i], sdy, hi, lo);
            *reinterpret_cast<half4_w*>(ds_h + c * DS + r * TWX + 4 * q) = hi;
            *reinterpret_cast<half4_w*>(ds_l + c * DS + r * TWX + 4 * q) = lo;
        }
#pragma unroll
        for (int i = 0; i < NXV; ++i) {
            const int e = tid + i * 256, c = e / (XR * 8), rem = e - c * (XR * 8), r = rem >> 3, q = rem & 7;
            half4_w hi, lo;
            split4(xv[i], sx, hi, lo);
            *reinterpret_cast<half4_w*>(xs_h + c * CS + r * RS + XOFF + 4 * q) = hi;
            *reinterpret_cast<half4_w*>(xs_l + c * CS + r * RS + XOFF + 4 * q) = lo;
        }
#pragma unroll
        for (int i = 0; i < NHV; ++i) {
            const int e = tid + i * 256, c = e / (XR * 2), rem = e - c * (XR * 2), r = rem >> 1, side = rem & 1;
            const float sv = hv[i] * sx;
            const float hf = __uint_as_float(__float_as_uint(sv) & 0xFFFFE000u);
            const int d = c * CS + r * RS + (side ? XOFF + TWX : XOFF - 1);
            xs_h[d] = (_Float16)hf;
            xs_l[d] = (_Float16)(sv - hf);
        }
    };
    int tile = split;
    if (tile < ntiles) load_tile(tile);
    for (; tile < ntiles; tile += nsplit) {
        __syncthreads();                              // previous tile's operands are consumed
        store_tile();
        __syncthreads();
        if (tile + nsplit < ntiles) load_tile(tile + nsplit);   // in flight under the MFMAs below
        const _Float16* ahp = ds_h + (wco * 32 + l31) * DS + 8 * kk;
        const _Float16* alp = ds_l + (wco * 32 + l31) * DS + 8 * kk;
        const _Float16* bhp = xs_h + (wci * 32 + l31) * CS + XOFF + 8 * kk;
        const _Float16* blp = xs_l + (wci * 32 + l31) * CS + XOFF + 8 * kk;
#pragma unroll 1   // (unrolled, hipcc hoists every fragment of the four k-steps: 259 spilled registers)
        for (int s = 0; s < 4; ++s) {
            const int r = s >> 1, px0 = (s & 1) * 16;
            const half8_w ah = *reinterpret_cast<const half8_w*>(ahp + r * TWX + px0);
            const half8_w al = *reinterpret_cast<const half8_w*>(alp + r * TWX + px0);
#pragma unroll
            for (int ky = 0; ky < KS; ++ky) {
                __builtin_amdgcn_sched_barrier(0);
                const _Float16* ph = bhp + (r + ky) * RS + px0;
                const _Float16* pl = blp + (r + ky) * RS + px0;
                const u32x4_w mh = *reinterpret_cast<const u32x4_w*>(ph);
                const u32x4_w ml = *reinterpret_cast<const u32x4_w*>(pl);
                if constexpr (KS == 1) {
                    const half8_w bh = __builtin_bit_cast(half8_w, mh), bl = __builtin_bit_cast(half8_w, ml);
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[0], 0, 0, 0);
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[0], 0, 0, 0);
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[0], 0, 0, 0);
                } else {
                    const unsigned Lh = *reinterpret_cast<const unsigned*>(ph - 2);
                    const unsigned Rh = *reinterpret_cast<const unsigned*>(ph + 8);
                    const unsigned Ll = *reinterpret_cast<const unsigned*>(pl - 2);
                    const unsigned Rl = *reinterpret_cast<const unsigned*>(pl + 8);
                    const unsigned h01 = __builtin_amdgcn_alignbit(mh.y, mh.x, 16), h12 = __builtin_amdgcn_alignbit(mh.z, mh.y, 16),
                                   h23 = __builtin_amdgcn_alignbit(mh.w, mh.z, 16);
                    const unsigned l01 = __builtin_amdgcn_alignbit(ml.y, ml.x, 16), l12 = __builtin_amdgcn_alignbit(ml.z, ml.y, 16),
                                   l23 = __builtin_amdgcn_alignbit(ml.w, ml.z, 16);
                    const u32x4_w lh = {__builtin_amdgcn_alignbit(mh.x, Lh, 16), h01, h12, h23};
                    const u32x4_w ll = {__builtin_amdgcn_alignbit(ml.x, Ll, 16), l01, l12, l23};
                    const u32x4_w rh = {h01, h12, h23, __builtin_amdgcn_alignbit(Rh, mh.w, 16)};
                    const u32x4_w rl = {l01, l12, l23, __builtin_amdgcn_alignbit(Rl, ml.w, 16)};
                    const half8_w bh[3] = {__builtin_bit_cast(half8_w, lh), __builtin_bit_cast(half8_w, mh),
                                           __builtin_bit_cast(half8_w, rh)};
                    const half8_w bl[3] = {__builtin_bit_cast(half8_w, ll), __builtin_bit_cast(half8_w, ml),
                                           __builtin_bit_cast(half8_w, rl)};
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const int t = ky * 3 + kx;
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[kx], acc[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[kx], acc[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[kx], acc[t], 0, 0, 0);
                    }
                }
            }
        }
    }
    // partial sums of this split, still carrying both scales: part[split][tap][co][ci] (ci contiguous:
    // coalesced; the fold kernel writes the [co][ci][tap] layout of the weight)
    float* pp = part + (long long)split * Co * Ci * NTAP;
#pragma unroll
    for (int t = 0; t < NTAP; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + wco * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
            const int ci = ci0 + wci * 32 + l31;
            if (co < Co && ci < Ci) pp[((long long)t * Co + co) * Ci + ci] = acc[t][r];
        }
}

// dW = unscale * sum over the splits (index order: deterministic) of the [split][tap][co][ci] partials.
// 64 consecutive elements x 4 split groups per block; the groups meet in LDS in a fixed order.
__global__ __launch_bounds__(256) void wgrad_fold_h_kernel(const float* __restrict__ part, long long n, int nsplit,
                                                          float* __restrict__ dw, int Co, int Ci, int ntap,
                                                          const lc_conv_range* __restrict__ rx,
                                                          const lc_conv_range* __restrict__ rdy, int accumulate) {
    __shared__ float sh[4][64];
    const int el = threadIdx.x & 63, g = threadIdx.x >> 6;
    const long long e = blockIdx.x * 64ll + el;
    float s = 0.0f;
    if (e < n)
        for (int k = g; k < nsplit; k += 4) s += part[k * n + e];
    sh[g][el] = s;
    __syncthreads();
    if (g == 0 && e < n) {
        const float v = ((sh[0][el] + sh[1][el]) + (sh[2][el] + sh[3][el])) * (rx->x_unscale * rdy->x_unscale);
        const int ci = (int)(e % Ci);
        const long long q = e / Ci;
        const int co = (int)(q % Co), t = (int)(q / Co);
        float* d = dw + ((long long)co * Ci + ci) * ntap + t;
        *d = accumulate ? *d + v : v;
    }
}

__global__ void wgrad_reduce_kernel(const float* __restrict__ part, long long n, int nsplit,
                                    float* __restrict__ dw, int accumulate) {
    const long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (e >= n) return;
    float s = 0.0f;
    for (int k = 0; k < nsplit; ++k) s += part[k * n + e];
    dw[e] = accumulate ? dw[e] + s : s;
}

// db[co] = sum over batch and plane of dY: one block per (channel, sample) plane (float4 loads, fp32
// per thread, fp64 across the block) writes a partial; a second launch adds the B partials of a
// channel in index order (deterministic).  (First version: one block per channel looping over
// the batch with scalar loads -- 6 ms of a 74 ms training step.)
__global__ __launch_bounds__(256) void bias_grad_plane_kernel(const float* __restrict__ dy, long long dy_bs,
                                                             double* __restrict__ part, int B, long long HW) {
    const int co = blockIdx.x, b = blockIdx.y;
    const float* p = dy + b * dy_bs + (long long)co * HW;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if ((HW & 3) == 0 && (reinterpret_cast<unsigned long long>(p) & 15) == 0) {
        const f32x4* p4 = reinterpret_cast<const f32x4*>(p);
        for (long long i = threadIdx.x; i < (HW >> 2); i += 256) {
            const f32x4 v = p4[i];
            a0 += v.x; a1 += v.y; a2 += v.z; a3 += v.w;
        }
    } else {
        for (long long i = threadIdx.x; i < HW; i += 256) a0 += p[i];
    }
    double s = ((double)a0 + (double)a1) + ((double)a2 + (double)a3);
    s = lc_wave_sum(s);
    __shared__ double sh[4];
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[(long long)co * B + b] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}
__global__ void bias_grad_fold_kernel(const double* __restrict__ part, float* __restrict__ db, int Co, int B,
                                      int accumulate) {
    const int co = blockIdx.x * blockDim.x + threadIdx.x;
    if (co >= Co) return;
    double s = 0.0;
    for (int b = 0; b < B; ++b) s += part[(long long)co * B + b];
    db[co] = accumulate ? db[co] + (float)s : (float)s;
}

int wgrad_splits(int B, int Ci, int Co, int H, int W) {
    const int tiles = B * H * ((W + 63) / 64);
    const int ctiles = ((Co + 63) / 64) * ((Ci + 63) / 64);
    int n = (512 + ctiles - 1) / ctiles;              // one full wave of blocks (2 resident per CU): equal
                                                      // work per block, and half the partials to reduce
    if (n > tiles) n = tiles;
    if (n < 1) n = 1;
    return n;
}

}  // namespace

extern "C" int64_t lc_conv2d_ring_wgrad_scratch_elems(int B, int Ci, int Co, int H, int W, int ks) {
    if (B <= 0 || Ci <= 0 || Co <= 0 || H <= 0 || W <= 0 || (ks != 1 && ks != 3)) return 0;
    // weight partials (rounded up to an even count) + Co * B fp64 bias partials
    const int64_t wp = (int64_t)wgrad_splits(B, Ci, Co, H, W) * Co * Ci * ks * ks;
    return ((wp + 1) & ~(int64_t)1) + 2 * (int64_t)Co * B;
}

extern "C" int lc_conv2d_ring_wgrad(const float* x, int64_t x_bs, const float* dy, int64_t dy_bs,
                                    float* scratch, float* dw, float* dbias, int B, int Ci, int Co,
                                    int H, int W, int ks, int accumulate, lc_stream_t s) {
    if (!x || !dy || !scratch || !dw || B <= 0 || Ci <= 0 || Co <= 0 || H <= 0 || W <= 0)
        return LC_EINVAL;
    if (ks != 1 && ks != 3) return LC_EUNSUP;
    const int ncib = (Ci + 63) / 64, ncob = (Co + 63) / 64;
    const int nsplit = wgrad_splits(B, Ci, Co, H, W);
    dim3 grid(ncob * ncib, nsplit);
    if (ks == 3)
        hipLaunchKernelGGL(conv_wgrad_kernel<3>, grid, dim3(256), 0, lc_s(s), x, (long long)x_bs, dy,
                           (long long)dy_bs, scratch, B, Ci, Co, H, W, ncib, nsplit);
    else
        hipLaunchKernelGGL(conv_wgrad_kernel<1>, grid, dim3(256), 0, lc_s(s), x, (long long)x_bs, dy,
                           (long long)dy_bs, scratch, B, Ci, Co, H, W, ncib, nsplit);
    const long long n = (long long)Co * Ci * ks * ks;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, lc_s(s),
                       scratch, n, nsplit, dw, accumulate);
    if (dbias) {
        const long long wp = ((long long)nsplit * n + 1) & ~1LL;
        double* bpart = reinterpret_cast<double*>(scratch + wp);
        hipLaunchKernelGGL(bias_grad_plane_kernel, dim3(Co, B), dim3(256), 0, lc_s(s), dy, (long long)dy_bs,
                           bpart, B, (long long)H * W);
        hipLaunchKernelGGL(bias_grad_fold_kernel, dim3((Co + 255) / 256), dim3(256), 0, lc_s(s), bpart, dbias,
                           Co, B, accumulate);
    }
    return lc_launch_status();
}

extern "C" int lc_conv2d_ring_wgrad_f16x2(const float* x, int64_t x_bs, const float* dy, int64_t dy_bs,
                                          const lc_conv_range* x_range, const lc_conv_range* dy_range,
                                          float* scratch, float* dw, float* dbias, int B, int Ci, int Co,
                                          int H, int W, int ks, int accumulate, lc_stream_t s) {
    if (!x || !dy || !scratch || !dw || !x_range || !dy_range || B <= 0 || Ci <= 0 || Co <= 0 || H <= 0 || W <= 0)
        return LC_EINVAL;
    if (ks != 1 && ks != 3) return LC_EUNSUP;
    // whole 2 x 32 tiles, 16-byte aligned rows (anything else: lc_conv2d_ring_wgrad)
    if ((H & 1) || (W & 31) || (x_bs & 3) || (dy_bs & 3) || (reinterpret_cast<uintptr_t>(x) & 15) ||
        (reinterpret_cast<uintptr_t>(dy) & 15))
        return LC_EUNSUP;
    const int ncib = (Ci + 63) / 64, ncob = (Co + 63) / 64;
    const int nsplit = wgrad_splits(B, Ci, Co, H, W);
    dim3 grid(ncob * ncib, nsplit);
    if (ks == 3)
        hipLaunchKernelGGL(conv_wgrad_h_kernel<3>, grid, dim3(256), 0, lc_s(s), x, (long long)x_bs, dy,
                           (long long)dy_bs, scratch, x_range, dy_range, B, Ci, Co, H, W, ncib, nsplit);
    else
        hipLaunchKernelGGL(conv_wgrad_h_kernel<1>, grid, dim3(256), 0, lc_s(s), x, (long long)x_bs, dy,
                           (long long)dy_bs, scratch, x_range, dy_range, B, Ci, Co, H, W, ncib, nsplit);
    const long long n = (long long)Co * Ci * ks * ks;
    hipLaunchKernelGGL(wgrad_fold_h_kernel, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, lc_s(s), scratch, n,
                       nsplit, dw, Co, Ci, ks * ks, x_range, dy_range, accumulate);
    if (dbias) {
        const long long wp = ((long long)nsplit * n + 1) & ~1LL;
        double* bpart = reinterpret_cast<double*>(scratch + wp);
        hipLaunchKernelGGL(bias_grad_plane_kernel, dim3(Co, B), dim3(256), 0, lc_s(s), dy, (long long)dy_bs,
                           bpart, B, (long long)H * W);
        hipLaunchKernelGGL(bias_grad_fold_kernel, dim3((Co + 255) / 256), dim3(256), 0, lc_s(s), bpart, dbias,
                           Co, B, accumulate);
    }
    return lc_launch_status();
}

LC_TOUCH_TU(conv_bwd, conv_wgrad_kernel<3>)
