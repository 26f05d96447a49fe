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
