// GroupNorm statistics + fused normalise / affine / AdaGN scale-shift / SiLU.
// Reference: nn.GroupNorm(8,C,1e-6) efficient_unet.py:37,77; ops.AdaGN ops.py:176-200;
// GroupNorm32 + scale-shift layout_unet_v1.py:243-245 / nn.py:17-19.  HBM-bound: stats reads the
// tensor once (float4, fp32 lane partials over <=64 values, fp64 wave/block reduction),
// apply reads once and writes once.  A group is a contiguous span of (C/G)*H*W floats in NCHW.
#include "common.h"

namespace {

// elements per statistics block: 64 KiB normally, 16 KiB when the tensor is so small that 64 KiB
// blocks would leave most CUs idle (deep levels at small batch: 128 blocks for [8,512,4,128])
__host__ __device__ inline int gn_chunk_elems(int B, int G, long long n) {
    const long long blocks = (long long)B * G * ((n + 16383) / 16384);
    return blocks >= 512 ? 16384 : 4096;
}
__host__ __device__ inline int gn_chunks(int B, int G, long long n) {
    const int ce = gn_chunk_elems(B, G, n);
    return (int)((n + ce - 1) / ce);
}

__global__ __launch_bounds__(256) void gn_stats_kernel(const float* __restrict__ x, long long x_bs,
                                                      double* __restrict__ part, int C, int G,
                                                      long long HW, int nch, int chunk_elems) {
    const int chunk = blockIdx.x, g = blockIdx.y, b = blockIdx.z;
    const int cpg = C / G;
    const long long n = (long long)cpg * HW;
    const float* p = x + b * x_bs + (long long)g * n;
    const long long lo = (long long)chunk * chunk_elems;
    const long long hi = (lo + chunk_elems < n) ? lo + chunk_elems : n;
    // shifted sums (pivot = first element of the group) -> no cancellation in E[d^2]-E[d]^2
    const float piv = p[0];
    float s = 0.f, q = 0.f;
    if ((n & 3) == 0 && ((reinterpret_cast<uintptr_t>(p) & 15) == 0)) {
        for (long long i = lo + threadIdx.x * 4; i < hi; i += 1024) {
            f32x4 v = *reinterpret_cast<const f32x4*>(p + i);
            v.x -= piv; v.y -= piv; v.z -= piv; v.w -= piv;
            s += (v.x + v.y) + (v.z + v.w);
            q += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
        }
    } else {
        for (long long i = lo + threadIdx.x; i < hi; i += 256) {
            const float v = p[i] - piv;
            s += v; q += v * v;
        }
    }
    double ds = lc_wave_sum((double)s), dq = lc_wave_sum((double)q);
    __shared__ double sh[8];
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sh[w] = ds; sh[4 + w] = dq; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double* o = part + (((long long)b * G + g) * nch + chunk) * 2;
        o[0] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
        o[1] = (sh[4] + sh[5]) + (sh[6] + sh[7]);
    }
}

// one block per (b, cpb channels of one group, slab of the H*W plane)
__global__ __launch_bounds__(256) void gn_apply_kernel(
    const float* __restrict__ x, long long x_bs, const double* __restrict__ part,
    const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ scale,
    const float* __restrict__ shift, long long ss_bs, float* __restrict__ y, long long y_bs, int C,
    int G, long long HW, int nch, float eps, int act, int cpb, float* mr_out, float* amax_out) {
    const int c_first = blockIdx.y * cpb, b = blockIdx.z;     // cpb channels of ONE group per block
    float am = 0.0f;                                          // max |y| of this thread (amax_out != NULL)
    const int cpg = C / G, g = c_first / cpg;
    const double* pp = part + ((long long)b * G + g) * nch * 2;
    double s = 0.0, q = 0.0;
    for (int i = 0; i < nch; ++i) { s += pp[2 * i]; q += pp[2 * i + 1]; }
    const double n = (double)cpg * (double)HW;
    const double dm = s / n;                                   // mean of (x - pivot)
    double var = q / n - dm * dm;
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const float mu = (float)((double)x[b * x_bs + (long long)g * cpg * HW] + dm);
    // training: the (mean, rstd) this pass normalised with, for the backward (the group's first block writes them)
    if (mr_out && blockIdx.x == 0 && c_first == g * cpg && threadIdx.x == 0) {
        mr_out[2 * (b * G + g)] = mu;
        mr_out[2 * (b * G + g) + 1] = rstd;
    }
    const long long per = (HW + gridDim.x - 1) / gridDim.x;
    const long long lo = blockIdx.x * per;
    const long long hi = lo + per < HW ? lo + per : HW;
    for (int c = c_first; c < c_first + cpb; ++c) {
        const float ga = gamma ? gamma[c] : 1.0f, be = beta ? beta[c] : 0.0f;
        const float sc = scale ? 1.0f + scale[b * ss_bs + c] : 1.0f;
        const float sh = shift ? shift[b * ss_bs + c] : 0.0f;
        const float* xp = x + b * x_bs + (long long)c * HW;
        float* yp = y + b * y_bs + (long long)c * HW;
        auto f = [&](float v) {
            float t = (v - mu) * rstd;
            t = t * ga + be;
            t = t * sc + sh;
            return act ? lc_silu(t) : t;
        };
        const bool vec = (HW & 3) == 0 && (per & 3) == 0 &&
                         ((reinterpret_cast<uintptr_t>(xp) | reinterpret_cast<uintptr_t>(yp)) & 15) == 0;
        const __amdgpu_buffer_rsrc_t rs_yp = lc_wt_buf(yp);
        if (vec) {
            for (long long i = lo + threadIdx.x * 4; i < hi; i += 1024) {
                f32x4 v = *reinterpret_cast<const f32x4*>(xp + i);
                v.x = f(v.x); v.y = f(v.y); v.z = f(v.z); v.w = f(v.w);
                lc_st4(rs_yp, (unsigned)i * 4u, v);
                am = fmaxf(fmaxf(am, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
            }
        } else {
            for (long long i = lo + threadIdx.x; i < hi; i += 256) {
                const float r = f(xp[i]);
                yp[i] = r;
                am = fmaxf(am, fabsf(r));
            }
        }
    }
    if (amax_out)
        lc_block_amax_store(am, amax_out + ((long long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x);
}

// Per-(b, channel) affine form of GroupNorm(+AdaGN): y = (x - mu) * A + Bc, stored as float4
// (mu, A, Bc, 0) rows so a consumer (the conv staging pass) can apply GN + SiLU on the fly and the
// normalised tensor never exists in HBM.
__global__ void gn_coeffs_kernel(const double* __restrict__ part, const float* __restrict__ x,
                                 long long x_bs, const float* __restrict__ gamma,
                                 const float* __restrict__ beta, const float* __restrict__ scale,
                                 const float* __restrict__ shift, long long ss_bs,
                                 f32x4* __restrict__ out, int B, int C, int Cpad, int G,
                                 long long HW, int nch, float eps) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * Cpad) return;
    const int b = i / Cpad, c = i - b * Cpad;
    f32x4 r = {0.f, 0.f, 0.f, 0.f};
    if (c < C) {
        const int cpg = C / G, g = c / cpg;
        const double* pp = part + ((long long)b * G + g) * nch * 2;
        double s = 0.0, q = 0.0;
        for (int k = 0; k < nch; ++k) { s += pp[2 * k]; q += pp[2 * k + 1]; }
        const double n = (double)cpg * (double)HW;
        const double dm = s / n;
        double var = q / n - dm * dm;
        if (var < 0.0) var = 0.0;
        const float rstd = (float)(1.0 / sqrt(var + (double)eps));
        const float mu = (float)((double)x[b * x_bs + (long long)g * cpg * HW] + dm);
        const float ga = gamma ? gamma[c] : 1.0f, be = beta ? beta[c] : 0.0f;
        const float sc = scale ? 1.0f + scale[b * ss_bs + c] : 1.0f;
        const float sh = shift ? shift[b * ss_bs + c] : 0.0f;
        r.x = mu; r.y = rstd * ga * sc; r.z = be * sc + sh;
    }
    out[i] = r;
}


// ---------------------------------------------------------------------------------------------
// GroupNorm apply fed by the PRODUCER's statistics (conv epilogue, conv_f16x2.hip): entries
// (pivot, n, S = sum(y - pivot), Q = sum((y - pivot)^2)) per (sample, octet of 8 channels, wave
// tile); a group's entries are contiguous.  Every entry is re-centred on ONE common pivot P0 (the
// group's first entry's pivot, a typical value of the tensor):
//   S' = S + n d,  Q' = Q + d (2 S + n d),  d = pivot - P0
// and fp64 sums of N, S', Q' give mean = P0 + S'/N, var = Q'/N - (S'/N)^2: no division per entry,
// no cancellation beyond |mean - P0| / std (a few), fixed summation order (deterministic).
struct OctStats2 {
    const f32x4* p0; const f32x4* p1;     // segment 0: channels [0, c0), segment 1: [c0, c0 + c1)
    int c0, slots0, c1, slots1;
    int ush0, ush1;                       // log2(channels per entry) of each segment: 3 octets, 2 quads, 1 pairs, 0 channels
};
// lc_oct_stats -> one OctStats2 segment; false: not a unit this library writes
static bool os_unit_shift(int unit, int& ush) {
    switch (unit) { case 8: ush = 3; return true; case 4: ush = 2; return true; case 2: ush = 1; return true;
                    case 1: ush = 0; return true; default: return false; }
}
// The segments of a statistics-fed GroupNorm: every group must be whole entries of ONE segment.
static int os_from_segments(const lc_oct_stats* s0, const lc_oct_stats* s1, int C, int cpg, OctStats2& os) {
    if (!s0 || !s0->p || s0->channels <= 0 || s0->slots <= 0) return LC_EINVAL;
    if (s1 && (!s1->p || s1->channels <= 0 || s1->slots <= 0)) return LC_EINVAL;
    os.p0 = reinterpret_cast<const f32x4*>(s0->p); os.c0 = s0->channels; os.slots0 = s0->slots;
    os.p1 = nullptr; os.c1 = 0; os.slots1 = 0; os.ush0 = os.ush1 = 3;
    if (!os_unit_shift(s0->unit, os.ush0)) return LC_EUNSUP;
    if (s1) {
        os.p1 = reinterpret_cast<const f32x4*>(s1->p); os.c1 = s1->channels; os.slots1 = s1->slots;
        if (!os_unit_shift(s1->unit, os.ush1)) return LC_EUNSUP;
    }
    if (os.c0 + os.c1 != C || os.c0 % cpg || cpg % s0->unit || os.c0 % s0->unit ||
        (s1 && (cpg % s1->unit || os.c1 % s1->unit)))
        return LC_EUNSUP;
    return LC_OK;
}

__global__ __launch_bounds__(256) void gn_apply_os_kernel(
    const float* __restrict__ x, long long x_bs, OctStats2 os, const float* __restrict__ gamma,
    const float* __restrict__ beta, const float* __restrict__ scale, const float* __restrict__ shift,
    long long ss_bs, float* __restrict__ y, long long y_bs, int C, int G, long long HW, float eps,
    int act, int cpb) {
    __shared__ double sh[12];
    const int c_first = blockIdx.y * cpb, b = blockIdx.z;
    const int cpg = C / G, g = c_first / cpg;
    const int cg0 = g * cpg;                                   // first channel of the group
    const bool seg1 = cg0 >= os.c0;
    const int slots = seg1 ? os.slots1 : os.slots0;
    const int ush = seg1 ? os.ush1 : os.ush0;
    const f32x4* e = seg1 ? os.p1 + ((long long)b * (os.c1 >> ush) + ((cg0 - os.c0) >> ush)) * slots
                          : os.p0 + ((long long)b * (os.c0 >> ush) + (cg0 >> ush)) * slots;
    const int n_ent = (cpg >> ush) * slots;
    const double P0 = (double)e[0].x;
    double N = 0.0, S = 0.0, Q = 0.0;
    for (int base = threadIdx.x; base < n_ent; base += 256 * 4) {   // 4 loads in flight per thread
        f32x4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            v[k] = base + 256 * k < n_ent ? e[base + 256 * k] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 4; ++k) {                               // an absent entry (n = 0) adds nothing
            const double n = v[k].y, d = (double)v[k].x - P0, s_ = v[k].z;
            N += n;
            S += s_ + n * d;
            Q += (double)v[k].w + d * (2.0 * s_ + n * d);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        N += __shfl_xor(N, o, 64); S += __shfl_xor(S, o, 64); Q += __shfl_xor(Q, o, 64);
    }
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sh[3 * wv] = N; sh[3 * wv + 1] = S; sh[3 * wv + 2] = Q; }
    __syncthreads();
    N = (sh[0] + sh[3]) + (sh[6] + sh[9]);
    S = (sh[1] + sh[4]) + (sh[7] + sh[10]);
    Q = (sh[2] + sh[5]) + (sh[8] + sh[11]);
    const double m = N > 0.0 ? S / N : 0.0;
    double var = N > 0.0 ? Q / N - m * m : 0.0;
    if (var < 0.0) var = 0.0;
    const float mu = (float)(P0 + m);
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const long long per = (HW + gridDim.x - 1) / gridDim.x;
    const long long lo = blockIdx.x * per;
    const long long hi = lo + per < HW ? lo + per : HW;
    for (int c = c_first; c < c_first + cpb; ++c) {
        const float ga = gamma ? gamma[c] : 1.0f, be = beta ? beta[c] : 0.0f;
        const float sc = scale ? 1.0f + scale[b * ss_bs + c] : 1.0f;
        const float sf = shift ? shift[b * ss_bs + c] : 0.0f;
        const float* xp = x + b * x_bs + (long long)c * HW;
        float* yp = y + b * y_bs + (long long)c * HW;
        auto f = [&](float v) {
            float t = (v - mu) * rstd;
            t = t * ga + be;
            t = t * sc + sf;
            return act ? lc_silu(t) : t;
        };
        const bool vec = (HW & 3) == 0 && (per & 3) == 0 &&
                         ((reinterpret_cast<uintptr_t>(xp) | reinterpret_cast<uintptr_t>(yp)) & 15) == 0;
        const __amdgpu_buffer_rsrc_t rs_yp = lc_wt_buf(yp);
        if (vec) {
            for (long long i = lo + threadIdx.x * 4; i < hi; i += 1024) {
                f32x4 v = *reinterpret_cast<const f32x4*>(xp + i);
                v.x = f(v.x); v.y = f(v.y); v.z = f(v.z); v.w = f(v.w);
                lc_st4(rs_yp, (unsigned)i * 4u, v);
            }
        } else {
            for (long long i = lo + threadIdx.x; i < hi; i += 256) lc_st(yp + i, f(xp[i]));
        }
    }
}

// ---------------------------------------------------------------------------------------------
// GroupNorm apply with PRE-SPLIT output for conv_f16x2_ps_kernel: the same normalisation
// (two-pass partials, OS = false, or the producer's octet statistics, OS = true), but every thread
// owns ONE pixel of ONE channel octet: 8 channel loads (coalesced across the wave along W), the
// affine / AdaGN / SiLU arithmetic of gn_apply_kernel, multiplication by the consumer layer's
// x_scale, the fp16 hi/lo split of conv_f16x2.hip (split rule restated: hi = 11 significant bits
// truncated, packed toward zero; lo = fp16(s - hi)) and two 16-byte stores into
// ysp[b][plane][c/8][h][w][8].  max |y * x_scale| goes to the consumer's lc_conv_range.  The split
// therefore runs ONCE per element in a memory-bound kernel instead of once per output-channel
// block inside the conv's K loop.  Needs C % 16 == 0 and whole octets per group.
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef float float2_t __attribute__((ext_vector_type(2)));

// ---------------------------------------------------------------------------------------------
// BACKWARD of GroupNorm (+affine) (+AdaGN scale/shift) (+SiLU)  -- training, SURVEY.md section 8f-4.
//   forward:  xh = (x - mu) rstd;  t1 = xh g + be;  t2 = t1 (1 + sc) + sf;  y = silu?(t2)
//   backward: dt2 = dy silu'(t2);  r1[b,c] = sum_hw dt2,  r3[b,c] = sum_hw dt2 xh      (rows kernel)
//             dxh = g (1 + sc) dt2;   dx = rstd (dxh - mean_g(dxh) - xh mean_g(dxh xh))   (apply kernel)
//   the parameter gradients are small contractions of the rows (host side of the C ABI's caller):
//     dsf = r1, dsc = g r3 + be r1, dbe = sum_b (1+sc) r1, dg = sum_b (1+sc) r3.
__global__ void gn_meanrstd_kernel(const double* __restrict__ part, const float* __restrict__ x,
                                   long long x_bs, float* __restrict__ out, int B, int C, int G,
                                   long long HW, int nch, float eps) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * G) return;
    const int b = i / G, g = i - b * G, cpg = C / G;
    const double* pp = part + (long long)i * nch * 2;
    double s = 0.0, q = 0.0;
    for (int k = 0; k < nch; ++k) { s += pp[2 * k]; q += pp[2 * k + 1]; }
    const double n = (double)cpg * (double)HW;
    const double dm = s / n;
    double var = q / n - dm * dm;
    if (var < 0.0) var = 0.0;
    out[2 * i] = (float)((double)x[b * x_bs + (long long)g * cpg * HW] + dm);
    out[2 * i + 1] = (float)(1.0 / sqrt(var + (double)eps));
}

__device__ __forceinline__ float silu_grad(float t) {
    const float sg = 1.0f / (1.0f + __expf(-t));
    return sg * (1.0f + t * (1.0f - sg));
}

__global__ __launch_bounds__(256) void gn_bwd_rows_kernel(
    const float* __restrict__ x, long long x_bs, const float* __restrict__ dy, long long dy_bs,
    const float* __restrict__ mr, const float* __restrict__ gamma, const float* __restrict__ beta,
    const float* __restrict__ scale, const float* __restrict__ shift, long long ss_bs,
    double* __restrict__ rows, int C, int G, long long HW, int act) {
    const int c = blockIdx.x, b = blockIdx.y, g = c / (C / G);
    const float mu = mr[2 * (b * G + g)], rstd = mr[2 * (b * G + g) + 1];
    const float ga = gamma ? gamma[c] : 1.0f, be = beta ? beta[c] : 0.0f;
    const float sc = scale ? 1.0f + scale[b * ss_bs + c] : 1.0f;
    const float sf = shift ? shift[b * ss_bs + c] : 0.0f;
    const float* xp = x + b * x_bs + (long long)c * HW;
    const float* dp = dy + b * dy_bs + (long long)c * HW;
    double r1 = 0.0, r3 = 0.0;
    for (long long i0 = threadIdx.x; i0 < HW; i0 += 256 * 16) {
        float a1 = 0.f, a3 = 0.f;                              // fp32 over <= 16 values, then fp64
        for (long long i = i0; i < HW && i < i0 + 256 * 16; i += 256) {
            const float xh = (xp[i] - mu) * rstd;
            float d = dp[i];
            if (act) d *= silu_grad((xh * ga + be) * sc + sf);
            a1 += d; a3 += d * xh;
        }
        r1 += (double)a1; r3 += (double)a3;
    }
    r1 = lc_wave_sum(r1); r3 = lc_wave_sum(r3);
    __shared__ double sh[8];
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sh[w] = r1; sh[4 + w] = r3; }
    __syncthreads();
    if (threadIdx.x == 0) {
        rows[2 * ((long long)b * C + c)] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
        rows[2 * ((long long)b * C + c) + 1] = (sh[4] + sh[5]) + (sh[6] + sh[7]);
    }
}

__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(
    const float* __restrict__ x, long long x_bs, const float* __restrict__ dy, long long dy_bs,
    const float* __restrict__ mr, const float* __restrict__ gamma, const float* __restrict__ beta,
    const float* __restrict__ scale, const float* __restrict__ shift, long long ss_bs,
    const double* __restrict__ rows, float* __restrict__ dx, long long dx_bs, int C, int G,
    long long HW, int act, float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ dscale,
    float* __restrict__ dshift, int B, float* amax_out) {
    const int c = blockIdx.y, b = blockIdx.z, cpg = C / G, g = c / cpg;
    // The small parameter gradients from the rows (fp64, samples in ascending order: deterministic), by the first slab's
    // thread 0:  dshift[b,c] = r1,  dscale[b,c] = g r3 + be r1;  the b = 0 block: dbeta[c] = sum_b (1 + sc) r1,
    // dgamma[c] = sum_b (1 + sc) r3.
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const double gd = gamma ? (double)gamma[c] : 1.0, bd = beta ? (double)beta[c] : 0.0;
        if (dscale || dshift) {
            const double r1 = rows[2 * ((long long)b * C + c)], r3 = rows[2 * ((long long)b * C + c) + 1];
            if (dscale) dscale[(long long)b * C + c] = (float)(gd * r3 + bd * r1);
            if (dshift) dshift[(long long)b * C + c] = (float)r1;
        }
        if (b == 0 && (dgamma || dbeta)) {
            double ag = 0.0, ab = 0.0;
            for (int bb = 0; bb < B; ++bb) {
                const double one_sc = scale ? 1.0 + (double)scale[bb * ss_bs + c] : 1.0;
                ab += one_sc * rows[2 * ((long long)bb * C + c)];
                ag += one_sc * rows[2 * ((long long)bb * C + c) + 1];
            }
            if (dgamma) dgamma[c] = (float)ag;
            if (dbeta) dbeta[c] = (float)ab;
        }
    }
    const float mu = mr[2 * (b * G + g)], rstd = mr[2 * (b * G + g) + 1];
    double m1 = 0.0, m2 = 0.0;                                  // group means of dxh and dxh * xh
    for (int k = 0; k < cpg; ++k) {
        const int cc = g * cpg + k;
        const double w = (double)(gamma ? gamma[cc] : 1.0f) *
                         (double)(scale ? 1.0f + scale[b * ss_bs + cc] : 1.0f);
        m1 += w * rows[2 * ((long long)b * C + cc)];
        m2 += w * rows[2 * ((long long)b * C + cc) + 1];
    }
    const double n = (double)cpg * (double)HW;
    const float fm1 = (float)(m1 / n), fm2 = (float)(m2 / n);
    const float ga = gamma ? gamma[c] : 1.0f, be = beta ? beta[c] : 0.0f;
    const float sc = scale ? 1.0f + scale[b * ss_bs + c] : 1.0f;
    const float sf = shift ? shift[b * ss_bs + c] : 0.0f;
    const float* xp = x + b * x_bs + (long long)c * HW;
    const float* dp = dy + b * dy_bs + (long long)c * HW;
    float* op = dx + b * dx_bs + (long long)c * HW;
    const long long per = (HW + gridDim.x - 1) / gridDim.x;
    const long long lo = blockIdx.x * per, hi = lo + per < HW ? lo + per : HW;
    float am = 0.0f;
    const auto g1 = [&](float xv, float d) {
        const float xh = (xv - mu) * rstd;
        if (act) d *= silu_grad((xh * ga + be) * sc + sf);
        const float r = rstd * (ga * sc * d - fm1 - xh * fm2);
        am = fmaxf(am, fabsf(r));
        return r;
    };
    if ((HW & 3) == 0 && (per & 3) == 0 && ((reinterpret_cast<uintptr_t>(xp) | reinterpret_cast<uintptr_t>(dp) |
                                             reinterpret_cast<uintptr_t>(op)) & 15) == 0) {
        for (long long i = lo + threadIdx.x * 4; i < hi; i += 1024) {
            const f32x4 xv = *reinterpret_cast<const f32x4*>(xp + i), dv = *reinterpret_cast<const f32x4*>(dp + i);
            f32x4 r;
            r.x = g1(xv.x, dv.x); r.y = g1(xv.y, dv.y); r.z = g1(xv.z, dv.z); r.w = g1(xv.w, dv.w);
            *reinterpret_cast<f32x4*>(op + i) = r;
        }
    } else {
        for (long long i = lo + threadIdx.x; i < hi; i += 256) op[i] = g1(xp[i], dp[i]);
    }
    if (amax_out)
        lc_block_amax_store(am, amax_out + ((long long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x);
}

// OS: 0 = statistics-pass partials; 1 = producer entries, groups of whole channel octets (the block folds ONE group);
// 2 = producer entries, 2 or 4 channels per group (GroupNorm32 at 64 / 128 channels, round 5): the octet holds 4 / 2
// groups, wave w of the block folds group w.
// Store form of the hi / lo planes: 0 = plain (write-back) stores -- shipped; 2 = write-through (sc1) buffer stores.  Round 5
// measured 2 and four more write-through forms (inline assembly with v_nop / s_nop wait states, four pixels then eight stores
// back to back, nontemporal): every correct one costs the pass +9 us at 8 x 128 x 16 x 512 and the C2 step 3-5 %
// (profiles/r05_level0.txt section 8).
#ifndef LC_GNS_STORE
#define LC_GNS_STORE 0
#endif
template <int OS>
__global__ __launch_bounds__(256) void gn_apply_split_kernel(
    const float* __restrict__ x, long long x_bs, const double* __restrict__ part, OctStats2 os,
    const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ scale,
    const float* __restrict__ shift, long long ss_bs, half8_t* __restrict__ ysp, long long ysp_bs,
    int C, int G, long long HW, int nch, float eps, int act, lc_conv_range* range) {
    __shared__ double sh[12];
    const int oct = blockIdx.y, b = blockIdx.z;
    const int c0 = oct * 8, cpg = C / G, g = c0 / cpg;
    // The first 8 x 16-byte loads of this thread go out BEFORE the statistics fold (round 5): they do not depend on
    // it, and the fold (entry loads, fp64 shuffles, one barrier: ~1.5 us) otherwise sits in front of every block's
    // first HBM access -- 30 launches of this kernel per C2 step.
    const float* xp = x + b * x_bs + (long long)c0 * HW;
    const long long per = (HW + gridDim.x - 1) / gridDim.x;
    const long long lo = blockIdx.x * per;
    const long long hi = lo + per < HW ? lo + per : HW;
    const bool vec = (HW & 3) == 0 && (per & 3) == 0 && (reinterpret_cast<uintptr_t>(xp) & 15) == 0;
    f32x4 c4_first[8];
    auto issue_first = [&]() {      // (behind the fold's own entry loads: VMEM returns in order, the fold must not wait for x)
        asm volatile("" ::: "memory");
        // unconditional loads (a predicated load is an exec-mask branch, and hipcc's waits across branches are vmcnt(0)):
        // a thread without a first quad re-reads the slab's last quad and never uses it (C % 16 == 0: 16 channels x HW
        // floats behind x always hold 16 bytes)
        // Only the vector path consumes them.  Off it (HW % 4 != 0 or a slab / pointer that is not 16-byte aligned) a typed
        // 16-byte load from xp would be misaligned: every thread then reads the 16-byte aligned word at or below x instead
        // (inside x's allocation: allocation bases are 16-byte aligned) -- still no branch, and nothing is read for nothing
        // beyond that one cached line.
        long long p0 = lo + threadIdx.x * 4;
        p0 = p0 < hi - 4 ? p0 : hi - 4;
        const float* x_al = reinterpret_cast<const float*>(reinterpret_cast<uintptr_t>(x) & ~static_cast<uintptr_t>(15));
        const float* src = vec ? xp + (p0 > 0 ? p0 : 0) : x_al;
        const long long cs = vec ? HW : 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) c4_first[k] = *reinterpret_cast<const f32x4*>(src + (long long)k * cs);
        asm volatile("" ::: "memory");
    };
    // per-channel (mean, rstd): one group per octet when the groups are whole octets; otherwise
    // (GroupNorm32 at widths 32 ... 128: 1 / 2 / 4 channels per group; concatenated inputs of 192 /
    // 384 channels: 6 / 12) the octet spans several groups (partials route only)
    float mu[8], rstd[8];
    if constexpr (!OS) {
        issue_first();
        const int ng = (c0 + 7) / cpg - g + 1;
        for (int j = 0; j < ng; ++j) {
            const int gj = g + j;
            const double* pp = part + ((long long)b * G + gj) * nch * 2;
            double s = 0.0, q = 0.0;
            for (int i = 0; i < nch; ++i) { s += pp[2 * i]; q += pp[2 * i + 1]; }
            const double n = (double)cpg * (double)HW;
            const double dm = s / n;
            double var = q / n - dm * dm;
            if (var < 0.0) var = 0.0;
            const float r_ = (float)(1.0 / sqrt(var + (double)eps));
            const float m_ = (float)((double)x[b * x_bs + (long long)gj * cpg * HW] + dm);
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if ((c0 + k) / cpg == gj) { mu[k] = m_; rstd[k] = r_; }
        }
    } else if constexpr (OS == 2) {
        __shared__ float shm[8];
        const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
        const int ng = 8 / cpg;                                    // 4 or 2 groups in this octet
        const int cg0 = (g + (wv < ng ? wv : ng - 1)) * cpg;       // (waves past the last group redo it: no divergence)
        const bool seg1 = cg0 >= os.c0;
        const int slots = seg1 ? os.slots1 : os.slots0;
        const int ush = seg1 ? os.ush1 : os.ush0;
        const f32x4* e = seg1 ? os.p1 + ((long long)b * (os.c1 >> ush) + ((cg0 - os.c0) >> ush)) * slots
                              : os.p0 + ((long long)b * (os.c0 >> ush) + (cg0 >> ush)) * slots;
        const int n_ent = (cpg >> ush) * slots;
        const float P0f = e[0].x;
        double N = 0.0, S = 0.0, Q = 0.0;
        f32x4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = lane + 64 * k < n_ent ? e[lane + 64 * k] : f32x4{0.f, 0.f, 0.f, 0.f};
        issue_first();
        const double P0 = (double)P0f;
        auto fold4 = [&](const f32x4 (&w)[4]) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const double n = w[k].y, d = (double)w[k].x - P0, s_ = w[k].z;
                N += n;
                S += s_ + n * d;
                Q += (double)w[k].w + d * (2.0 * s_ + n * d);
            }
        };
        fold4(v);
        for (int base = lane + 64 * 4; base < n_ent; base += 64 * 4) {
            f32x4 w[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) w[k] = base + 64 * k < n_ent ? e[base + 64 * k] : f32x4{0.f, 0.f, 0.f, 0.f};
            fold4(w);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            N += __shfl_xor(N, o, 64); S += __shfl_xor(S, o, 64); Q += __shfl_xor(Q, o, 64);
        }
        if (lane == 0) {
            const double m = N > 0.0 ? S / N : 0.0;
            double var = N > 0.0 ? Q / N - m * m : 0.0;
            if (var < 0.0) var = 0.0;
            shm[2 * wv] = (float)(P0 + m);
            shm[2 * wv + 1] = (float)(1.0 / sqrt(var + (double)eps));
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 8; ++k) { const int j = k >> (cpg == 4 ? 2 : 1); mu[k] = shm[2 * j]; rstd[k] = shm[2 * j + 1]; }
    } else {
        const int cg0 = g * cpg;
        const bool seg1 = cg0 >= os.c0;
        const int slots = seg1 ? os.slots1 : os.slots0;
        const int ush = seg1 ? os.ush1 : os.ush0;
        const f32x4* e = seg1 ? os.p1 + ((long long)b * (os.c1 >> ush) + ((cg0 - os.c0) >> ush)) * slots
                              : os.p0 + ((long long)b * (os.c0 >> ush) + (cg0 >> ush)) * slots;
        const int n_ent = (cpg >> ush) * slots;
        const float P0f = e[0].x;
        double N = 0.0, S = 0.0, Q = 0.0;
        f32x4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            v[k] = (int)threadIdx.x + 256 * k < n_ent ? e[threadIdx.x + 256 * k] : f32x4{0.f, 0.f, 0.f, 0.f};
        issue_first();                                            // the x loads ride behind the first batch of entries
        const double P0 = (double)P0f;
        auto fold4 = [&](const f32x4 (&w)[4]) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {                         // an absent entry (n = 0) adds nothing
                const double n = w[k].y, d = (double)w[k].x - P0, s_ = w[k].z;
                N += n;
                S += s_ + n * d;
                Q += (double)w[k].w + d * (2.0 * s_ + n * d);
            }
        };
        fold4(v);                                                 // straight-line: waits for the entries only (vmcnt(8))
        for (int base = threadIdx.x + 256 * 4; base < n_ent; base += 256 * 4) {   // > 1024 entries per group: 64 x 2048 images
            f32x4 w[4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                w[k] = base + 256 * k < n_ent ? e[base + 256 * k] : f32x4{0.f, 0.f, 0.f, 0.f};
            fold4(w);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            N += __shfl_xor(N, o, 64); S += __shfl_xor(S, o, 64); Q += __shfl_xor(Q, o, 64);
        }
        const int wv = threadIdx.x >> 6;
        if ((threadIdx.x & 63) == 0) { sh[3 * wv] = N; sh[3 * wv + 1] = S; sh[3 * wv + 2] = Q; }
        __syncthreads();
        N = (sh[0] + sh[3]) + (sh[6] + sh[9]);
        S = (sh[1] + sh[4]) + (sh[7] + sh[10]);
        Q = (sh[2] + sh[5]) + (sh[8] + sh[11]);
        const double m = N > 0.0 ? S / N : 0.0;
        double var = N > 0.0 ? Q / N - m * m : 0.0;
        if (var < 0.0) var = 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k) { mu[k] = (float)(P0 + m); rstd[k] = (float)(1.0 / sqrt(var + (double)eps)); }
    }
    // per channel: y = (x - mean) * A + Bc with A = rstd * gamma * (1 + scale), Bc = beta * (1 + scale) + shift -- the
    // form (and rounding order) of the convolutions' fused input norm (conv_f16x2_common.h gn_act)
    float cA[8], cB[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int c = c0 + k;
        const float ga = gamma ? gamma[c] : 1.0f, be = beta ? beta[c] : 0.0f;
        const float sc = scale ? 1.0f + scale[b * ss_bs + c] : 1.0f;
        const float sf = shift ? shift[b * ss_bs + c] : 0.0f;
        cA[k] = rstd[k] * ga * sc;
        cB[k] = be * sc + sf;
    }
    const float xs = range->x_scale;
    const float seen = range->amax_scaled;
    float am = 0.0f;
    const int C8 = C >> 3;
    half8_t* yh = ysp + b * ysp_bs + (long long)oct * HW;
    half8_t* yl = yh + (long long)C8 * HW;
    auto one = [&](const float (&v)[8], half8_t& h8, half8_t& l8) {
#pragma unroll
        for (int k = 0; k < 8; k += 2) {
            float t0 = fmaf(v[k] - mu[k], cA[k], cB[k]), t1 = fmaf(v[k + 1] - mu[k + 1], cA[k + 1], cB[k + 1]);
            if (act) { t0 = lc_silu(t0); t1 = lc_silu(t1); }
            const float s0 = t0 * xs, s1 = t1 * xs;
            am = fmaxf(am, fmaxf(fabsf(s0), fabsf(s1)));
            const float h0 = __uint_as_float(__float_as_uint(s0) & 0xFFFFE000u);
            const float h1 = __uint_as_float(__float_as_uint(s1) & 0xFFFFE000u);
            const half2_t ph = __builtin_bit_cast(half2_t, __builtin_amdgcn_cvt_pkrtz(h0, h1));
            float2_t r; r.x = s0 - h0; r.y = s1 - h1;
            const half2_t pl = __builtin_convertvector(r, half2_t);
            h8[k] = ph.x; h8[k + 1] = ph.y; l8[k] = pl.x; l8[k + 1] = pl.y;
        }
    };
    if (vec) {
        // four consecutive pixels per thread: 8 float4 channel loads, 2 x 4 contiguous 16-byte stores; the first quad is
        // the one issued ahead of the fold (peeled: a select per load inside the loop compiled to 8 branches)
        const __amdgpu_buffer_rsrc_t rs_yh = lc_wt_buf(yh);
        const unsigned lo_off = (unsigned)((long long)C8 * HW * 16);          // the lo plane behind the hi plane (< 4 GiB: host check)
        (void)rs_yh; (void)lo_off;
        auto quad = [&](const f32x4 (&c4)[8], long long p) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = c4[k][q];
                half8_t h8, l8;
                one(v, h8, l8);
#if LC_GNS_STORE == 2      // write-through (sc1) buffer stores: developer A/B (profiles/r05_level0.txt section 8)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(lc_u32x4, h8), rs_yh, (unsigned)(p + q) * 16u, 0, 16);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(lc_u32x4, l8), rs_yh,
                                                       (unsigned)(p + q) * 16u + lo_off, 0, 16);
#else                      // write-back: shipped
                yh[p + q] = h8;
                yl[p + q] = l8;
#endif
            }
        };
        long long p = lo + threadIdx.x * 4;
        if (p < hi) quad(c4_first, p);
        for (p += 1024; p < hi; p += 1024) {
            f32x4 c4[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) c4[k] = *reinterpret_cast<const f32x4*>(xp + (long long)k * HW + p);
            quad(c4, p);
        }
    } else {
        for (long long p = lo + threadIdx.x; p < hi; p += 256) {
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = xp[(long long)k * HW + p];
            half8_t h8, l8;
            one(v, h8, l8);
            yh[p] = h8;
            yl[p] = l8;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) am = fmaxf(am, __shfl_xor(am, o, 64));
    if ((threadIdx.x & 63) == 0 && am > seen)
        atomicMax(reinterpret_cast<unsigned*>(&range->amax_scaled), __float_as_uint(am));
}

}  // namespace

extern "C" int lc_groupnorm_coeffs(const float* x, int64_t x_bs, const double* partials,
                                   const float* gamma, const float* beta, const float* scale,
                                   const float* shift, int64_t ss_bs, float* coeffs, int B, int C,
                                   int Cpad, int H, int W, int G, float eps, lc_stream_t s) {
    if (!x || !partials || !coeffs || B <= 0 || G <= 0 || C % G || Cpad < C) return LC_EINVAL;
    const long long HW = (long long)H * W;
    const int nch = gn_chunks(B, G, (long long)(C / G) * HW);
    const int n = B * Cpad;
    hipLaunchKernelGGL(gn_coeffs_kernel, dim3((n + 255) / 256), dim3(256), 0, lc_s(s), partials, x,
                       (long long)x_bs, gamma, beta, scale, shift, (long long)ss_bs,
                       reinterpret_cast<f32x4*>(coeffs), B, C, Cpad, G, HW, nch, eps);
    return lc_launch_status();
}

extern "C" int64_t lc_groupnorm_partials_elems(int B, int C, int H, int W, int G) {
    if (G <= 0 || C % G) return 0;
    return (int64_t)B * G * gn_chunks(B, G, (long long)(C / G) * H * W) * 2;
}

extern "C" int lc_groupnorm_stats(const float* x, int64_t x_bs, double* partials, int B, int C,
                                  int H, int W, int G, lc_stream_t s) {
    if (!x || !partials || B <= 0 || G <= 0 || C % G) return LC_EINVAL;
    const long long HW = (long long)H * W;
    const int nch = gn_chunks(B, G, (long long)(C / G) * HW);
    hipLaunchKernelGGL(gn_stats_kernel, dim3(nch, G, B), dim3(256), 0, lc_s(s), x, (long long)x_bs,
                       partials, C, G, HW, nch, gn_chunk_elems(B, G, (long long)(C / G) * HW));
    return lc_launch_status();
}

namespace {
// launch shape of gn_apply_kernel: slabs of >= 4096 elements; small planes: several channels of a group per block, so the
// per-block fold of the partials (fp64 divide + sqrt) is paid once per >= 4096 elements while >= 512 blocks remain
inline void gn_apply_grid(int B, int C, int G, long long HW, int* slabs, int* cpb) {
    int sl = (int)((HW + 4095) / 4096);
    if (sl < 1) sl = 1;
    int c = 1;
    const int cpg = C / G;
    while (c * 2 <= cpg && cpg % (c * 2) == 0 && HW * c * 2 <= 4096 && (long long)B * (C / (c * 2)) * sl >= 512) c *= 2;
    *slabs = sl; *cpb = c;
}
}  // namespace

// floats lc_groupnorm_apply_train (backward == 0) / lc_groupnorm_bwd_train (backward != 0) write through amax_out: one
// partial maximum per block of their apply pass
extern "C" int64_t lc_groupnorm_amax_partials(int B, int C, int H, int W, int G, int backward) {
    if (B <= 0 || C <= 0 || G <= 0 || C % G || H <= 0 || W <= 0) return 0;
    int slabs, cpb;
    gn_apply_grid(B, C, G, (long long)H * W, &slabs, &cpb);
    return (int64_t)slabs * (backward ? C : C / cpb) * B;
}

extern "C" int lc_groupnorm_apply_train(const float* x, int64_t x_bs, const double* partials,
                                        const float* gamma, const float* beta, const float* scale,
                                        const float* shift, int64_t ss_bs, float* y, int64_t y_bs, int B,
                                        int C, int H, int W, int G, float eps, int act_silu,
                                        float* mean_rstd_out, float* amax_out, lc_stream_t s) {
    if (!x || !y || !partials || B <= 0 || G <= 0 || C % G) return LC_EINVAL;
    const long long HW = (long long)H * W;
    const int nch = gn_chunks(B, G, (long long)(C / G) * HW);
    int slabs, cpb;
    gn_apply_grid(B, C, G, HW, &slabs, &cpb);
    hipLaunchKernelGGL(gn_apply_kernel, dim3(slabs, C / cpb, B), dim3(256), 0, lc_s(s), x,
                       (long long)x_bs, partials, gamma, beta, scale, shift, (long long)ss_bs, y,
                       (long long)y_bs, C, G, HW, nch, eps, act_silu, cpb, mean_rstd_out, amax_out);
    return lc_launch_status();
}

extern "C" int lc_groupnorm_apply(const float* x, int64_t x_bs, const double* partials,
                                  const float* gamma, const float* beta, const float* scale,
                                  const float* shift, int64_t ss_bs, float* y, int64_t y_bs, int B,
                                  int C, int H, int W, int G, float eps, int act_silu,
                                  lc_stream_t s) {
    return lc_groupnorm_apply_train(x, x_bs, partials, gamma, beta, scale, shift, ss_bs, y, y_bs, B, C, H, W, G, eps,
                                    act_silu, nullptr, nullptr, s);
}


extern "C" int lc_groupnorm_apply_os(const float* x, int64_t x_bs, const lc_oct_stats* s0,
                                     const lc_oct_stats* s1, const float* gamma, const float* beta,
                                     const float* scale, const float* shift, int64_t ss_bs, float* y,
                                     int64_t y_bs, int B, int C, int H, int W, int G, float eps,
                                     int act_silu, lc_stream_t s) {
    if (!x || !y || B <= 0 || G <= 0 || C % G) return LC_EINVAL;
    const int cpg = C / G;
    OctStats2 os;
    // groups are whole entries (octets, quads, pairs or single channels: round 5) and lie inside one segment
    if (const int rc = os_from_segments(s0, s1, C, cpg, os)) return rc;
    const long long HW = (long long)H * W;
    int slabs = (int)((HW + 4095) / 4096);
    if (slabs < 1) slabs = 1;
    // every block folds its group's entries (hundreds of 16-byte loads + a reduction) before it
    // streams: give it as long a slab as possible while >= 2048 blocks remain
    while (slabs % 2 == 0 && (long long)B * C * (slabs / 2) >= 2048) slabs /= 2;
    int cpb = 1;
    while (cpb * 2 <= cpg && cpg % (cpb * 2) == 0 && HW * cpb * 2 <= 4096 &&
           (long long)B * (C / (cpb * 2)) * slabs >= 512)
        cpb *= 2;
    hipLaunchKernelGGL(gn_apply_os_kernel, dim3(slabs, C / cpb, B), dim3(256), 0, lc_s(s), x,
                       (long long)x_bs, os, gamma, beta, scale, shift, (long long)ss_bs, y,
                       (long long)y_bs, C, G, HW, eps, act_silu, cpb);
    return lc_launch_status();
}

// Per-(sample, channel) rows (mu, A, Bc, 0) from the PRODUCER's statistics entries -- what gn_coeffs_kernel derives from the
// partials of a statistics pass: y = (x - mu) * A + Bc.  One block per (group, sample): the fold of gn_apply_os_kernel, then
// one row per channel of the group.  Consumer (round 6): the paired resampler of a resampling ResBlock (resample.hip).
__global__ __launch_bounds__(256) void gn_coeffs_os_kernel(OctStats2 os, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, long long ss_bs,
                                                          f32x4* __restrict__ out, int C, int Cpad, int G, float eps) {
    __shared__ double sh[12];
    const int g = blockIdx.x, b = blockIdx.y;
    const int cpg = C / G, cg0 = g * cpg;
    const bool seg1 = cg0 >= os.c0;
    const int slots = seg1 ? os.slots1 : os.slots0;
    const int ush = seg1 ? os.ush1 : os.ush0;
    const f32x4* e = seg1 ? os.p1 + ((long long)b * (os.c1 >> ush) + ((cg0 - os.c0) >> ush)) * slots
                          : os.p0 + ((long long)b * (os.c0 >> ush) + (cg0 >> ush)) * slots;
    const int n_ent = (cpg >> ush) * slots;
    const double P0 = (double)e[0].x;
    double N = 0.0, S = 0.0, Q = 0.0;
    for (int base = threadIdx.x; base < n_ent; base += 256 * 4) {
        f32x4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            v[k] = base + 256 * k < n_ent ? e[base + 256 * k] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const double n = v[k].y, d = (double)v[k].x - P0, s_ = v[k].z;
            N += n;
            S += s_ + n * d;
            Q += (double)v[k].w + d * (2.0 * s_ + n * d);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        N += __shfl_xor(N, o, 64); S += __shfl_xor(S, o, 64); Q += __shfl_xor(Q, o, 64);
    }
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sh[3 * wv] = N; sh[3 * wv + 1] = S; sh[3 * wv + 2] = Q; }
    __syncthreads();
    N = (sh[0] + sh[3]) + (sh[6] + sh[9]);
    S = (sh[1] + sh[4]) + (sh[7] + sh[10]);
    Q = (sh[2] + sh[5]) + (sh[8] + sh[11]);
    const double m = N > 0.0 ? S / N : 0.0;
    double var = N > 0.0 ? Q / N - m * m : 0.0;
    if (var < 0.0) var = 0.0;
    const float mu = (float)(P0 + m);
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    for (int k = threadIdx.x; k < cpg; k += 256) {
        const int c = cg0 + k;
        const float ga = gamma ? gamma[c] : 1.0f, be = beta ? beta[c] : 0.0f;
        const float sc = scale ? 1.0f + scale[b * ss_bs + c] : 1.0f;
        const float sf = shift ? shift[b * ss_bs + c] : 0.0f;
        out[(long long)b * Cpad + c] = f32x4{mu, rstd * ga * sc, be * sc + sf, 0.f};
    }
}

extern "C" int lc_groupnorm_coeffs_os(const lc_oct_stats* s0, const lc_oct_stats* s1, const float* gamma, const float* beta,
                                      const float* scale, const float* shift, int64_t ss_bs, float* coeffs, int B, int C,
                                      int Cpad, int G, float eps, lc_stream_t s) {
    if (!coeffs || B <= 0 || G <= 0 || C % G || Cpad < C) return LC_EINVAL;
    OctStats2 os;
    if (const int rc = os_from_segments(s0, s1, C, C / G, os)) return rc;
    hipLaunchKernelGGL(gn_coeffs_os_kernel, dim3(G, B), dim3(256), 0, lc_s(s), os, gamma, beta, scale, shift,
                       (long long)ss_bs, reinterpret_cast<f32x4*>(coeffs), C, Cpad, G, eps);
    return lc_launch_status();
}

// ---- pre-split output (see gn_apply_split_kernel) -------------------------------------------------
// (round 5 sweep of the pixels per block, 256 ... 4096, on the C2 shapes: the choices below are within 1 us of the best
//  everywhere -- profiles/r05_second_half_raw.txt)
static int split_slabs(int B, int C, long long HW) {
    int slabs = (int)((HW + 2047) / 2048);                 // >= 2048 pixels (x 8 channels) per block
    if (slabs < 1) slabs = 1;
    // small batches: 1024 pixels (one pass of the vector loop), then 512, while < 4 blocks per CU exist
    // (batch 1: 32-64 blocks of two serial passes each took 8.5 us per launch, profiles/r03_small_batch.txt)
    if ((long long)B * (C / 8) * slabs < 1024 && HW % 1024 == 0) slabs = (int)(HW / 1024);
    if ((long long)B * (C / 8) * slabs < 1024 && HW % 512 == 0) slabs = (int)(HW / 512);
    while (slabs > 1 && (long long)B * (C / 8) * slabs > 8192) slabs = (slabs + 1) / 2;
    return slabs;
}

extern "C" int64_t lc_split_act_units(int B, int C, int H, int W) {
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return 0;
    return (int64_t)B * 2 * ((C + 15) / 16 * 2) * H * W;    // 16-byte units
}

extern "C" int lc_groupnorm_apply_split(const float* x, int64_t x_bs, const double* partials,
                                        const float* gamma, const float* beta, const float* scale,
                                        const float* shift, int64_t ss_bs, void* y_split, int B, int C,
                                        int H, int W, int G, float eps, int act_silu,
                                        lc_conv_range* range, lc_stream_t s) {
    if (!x || !y_split || !partials || !range || B <= 0 || G <= 0 || C % G) return LC_EINVAL;
    if (C % 16) return LC_EUNSUP;
    const long long HW = (long long)H * W;
    if (2ll * (C / 8) * HW * 16 >= (1ll << 31)) return LC_EUNSUP;       // 32-bit byte offsets into one sample's planes (as the consumer conv)
    const int nch = gn_chunks(B, G, (long long)(C / G) * HW);
    OctStats2 os{nullptr, nullptr, 0, 0, 0, 0, 3, 3};
    hipLaunchKernelGGL(gn_apply_split_kernel<0>, dim3(split_slabs(B, C, HW), C / 8, B), dim3(256), 0,
                       lc_s(s), x, (long long)x_bs, partials, os, gamma, beta, scale, shift,
                       (long long)ss_bs, reinterpret_cast<half8_t*>(y_split),
                       (long long)2 * (C / 8) * HW, C, G, HW, nch, eps, act_silu, range);
    return lc_launch_status();
}

extern "C" int lc_groupnorm_apply_os_split(const float* x, int64_t x_bs, const lc_oct_stats* s0,
                                           const lc_oct_stats* s1, const float* gamma,
                                           const float* beta, const float* scale, const float* shift,
                                           int64_t ss_bs, void* y_split, int B, int C, int H, int W,
                                           int G, float eps, int act_silu, lc_conv_range* range,
                                           lc_stream_t s) {
    if (!x || !y_split || !range || B <= 0 || G <= 0 || C % G) return LC_EINVAL;
    const int cpg = C / G;
    OctStats2 os;
    if (const int rc = os_from_segments(s0, s1, C, cpg, os)) return rc;
    // (a block of this kernel owns a channel OCTET: groups of whole octets -- one (mean, rstd) per block -- or 2 / 4
    // channels per group -- one group per wave --, whatever the entries' unit)
    if ((cpg % 8 && cpg != 2 && cpg != 4) || os.c0 % 8 || C % 16) return LC_EUNSUP;
    const long long HW = (long long)H * W;
    if (2ll * (C / 8) * HW * 16 >= (1ll << 31)) return LC_EUNSUP;
    const dim3 grid(split_slabs(B, C, HW), C / 8, B);
    if (cpg % 8 == 0)
        hipLaunchKernelGGL(gn_apply_split_kernel<1>, grid, dim3(256), 0, lc_s(s), x, (long long)x_bs, nullptr, os, gamma,
                           beta, scale, shift, (long long)ss_bs, reinterpret_cast<half8_t*>(y_split),
                           (long long)2 * (C / 8) * HW, C, G, HW, 0, eps, act_silu, range);
    else
        hipLaunchKernelGGL(gn_apply_split_kernel<2>, grid, dim3(256), 0, lc_s(s), x, (long long)x_bs, nullptr, os, gamma,
                           beta, scale, shift, (long long)ss_bs, reinterpret_cast<half8_t*>(y_split),
                           (long long)2 * (C / 8) * HW, C, G, HW, 0, eps, act_silu, range);
    return lc_launch_status();
}

// ---- backward (training) ----------------------------------------------------------------------------
extern "C" int lc_groupnorm_meanrstd(const float* x, int64_t x_bs, const double* partials,
                                     float* mean_rstd, int B, int C, int H, int W, int G, float eps,
                                     lc_stream_t s) {
    if (!x || !partials || !mean_rstd || B <= 0 || G <= 0 || C % G) return LC_EINVAL;
    const long long HW = (long long)H * W;
    const int nch = gn_chunks(B, G, (long long)(C / G) * HW);
    hipLaunchKernelGGL(gn_meanrstd_kernel, dim3((B * G + 63) / 64), dim3(64), 0, lc_s(s), partials, x,
                       (long long)x_bs, mean_rstd, B, C, G, HW, nch, eps);
    return lc_launch_status();
}

extern "C" int lc_groupnorm_bwd_train(const float* x, int64_t x_bs, const float* dy, int64_t dy_bs,
                                      const float* mean_rstd, const float* gamma, const float* beta,
                                      const float* scale, const float* shift, int64_t ss_bs, double* rows,
                                      float* dx, int64_t dx_bs, float* dgamma, float* dbeta, float* dscale,
                                      float* dshift, int B, int C, int H, int W, int G, int act_silu,
                                      float* amax_out, lc_stream_t s) {
    if (!x || !dy || !mean_rstd || !rows || !dx || B <= 0 || G <= 0 || C % G) return LC_EINVAL;
    const long long HW = (long long)H * W;
    hipLaunchKernelGGL(gn_bwd_rows_kernel, dim3(C, B), dim3(256), 0, lc_s(s), x, (long long)x_bs, dy,
                       (long long)dy_bs, mean_rstd, gamma, beta, scale, shift, (long long)ss_bs, rows, C,
                       G, HW, act_silu);
    int slabs = (int)((HW + 4095) / 4096);
    if (slabs < 1) slabs = 1;
    hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3(slabs, C, B), dim3(256), 0, lc_s(s), x, (long long)x_bs,
                       dy, (long long)dy_bs, mean_rstd, gamma, beta, scale, shift, (long long)ss_bs, rows,
                       dx, (long long)dx_bs, C, G, HW, act_silu, dgamma, dbeta, dscale, dshift, B, amax_out);
    return lc_launch_status();
}

extern "C" int lc_groupnorm_bwd(const float* x, int64_t x_bs, const float* dy, int64_t dy_bs,
                                const float* mean_rstd, const float* gamma, const float* beta,
                                const float* scale, const float* shift, int64_t ss_bs, double* rows,
                                float* dx, int64_t dx_bs, int B, int C, int H, int W, int G,
                                int act_silu, lc_stream_t s) {
    return lc_groupnorm_bwd_train(x, x_bs, dy, dy_bs, mean_rstd, gamma, beta, scale, shift, ss_bs, rows, dx, dx_bs,
                                  nullptr, nullptr, nullptr, nullptr, B, C, H, W, G, act_silu, nullptr, s);
}

LC_TOUCH_TU(norm, gn_stats_kernel)
