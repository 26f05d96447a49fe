// f16x2-split variant of the attention backward pass (attention_bwd.hip): the same two kernels -- queries outside for
// dQ, keys outside for dK / dV, P recomputed from the forward's log2-sum-exp, deterministic -- on
// v_mfma_f32_32x32x16_f16 with the hi/lo operand split of the forward kernel (attention.hip, attn_h_kernel): every
// product a*b is ah*bh + ah*bl + al*bh with fp16 hi/lo parts of the pre-scaled fp32 values and fp32 accumulation
// (per-product error ~5e-7): 3 MFMAs of 32 clk per 16-deep k-step instead of 8 fp32 MFMAs of 64 clk.
//
// Operand forms (A rows / B columns = lane & 31, k = 8 * (lane >> 5) + 0..7), X in {K, V, Q, dO}:
//   channel units  X_c[cb][pos]        = X[8 cb + 0..7][pos]                    products that contract over channels
//   position units X_p[step][half][c]  = X[c][kappa(8 step + 0..7, half)]       products that contract over keys / queries:
//                                        their B operand is the previous product's accumulator registers 8 step .. 8 step + 7
//                                        (kappa(r, half) = (r & 3) + 8 (r >> 2) + 4 half), split in registers
// dQ kernel   (block = 4 waves x 32 queries; per 32-key tile: K_c, K_p, V_c staged):
//     S^T = K_c Q, dP^T = V_c dO, dS^T = scale P (dP - D) in registers, dQ^T += K_p dS^T
// dK/dV kernel (block = 4 waves x 32 keys; per 32-query tile: Q_c, Q_p, dO_c, dO_p, lse, D staged):
//     S = Q_c K, dV^T += dO_p P, dP = dO_c V, dK^T += Q_p dS
// Ranges: q, k, v take the pre-scales the forward pass used -- derived from max |q|, |k|, |v| as measured by the forward
// entry (attention_pre.h: 16 for activations of a normalised network, a power of two that fits otherwise; round 6, before:
// the constant 16, which saturated |x| >= 4094 silently); dO is a gradient of arbitrary magnitude: attn_dsum_h_kernel measures max |dO| next to D and every block derives the power of two that puts
// it at 2^6, so dP, dS and the outputs carry that factor exactly until the epilogue divides it out.
#include "common.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

#include "attention_pre.h"

constexpr float P_PRE = 2048.0f;

struct AttnBwdHArgs {
    const float *q, *k, *v, *dout, *lse, *dsum;
    const unsigned* do_amax;     // bit pattern of max |dO| (attn_dsum_h_kernel)
    const unsigned* qkv_amax;    // bit patterns of max |q|, |k|, |v| (the forward entry's measurement), or NULL: pre-scale 16
    float *dq, *dk, *dv;
    int Lq, Lk, dqk, dv_;
    float qscale, scale;
};

__device__ __forceinline__ int kappa(int r, int kh) { return (r & 3) + 8 * (r >> 2) + 4 * kh; }

// the split rule of the forward kernel: s = v * scale, hi = 11 significant bits (truncated), lo = fp16(s - hi)
__device__ __forceinline__ void split8(const float (&v)[8], float scale, half8& hi, half8& lo) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int k = 0; k < 8; k += 2) {
        const float s0 = v[k] * scale, s1 = v[k + 1] * scale;
        const float h0 = __uint_as_float(__float_as_uint(s0) & 0xFFFFE000u);
        const float h1 = __uint_as_float(__float_as_uint(s1) & 0xFFFFE000u);
        const h2 ph = __builtin_bit_cast(h2, __builtin_amdgcn_cvt_pkrtz(h0, h1));
        f2 r; r.x = s0 - h0; r.y = s1 - h1;
        const h2 pl = __builtin_convertvector(r, h2);
        hi[k] = ph.x; hi[k + 1] = ph.y; lo[k] = pl.x; lo[k + 1] = pl.y;
    }
}

// the power of two that puts max |dO| into [2^6, 2^7)
__device__ __forceinline__ float do_scale_from(unsigned amax_bits) {
    const float amax = __uint_as_float(amax_bits);
    if (!(amax > 0.0f) || !(amax < 3.0e38f)) return 1.0f;
    int e;
    frexpf(amax, &e);                       // amax = m * 2^e, m in [0.5, 1)
    int kx = 7 - e;
    kx = kx < -100 ? -100 : (kx > 100 ? 100 : kx);
    return ldexpf(1.0f, kx);
}

// D[bh][t] = sum_c do[c][t] * o[c][t], and max |dO| over the whole tensor (atomicMax on the bit pattern)
__global__ __launch_bounds__(256) void attn_dsum_h_kernel(const float* __restrict__ o, const float* __restrict__ dout,
                                                         float* __restrict__ dsum, unsigned* __restrict__ amax, int L, int d) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    float acc = 0.f, am = 0.f;
    if (t < L) {
        const long long base = (long long)blockIdx.y * d * L + t;
        for (int c = 0; c < d; ++c) {
            const float g = dout[base + (long long)c * L];
            acc = fmaf(g, o[base + (long long)c * L], acc);
            am = fmaxf(am, fabsf(g));
        }
        dsum[(long long)blockIdx.y * L + t] = acc;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) am = fmaxf(am, __shfl_xor(am, off, 64));
    if ((threadIdx.x & 63) == 0 && am > 0.f) atomicMax(amax, __float_as_uint(am));
}

// ---- staging helpers: one thread = one unit ------------------------------------------------------
// channel unit (cb, pos): 8 channels of one position, strided loads
__device__ __forceinline__ void load_cunit(float (&r)[8], const float* base, int L, int d, int cb, int pos, bool ok) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = cb * 8 + j;
        r[j] = (ok && c < d && pos < L) ? base[(long long)c * L + pos] : 0.f;
    }
}
// position octet (c, o): 8 consecutive positions of one channel
__device__ __forceinline__ void load_poct(float (&r)[8], const float* base, int L, int d, int c, int p0, bool ok) {
#pragma unroll
    for (int m = 0; m < 8; ++m) r[m] = (ok && c < d && p0 + m < L) ? base[(long long)c * L + p0 + m] : 0.f;
}
// ... and its two 4-element pieces in the position units (step = o >> 1, half 0 / 1, piece o & 1)
template <int D>
__device__ __forceinline__ void store_poct(half8* u_hi, half8* u_lo, const float (&r)[8], float scale, int c, int o) {
    half8 hi, lo;
    split8(r, scale, hi, lo);
    const int u0 = ((o >> 1) * 2 + 0) * D + c, u1 = u0 + D;
    half4* h0 = reinterpret_cast<half4*>(&u_hi[u0]) + (o & 1);
    half4* h1 = reinterpret_cast<half4*>(&u_hi[u1]) + (o & 1);
    half4* l0 = reinterpret_cast<half4*>(&u_lo[u0]) + (o & 1);
    half4* l1 = reinterpret_cast<half4*>(&u_lo[u1]) + (o & 1);
    *h0 = __builtin_shufflevector(hi, hi, 0, 1, 2, 3);
    *h1 = __builtin_shufflevector(hi, hi, 4, 5, 6, 7);
    *l0 = __builtin_shufflevector(lo, lo, 0, 1, 2, 3);
    *l1 = __builtin_shufflevector(lo, lo, 4, 5, 6, 7);
}

#define LC_MFMA3(ACC, AH, AL, BH, BL)                                           \
    do {                                                                        \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_f16(AL, BH, ACC, 0, 0, 0);     \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_f16(AH, BL, ACC, 0, 0, 0);     \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_f16(AH, BH, ACC, 0, 0, 0);     \
    } while (0)

template <int DQK, int NDV>
__global__ __launch_bounds__(256) void attn_bwd_dq_h_kernel(AttnBwdHArgs a) {
    constexpr int DV = NDV * 32, NDQ = DQK / 32, NST = DQK / 16, NSV = DV / 16;
    constexpr int NKC = DQK / 8 * 32, NKP = 4 * DQK, NVC = DV / 8 * 32;
    __shared__ half8 kc_hi[NKC], kc_lo[NKC], kp_hi[NKP], kp_lo[NKP], vc_hi[NVC], vc_lo[NVC];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, kh = lane >> 5, wave = tid >> 6;
    const long long bh = blockIdx.y;
    const int t = blockIdx.x * 128 + wave * 32 + l31;
    const bool tok = t < a.Lq;
    const float* qb = a.q + bh * a.dqk * a.Lq;
    const float* kb = a.k + bh * a.dqk * a.Lk;
    const float* vb = a.v + bh * a.dv_ * a.Lk;
    const float* dob = a.dout + bh * a.dv_ * a.Lq;
    const float sdo = do_scale_from(*a.do_amax);
    // operand pre-scales (q carries the softmax scale here: its maximum is taken after that factor)
    const float QP = a.qkv_amax ? attn_pre_from(__uint_as_float(a.qkv_amax[0]) * fabsf(a.qscale)) : 16.0f;
    const float KP = a.qkv_amax ? attn_pre_from(__uint_as_float(a.qkv_amax[1])) : 16.0f;
    const float VP = a.qkv_amax ? attn_pre_from(__uint_as_float(a.qkv_amax[2])) : 16.0f;
    const float S_UN = 1.0f / (QP * KP);
    half8 qh[NST], ql[NST], doh[NSV], dol[NSV];
#pragma unroll
    for (int st = 0; st < NST; ++st) {
        float v[8];
        load_cunit(v, qb, a.Lq, a.dqk, 2 * st + kh, t, tok);
        split8(v, a.qscale * QP, qh[st], ql[st]);
    }
#pragma unroll
    for (int st = 0; st < NSV; ++st) {
        float v[8];
        load_cunit(v, dob, a.Lq, a.dv_, 2 * st + kh, t, tok);
        split8(v, sdo, doh[st], dol[st]);
    }
    const float lse_t = tok ? a.lse[bh * a.Lq + t] : 0.f;
    const float VP_ = a.qkv_amax ? attn_pre_from(__uint_as_float(a.qkv_amax[2])) : 16.0f;
    const float d_t = tok ? a.dsum[bh * a.Lq + t] * sdo * (VP_ == 16.0f ? 1.0f : VP_ * (1.0f / 128.0f)) : 0.f;
    f32x16 dqacc[NDQ];
#pragma unroll
    for (int i = 0; i < NDQ; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) dqacc[i][r] = 0.f;

    // staging roles of this thread
    const int c_cb = tid >> 5, c_pos = tid & 31;       // channel unit
    const int p_c = tid >> 2, p_o = tid & 3;           // position octet
    float rkc[8], rkp[8], rvc[8];
    auto load_tile = [&](int s0) {
        load_cunit(rkc, kb, a.Lk, a.dqk, c_cb, s0 + c_pos, tid < NKC);
        load_poct(rkp, kb, a.Lk, a.dqk, p_c, s0 + 8 * p_o, tid < NKP);
        load_cunit(rvc, vb, a.Lk, a.dv_, c_cb, s0 + c_pos, tid < NVC);
    };
    auto store_tile = [&]() {
        if (tid < NKC) { half8 hi, lo; split8(rkc, KP, hi, lo); kc_hi[tid] = hi; kc_lo[tid] = lo; }
        if (tid < NKP) store_poct<DQK>(kp_hi, kp_lo, rkp, KP, p_c, p_o);
        if (tid < NVC) { half8 hi, lo; split8(rvc, VP, hi, lo); vc_hi[tid] = hi; vc_lo[tid] = lo; }
    };
    // dS = scale P (dP - D) grows with |v|: away from the default window it is carried times DS = VP / 128 (as if max |v|
    // were ~16), so that its fp16 split neither saturates (|v| ~ 3e4) nor loses its lo half (|v| ~ 2e-5); 1 by default
    const float DS = VP == 16.0f ? 1.0f : VP * (1.0f / 128.0f);
    const float c2 = DS / VP;                           // dP' = VP * sdo * dP
    load_tile(0);
    for (int s0 = 0; s0 < a.Lk; s0 += 32) {
        __syncthreads();
        store_tile();
        __syncthreads();
        if (s0 + 32 < a.Lk) load_tile(s0 + 32);
        f32x16 sacc, dpacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) { sacc[r] = 0.f; dpacc[r] = 0.f; }
#pragma unroll
        for (int st = 0; st < NST; ++st)
            LC_MFMA3(sacc, kc_hi[(2 * st + kh) * 32 + l31], kc_lo[(2 * st + kh) * 32 + l31], qh[st], ql[st]);
#pragma unroll
        for (int st = 0; st < NSV; ++st)
            LC_MFMA3(dpacc, vc_hi[(2 * st + kh) * 32 + l31], vc_lo[(2 * st + kh) * 32 + l31], doh[st], dol[st]);
        float ds[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = s0 + kappa(r, kh);
            const float p = key < a.Lk ? __builtin_amdgcn_exp2f(fmaf(sacc[r], S_UN, -lse_t)) : 0.f;
            ds[r] = p * (dpacc[r] * c2 - d_t) * a.scale;        // = sdo * dS
        }
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = ds[8 * st + j];
            half8 dsh, dsl;
            split8(v, 1.0f, dsh, dsl);
#pragma unroll
            for (int i = 0; i < NDQ; ++i)
                LC_MFMA3(dqacc[i], kp_hi[(st * 2 + kh) * DQK + i * 32 + l31], kp_lo[(st * 2 + kh) * DQK + i * 32 + l31], dsh, dsl);
        }
    }
    if (tok) {
        const float un = 1.0f / (KP * sdo * DS);
        float* dqb = a.dq + bh * a.dqk * a.Lq;
#pragma unroll
        for (int i = 0; i < NDQ; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = i * 32 + kappa(r, kh);
                if (c < a.dqk) dqb[(long long)c * a.Lq + t] = dqacc[i][r] * un;
            }
    }
}

template <int DQK, int NDV>
__global__ __launch_bounds__(256) void attn_bwd_dkv_h_kernel(AttnBwdHArgs a) {
    constexpr int DV = NDV * 32, NDQ = DQK / 32, NST = DQK / 16, NSV = DV / 16;
    constexpr int NQC = DQK / 8 * 32, NQP = 4 * DQK, NDC = DV / 8 * 32, NDP = 4 * DV;
    __shared__ half8 qc_hi[NQC], qc_lo[NQC], qp_hi[NQP], qp_lo[NQP], dc_hi[NDC], dc_lo[NDC], dp_hi[NDP], dp_lo[NDP];
    __shared__ float lse_s[32], d_s[32];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, kh = lane >> 5, wave = tid >> 6;
    const long long bh = blockIdx.y;
    const int s = blockIdx.x * 128 + wave * 32 + l31;      // this lane's key
    const bool sok = s < a.Lk;
    const float* qb = a.q + bh * a.dqk * a.Lq;
    const float* kb = a.k + bh * a.dqk * a.Lk;
    const float* vb = a.v + bh * a.dv_ * a.Lk;
    const float* dob = a.dout + bh * a.dv_ * a.Lq;
    const float sdo = do_scale_from(*a.do_amax);
    const float QP = a.qkv_amax ? attn_pre_from(__uint_as_float(a.qkv_amax[0])) : 16.0f;   // (q without the softmax scale here)
    const float KP = a.qkv_amax ? attn_pre_from(__uint_as_float(a.qkv_amax[1])) : 16.0f;
    const float VP = a.qkv_amax ? attn_pre_from(__uint_as_float(a.qkv_amax[2])) : 16.0f;
    const float DS = VP == 16.0f ? 1.0f : VP * (1.0f / 128.0f);     // dS is carried times DS (see the dQ kernel)
    half8 kh_[NST], kl_[NST], vh_[NSV], vl_[NSV];
#pragma unroll
    for (int st = 0; st < NST; ++st) {
        float v[8];
        load_cunit(v, kb, a.Lk, a.dqk, 2 * st + kh, s, sok);
        split8(v, KP, kh_[st], kl_[st]);
    }
#pragma unroll
    for (int st = 0; st < NSV; ++st) {
        float v[8];
        load_cunit(v, vb, a.Lk, a.dv_, 2 * st + kh, s, sok);
        split8(v, VP, vh_[st], vl_[st]);
    }
    f32x16 dkacc[NDQ], dvacc[NDV];
#pragma unroll
    for (int i = 0; i < NDQ; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) dkacc[i][r] = 0.f;
#pragma unroll
    for (int i = 0; i < NDV; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) dvacc[i][r] = 0.f;

    const int c_cb = tid >> 5, c_pos = tid & 31;
    const int p_c = tid >> 2, p_o = tid & 3;
    float rqc[8], rqp[8], rdc[8], rdp[8], rl = 0.f, rd = 0.f;
    auto load_tile = [&](int t0) {
        load_cunit(rqc, qb, a.Lq, a.dqk, c_cb, t0 + c_pos, tid < NQC);
        load_poct(rqp, qb, a.Lq, a.dqk, p_c, t0 + 8 * p_o, tid < NQP);
        load_cunit(rdc, dob, a.Lq, a.dv_, c_cb, t0 + c_pos, tid < NDC);
        load_poct(rdp, dob, a.Lq, a.dv_, p_c, t0 + 8 * p_o, tid < NDP);
        if (tid < 32) {
            const int t = t0 + tid;
            rl = t < a.Lq ? a.lse[bh * a.Lq + t] : 0.f;
            rd = t < a.Lq ? a.dsum[bh * a.Lq + t] : 0.f;
        }
    };
    auto store_tile = [&]() {
        if (tid < NQC) { half8 hi, lo; split8(rqc, QP, hi, lo); qc_hi[tid] = hi; qc_lo[tid] = lo; }
        if (tid < NQP) store_poct<DQK>(qp_hi, qp_lo, rqp, QP, p_c, p_o);
        if (tid < NDC) { half8 hi, lo; split8(rdc, sdo, hi, lo); dc_hi[tid] = hi; dc_lo[tid] = lo; }
        if (tid < NDP) store_poct<DV>(dp_hi, dp_lo, rdp, sdo, p_c, p_o);
        if (tid < 32) { lse_s[tid] = rl; d_s[tid] = rd * sdo * DS; }
    };
    const float c1 = a.qscale / (QP * KP), c2 = DS / VP;
    load_tile(0);
    for (int t0 = 0; t0 < a.Lq; t0 += 32) {
        __syncthreads();
        store_tile();
        __syncthreads();
        if (t0 + 32 < a.Lq) load_tile(t0 + 32);
        f32x16 sacc, dpacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) { sacc[r] = 0.f; dpacc[r] = 0.f; }
#pragma unroll
        for (int st = 0; st < NST; ++st)      // S: rows = queries of the tile, columns = this lane's key
            LC_MFMA3(sacc, qc_hi[(2 * st + kh) * 32 + l31], qc_lo[(2 * st + kh) * 32 + l31], kh_[st], kl_[st]);
#pragma unroll
        for (int st = 0; st < NSV; ++st)      // dP (x V_PRE x sdo)
            LC_MFMA3(dpacc, dc_hi[(2 * st + kh) * 32 + l31], dc_lo[(2 * st + kh) * 32 + l31], vh_[st], vl_[st]);
        float p[16], ds[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int tq = kappa(r, kh);
            p[r] = (sok && t0 + tq < a.Lq) ? __builtin_amdgcn_exp2f(fmaf(sacc[r], c1, -lse_s[tq])) : 0.f;
            ds[r] = p[r] * (dpacc[r] * c2 - d_s[tq]) * a.scale;      // = sdo * dS
        }
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            float v[8];
            half8 bh_, bl_;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = p[8 * st + j];
            split8(v, P_PRE, bh_, bl_);
#pragma unroll
            for (int i = 0; i < NDV; ++i)     // dV^T += dO_p P   (x sdo x P_PRE)
                LC_MFMA3(dvacc[i], dp_hi[(st * 2 + kh) * DV + i * 32 + l31], dp_lo[(st * 2 + kh) * DV + i * 32 + l31], bh_, bl_);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = ds[8 * st + j];
            split8(v, 1.0f, bh_, bl_);
#pragma unroll
            for (int i = 0; i < NDQ; ++i)     // dK^T += Q_p dS   (x QP x sdo)
                LC_MFMA3(dkacc[i], qp_hi[(st * 2 + kh) * DQK + i * 32 + l31], qp_lo[(st * 2 + kh) * DQK + i * 32 + l31], bh_, bl_);
        }
    }
    if (sok) {
        const float unk = 1.0f / (QP * sdo * DS), unv = 1.0f / (P_PRE * sdo);
        float* dkb = a.dk + bh * a.dqk * a.Lk;
        float* dvb = a.dv + bh * a.dv_ * a.Lk;
#pragma unroll
        for (int i = 0; i < NDQ; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = i * 32 + kappa(r, kh);
                if (c < a.dqk) dkb[(long long)c * a.Lk + s] = dkacc[i][r] * unk;
            }
#pragma unroll
        for (int i = 0; i < NDV; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = i * 32 + kappa(r, kh);
                if (c < a.dv_) dvb[(long long)c * a.Lk + s] = dvacc[i][r] * unv;
            }
    }
}
#undef LC_MFMA3

}  // namespace

// dsum_scratch: float [BH * Lq + 1]: D per (head, query) and, in the last word, the bit pattern of max |dO|
// qkv_amax: the 3 words the forward entry (lc_attention_train_fwd) measured, or NULL (constant pre-scale 16)
extern "C" int lc_attention_bwd_f16x2(const float* q, const float* k, const float* v, const float* o, const float* dout,
                                      const float* lse, float* dsum_scratch, float* dq, float* dk, float* dv, int BH,
                                      int Lq, int Lk, int dqk, int dv_ch, float scale, const float* qkv_amax, lc_stream_t s) {
    if (!q || !k || !v || !o || !dout || !lse || !dsum_scratch || !dq || !dk || !dv || BH <= 0 || Lq <= 0 || Lk <= 0)
        return LC_EINVAL;
    if (dqk <= 0 || dqk > 64 || dv_ch <= 0 || dv_ch > 64) return LC_EUNSUP;
    AttnBwdHArgs a;
    unsigned* amax = reinterpret_cast<unsigned*>(dsum_scratch + (long long)BH * Lq);
    a.q = q; a.k = k; a.v = v; a.dout = dout; a.lse = lse; a.dsum = dsum_scratch; a.do_amax = amax;
    a.qkv_amax = reinterpret_cast<const unsigned*>(qkv_amax);
    a.dq = dq; a.dk = dk; a.dv = dv;
    a.Lq = Lq; a.Lk = Lk; a.dqk = dqk; a.dv_ = dv_ch;
    a.scale = scale; a.qscale = scale * 1.4426950408889634f;
    if (hipMemsetAsync(amax, 0, sizeof(unsigned), lc_s(s)) != hipSuccess) return lc_launch_status();
    hipLaunchKernelGGL(attn_dsum_h_kernel, dim3((Lq + 255) / 256, BH), dim3(256), 0, lc_s(s), o, dout, dsum_scratch, amax, Lq,
                       dv_ch);
    const int dqp = dqk <= 32 ? 32 : 64, nd = dv_ch <= 32 ? 1 : 2;
    const dim3 gq((Lq + 127) / 128, BH), gk((Lk + 127) / 128, BH);
#define LC_BWD(DQ, ND)                                                                           \
    do {                                                                                         \
        hipLaunchKernelGGL((attn_bwd_dq_h_kernel<DQ, ND>), gq, dim3(256), 0, lc_s(s), a);        \
        hipLaunchKernelGGL((attn_bwd_dkv_h_kernel<DQ, ND>), gk, dim3(256), 0, lc_s(s), a);       \
    } while (0)
    if (dqp == 32 && nd == 1) LC_BWD(32, 1);
    else if (dqp == 64 && nd == 1) LC_BWD(64, 1);
    else if (dqp == 32 && nd == 2) LC_BWD(32, 2);
    else LC_BWD(64, 2);
#undef LC_BWD
    return lc_launch_status();
}

LC_TOUCH_TU(attention_bwd_h, attn_dsum_h_kernel)
