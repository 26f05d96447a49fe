// Backward pass of the flash attention of attention.hip (training): exact fp32 MFMA
// (v_mfma_f32_32x32x2_f32), no score matrix in HBM, deterministic (no atomics).
//
// Reference: autograd of nn.MultiheadAttention (efficient_unet.py:28-58) and of
// ObjectAwareCrossAttention.forward (layout_unet_v1.py:489-506: softmax in fp32 at :502), as
// tools/train/train_lidm_cond.py:259-322 runs them.  The reference materialises the [B*heads, Lq, Lk]
// scores and their gradient (1 GB per layer and direction at 8 x 8 heads x 2048 x 2061); here P is
// recomputed tile by tile from the per-query log2-sum-exp the forward kernels leave (lc_attention_train_fwd).
//
// Operands: per head a [d, L] matrix with L contiguous (what a 1x1 conv on NCHW produces), heads
// back to back: q [BH, dqk, Lq], k [BH, dqk, Lk], v [BH, dv, Lk], o / do [BH, dv, Lq], lse [BH, Lq].
//   S2[s][t] = qscale * sum_c k[c][s] q[c][t]   (qscale = scale * log2 e)      P = exp2(S2 - lse[t])
//   D[t]  = sum_c do[c][t] o[c][t]              dP[s][t] = sum_c v[c][s] do[c][t]
//   dS    = scale * P (dP - D[t])
//   dq[c][t] = sum_s k[c][s] dS[s][t]    dk[c][s] = sum_t q[c][t] dS[s][t]    dv[c][s] = sum_t do[c][t] P[s][t]
// Two kernels, both with the register trick of attention.hip: an S tile's accumulator register r of a lane IS the
// B operand of the next product's k-step r (rows kappa(r, half) = (r & 3) + 8 (r >> 2) + 4 half).
//   attn_bwd_dq_kernel   block = 4 waves x 32 queries, loops over 32-key tiles (K, V staged in LDS):
//                        S^T = K^T Q, dP^T = V^T dO (lane = query: lse, D lane-local), dQ^T += K dS^T
//   attn_bwd_dkv_kernel  block = 4 waves x 32 keys, loops over 32-query tiles (Q, dO, lse, D staged in LDS):
//                        S = Q^T K, dV^T += dO P, dP = dO^T V, dK^T += Q dS
#include "common.h"

namespace {

struct AttnBwdArgs {
    const float *q, *k, *v, *dout, *lse, *dsum;
    float *dq, *dk, *dv;
    int Lq, Lk, dqk, dv_;
    float qscale, scale;
};

constexpr int TS = 33;   // LDS row stride of a 32-wide tile: conflict-free along rows and along columns

__device__ __forceinline__ int kappa(int r, int kh) { return (r & 3) + 8 * (r >> 2) + 4 * kh; }

// D[bh][t] = sum_c do[c][t] * o[c][t]
__global__ __launch_bounds__(256) void attn_dsum_kernel(const float* __restrict__ o, const float* __restrict__ dout,
                                                       float* __restrict__ dsum, int L, int d) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= L) return;
    const long long base = (long long)blockIdx.y * d * L + t;
    float acc = 0.f;
    for (int c = 0; c < d; ++c) acc = fmaf(dout[base + (long long)c * L], o[base + (long long)c * L], acc);
    dsum[(long long)blockIdx.y * L + t] = acc;
}

template <int DQK, int NDV>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(AttnBwdArgs a) {
    constexpr int DV = NDV * 32, NDQ = DQK / 32;
    __shared__ float ks[DQK * TS];
    __shared__ float vs[DV * TS];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, kh = lane >> 5, wave = tid >> 6;
    const long long bh = blockIdx.y;
    const int t = blockIdx.x * 128 + wave * 32 + l31;
    const bool tok = t < a.Lq;
    const float* qb = a.q + bh * a.dqk * a.Lq;
    const float* kb = a.k + bh * a.dqk * a.Lk;
    const float* vb = a.v + bh * a.dv_ * a.Lk;
    const float* dob = a.dout + bh * a.dv_ * a.Lq;
    float qreg[DQK / 2], doreg[DV / 2];
#pragma unroll
    for (int kk = 0; kk < DQK / 2; ++kk) {
        const int c = 2 * kk + kh;
        qreg[kk] = (tok && c < a.dqk) ? qb[(long long)c * a.Lq + t] : 0.f;
    }
#pragma unroll
    for (int kk = 0; kk < DV / 2; ++kk) {
        const int c = 2 * kk + kh;
        doreg[kk] = (tok && c < a.dv_) ? dob[(long long)c * a.Lq + t] : 0.f;
    }
    const float lse_t = tok ? a.lse[bh * a.Lq + t] : 0.f;
    const float d_t = tok ? a.dsum[bh * a.Lq + t] : 0.f;
    f32x16 dqacc[NDQ];
#pragma unroll
    for (int i = 0; i < NDQ; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) dqacc[i][r] = 0.f;

    for (int s0 = 0; s0 < a.Lk; s0 += 32) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < DQK * 32 / 256; ++i) {
            const int e = tid + i * 256, c = e >> 5, sl = e & 31, s = s0 + sl;
            ks[c * TS + sl] = (c < a.dqk && s < a.Lk) ? kb[(long long)c * a.Lk + s] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < DV * 32 / 256; ++i) {
            const int e = tid + i * 256, c = e >> 5, sl = e & 31, s = s0 + sl;
            vs[c * TS + sl] = (c < a.dv_ && s < a.Lk) ? vb[(long long)c * a.Lk + s] : 0.f;
        }
        __syncthreads();
        f32x16 sacc, dpacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) { sacc[r] = 0.f; dpacc[r] = 0.f; }
#pragma unroll
        for (int kk = 0; kk < DQK / 2; ++kk)      // S^T: A[i = key][k = c] = K, B[k = c][j = query] = Q
            sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(ks[(2 * kk + kh) * TS + l31], qreg[kk], sacc, 0, 0, 0);
#pragma unroll
        for (int kk = 0; kk < DV / 2; ++kk)       // dP^T: A = V, B = dO
            dpacc = __builtin_amdgcn_mfma_f32_32x32x2f32(vs[(2 * kk + kh) * TS + l31], doreg[kk], dpacc, 0, 0, 0);
        float ds[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = s0 + kappa(r, kh);
            const float p = key < a.Lk ? exp2f(fmaf(sacc[r], a.qscale, -lse_t)) : 0.f;
            ds[r] = p * (dpacc[r] - d_t) * a.scale;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {            // dQ^T += K dS^T: A[i = c][k = key kappa] = K, B = dS register r
            const int kap = kappa(r, kh);
#pragma unroll
            for (int i = 0; i < NDQ; ++i)
                dqacc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(ks[(i * 32 + l31) * TS + kap], ds[r], dqacc[i], 0, 0, 0);
        }
    }
    if (tok) {
        float* dqb = a.dq + bh * a.dqk * a.Lq;
#pragma unroll
        for (int i = 0; i < NDQ; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = i * 32 + kappa(r, kh);
                if (c < a.dqk) dqb[(long long)c * a.Lq + t] = dqacc[i][r];
            }
    }
}

template <int DQK, int NDV>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(AttnBwdArgs a) {
    constexpr int DV = NDV * 32, NDQ = DQK / 32;
    __shared__ float qs[DQK * TS];
    __shared__ float dos[DV * TS];
    __shared__ float lse_s[32], d_s[32];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, kh = lane >> 5, wave = tid >> 6;
    const long long bh = blockIdx.y;
    const int s = blockIdx.x * 128 + wave * 32 + l31;     // this lane's key
    const bool sok = s < a.Lk;
    const float* qb = a.q + bh * a.dqk * a.Lq;
    const float* kb = a.k + bh * a.dqk * a.Lk;
    const float* vb = a.v + bh * a.dv_ * a.Lk;
    const float* dob = a.dout + bh * a.dv_ * a.Lq;
    float kreg[DQK / 2], vreg[DV / 2];
#pragma unroll
    for (int kk = 0; kk < DQK / 2; ++kk) {
        const int c = 2 * kk + kh;
        kreg[kk] = (sok && c < a.dqk) ? kb[(long long)c * a.Lk + s] : 0.f;
    }
#pragma unroll
    for (int kk = 0; kk < DV / 2; ++kk) {
        const int c = 2 * kk + kh;
        vreg[kk] = (sok && c < a.dv_) ? vb[(long long)c * a.Lk + s] : 0.f;
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

    for (int t0 = 0; t0 < a.Lq; t0 += 32) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < DQK * 32 / 256; ++i) {
            const int e = tid + i * 256, c = e >> 5, tl = e & 31, t = t0 + tl;
            qs[c * TS + tl] = (c < a.dqk && t < a.Lq) ? qb[(long long)c * a.Lq + t] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < DV * 32 / 256; ++i) {
            const int e = tid + i * 256, c = e >> 5, tl = e & 31, t = t0 + tl;
            dos[c * TS + tl] = (c < a.dv_ && t < a.Lq) ? dob[(long long)c * a.Lq + t] : 0.f;
        }
        if (tid < 32) {
            const int t = t0 + tid;
            lse_s[tid] = t < a.Lq ? a.lse[bh * a.Lq + t] : 0.f;
            d_s[tid] = t < a.Lq ? a.dsum[bh * a.Lq + t] : 0.f;
        }
        __syncthreads();
        f32x16 sacc, dpacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) { sacc[r] = 0.f; dpacc[r] = 0.f; }
#pragma unroll
        for (int kk = 0; kk < DQK / 2; ++kk)      // S: A[i = query][k = c] = Q, B[k = c][j = key] = K
            sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(qs[(2 * kk + kh) * TS + l31], kreg[kk], sacc, 0, 0, 0);
#pragma unroll
        for (int kk = 0; kk < DV / 2; ++kk)       // dP: A = dO, B = V
            dpacc = __builtin_amdgcn_mfma_f32_32x32x2f32(dos[(2 * kk + kh) * TS + l31], vreg[kk], dpacc, 0, 0, 0);
        float p[16], ds[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {            // row = query kappa(r, half) of the tile, column = this lane's key
            const int tq = kappa(r, kh);
            p[r] = (sok && t0 + tq < a.Lq) ? exp2f(fmaf(sacc[r], a.qscale, -lse_s[tq])) : 0.f;
            ds[r] = p[r] * (dpacc[r] - d_s[tq]) * a.scale;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int kap = kappa(r, kh);
#pragma unroll
            for (int i = 0; i < NDV; ++i)         // dV^T += dO P: A[i = c][k = query kappa] = dO, B = P register r
                dvacc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(dos[(i * 32 + l31) * TS + kap], p[r], dvacc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < NDQ; ++i)         // dK^T += Q dS
                dkacc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(qs[(i * 32 + l31) * TS + kap], ds[r], dkacc[i], 0, 0, 0);
        }
    }
    if (sok) {
        float* dkb = a.dk + bh * a.dqk * a.Lk;
        float* dvb = a.dv + bh * a.dv_ * a.Lk;
#pragma unroll
        for (int i = 0; i < NDQ; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = i * 32 + kappa(r, kh);
                if (c < a.dqk) dkb[(long long)c * a.Lk + s] = dkacc[i][r];
            }
#pragma unroll
        for (int i = 0; i < NDV; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = i * 32 + kappa(r, kh);
                if (c < a.dv_) dvb[(long long)c * a.Lk + s] = dvacc[i][r];
            }
    }
}

}  // namespace

extern "C" int lc_attention_bwd(const float* q, const float* k, const float* v, const float* o, const float* dout,
                                const float* lse, float* dsum_scratch, float* dq, float* dk, float* dv, int BH, int Lq,
                                int Lk, int dqk, int dv_ch, float scale, lc_stream_t s) {
    if (!q || !k || !v || !o || !dout || !lse || !dsum_scratch || !dq || !dk || !dv || BH <= 0 || Lq <= 0 || Lk <= 0)
        return LC_EINVAL;
    if (dqk <= 0 || dqk > 64 || dv_ch <= 0 || dv_ch > 64) return LC_EUNSUP;
    AttnBwdArgs a;
    a.q = q; a.k = k; a.v = v; a.dout = dout; a.lse = lse; a.dsum = dsum_scratch;
    a.dq = dq; a.dk = dk; a.dv = dv;
    a.Lq = Lq; a.Lk = Lk; a.dqk = dqk; a.dv_ = dv_ch;
    a.scale = scale; a.qscale = scale * 1.4426950408889634f;
    hipLaunchKernelGGL(attn_dsum_kernel, dim3((Lq + 255) / 256, BH), dim3(256), 0, lc_s(s), o, dout, dsum_scratch, Lq, dv_ch);
    const int dqp = dqk <= 32 ? 32 : 64, nd = dv_ch <= 32 ? 1 : 2;
    const dim3 gq((Lq + 127) / 128, BH), gk((Lk + 127) / 128, BH);
#define LC_BWD(DQ, ND)                                                                         \
    do {                                                                                       \
        hipLaunchKernelGGL((attn_bwd_dq_kernel<DQ, ND>), gq, dim3(256), 0, lc_s(s), a);        \
        hipLaunchKernelGGL((attn_bwd_dkv_kernel<DQ, ND>), gk, dim3(256), 0, lc_s(s), a);       \
    } while (0)
    if (dqp == 32 && nd == 1) LC_BWD(32, 1);
    else if (dqp == 64 && nd == 1) LC_BWD(64, 1);
    else if (dqp == 32 && nd == 2) LC_BWD(32, 2);
    else LC_BWD(64, 2);
#undef LC_BWD
    return lc_launch_status();
}

LC_TOUCH_TU(attention_bwd, attn_dsum_kernel)
