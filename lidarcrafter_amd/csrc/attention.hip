// Flash-style attention over range-image tokens with channel-major operands, fp32 MFMA.
//
// Reference: nn.MultiheadAttention inside SelfAttentionBlock (efficient_unet.py:28-58),
// QKVAttentionLegacy (layout_unet_v1.py:555-596) and ObjectAwareCrossAttention.forward
// (layout_unet_v1.py:489-506), which materialise the [B*heads, Lq, Lk] score tensor in HBM
// (135 MB/sample at ds=4).  Here the scores live in MFMA accumulators only.
//
// Per head, operands are [d, L] matrices with L contiguous (exactly what a 1x1 conv on NCHW
// produces), so both MFMA operands of S^T = K^T Q are coalesced row reads:
//   S^T tile (32 keys x 32 queries):  A[i=key][k=c] = K[c][s],  B[k=c][j=query] = Q[c][t]
//   -> lane l holds query j=l&31 and 16 key rows: softmax reductions are in-lane + ONE
//      cross-half shuffle (lane ^ 32); the running max / sum / rescale are lane-local.
//   O^T tile (32 ch x 32 queries):   A[i=c][k=key] = V[c][s],  B[k=key][j=query] = P^T[s][t]
//   -> B is register r of the S^T accumulator as is: k-step r contracts keys
//      kappa(r,half) = (r&3)+8*(r>>2)+4*half, and V is read from LDS with the same kappa.
// Block = 4 waves x 32 queries of one head; K/V tiles of 32 keys staged in LDS.
#include "common.h"

namespace {

struct AttnArgs {
    const float *q, *k, *v, *k2, *v2;
    float* o;
    long long q_bs, q_hs, q_cs, k_bs, k_hs, k_cs, v_bs, v_hs, v_cs;
    long long k2_bs, k2_hs, k2_cs, v2_bs, v2_hs, v2_cs, o_bs, o_hs, o_cs;
    int heads, Lq, Lk0, Lk1, dqk, dv;
    float qscale;  // scale * log2(e)
};

template <int DQK, int NDV>
__global__ __launch_bounds__(256) void attn_kernel(AttnArgs a) {
    constexpr int DV = NDV * 32;
    constexpr int VS = 33;
    __shared__ float ks[DQK * 32];
    __shared__ float vs[DV * VS];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, kh = lane >> 5;
    const int wave = tid >> 6;
    const int bh = blockIdx.y, b = bh / a.heads, h = bh - b * a.heads;
    const int t0 = blockIdx.x * 128 + wave * 32;
    const int t = t0 + l31;
    const int Lk = a.Lk0 + a.Lk1;

    const float* qp = a.q + b * a.q_bs + h * a.q_hs;
    float qreg[DQK / 2];
#pragma unroll
    for (int kk = 0; kk < DQK / 2; ++kk) {
        const int c = 2 * kk + kh;
        qreg[kk] = (c < a.dqk && t < a.Lq) ? qp[c * a.q_cs + t] * a.qscale : 0.f;
    }
    f32x16 oacc[NDV];
#pragma unroll
    for (int i = 0; i < NDV; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[i][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    const float* kp0 = a.k + b * a.k_bs + h * a.k_hs;
    const float* vp0 = a.v + b * a.v_bs + h * a.v_hs;
    const float* kp1 = a.k2 ? a.k2 + b * a.k2_bs + h * a.k2_hs : nullptr;
    const float* vp1 = a.v2 ? a.v2 + b * a.v2_bs + h * a.v2_hs : nullptr;

    for (int s0 = 0; s0 < Lk; s0 += 32) {
        __syncthreads();  // previous tile fully consumed
        // ---- stage K [DQK][32] and V [DV][32] -------------------------------------------------
#pragma unroll
        for (int i = 0; i < DQK * 32 / 256; ++i) {
            const int e = tid + i * 256;
            const int c = e >> 5, sl = e & 31, s = s0 + sl;
            float val = 0.f;
            if (c < a.dqk) {
                if (s < a.Lk0) val = kp0[c * a.k_cs + s];
                else if (s < Lk) val = kp1[c * a.k2_cs + (s - a.Lk0)];
            }
            ks[e] = val;
        }
#pragma unroll
        for (int i = 0; i < DV * 32 / 256; ++i) {
            const int e = tid + i * 256;
            const int c = e >> 5, sl = e & 31, s = s0 + sl;
            float val = 0.f;
            if (c < a.dv) {
                if (s < a.Lk0) val = vp0[c * a.v_cs + s];
                else if (s < Lk) val = vp1[c * a.v2_cs + (s - a.Lk0)];
            }
            vs[c * VS + sl] = val;
        }
        __syncthreads();
        // ---- S^T = K^T Q ----------------------------------------------------------------------
        f32x16 sacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
#pragma unroll
        for (int kk = 0; kk < DQK / 2; ++kk) {
            const float av = ks[(2 * kk + kh) * 32 + l31];
            sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, qreg[kk], sacc, 0, 0, 0);
        }
        // ---- online softmax (base 2) ------------------------------------------------------------
        float mt = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = s0 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            if (key >= Lk) sacc[r] = -INFINITY;
            mt = fmaxf(mt, sacc[r]);
        }
        mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
        const float m_new = fmaxf(m_run, mt);
        const float alpha = exp2f(m_run - m_new);
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            sacc[r] = exp2f(sacc[r] - m_new);
            psum += sacc[r];
        }
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int i = 0; i < NDV; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[i][r] *= alpha;
        // ---- O^T += V P^T ---------------------------------------------------------------------
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int kap = (r & 3) + 8 * (r >> 2) + 4 * kh;
#pragma unroll
            for (int i = 0; i < NDV; ++i) {
                const float av = vs[(i * 32 + l31) * VS + kap];
                oacc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, sacc[r], oacc[i], 0, 0, 0);
            }
        }
    }
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    float* op = a.o + b * a.o_bs + h * a.o_hs;
    if (t < a.Lq) {
#pragma unroll
        for (int i = 0; i < NDV; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (c < a.dv) op[c * a.o_cs + t] = oacc[i][r] * inv;
            }
    }
}

}  // namespace

extern "C" int lc_attention_fwd(const float* q, int64_t q_bs, int64_t q_hs, int64_t q_cs,
                                const float* k, int64_t k_bs, int64_t k_hs, int64_t k_cs,
                                const float* v, int64_t v_bs, int64_t v_hs, int64_t v_cs,
                                const float* k2, int64_t k2_bs, int64_t k2_hs, int64_t k2_cs,
                                const float* v2, int64_t v2_bs, int64_t v2_hs, int64_t v2_cs,
                                float* o, int64_t o_bs, int64_t o_hs, int64_t o_cs, int B, int heads,
                                int Lq, int Lk0, int Lk1, int dqk, int dv, float scale,
                                lc_stream_t s) {
    if (!q || !k || !v || !o || B <= 0 || heads <= 0 || Lq <= 0 || Lk0 <= 0 || Lk1 < 0)
        return LC_EINVAL;
    if (Lk1 > 0 && (!k2 || !v2)) return LC_EINVAL;
    if (dqk <= 0 || dqk > 64 || dv <= 0 || dv > 64) return LC_EUNSUP;
    AttnArgs a;
    a.q = q; a.k = k; a.v = v; a.k2 = k2; a.v2 = v2; a.o = o;
    a.q_bs = q_bs; a.q_hs = q_hs; a.q_cs = q_cs; a.k_bs = k_bs; a.k_hs = k_hs; a.k_cs = k_cs;
    a.v_bs = v_bs; a.v_hs = v_hs; a.v_cs = v_cs; a.k2_bs = k2_bs; a.k2_hs = k2_hs; a.k2_cs = k2_cs;
    a.v2_bs = v2_bs; a.v2_hs = v2_hs; a.v2_cs = v2_cs; a.o_bs = o_bs; a.o_hs = o_hs; a.o_cs = o_cs;
    a.heads = heads; a.Lq = Lq; a.Lk0 = Lk0; a.Lk1 = Lk1; a.dqk = dqk; a.dv = dv;
    a.qscale = scale * 1.4426950408889634f;
    dim3 grid((Lq + 127) / 128, B * heads);
    const int dq = dqk <= 32 ? 32 : 64, nd = dv <= 32 ? 1 : 2;
    if (dq == 32 && nd == 1) hipLaunchKernelGGL((attn_kernel<32, 1>), grid, dim3(256), 0, lc_s(s), a);
    else if (dq == 64 && nd == 1) hipLaunchKernelGGL((attn_kernel<64, 1>), grid, dim3(256), 0, lc_s(s), a);
    else if (dq == 32 && nd == 2) hipLaunchKernelGGL((attn_kernel<32, 2>), grid, dim3(256), 0, lc_s(s), a);
    else hipLaunchKernelGGL((attn_kernel<64, 2>), grid, dim3(256), 0, lc_s(s), a);
    return lc_launch_status();
}
