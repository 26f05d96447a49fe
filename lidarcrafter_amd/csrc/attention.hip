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
    lc_cm_operand q, qp, k, kp, v, k2, k2p, v2;
    float* o;
    long long o_bs, o_hs, o_cs;
    int heads, Lq, Lk0, Lk1, dqk, dpos, dv;
    float qscale;  // scale * log2(e)
};

__device__ __forceinline__ const float* head_ptr(const lc_cm_operand& x, int b, int h) {
    return x.p ? x.p + b * x.bs + h * x.hs : nullptr;
}

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

    const int dq = a.dqk + a.dpos;  // channels of a head's q/k: content ++ positional
    const float* qc = head_ptr(a.q, b, h);
    const float* qpos = head_ptr(a.qp, b, h);
    float qreg[DQK / 2];
#pragma unroll
    for (int kk = 0; kk < DQK / 2; ++kk) {
        const int c = 2 * kk + kh;
        float val = 0.f;
        if (t < a.Lq) {
            if (c < a.dqk) val = qc[c * a.q.cs + t];
            else if (c < dq) val = qpos[(c - a.dqk) * a.qp.cs + t];
        }
        qreg[kk] = val * a.qscale;
    }
    f32x16 oacc[NDV];
#pragma unroll
    for (int i = 0; i < NDV; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[i][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    const float* kc0 = head_ptr(a.k, b, h);
    const float* kq0 = head_ptr(a.kp, b, h);
    const float* vp0 = head_ptr(a.v, b, h);
    const float* kc1 = head_ptr(a.k2, b, h);
    const float* kq1 = head_ptr(a.k2p, b, h);
    const float* vp1 = head_ptr(a.v2, b, h);

    for (int s0 = 0; s0 < Lk; s0 += 32) {
        __syncthreads();  // previous tile fully consumed
        // ---- stage K [DQK][32] and V [DV][32] -------------------------------------------------
#pragma unroll
        for (int i = 0; i < DQK * 32 / 256; ++i) {
            const int e = tid + i * 256;
            const int c = e >> 5, sl = e & 31, s = s0 + sl;
            float val = 0.f;
            if (s < a.Lk0) {
                if (c < a.dqk) val = kc0[c * a.k.cs + s];
                else if (c < dq) val = kq0[(c - a.dqk) * a.kp.cs + s];
            } else if (s < Lk) {
                if (c < a.dqk) val = kc1[c * a.k2.cs + (s - a.Lk0)];
                else if (c < dq) val = kq1[(c - a.dqk) * a.k2p.cs + (s - a.Lk0)];
            }
            ks[e] = val;
        }
#pragma unroll
        for (int i = 0; i < DV * 32 / 256; ++i) {
            const int e = tid + i * 256;
            const int c = e >> 5, sl = e & 31, s = s0 + sl;
            float val = 0.f;
            if (c < a.dv) {
                if (s < a.Lk0) val = vp0[c * a.v.cs + s];
                else if (s < Lk) val = vp1[c * a.v2.cs + (s - a.Lk0)];
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

extern "C" int lc_attention_fwd(const lc_cm_operand* q, const lc_cm_operand* q_pos,
                                const lc_cm_operand* k, const lc_cm_operand* k_pos,
                                const lc_cm_operand* v, const lc_cm_operand* k2,
                                const lc_cm_operand* k2_pos, const lc_cm_operand* v2, float* o,
                                int64_t o_bs, int64_t o_hs, int64_t o_cs, int B, int heads, int Lq,
                                int Lk0, int Lk1, int dqk, int dpos, int dv, float scale,
                                lc_stream_t s) {
    if (!q || !k || !v || !q->p || !k->p || !v->p || !o || B <= 0 || heads <= 0 || Lq <= 0 ||
        Lk0 <= 0 || Lk1 < 0 || dpos < 0)
        return LC_EINVAL;
    if (Lk1 > 0 && (!k2 || !v2 || !k2->p || !v2->p)) return LC_EINVAL;
    if (dpos > 0 && (!q_pos || !k_pos || !q_pos->p || !k_pos->p)) return LC_EINVAL;
    if (dpos > 0 && Lk1 > 0 && (!k2_pos || !k2_pos->p)) return LC_EINVAL;
    if (dqk <= 0 || dqk + dpos > 64 || dv <= 0 || dv > 64) return LC_EUNSUP;
    const lc_cm_operand none = {nullptr, 0, 0, 0};
    AttnArgs a;
    a.q = *q; a.qp = q_pos ? *q_pos : none; a.k = *k; a.kp = k_pos ? *k_pos : none; a.v = *v;
    a.k2 = k2 ? *k2 : none; a.k2p = k2_pos ? *k2_pos : none; a.v2 = v2 ? *v2 : none;
    a.o = o; a.o_bs = o_bs; a.o_hs = o_hs; a.o_cs = o_cs;
    a.heads = heads; a.Lq = Lq; a.Lk0 = Lk0; a.Lk1 = Lk1; a.dqk = dqk; a.dpos = dpos; a.dv = dv;
    a.qscale = scale * 1.4426950408889634f;
    dim3 grid((Lq + 127) / 128, B * heads);
    const int dq = (dqk + dpos) <= 32 ? 32 : 64, nd = dv <= 32 ? 1 : 2;
    if (dq == 32 && nd == 1) hipLaunchKernelGGL((attn_kernel<32, 1>), grid, dim3(256), 0, lc_s(s), a);
    else if (dq == 64 && nd == 1) hipLaunchKernelGGL((attn_kernel<64, 1>), grid, dim3(256), 0, lc_s(s), a);
    else if (dq == 32 && nd == 2) hipLaunchKernelGGL((attn_kernel<32, 2>), grid, dim3(256), 0, lc_s(s), a);
    else hipLaunchKernelGGL((attn_kernel<64, 2>), grid, dim3(256), 0, lc_s(s), a);
    return lc_launch_status();
}
