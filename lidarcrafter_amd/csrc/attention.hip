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
#include <cstdlib>

#include "common.h"

namespace {

struct AttnArgs {
    lc_cm_operand q, qp, k, kp, v, k2, k2p, v2;
    float* o;
    long long o_bs, o_hs, o_cs;
    int heads, Lq, Lk0, Lk1, dqk, dpos, dv;
    float qscale;  // scale * log2(e)
    float* lse;    // optional [B * heads, Lq]: log2-sum-exp of the scaled scores (base 2, qscale included) for the backward pass
    const unsigned* qkv_amax;   // training entry (attn_h_kernel<..., PRE = true>): bit patterns of max |q|, |k|, |v| (attention_pre.h)
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
    if (a.lse && t < a.Lq && kh == 0) a.lse[(long long)bh * a.Lq + t] = m_run + log2f(l_tot);
    if (t < a.Lq) {
#pragma unroll
        for (int i = 0; i < NDV; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (c < a.dv) lc_st(op + c * a.o_cs + t, oacc[i][r] * inv);
            }
    }
}


// ---------------------------------------------------------------------------------------------
// f16x2-split variant (default): the same flash loop with v_mfma_f32_32x32x16_f16 and the
// hi/lo operand split of conv_f16x2.hip -- every product a*b is evaluated as
// ah*bh + ah*bl + al*bh with fp16 hi/lo parts of the pre-scaled fp32 values and fp32
// accumulation (per-product error ~5e-7), 3 MFMAs of 32 clk per 16-deep k-step instead of 8
// fp32 MFMAs of 64 clk.  Operand mapping (A rows / B cols = lane&31, k = 8*(lane>>5)+0..7):
//   S^T: A = K unit [cb = 2*step + half][key] (8 channels of one key), B = the wave's Q
//        fragments (registers, split once).
//   O^T: B = the S^T accumulator registers 8*step..8*step+7 of the lane (keys kappa(r, half)),
//        A = V unit [step][half][c] holding V[c][kappa(8*step + 0..7, half)] -- the staging
//        pass writes each thread's 8 consecutive keys as two 8-byte pieces into that order.
// K/V of the next tile are prefetched into registers while the current tile is computed.
#include "attention_split.h"
#include "attention_pre.h"

// NW = waves per block (32 queries each).  4: the block of rounds 1-4.  8 (round 5): the K / V staging of a tile is
// per-BLOCK work (~75 of the ~250 VALU instructions a wave spends per 32-key tile, profiles/r04_pmc_attn.txt); 256
// queries per block halve it per query, and the roles are dealt so that no wave splits more than one operand: waves
// 0-3 stage K, waves 4-5 stage V, waves 6-7 only compute.  Bit-identical results; 2048 + 13 keys 241.8 -> 227.4 us,
// 512 + 13 keys 45.6 -> 40.0, 512 keys 29.1 -> 25.0 (profiles/r05_level0.txt section 1).
// PRE (round 6, training entry only): the three pre-scales are derived from the measured maxima of q, k, v instead of the
// constants -- the inference instantiations are unchanged.
template <int DQK, int NDV, int NW = 4, bool PRE = false>
__global__ __launch_bounds__(64 * NW) void attn_h_kernel(AttnArgs a) {
    constexpr int DV = NDV * 32;
    constexpr int NKU = DQK / 8 * 32;    // K units [cb][key]
    constexpr int NVU = 4 * DV;          // V units [step][half][c]
    constexpr int NST = DQK / 16;        // k-steps of S^T
    __shared__ half8 k_hi[NKU], k_lo[NKU], v_hi[NVU], v_lo[NVU];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, kh = lane >> 5;
    const int wave = tid >> 6;
    const int bh = blockIdx.y, b = bh / a.heads, h = bh - b * a.heads;
    const int t = blockIdx.x * (32 * NW) + wave * 32 + l31;
    const int Lk = a.Lk0 + a.Lk1;
    const int dq = a.dqk + a.dpos;

    half8 qh[NST], ql[NST];
    {
        const float* qc = head_ptr(a.q, b, h);
        const float* qpos = head_ptr(a.qp, b, h);
        const float qs = a.qscale * (PRE ? attn_pre_from(__uint_as_float(a.qkv_amax[0]) * fabsf(a.qscale)) : QK_PRE);
#pragma unroll
        for (int st = 0; st < NST; ++st) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int c = 16 * st + 8 * kh + j;
                float val = 0.f;
                if (t < a.Lq) {
                    if (c < a.dqk) val = qc[c * a.q.cs + t];
                    else if (c < dq) val = qpos[(c - a.dqk) * a.qp.cs + t];
                }
                v[j] = val;
            }
            split8(v, qs, qh[st], ql[st]);
        }
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

    // staging roles: K unit (cb, key) for tid < NKU; V item (channel c, key octet o) for tid < NVU.
    // dqk % 8 == 0 (checked by the launcher): the 8 channels of a K unit are all content or all
    // positional, so each thread keeps ONE base pointer + channel stride per key segment and the
    // K loop has no per-element address logic.  Threads without a role read a valid dummy
    // address and store zeros.
    static_assert(NW == 4 || (NW == 8 && NKU <= 256 && NVU <= 128), "8 waves: K roles in waves 0-3, V roles in waves 4-5");
    const int k_cb = tid >> 5, k_key = tid & 31;
    const int vt = NW == 8 ? tid - 256 : tid;            // V roles: threads 256 ... 256 + NVU of an 8-wave block
    const bool v_role = vt >= 0 && vt < NVU, k_role = tid < NKU;     // (both wave-uniform: NKU, NVU are multiples of 64)
    const int v_c = (v_role ? vt : 0) >> 2, v_o = vt & 3;
    const bool k_ok = k_role && k_cb * 8 < dq;
    const bool v_ok = v_role && v_c < a.dv;
    const float *kp0 = kc0, *kp1 = kc0, *vq0 = vp0, *vq1 = vp0;
    long long kcs0 = 0, kcs1 = 0;
    if (k_ok) {
        const int c0 = k_cb * 8;
        if (c0 < a.dqk) {
            kp0 = kc0 + c0 * a.k.cs; kcs0 = a.k.cs;
            if (a.Lk1 > 0) { kp1 = kc1 + c0 * a.k2.cs; kcs1 = a.k2.cs; }
        } else {
            kp0 = kq0 + (c0 - a.dqk) * a.kp.cs; kcs0 = a.kp.cs;
            if (a.Lk1 > 0) { kp1 = kq1 + (c0 - a.dqk) * a.k2p.cs; kcs1 = a.k2p.cs; }
        }
        if (a.Lk1 == 0) { kp1 = kp0; kcs1 = kcs0; }
    }
    if (v_ok) {
        vq0 = vp0 + v_c * a.v.cs;
        vq1 = a.Lk1 > 0 ? vp1 + v_c * a.v2.cs : vq0;
    }
    // 16-byte loads of V are legal when every (channel row, key octet) start is 16-byte aligned
    const bool v_vec = ((reinterpret_cast<uintptr_t>(vp0) & 15) == 0) && ((a.v.cs & 3) == 0);
    float kreg[8], vreg[8];
    auto load_tile = [&](int s0) {
        if (NW == 8 && !k_role && !v_role) return;       // waves 6-7 of an 8-wave block stage nothing
        if (s0 + 32 <= a.Lk0) {                 // whole tile inside segment 0 (uniform branch)
            if (NW == 4 || k_role) {
                const float* kp = kp0 + (s0 + k_key);
#pragma unroll
                for (int j = 0; j < 8; ++j) kreg[j] = kp[j * kcs0];
            }
            const float* vp = vq0 + (s0 + 8 * v_o);
            if (NW == 8 && !v_role) {
            } else if (v_vec) {
                const f32x4 x0 = *reinterpret_cast<const f32x4*>(vp);
                const f32x4 x1 = *reinterpret_cast<const f32x4*>(vp + 4);
                vreg[0] = x0.x; vreg[1] = x0.y; vreg[2] = x0.z; vreg[3] = x0.w;
                vreg[4] = x1.x; vreg[5] = x1.y; vreg[6] = x1.z; vreg[7] = x1.w;
            } else {
#pragma unroll
                for (int m = 0; m < 8; ++m) vreg[m] = vp[m];
            }
        } else {                                 // segment boundary and/or ragged end
            {
                const int s = s0 + k_key;
                const bool in0 = s < a.Lk0, in1 = !in0 && s < Lk;
                const float* kp = in0 ? kp0 + s : kp1 + (in1 ? s - a.Lk0 : 0);
                const long long cs = in0 ? kcs0 : kcs1;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float val = kp[j * cs];
                    kreg[j] = (in0 || in1) ? val : 0.f;
                }
            }
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                const int s = s0 + 8 * v_o + m;
                const bool in0 = s < a.Lk0, in1 = !in0 && s < Lk;
                const float val = in0 ? vq0[s] : vq1[in1 ? s - a.Lk0 : 0];
                vreg[m] = (in0 || in1) ? val : 0.f;
            }
        }
        if (!k_ok) {
#pragma unroll
            for (int j = 0; j < 8; ++j) kreg[j] = 0.f;
        }
        if (!v_ok) {
#pragma unroll
            for (int m = 0; m < 8; ++m) vreg[m] = 0.f;
        }
    };
    const float k_pre = PRE ? attn_pre_from(__uint_as_float(a.qkv_amax[1])) : QK_PRE;
    const float v_pre = PRE ? attn_pre_from(__uint_as_float(a.qkv_amax[2])) : V_PRE;
    auto store_tile = [&]() {
        if (tid < NKU) {
            half8 hi, lo;
            split8(kreg, k_pre, hi, lo);
            k_hi[tid] = hi;                       // unit index = cb*32 + key = tid
            k_lo[tid] = lo;
        }
        if (v_role) {
            half8 hi, lo;
            split8(vreg, v_pre, hi, lo);
            // keys 8o+0..3 -> (step o>>1, half 0), keys 8o+4..7 -> (step o>>1, half 1); both land
            // in the 4-element piece (o&1) of their unit
            const int u0 = ((v_o >> 1) * 2 + 0) * DV + v_c, u1 = u0 + DV;
            half4* h0 = reinterpret_cast<half4*>(&v_hi[u0]) + (v_o & 1);
            half4* h1 = reinterpret_cast<half4*>(&v_hi[u1]) + (v_o & 1);
            half4* l0 = reinterpret_cast<half4*>(&v_lo[u0]) + (v_o & 1);
            half4* l1 = reinterpret_cast<half4*>(&v_lo[u1]) + (v_o & 1);
            *h0 = __builtin_shufflevector(hi, hi, 0, 1, 2, 3);
            *h1 = __builtin_shufflevector(hi, hi, 4, 5, 6, 7);
            *l0 = __builtin_shufflevector(lo, lo, 0, 1, 2, 3);
            *l1 = __builtin_shufflevector(lo, lo, 4, 5, 6, 7);
        }
    };

    const float S_UN = PRE ? 1.0f / (attn_pre_from(__uint_as_float(a.qkv_amax[0]) * fabsf(a.qscale)) * k_pre)
                           : 1.0f / (QK_PRE * QK_PRE);
    load_tile(0);
    for (int s0 = 0; s0 < Lk; s0 += 32) {
        __syncthreads();   // previous tile fully consumed
        store_tile();
        __syncthreads();
        if (s0 + 32 < Lk) load_tile(s0 + 32);   // in flight while this tile is computed
        // ---- S^T = K^T Q (scaled by QK_PRE^2) -------------------------------------------------
        f32x16 sacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
#pragma unroll
        for (int st = 0; st < NST; ++st) {
            const half8 ah = k_hi[(2 * st + kh) * 32 + l31];
            const half8 al = k_lo[(2 * st + kh) * 32 + l31];
            sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, qh[st], sacc, 0, 0, 0);
            sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, ql[st], sacc, 0, 0, 0);
            sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, qh[st], sacc, 0, 0, 0);
        }
        // ---- online softmax (base 2) ------------------------------------------------------------
        if (s0 + 32 > Lk) {    // ragged last tile (uniform branch)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = s0 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (key >= Lk) sacc[r] = -INFINITY;
            }
        }
        float mt = sacc[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mt = fmaxf(mt, sacc[r]);
        mt = fmaxf(mt, __shfl_xor(mt, 32, 64)) * S_UN;
        const float m_new = fmaxf(m_run, mt);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        // p' = P_PRE * 2^(s - m): the pre-scale of the split is folded into the exponent, the
        // running sum carries the same factor and it cancels in the final division
        const float m_off = P_LOG2 - m_new;
        float psum = 0.f;
        float p[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            p[r] = __builtin_amdgcn_exp2f(fmaf(sacc[r], S_UN, m_off));
            psum += p[r];
        }
        l_run = l_run * alpha + psum;
        m_run = m_new;
        if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {   // uniform: some lane's max moved
#pragma unroll
            for (int i = 0; i < NDV; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[i][r] *= alpha;
        }
        // ---- O^T += V P^T (scaled by P_PRE * V_PRE) -----------------------------------------------
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            float pv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) pv[j] = p[8 * st + j];
            half8 ph, pl;
            split8(pv, 1.0f, ph, pl);
#pragma unroll
            for (int i = 0; i < NDV; ++i) {
                const half8 avh = v_hi[(st * 2 + kh) * DV + i * 32 + l31];
                const half8 avl = v_lo[(st * 2 + kh) * DV + i * 32 + l31];
                oacc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(avl, ph, oacc[i], 0, 0, 0);
                oacc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(avh, pl, oacc[i], 0, 0, 0);
                oacc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(avh, ph, oacc[i], 0, 0, 0);
            }
        }
    }
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / (l_tot * v_pre);
    float* op = a.o + b * a.o_bs + h * a.o_hs;
    if (a.lse && t < a.Lq && kh == 0) a.lse[(long long)bh * a.Lq + t] = m_run + log2f(l_tot) - P_LOG2;   // (l_run carries P_PRE)
    if (t < a.Lq) {
#pragma unroll
        for (int i = 0; i < NDV; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (c < a.dv) lc_st(op + c * a.o_cs + t, oacc[i][r] * inv);
            }
    }
}

}  // namespace

static int attention_launch(int split, float* lse, const unsigned* qkv_amax, const lc_cm_operand* q, const lc_cm_operand* q_pos,
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
    a.lse = lse;
    a.qkv_amax = qkv_amax;
    dim3 grid((Lq + 127) / 128, B * heads);
    const int dq = (dqk + dpos) <= 32 ? 32 : 64, nd = dv <= 32 ? 1 : 2;
    if (split && dqk % 8 == 0) {   // (a K unit = 8 channels of ONE operand; else the fp32 kernel)
        // 256 queries per block (8 waves) where d_v <= 32 and the grid still covers the chip (measured: every such layer
        // of the layout model and the 512-key self-attention gain 6-14 %; a 300-token batch-2 case loses 12 %)
        static const int w8_env = [] { const char* e = getenv("LC_ATTN_WAVES"); return e ? atoi(e) : 0; }();
        const long long blocks8 = (long long)((Lq + 255) / 256) * B * heads;
        if (qkv_amax) {            // training entry: measured pre-scales
            if (dq == 32 && nd == 1) hipLaunchKernelGGL((attn_h_kernel<32, 1, 4, true>), grid, dim3(256), 0, lc_s(s), a);
            else if (dq == 64 && nd == 1) hipLaunchKernelGGL((attn_h_kernel<64, 1, 4, true>), grid, dim3(256), 0, lc_s(s), a);
            else if (dq == 32 && nd == 2) hipLaunchKernelGGL((attn_h_kernel<32, 2, 4, true>), grid, dim3(256), 0, lc_s(s), a);
            else hipLaunchKernelGGL((attn_h_kernel<64, 2, 4, true>), grid, dim3(256), 0, lc_s(s), a);
            return lc_launch_status();
        }
        if (nd == 1 && (w8_env == 8 || (w8_env == 0 && blocks8 >= 128))) {
            dim3 grid8((Lq + 255) / 256, B * heads);
            if (dq == 32) hipLaunchKernelGGL((attn_h_kernel<32, 1, 8>), grid8, dim3(512), 0, lc_s(s), a);
            else hipLaunchKernelGGL((attn_h_kernel<64, 1, 8>), grid8, dim3(512), 0, lc_s(s), a);
            return lc_launch_status();
        }
        if (dq == 32 && nd == 1) hipLaunchKernelGGL((attn_h_kernel<32, 1>), grid, dim3(256), 0, lc_s(s), a);
        else if (dq == 64 && nd == 1) hipLaunchKernelGGL((attn_h_kernel<64, 1>), grid, dim3(256), 0, lc_s(s), a);
        else if (dq == 32 && nd == 2) hipLaunchKernelGGL((attn_h_kernel<32, 2>), grid, dim3(256), 0, lc_s(s), a);
        else hipLaunchKernelGGL((attn_h_kernel<64, 2>), grid, dim3(256), 0, lc_s(s), a);
        return lc_launch_status();
    }
    if (dq == 32 && nd == 1) hipLaunchKernelGGL((attn_kernel<32, 1>), grid, dim3(256), 0, lc_s(s), a);
    else if (dq == 64 && nd == 1) hipLaunchKernelGGL((attn_kernel<64, 1>), grid, dim3(256), 0, lc_s(s), a);
    else if (dq == 32 && nd == 2) hipLaunchKernelGGL((attn_kernel<32, 2>), grid, dim3(256), 0, lc_s(s), a);
    else hipLaunchKernelGGL((attn_kernel<64, 2>), grid, dim3(256), 0, lc_s(s), a);
    return lc_launch_status();
}

#define LC_ATTN_PARAMS                                                                            \
    const lc_cm_operand *q, const lc_cm_operand *q_pos, const lc_cm_operand *k,                  \
        const lc_cm_operand *k_pos, const lc_cm_operand *v, const lc_cm_operand *k2,             \
        const lc_cm_operand *k2_pos, const lc_cm_operand *v2, float *o, int64_t o_bs,            \
        int64_t o_hs, int64_t o_cs, int B, int heads, int Lq, int Lk0, int Lk1, int dqk,         \
        int dpos, int dv, float scale, lc_stream_t s
#define LC_ATTN_ARGS                                                                              \
    q, q_pos, k, k_pos, v, k2, k2_pos, v2, o, o_bs, o_hs, o_cs, B, heads, Lq, Lk0, Lk1, dqk,     \
        dpos, dv, scale, s

extern "C" int lc_attention_fwd(LC_ATTN_PARAMS) { return attention_launch(0, nullptr, nullptr, LC_ATTN_ARGS); }
extern "C" int lc_attention_f16x2_fwd(LC_ATTN_PARAMS) { return attention_launch(1, nullptr, nullptr, LC_ATTN_ARGS); }
// Training forward: the same kernels, plus the per-query log2-sum-exp the backward pass recomputes P from
// (qkv_amax: 3 words of scratch that stay alive until the backward pass has run -- the forward measures max |q|, |k|, |v| into
//  them and both passes derive their operand pre-scales from them, attention_pre.h; NULL: the constant pre-scale 16)
extern "C" int lc_attention_train_fwd(const float* q, const float* k, const float* v, float* o, float* lse, int BH,
                                      int Lq, int Lk, int dqk, int dv, float scale, int f16x2, float* qkv_amax,
                                      lc_stream_t s) {
    if (!q || !k || !v || !o || !lse || BH <= 0) return LC_EINVAL;
    unsigned* am = reinterpret_cast<unsigned*>(qkv_amax);
    if (am && f16x2) {
        if (hipMemsetAsync(am, 0, 3 * sizeof(unsigned), lc_s(s)) != hipSuccess) return lc_launch_status();
        const long long nq = (long long)BH * dqk * Lq, nk = (long long)BH * dqk * Lk, nv = (long long)BH * dv * Lk;
        const long long nmax = nq > nk ? (nq > nv ? nq : nv) : (nk > nv ? nk : nv);
        long long g = (nmax / 4 + 255) / 256;
        g = g < 1 ? 1 : (g > 1024 ? 1024 : g);
        hipLaunchKernelGGL(attn_amax3_kernel, dim3((unsigned)g, 3), dim3(256), 0, lc_s(s), q, nq, k, nk, v, nv, am);
    }
    const lc_cm_operand oq = {q, (int64_t)dqk * Lq, 0, Lq}, ok = {k, (int64_t)dqk * Lk, 0, Lk}, ov = {v, (int64_t)dv * Lk, 0, Lk};
    return attention_launch(f16x2, lse, f16x2 ? am : nullptr, &oq, nullptr, &ok, nullptr, &ov, nullptr, nullptr, nullptr, o,
                            (int64_t)dv * Lq, 0, Lq, BH, 1, Lq, Lk, 0, dqk, 0, dv, scale, s);
}

LC_TOUCH_TU(attention, attn_kernel<32, 1>)
