// Attention over PRE-SPLIT keys and values (round 6): the layout model's ObjectAwareCrossAttention at ds 4 / ds 8.
//
// Reference: ObjectAwareCrossAttention.forward (layout_unet_v1.py:416-532) -- image queries against image keys ++ the
// 13 layout keys, content and positional channels concatenated per head.  attn_h_kernel (attention.hip) stages a 32-key
// tile of K and V from fp32 channel-major operands and SPLITS it into fp16 hi / lo planes -- once per query block, i.e.
// 8 times per tile at 2048 queries (256 queries per block), ~75 of the ~250 VALU instructions a wave spends per tile
// (profiles/r04_pmc_attn.txt), and the waves that stage are the ones every barrier waits for.  Here K and V live in the
// form the MFMA fragments are read in -- "units" of 8 halves -- written ONCE per step (lc_attention_pack_units, or the qkv
// projection's own epilogue) or once per CONDITION (the positional half of every key and the layout keys are
// step-invariant), and a tile reaches LDS by LDS-DMA: no staging arithmetic, no staging registers, one barrier per tile.
//
// Unit image of one 32-key tile of one (sample, head) -- 768 units of 16 bytes, the LDS image and the global layout alike:
//     K hi [cb 8][key 32]   K lo [cb 8][key 32]            cb = 8 channels of a head's (content ++ positional) 64
//     V hi [step 2][half 2][c 32]   V lo [...]             unit (step, half, c) = V[c][keys 16 step + 4 half + {0..3, 8..11}]
// (the V order is the accumulator order of S^T = K^T Q, attention.hip).  kv = [B][heads][tiles][768] units; keys beyond
// Lk0 + Lk1 and channels a head does not have are zero.
// Same arithmetic and constants as attn_h_kernel<64, 1>: results are bit-identical (tests/test_hip_parity.py).
#include "conv_f16x2_common.h"      // lds_dma16
#include "attention_split.h"

namespace {

constexpr int TILE_UNITS = 768, K_LO = 256, V_HI = 512, V_LO = 640;

struct UArgs {
    lc_cm_operand q, qp;
    const half8* kv;
    float* o;
    long long o_bs, o_hs, o_cs;
    int heads, Lq, Lk, tiles, dqk, dpos, dv;
    float qscale;
};

__device__ __forceinline__ const float* head_ptr(const lc_cm_operand& x, int b, int h) {
    return x.p ? x.p + b * x.bs + h * x.hs : nullptr;
}

template <int NW>
__global__ __launch_bounds__(64 * NW) void attn_u_kernel(UArgs a) {
    constexpr int NST = 4;
    __shared__ half8 lds[2 * TILE_UNITS];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, kh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bh = blockIdx.y, b = bh / a.heads, h = bh - b * a.heads;
    const int t = blockIdx.x * (32 * NW) + wave * 32 + l31;
    const int dq = a.dqk + a.dpos;

    half8 qh[NST], ql[NST];
    {
        const float* qc = head_ptr(a.q, b, h);
        const float* qpos = head_ptr(a.qp, b, h);
        const float qs = a.qscale * QK_PRE;
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
    f32x16 oacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    // this head's tiles: tiles * 12 KB, moved by 12 wave-instructions of 1 KB per tile
    const half8* base = a.kv + (long long)bh * a.tiles * TILE_UNITS;
    __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (unsigned)a.tiles * TILE_UNITS * 16u, 0x00020000);
    auto issue = [&](int tile, half8* buf) {
#pragma unroll
        for (int k = 0; k < (12 + NW - 1) / NW; ++k) {
            const int j = wave + k * NW;                     // (wave-uniform: the LDS address of the piece is a scalar)
            if (j < 12)
                lcconv::lds_dma16(rs, (lcconv::lds_vptr)(buf + j * 64), (unsigned)(tile * TILE_UNITS + j * 64 + lane) * 16u, 0u);
        }
    };
    constexpr float S_UN = 1.0f / (QK_PRE * QK_PRE);
    issue(0, lds);
    for (int tile = 0; tile < a.tiles; ++tile) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();        // every wave's pieces of this tile have landed; the other buffer is consumed
        half8* cur = lds + (tile & 1) * TILE_UNITS;
        if (tile + 1 < a.tiles) issue(tile + 1, lds + ((tile + 1) & 1) * TILE_UNITS);
        const int s0 = tile * 32;
        // ---- S^T = K^T Q (scaled by QK_PRE^2) -------------------------------------------------
        f32x16 sacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
        half8 ah[NST], al[NST], avh[2], avl[2];
#pragma unroll
        for (int st = 0; st < NST; ++st) {                   // all fragment reads of the tile up front: the V reads land
            ah[st] = cur[(2 * st + kh) * 32 + l31];          // behind the softmax arithmetic
            al[st] = cur[K_LO + (2 * st + kh) * 32 + l31];
        }
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            avh[st] = cur[V_HI + (st * 2 + kh) * 32 + l31];
            avl[st] = cur[V_LO + (st * 2 + kh) * 32 + l31];
        }
#pragma unroll
        for (int st = 0; st < NST; ++st) {
            sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[st], qh[st], sacc, 0, 0, 0);
            sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[st], ql[st], sacc, 0, 0, 0);
            sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[st], qh[st], sacc, 0, 0, 0);
        }
        // ---- online softmax (base 2) ------------------------------------------------------------
        if (s0 + 32 > a.Lk) {    // ragged last tile (uniform branch)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = s0 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (key >= a.Lk) sacc[r] = -INFINITY;
            }
        }
        float mt = sacc[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mt = fmaxf(mt, sacc[r]);
        mt = fmaxf(mt, __shfl_xor(mt, 32, 64)) * S_UN;
        const float m_new = fmaxf(m_run, mt);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
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
            for (int r = 0; r < 16; ++r) oacc[r] *= alpha;
        }
        // ---- O^T += V P^T (scaled by P_PRE * V_PRE) -----------------------------------------------
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            float pv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) pv[j] = p[8 * st + j];
            half8 ph, pl;
            split8(pv, 1.0f, ph, pl);
            oacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(avl[st], ph, oacc, 0, 0, 0);
            oacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(avh[st], pl, oacc, 0, 0, 0);
            oacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(avh[st], ph, oacc, 0, 0, 0);
        }
    }
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / (l_tot * V_PRE);
    float* op = a.o + b * a.o_bs + h * a.o_hs;
    if (t < a.Lq) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int c = (r & 3) + 8 * (r >> 2) + 4 * kh;
            if (c < a.dv) lc_st(op + c * a.o_cs + t, oacc[r] * inv);
        }
    }
}

// ---- writers of the unit form ---------------------------------------------------------------------------------------
struct PackArgs {
    lc_cm_operand src;      // fp32 [B][heads * d][L], unit key stride
    half8* kv;
    int heads, L, tiles, d, cb0, key0;
};

// K: one thread per (key, unit of 8 channels): 8 loads (channel stride) -> split -> K hi / K lo of unit (cb0 + cb, key0 + key)
__global__ __launch_bounds__(256) void pack_k_kernel(PackArgs a) {
    const int key = blockIdx.x * 256 + threadIdx.x;
    const int cb = blockIdx.y, bh = blockIdx.z, b = bh / a.heads, h = bh - b * a.heads;
    if (key >= a.L) return;
    const float* p = a.src.p + b * a.src.bs + h * a.src.hs + (long long)(cb * 8) * a.src.cs + key;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (cb * 8 + j < a.d) ? p[j * a.src.cs] : 0.f;
    half8 hi, lo;
    split8(v, QK_PRE, hi, lo);
    const int kd = a.key0 + key;
    half8* tile = a.kv + ((long long)bh * a.tiles + (kd >> 5)) * TILE_UNITS;
    tile[(a.cb0 + cb) * 32 + (kd & 31)] = hi;
    tile[K_LO + (a.cb0 + cb) * 32 + (kd & 31)] = lo;
}

// V: one thread per (channel, octet of keys): keys 8 o + 0..3 -> (step o >> 1, half 0), 8 o + 4..7 -> (step o >> 1, half 1),
// both in the 4-element piece (o & 1) of their unit (attention.hip, store_tile)
__global__ __launch_bounds__(256) void pack_v_kernel(PackArgs a) {
    const int o = blockIdx.x * 256 + threadIdx.x;
    const int c = blockIdx.y, bh = blockIdx.z, b = bh / a.heads, h = bh - b * a.heads;
    if (o * 8 >= a.L) return;
    const float* p = a.src.p + b * a.src.bs + h * a.src.hs + (long long)c * a.src.cs + o * 8;
    float v[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) v[m] = (o * 8 + m < a.L) ? p[m] : 0.f;
    half8 hi, lo;
    split8(v, V_PRE, hi, lo);
    const int kd = a.key0 + o * 8;                      // key0 % 32 == 0: the octet stays inside one tile
    half8* tile = a.kv + ((long long)bh * a.tiles + (kd >> 5)) * TILE_UNITS;
    const int vo = (kd & 31) >> 3;
    const int u0 = ((vo >> 1) * 2 + 0) * 32 + c, u1 = u0 + 32;
    half4* h0 = reinterpret_cast<half4*>(&tile[V_HI + u0]) + (vo & 1);
    half4* h1 = reinterpret_cast<half4*>(&tile[V_HI + u1]) + (vo & 1);
    half4* l0 = reinterpret_cast<half4*>(&tile[V_LO + u0]) + (vo & 1);
    half4* l1 = reinterpret_cast<half4*>(&tile[V_LO + u1]) + (vo & 1);
    *h0 = __builtin_shufflevector(hi, hi, 0, 1, 2, 3);
    *h1 = __builtin_shufflevector(hi, hi, 4, 5, 6, 7);
    *l0 = __builtin_shufflevector(lo, lo, 0, 1, 2, 3);
    *l1 = __builtin_shufflevector(lo, lo, 4, 5, 6, 7);
}

}  // namespace

extern "C" int64_t lc_attention_units_elems(int B, int heads, int Lk0, int Lk1) {
    if (B <= 0 || heads <= 0 || Lk0 <= 0 || Lk0 % 32 || Lk1 < 0 || Lk1 > 32) return -1;
    return (int64_t)B * heads * (Lk0 / 32 + (Lk1 > 0)) * TILE_UNITS * 8;
}

// which = 0: keys (src = [B][heads * d][L] fp32; channels land in units cb0 .. cb0 + ceil(d / 8) - 1 of a head's 8),
// which = 1: values (d <= 32 channels).  key0 (a multiple of 32): the first destination key.
extern "C" int lc_attention_pack_units(const lc_cm_operand* src, void* kv, int B, int heads, int L, int Lk0, int Lk1, int d,
                                       int cb0, int key0, int which, lc_stream_t s) {
    if (!src || !src->p || !kv || B <= 0 || heads <= 0 || L <= 0 || d <= 0 || cb0 < 0 || key0 < 0) return LC_EINVAL;
    if (Lk0 <= 0 || Lk0 % 32 || Lk1 < 0 || Lk1 > 32 || key0 % 32 || key0 + L > Lk0 + 32 * (Lk1 > 0)) return LC_EUNSUP;
    PackArgs a;
    a.src = *src; a.kv = (half8*)kv; a.heads = heads; a.L = L; a.tiles = Lk0 / 32 + (Lk1 > 0); a.d = d; a.cb0 = cb0; a.key0 = key0;
    if (which == 0) {
        const int ncb = (d + 7) / 8;
        if (cb0 + ncb > 8) return LC_EUNSUP;
        hipLaunchKernelGGL(pack_k_kernel, dim3((L + 255) / 256, ncb, B * heads), dim3(256), 0, lc_s(s), a);
    } else {
        if (d > 32) return LC_EUNSUP;
        hipLaunchKernelGGL(pack_v_kernel, dim3(((L + 7) / 8 + 255) / 256, d, B * heads), dim3(256), 0, lc_s(s), a);
    }
    return lc_launch_status();
}

// q / q_pos: fp32 channel-major operands as lc_attention_f16x2_fwd takes them; kv: the unit form of ALL keys and values
// (lc_attention_units_elems halves).  dqk + dpos <= 64, both multiples of 8, dv <= 32, Lk0 % 32 == 0, Lk1 <= 32.
extern "C" int lc_attention_units_fwd(const lc_cm_operand* q, const lc_cm_operand* q_pos, const void* kv, float* o,
                                      int64_t o_bs, int64_t o_hs, int64_t o_cs, int B, int heads, int Lq, int Lk0, int Lk1,
                                      int dqk, int dpos, int dv, float scale, lc_stream_t s) {
    if (!q || !q->p || !kv || !o || B <= 0 || heads <= 0 || Lq <= 0 || Lk0 <= 0 || Lk1 < 0 || dpos < 0) return LC_EINVAL;
    if (dpos > 0 && (!q_pos || !q_pos->p)) return LC_EINVAL;
    if (dqk <= 0 || dqk % 8 || dpos % 8 || dqk + dpos > 64 || dv <= 0 || dv > 32 || Lk0 % 32 || Lk1 > 32) return LC_EUNSUP;
    const lc_cm_operand none = {nullptr, 0, 0, 0};
    UArgs a;
    a.q = *q; a.qp = q_pos ? *q_pos : none; a.kv = (const half8*)kv;
    a.o = o; a.o_bs = o_bs; a.o_hs = o_hs; a.o_cs = o_cs;
    a.heads = heads; a.Lq = Lq; a.Lk = Lk0 + Lk1; a.tiles = Lk0 / 32 + (Lk1 > 0); a.dqk = dqk; a.dpos = dpos; a.dv = dv;
    a.qscale = scale * 1.4426950408889634f;
    if ((long long)a.tiles * TILE_UNITS * 16 >= (1ll << 31)) return LC_EUNSUP;
    static const int w_env = [] { const char* e = getenv("LC_ATTN_U_WAVES"); return e ? atoi(e) : 0; }();
    const long long blocks8 = (long long)((Lq + 255) / 256) * B * heads;
    if (w_env == 8 || (w_env == 0 && blocks8 >= 128)) {
        hipLaunchKernelGGL(attn_u_kernel<8>, dim3((Lq + 255) / 256, B * heads), dim3(512), 0, lc_s(s), a);
    } else {
        hipLaunchKernelGGL(attn_u_kernel<4>, dim3((Lq + 127) / 128, B * heads), dim3(256), 0, lc_s(s), a);
    }
    return lc_launch_status();
}

LC_TOUCH_TU(attention_units, attn_u_kernel<8>)
