// Shared pieces of the f16x2 convolution kernels (conv_f16x2.hip, conv_f16x2_tall.hip): argument block, tile
// configuration, THE split rule, fused-GroupNorm row derivation, DPP sums, LDS-DMA and wait helpers.
// See the header comment of conv_f16x2.hip for the arithmetic.
#pragma once
#include <cstdlib>
#include <type_traits>

#include "common.h"

#ifndef LC_EPI_MODE
#define LC_EPI_MODE 2   // 0 plain stores, 2 write-through (sc1) stores (developer A/B)
#endif

namespace lcconv {


// Output store of the conv epilogues.  Mode 2 marks the store write-through (sc1): the lines do
// not stay dirty in L2, so the kernel-end release has less to flush.
__device__ __forceinline__ void epi_store(float* p, float v) {
#if LC_EPI_MODE == 2
    asm volatile("global_store_dword %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
#elif LC_EPI_MODE == 3
    asm volatile("global_store_dword %0, %1, off nt" :: "v"(p), "v"(v) : "memory");
#else
    *p = v;
#endif
}


typedef _Float16 half8 __attribute__((ext_vector_type(8)));

#ifndef LC_ABLATE
#define LC_ABLATE 0   // developer ablation switches (devtools/ablate_conv.sh); 0 in the product
                      // 1 no split/ds_write, 2 no global loads, 4 no fragment pipelining, 8 no stores,
                      // 16 no residual loads, 32 no MFMAs
#endif
#ifndef LC_PS_ABL
#define LC_PS_ABL 0   // pre-split kernel ablation: 1 no DMA in the K loop, 2 no MFMAs, 4 no x DMA, 8 no w DMA, 16 no epilogue stores / residual loads
#endif
#ifndef LC_EMIT_ABL
#define LC_EMIT_ABL 0   // pre-split kernel, statistics epilogue ablation: 1 no per-element sums, 2 no reductions / stores
#endif
#ifndef LC_DEF_AUX
#define LC_DEF_AUX 16   // cache policy of the deferred epilogue's output stores: 16 = sc1 (write-through), 0 = write-back, 2 = nt
#endif
#ifndef LC_DEF_ABL
#define LC_DEF_ABL 0    // developer ablation of the deferred epilogue (wrong results): 1 no output stores (values kept alive), 2 no residual loads
#endif
#ifndef LC_PIPE_ROWS
#define LC_PIPE_ROWS 1  // fused-GroupNorm rows are read ahead of the tap's fragment fetch (no lgkmcnt(0) drain)
#endif
#ifndef LC_DMA_FIRST
#define LC_DMA_FIRST 0   // pipelined kernel: 1 = the chunk's weight-DMA pieces are issued in front of its deferred-epilogue slots
#endif
#ifndef LC_TIMING
#define LC_TIMING 0     // developer build: s_memtime phase totals of the fp32-input pipelined kernel -> lc_dbg
#endif
#ifndef LC_PS_SCHED
#define LC_PS_SCHED 0   // pre-split kernel: 0 = fence per tap (reads of tap t+1, then MFMAs of tap t), 1 = 1:1 interleave
#endif
#ifndef LC_F16X2_TERMS
// which of the three products are accumulated: bit 0 wh*xh, bit 1 wl*xh, bit 2 wh*xl.  7 in the
// product library; 3 / 5 exist only to MEASURE what fewer passes cost in accuracy
// (devtools/passes_error.py, profiles/r02_passes_error.json); 1 = the single-product build
// liblidarcrafter_hip_p1.so that serves callers running under fp16 autocast (ops.conv_products).
#define LC_F16X2_TERMS 7
#endif
constexpr float X_PRESCALE_DEFAULT = 16.0f, W_PRESCALE_DEFAULT = 256.0f;

struct ConvArgsH {
    const float* x;
    const half8* wh;
    const half8* wl;
    const float* bias;
    const float* res;
    float* y;
    lc_conv_range* range;     // x pre-scale of this layer + the running max of what was staged
    const float* wmeta;       // {w_scale, 1 / w_scale} written by lc_pack_conv_weight_f16x2
    // pre-split input (conv_f16x2_ps_kernel): hi / lo fp16 planes written by the producer,
    // half8 units [B][2][xsp_c8][H][W]; batch stride in units.  x is unused then.
    const half8* xsp;
    long long xsp_bs;
    int xsp_c8;
    // split-K (pre-split kernel, small grids): blockIdx.z owns a contiguous range of the K chunks
    // and stores its raw partial sums to part[z][b][co][h][w]; lc_splitk_reduce finishes
    float* part;
    int ksplit;
    long long x_bs, res_bs, y_bs;
    int B, Ci, Co, H, W, Cib, Cop;
    int tiles_h, tiles_w;
    float out_scale;
    // optional fused GroupNorm(+AdaGN)+SiLU of the INPUT: rows (mu, A, B, 0) per (b, channel),
    // Cgn channels per sample (>= Ci rounded up to 16, zero rows beyond Ci); NULL = plain input
    const f32x4* gn;
    int Cgn, gn_silu;
    // ... or the statistics themselves (lc_groupnorm_stats partials): the block derives the rows of
    // its sample in its prologue, so no lc_groupnorm_coeffs launch sits between stats and conv
    lc_gn_stats_input gs;     // (os0 / os1 are host pointers: the kernels read seg[] instead)
    // ... or the octet statistics the input's producer(s) emitted (gs.partials == NULL):
    // seg[0] covers channels [0, seg[0].channels), seg[1] the rest
    struct OctSeg { const f32x4* p; int channels, slots, ush; } seg[2];   // ush: log2(channels per entry) = 3, 2, 1 or 0
    int tpb;   // pixel tiles per block (pipelined kernel): consecutive tiles of one sample
    int vert;  // 1: the block walks its tpb tiles down H (W-neighbours run concurrently), 0: along W
    int xcd;   // 1: blockIdx.x is remapped so that each XCD owns a contiguous range of tiles
    // optional GroupNorm statistics of the OUTPUT for the next GroupNorm (lc_groupnorm_apply_os):
    // per sample, channel octet (8 consecutive channels) and wave tile one entry
    // (pivot, n, sum(y - pivot), sum((y - pivot)^2));  ostats[(b*Co/8 + octet)*oslots + slot],
    // slot = (tile_row*tiles_w + tile_col)*WPX + wave_px.  Pipelined kernel only.
    // ounit = 2: one entry per channel PAIR instead (ostats[(b*Co/2 + pair)*oslots + slot]) -- for a consumer
    // GroupNorm with 2 / 4 / 6 channels per group (GroupNorm32 at 64 ... 192 channels); deferred epilogue only.
    f32x4* ostats;
    int oslots, ounit;
};

constexpr int GN_MAX_C = 1024;   // LDS table of fused GroupNorm rows: 16 KB

// (mu, A, Bc, 0) of channel c of sample b from the statistics partials -- the arithmetic of
// gn_coeffs_kernel (norm.hip), so both routes give bit-identical rows
__device__ __forceinline__ f32x4 gn_row_from_stats(const lc_gn_stats_input& gs, const float* xb,
                                                   int b, int c, int C, long long HW) {
    f32x4 r = {0.f, 0.f, 0.f, 0.f};
    if (c >= C) return r;
    const int cpg = C / gs.G, g = c / cpg;
    const double* pp = gs.partials + ((long long)b * gs.G + g) * gs.nch * 2;
    double s_ = 0.0, q_ = 0.0;
    typedef double f64x2 __attribute__((ext_vector_type(2)));
    const f64x2* pv = reinterpret_cast<const f64x2*>(pp);
    for (int k0 = 0; k0 < gs.nch; k0 += 8) {                    // 8 loads in flight, same add order
        f64x2 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = k0 + k < gs.nch ? pv[k0 + k] : f64x2{0.0, 0.0};
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (k0 + k < gs.nch) { s_ += v[k].x; q_ += v[k].y; }
    }
    const double n = (double)cpg * (double)HW;
    const double dm = s_ / n;
    double var = q_ / n - dm * dm;
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)gs.eps));
    const float mu = (float)((double)xb[(long long)g * cpg * HW] + dm);
    const float ga = gs.gamma ? gs.gamma[c] : 1.0f, be = gs.beta ? gs.beta[c] : 0.0f;
    const float sc = gs.scale ? 1.0f + gs.scale[b * gs.ss_bs + c] : 1.0f;
    const float sh = gs.shift ? gs.shift[b * gs.ss_bs + c] : 0.0f;
    r.x = mu; r.y = rstd * ga * sc; r.z = be * sc + sh;
    return r;
}

// The same rows from the PRODUCER's octet statistics (the fold of gn_apply_os_kernel, norm.hip):
// wave w folds the entries of groups w, w + nwaves, ... in fp64 around the group's first pivot,
// then every thread builds the rows of its channels.  All NT threads of the block call this.
constexpr int GN_MAX_G = 128;
template <int NT>
__device__ __forceinline__ void gn_rows_from_ostats(const ConvArgsH& a, int b, int tid, f32x4* ctab,
                                                    float2* gtab) {
    const int G = a.gs.G, cpg = a.Ci / G, lane = tid & 63;
    // the affine inputs of this thread's first channel do not depend on the fold: request them first
    // (as loads behind the fold's barrier they were one more exposed miss in every block's prologue)
    float ga0 = 1.0f, be0 = 0.0f, sc0 = 1.0f, sh0 = 0.0f;
    if (tid < a.Ci) {
        if (a.gs.gamma) ga0 = a.gs.gamma[tid];
        if (a.gs.beta) be0 = a.gs.beta[tid];
        if (a.gs.scale) sc0 = 1.0f + a.gs.scale[b * a.gs.ss_bs + tid];
        if (a.gs.shift) sh0 = a.gs.shift[b * a.gs.ss_bs + tid];
    }
    for (int g = tid >> 6; g < G; g += NT / 64) {
        const int cg0 = g * cpg;
        const bool s1 = cg0 >= a.seg[0].channels;
        const int slots = s1 ? a.seg[1].slots : a.seg[0].slots;
        const int ush = s1 ? a.seg[1].ush : a.seg[0].ush;
        const f32x4* e =
            s1 ? a.seg[1].p + ((long long)b * (a.seg[1].channels >> ush) + ((cg0 - a.seg[0].channels) >> ush)) * slots
               : a.seg[0].p + ((long long)b * (a.seg[0].channels >> ush) + (cg0 >> ush)) * slots;
        const int n_ent = (cpg >> ush) * slots;
        const double P0 = (double)e[0].x;
        double N = 0.0, S = 0.0, Q = 0.0;
        for (int base = lane; base < n_ent; base += 64 * 16) {   // 16 loads in flight per lane
            f32x4 v[16];
#pragma unroll
            for (int k = 0; k < 16; ++k)
                v[k] = base + 64 * k < n_ent ? e[base + 64 * k] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 16; ++k) {                      // an absent entry (n = 0) adds nothing
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
        const double m = N > 0.0 ? S / N : 0.0;
        double var = N > 0.0 ? Q / N - m * m : 0.0;
        if (var < 0.0) var = 0.0;
        if (lane == 0) gtab[g] = float2{(float)(P0 + m), (float)(1.0 / sqrt(var + (double)a.gs.eps))};
    }
    __syncthreads();
    for (int c = tid; c < a.Cgn; c += NT) {
        f32x4 r = {0.f, 0.f, 0.f, 0.f};
        if (c < a.Ci) {
            const float2 ms = gtab[c / cpg];
            float ga = ga0, be = be0, sc = sc0, sh = sh0;
            if (c != tid) {
                ga = a.gs.gamma ? a.gs.gamma[c] : 1.0f; be = a.gs.beta ? a.gs.beta[c] : 0.0f;
                sc = a.gs.scale ? 1.0f + a.gs.scale[b * a.gs.ss_bs + c] : 1.0f;
                sh = a.gs.shift ? a.gs.shift[b * a.gs.ss_bs + c] : 0.0f;
            }
            r.x = ms.x; r.y = ms.y * ga * sc; r.z = be * sc + sh;
        }
        ctab[c] = r;
    }
}

__device__ __forceinline__ float gn_act(float x, const f32x4 c, int silu) {
    float y = fmaf(x - c.x, c.y, c.z);
    if (silu) y = y * __builtin_amdgcn_rcpf(1.0f + __expf(-y));
    return y;
}

template <int WCO, int WPX, int TCO, int TPX, int TH, int TW, int KS>
struct HCfg {
    static constexpr int WCO_ = WCO, WPX_ = WPX, TCO_ = TCO, TPX_ = TPX, TH_ = TH, TW_ = TW;
    static constexpr int CB = 2;                        // 8-channel blocks per K chunk (16 ch)
    static constexpr int HALO = KS / 2;
    static constexpr int NTAP = KS * KS;
    static constexpr int BN = WCO * TCO * 32;
    static constexpr int NPT = WPX * TPX;
    static constexpr int TPR = TW / 32;
    static constexpr int XR = TH + 2 * HALO;
    static constexpr int XW = TW + 2 * HALO;
    static constexpr int NT = 64 * WCO * WPX;           // threads per block (4 or 8 waves)
    static constexpr int XU = CB * XR * XW;             // 16-byte x units per plane per chunk
    static constexpr int NXU = (XU + NT - 1) / NT;
    static constexpr int WU = NTAP * CB * BN;           // 16-byte weight units per plane
    static constexpr int NWU = (WU + NT - 1) / NT;
    static_assert(NPT * 32 == TH * TW, "tile shape");
    static_assert(WCO * WPX == 4 || WCO * WPX == 8, "4 or 8 waves");
};

// THE split rule of both kernels.  s = v * xs;  hi = s truncated to 11 significant bits (exact in
// fp32) and packed round-toward-zero -- exact below 65504, saturating (never inf) above;
// lo = fp16(s - hi) (RTNE; subnormal below 2^-14, zero below 2^-25).  `am` accumulates max |s|.
typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
typedef float f2_t __attribute__((ext_vector_type(2)));
#ifndef LC_SPLIT_ABL
#define LC_SPLIT_ABL 0   // developer ablation: 1 no amax accumulation, 2 round-to-nearest split
#endif
template <bool NOPACK>
__device__ __forceinline__ void split_pair(float v0, float v1, float xs, h2_t& ph, h2_t& pl, float& am) {
    float s0 = v0 * xs, s1 = v1 * xs;
    // NOPACK: keep hipcc's SLP vectoriser from fusing the two multiplies into v_pk_mul_f32 -- the
    // packed form needs an aligned register pair, and in the 2-blocks/CU kernel the v_mov_b64 that
    // builds it (with its s_waitcnt) lands between the global loads of the next chunk and
    // serialises them (1x1 convs 26 -> 35 us, 33 -> 61 us, measured r02g)
    if (NOPACK) asm volatile("" : "+v"(s0), "+v"(s1));
    if (!(LC_SPLIT_ABL & 1))
        am = __builtin_fmaxf(am, __builtin_fmaxf(__builtin_fabsf(s0), __builtin_fabsf(s1)));   // v_max3_f32
    if (LC_SPLIT_ABL & 2) {
        const _Float16 a0 = (_Float16)s0, a1 = (_Float16)s1;
        ph.x = a0; ph.y = a1;
        pl.x = (_Float16)(s0 - (float)a0); pl.y = (_Float16)(s1 - (float)a1);
        return;
    }
    const float h0 = __uint_as_float(__float_as_uint(s0) & 0xFFFFE000u);
    const float h1 = __uint_as_float(__float_as_uint(s1) & 0xFFFFE000u);
    ph = __builtin_bit_cast(h2_t, __builtin_amdgcn_cvt_pkrtz(h0, h1));
    f2_t r; r.x = s0 - h0; r.y = s1 - h1;
    pl = __builtin_convertvector(r, h2_t);
}
// the same rule for values that already carry the scale (fused GroupNorm rows are pre-multiplied)
__device__ __forceinline__ void split_pair_scaled(float s0, float s1, h2_t& ph, h2_t& pl, float& am) {
    am = __builtin_fmaxf(am, __builtin_fmaxf(__builtin_fabsf(s0), __builtin_fabsf(s1)));       // v_max3_f32
    const float h0 = __uint_as_float(__float_as_uint(s0) & 0xFFFFE000u);
    const float h1 = __uint_as_float(__float_as_uint(s1) & 0xFFFFE000u);
    ph = __builtin_bit_cast(h2_t, __builtin_amdgcn_cvt_pkrtz(h0, h1));
    f2_t r; r.x = s0 - h0; r.y = s1 - h1;
    pl = __builtin_convertvector(r, h2_t);
}
template <bool NOPACK = false>
__device__ __forceinline__ void split8(const float (&v)[8], float xs, half8& hi, half8& lo, float& am) {
#pragma unroll
    for (int k = 0; k < 8; k += 2) {
        h2_t ph, pl;
        split_pair<NOPACK>(v[k], v[k + 1], xs, ph, pl, am);
        hi[k] = ph.x; hi[k + 1] = ph.y; lo[k] = pl.x; lo[k + 1] = pl.y;
    }
}
// publish the wave's max |x * x_scale| (am >= 0: unsigned order == float order); a cached read
// keeps all but the first few blocks of a launch off the atomic
#ifndef LC_RANGE_ABL
#define LC_RANGE_ABL 0   // developer ablation: 1 no amax publish, 2 constant scales (no device loads)
#endif
__device__ __forceinline__ void publish_amax(lc_conv_range* rg, float am, float seen) {
    if (LC_RANGE_ABL & 1) return;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) am = __builtin_fmaxf(am, __shfl_xor(am, o, 64));
    if ((threadIdx.x & 63) == 0 && am > seen)
        atomicMax(reinterpret_cast<unsigned*>(&rg->amax_scaled), __float_as_uint(am));
}

typedef __attribute__((address_space(3))) void* lds_vptr;
typedef const __attribute__((address_space(1))) void* gbl_vptr;

__device__ __forceinline__ void split_store(const float (&v)[8], float xs, half8* dst_hi,
                                            half8* dst_lo, float& am) {
    half8 hi, lo;
    split8(v, xs, hi, lo, am);
    *dst_hi = hi;
    *dst_lo = lo;
}

// 64-lane sum with DPP adds (VALU rate, no LDS): row_shr 1,2,4,8, then row_bcast:15 into rows 1,3
// and row_bcast:31 into rows 2,3 -- lane 63 ends up with the total.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v),
                                                                     CTRL, ROW_MASK, 0xF, true));
}
__device__ __forceinline__ float wave_sum_to_lane63(float v) {
    v = dpp_add<0x111, 0xF>(v);
    v = dpp_add<0x112, 0xF>(v);
    v = dpp_add<0x114, 0xF>(v);
    v = dpp_add<0x118, 0xF>(v);
    v = dpp_add<0x142, 0xA>(v);
    v = dpp_add<0x143, 0xC>(v);
    return v;
}

// the same without the last step: lanes 31 / 63 hold the sums of lanes 0-31 / 32-63
__device__ __forceinline__ float half_sum_to_lane31_63(float v) {
    v = dpp_add<0x111, 0xF>(v);
    v = dpp_add<0x112, 0xF>(v);
    v = dpp_add<0x114, 0xF>(v);
    v = dpp_add<0x118, 0xF>(v);
    v = dpp_add<0x142, 0xA>(v);
    return v;
}

// One statistics entry (pivot, n, s, q) as ONE 32-bit buffer store: the sums sit in the reducing lane R (63, or 31 and 63
// for two half-wave entries), pivot and count are wave-uniform; lanes R-3 .. R carry the fields 0 .. 3 (their voffset =
// entry byte offset + 4 * (lane & 3); every other lane: an out-of-range offset).  No wide store data, no waits (DefEpi's
// store_entry_lanes, for epilogues outside that struct).
__device__ __forceinline__ void store_entry_4lanes(__amdgpu_buffer_rsrc_t rs, float p_, float n_, float s_, float q_,
                                                   unsigned voffset) {
    const float s1 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s_), 0x101, 0xF, 0xF, true));
    const int f = (int)(threadIdx.x & 3);
    const unsigned k0 = (unsigned)((f - 1) >> 31), k1 = (unsigned)(((f ^ 1) - 1) >> 31);
    const unsigned k2 = (unsigned)(((f ^ 2) - 1) >> 31), k3 = (unsigned)(((f ^ 3) - 1) >> 31);
    const unsigned vu = (__float_as_uint(p_) & k0) | (__float_as_uint(n_) & k1) | (__float_as_uint(s1) & k2) |
                        (__float_as_uint(q_) & k3);
    __builtin_amdgcn_raw_buffer_store_b32(vu, rs, voffset, 0, 0);
}

// one LDS-DMA wave-instruction: 64 lanes x 16 bytes, global (descriptor + per-lane voffset + uniform
// soffset; out of range -> zeros) -> LDS at dst + 16 * lane.  (The builtin exists in the device
// pass only; the host pass needs just the kernel's stub.)
__device__ __forceinline__ void lds_dma16(__amdgpu_buffer_rsrc_t rs, lds_vptr dst, unsigned voff,
                                          unsigned soff) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, dst, 16, voff, soff, 0, 0);
#endif
}

// s_waitcnt vmcnt(n) for a count that is a compile-time constant only after inlining
__device__ __forceinline__ void wait_vmcnt(int n) {
#define LC_WV(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
    switch (n) {
        LC_WV(0) LC_WV(1) LC_WV(2) LC_WV(3) LC_WV(4) LC_WV(5) LC_WV(6) LC_WV(7) LC_WV(8) LC_WV(9) LC_WV(10)
        LC_WV(11) LC_WV(12) LC_WV(13) LC_WV(14) LC_WV(15) LC_WV(16) LC_WV(17) LC_WV(18) LC_WV(19) LC_WV(20)
        LC_WV(21) LC_WV(22) LC_WV(23) LC_WV(24) LC_WV(25) LC_WV(26) LC_WV(27) LC_WV(28) LC_WV(29) LC_WV(30)
        LC_WV(31) LC_WV(32) LC_WV(33) LC_WV(34) LC_WV(35) LC_WV(36) LC_WV(37) LC_WV(38) LC_WV(39) LC_WV(40)
        LC_WV(41) LC_WV(42) LC_WV(43) LC_WV(44) LC_WV(45) LC_WV(46) LC_WV(47) LC_WV(48) LC_WV(49) LC_WV(50)
        LC_WV(51) LC_WV(52) LC_WV(53) LC_WV(54) LC_WV(55) LC_WV(56) LC_WV(57) LC_WV(58) LC_WV(59) LC_WV(60)
        LC_WV(61) LC_WV(62) LC_WV(63)
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
#undef LC_WV
}

// ---------------------------------------------------------------------------------------------
// DEFERRED EPILOGUE (round 3).  s_memtime phase totals of the level-0 launches (devtools/
// conv_phases.py, profiles/r03_conv_phases.txt): a wave spent 36 % of its lifetime ISSUING the 32
// write-through stores of a tile's epilogue -- ~375 cycles per `global_store_dword`: all 256 CUs
// reach their epilogue together and the burst (16.8 MB per tile round, + the residual reads) runs
// at the ~3.3 TB/s the fabric takes writes at, with every matrix pipe idle.  The bytes have to be
// written; what can change is WHEN.  A finished tile's raw accumulators are therefore parked in a
// second register set and finalised -- x out_unscale, + residual, x out_scale, GroupNorm statistics,
// one store -- VPT values per tap inside the MFMA stream of the NEXT tile's first DCH chunks, so
// the write traffic of tile t runs under the MFMAs of tile t+1; only the last tile of a block
// drains in the open.  The residual is fetched just in time (LAG slots ahead of its use, RING
// registers instead of a 32-register tile), the bias enters through the accumulator's initial
// value (bias / out_unscale: the scale is a power of two, exact) -- both were needed to make room
// for the parked accumulators at 256 registers per wave.
// Loads and stores are raw buffer operations on descriptors of sample b (residual: num_records 0
// when there is none -> zeros): out-of-image pixels, ragged channel tails and "no previous tile"
// are an out-of-range offset, so the per-value code has no branch and stays in the MFMAs' basic
// block.
template <class C, int EMIT>   // 0: no statistics, 1: octet entries, 2: pair entries (compile time: a runtime
struct DefEpi {                // branch here split the MFMAs' basic block -- level-0 launch 84 -> 100 us)
    static constexpr int TCO = C::TCO_, TPX = C::TPX_, NTAP = C::NTAP;
    static constexpr int NV = TCO * TPX * 16;              // values per thread and tile
    static constexpr int DCH = 4;                          // chunks of the next tile carrying deferred work
    static constexpr int NSLOT = DCH * NTAP;
    static constexpr int VPT = (NV + NSLOT - 1) / NSLOT;   // values per tap slot
    static constexpr int NUSED = (NV + VPT - 1) / VPT;     // slots that carry values
#ifndef LC_DEF_LAG
#define LC_DEF_LAG 10
#endif
    // residual loads run LAG slots ahead of their use (one slot per chunk for a 1x1 conv: many values per slot)
    static constexpr int LAG = NTAP == 1 ? 1 : LC_DEF_LAG, RING = (LAG + 1) * VPT;
    static constexpr unsigned OOB = 0x80000000u;
    // Value order: k = ((i * 4 + m) * TPX + j) * 4 + q  <->  accumulator (i, j, r = 4 m + q): the 4 * TPX
    // values of one channel OCTET (m; registers 4m .. 4m+3 of both lane halves) are consecutive, so
    // only one statistics triple is alive at a time and each octet's reduction + entry store follows
    // its last value (12 DPP adds every 4 * TPX values instead of 96 at the end of the tile).
    static constexpr int OCTV = 4 * TPX;

    f32x16 accp[TCO][TPX];
    float rq[RING];
    float st_p, st_s, st_q;    // the running octet: pivot, sum (v - p), sum (v - p)^2 of channels q = 0, 1 of each quad
    float st_s2, st_q2;        // pair entries: ... and of channels q = 2, 3 (unused for octet entries)
    static constexpr bool pairs = EMIT == 2;   // entries per channel pair (ConvArgsH::ounit == 2)
    unsigned voff[TPX];        // byte offset of (channel co_wave, pixel j) in the sample; OOB = nothing to do
    __amdgpu_buffer_rsrc_t rs_y, rs_r, rs_o;
    float out_unscale, out_scale;
    unsigned HW4;              // bytes per channel plane
    int co_wave, Co;
    float nv8;                 // (channels per entry) x valid pixels of the parked tile (entry field)
    unsigned ent_off;          // byte offset of the parked tile's entry of this lane's first unit (octets: lane 63;
                               // pairs: lanes 31 / 63 = channel quads 0 / 1 of the octet), or OOB
    unsigned oct_stride;       // bytes between the entries of consecutive units (octets or pairs)

    __device__ __forceinline__ void init(float* yb, const float* rb, int Co_, int HW, float unscale,
                                         float oscale, int co_wave_, f32x4* ostats_b, int oslots, int ounit) {
        const unsigned bytes = (unsigned)Co_ * (unsigned)HW * 4u;
        rs_y = __builtin_amdgcn_make_buffer_rsrc((void*)yb, 0, bytes, 0x00020000);
        rs_r = __builtin_amdgcn_make_buffer_rsrc((void*)rb, 0, rb ? bytes : 0u, 0x00020000);
        rs_o = __builtin_amdgcn_make_buffer_rsrc((void*)ostats_b, 0,
                                                 ostats_b ? (unsigned)(Co_ >> (pairs ? 1 : 3)) * (unsigned)oslots * 16u : 0u,
                                                 0x00020000);
        out_unscale = unscale; out_scale = oscale; HW4 = (unsigned)HW * 4u;
        co_wave = co_wave_; Co = Co_;
        oct_stride = (unsigned)oslots * 16u;
        nv8 = 0.f; ent_off = OOB;
        st_p = st_s = st_q = st_s2 = st_q2 = 0.f;
#pragma unroll
        for (int j = 0; j < TPX; ++j) voff[j] = OOB;
#pragma unroll
        for (int i = 0; i < TCO; ++i)
#pragma unroll
            for (int j = 0; j < TPX; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) accp[i][j][r] = 0.0f;
#pragma unroll
        for (int q = 0; q < RING; ++q) rq[q] = 0.0f;
    }
    static __device__ __forceinline__ int i_of(int k) { return k / (16 * TPX); }
    static __device__ __forceinline__ int m_of(int k) { return (k / OCTV) & 3; }
    static __device__ __forceinline__ int j_of(int k) { return (k >> 2) % TPX; }
    static __device__ __forceinline__ int cor_of(int k) {   // channel of value k relative to co_wave
        return i_of(k) * 32 + (k & 3) + 8 * m_of(k);
    }
    __device__ __forceinline__ unsigned off_of(int k) const {
        return (co_wave + cor_of(k) < Co) ? voff[j_of(k)] : OOB;
    }
    __device__ __forceinline__ void issue_res(int k) {
        if (LC_DEF_ABL & 2) return;
        rq[k % RING] = __builtin_bit_cast(
            float, __builtin_amdgcn_raw_buffer_load_b32(rs_r, off_of(k), (unsigned)cor_of(k) * HW4, 0));
    }
    __device__ __forceinline__ void finalize(int k) { finalize_with(k, rq[k % RING]); }
    __device__ __forceinline__ void finalize_with(int k, float res) {
        const int i = i_of(k), m = m_of(k), j = j_of(k), r = 4 * m + (k & 3);
        const float v = fmaf(accp[i][j][r], out_unscale, res) * out_scale;
#if LC_DEF_ABL & 1
        asm volatile("" ::"v"(v));
#else
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs_y, off_of(k),
                                              (unsigned)cor_of(k) * HW4, LC_DEF_AUX);
#endif
        if constexpr (EMIT) {
            const bool pok = voff[j] != OOB;
            if (k % OCTV == 0) {
                st_p = __builtin_amdgcn_readlane(pok ? v : 0.0f, 0);
                st_s = 0.f; st_q = 0.f; st_s2 = 0.f; st_q2 = 0.f;
            }
            const float d = pok ? v - st_p : 0.0f;
            if (pairs && (k & 2)) { st_s2 += d; st_q2 = fmaf(d, d, st_q2); }   // (k & 3 = channel inside the lane's quad)
            else { st_s += d; st_q = fmaf(d, d, st_q); }
            if (k % OCTV == OCTV - 1) {                  // the octet is complete
                const int oct = i * 4 + m;               // octet index inside this wave's channel rows
                const bool ok = ent_off != OOB && co_wave + i * 32 + 8 * m < Co;   // (co_wave carries 4 * kh <= 4)
                // The entry goes out as ONE 32-bit store in which the four lanes below the reducing lane carry one
                // field each (store_entry_lanes).  Rounds 1-3 wrote it as one 128-bit buffer store with an SGPR
                // soffset; ~3 entries in 10^7 then arrived with a foreign upper dword (a VALU write of the data
                // registers one or two instructions behind the store: LLVM exempts MUBUF stores with an SGPR soffset
                // from the ISA's ">64-bit store data" wait state, and under ~27 stores in flight per wave that
                // exemption does not hold on gfx950).  Measured in round 4 (devtools/entry_stress.py,
                // profiles/r04_entry_store.txt): 128-bit + SGPR soffset 17 bad entries in 5e7, 32-bit stores 0 in
                // 1e8.  32-bit stores fetch their data at issue -- the form every output value of this epilogue
                // has always used.
                const unsigned vo = ok ? ent_off : OOB;
                if constexpr (!pairs) {                  // one entry from lane 63
                    store_entry_lanes(st_p, nv8, wave_sum_to_lane63(st_s), wave_sum_to_lane63(st_q), vo,
                                  (unsigned)oct * oct_stride);
                } else {                                 // pairs (8m + 4kh + 0,1) and (+ 2,3) from lanes 31 / 63
                    store_entry_lanes(st_p, nv8, half_sum_to_lane31_63(st_s), half_sum_to_lane31_63(st_q), vo,
                                  (unsigned)(4 * oct) * oct_stride);
                    store_entry_lanes(st_p, nv8, half_sum_to_lane31_63(st_s2), half_sum_to_lane31_63(st_q2), vo,
                                  (unsigned)(4 * oct + 1) * oct_stride);
                }
            }
        }
    }
    // (pivot, n, s, q): the sums arrive in the reducing lane R (63, or 31 / 63 for pair entries), pivot and count are
    // wave-uniform; lanes R-3 .. R store the fields 0 .. 3 (ent_off of those lanes = entry + 4 * field, OOB elsewhere):
    // one store instruction per entry, no wide store data.  (First form of round 4: four 32-bit stores from lane R with
    // four different cache-policy bits to keep the load/store optimizer from re-merging them -- the sc0 sc1 one cost
    // the level-0 launch ~100 us.)
    __device__ __forceinline__ void store_entry_lanes(float p_, float n_, float s_, float q_, unsigned voffset,
                                                      unsigned soffset) {
        // row_shl:1 -- lane R-1 reads lane R's sum
        const float s1 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s_), 0x101, 0xF, 0xF, true));
        // field select by lane & 3 with masks, not selects: hipcc turned the nested ?: (one arm is a DPP result) into
        // exec-mask BRANCHES -- four basic-block splits per tile inside the MFMA stream (ISA of rounds 3-4)
        const int f = (int)(threadIdx.x & 3);
        const unsigned k0 = (unsigned)((f - 1) >> 31), k1 = (unsigned)(((f ^ 1) - 1) >> 31);
        const unsigned k2 = (unsigned)(((f ^ 2) - 1) >> 31), k3 = (unsigned)(((f ^ 3) - 1) >> 31);
        const unsigned vu = (__float_as_uint(p_) & k0) | (__float_as_uint(n_) & k1) | (__float_as_uint(s1) & k2) |
                            (__float_as_uint(q_) & k3);
        __builtin_amdgcn_raw_buffer_store_b32(vu, rs_o, voffset, soffset, 0);
    }
    // slot s of the deferred stream (s static after unrolling)
    __device__ __forceinline__ void slot(int s) {
#pragma unroll
        for (int u = 0; u < VPT; ++u)
            if ((s + LAG) * VPT + u < NV) issue_res((s + LAG) * VPT + u);
#pragma unroll
        for (int u = 0; u < VPT; ++u)
            if (s * VPT + u < NV) finalize(s * VPT + u);
    }
    // VMEM operations slot s issues (for the manual vmcnt in front of the chunk barrier)
    static __device__ __forceinline__ int ops_of(int s) {
        int n = 0;
#pragma unroll
        for (int u = 0; u < VPT; ++u) {
            n += ((s + LAG) * VPT + u < NV && !(LC_DEF_ABL & 2)) ? 1 : 0;
            const int k = s * VPT + u;
            n += (k < NV && !(LC_DEF_ABL & 1)) ? 1 : 0;
            n += (EMIT && k < NV && k % OCTV == OCTV - 1) ? (EMIT == 2 ? 2 : 1) : 0;
        }
        return n;
    }
    __device__ __forceinline__ void flush_from(int s0) {
#pragma unroll
        for (int s = 0; s < NUSED; ++s)
            if (s >= s0) slot(s);
    }
    // the block's LAST tile drains in the open: nothing hides a residual load's latency there, so all
    // of them are requested up front, into the (now dead) live accumulator registers
    __device__ __forceinline__ void drain(f32x16 (&tmp)[TCO][TPX]) {
#ifdef LC_NO_DRAIN   // developer timing build: what does the open drain cost?  (wrong results)
        return;
#endif
#pragma unroll
        for (int k = 0; k < NV; ++k)
            tmp[i_of(k)][j_of(k)][4 * m_of(k) + (k & 3)] = __builtin_bit_cast(
                float, __builtin_amdgcn_raw_buffer_load_b32(rs_r, off_of(k), (unsigned)cor_of(k) * HW4, 0));
#pragma unroll
        for (int k = 0; k < NV; ++k) finalize_with(k, tmp[i_of(k)][j_of(k)][4 * m_of(k) + (k & 3)]);
    }
    // park the finished tile at (h0, w0): accumulators, pixel offsets, statistics entry, first residuals
    __device__ __forceinline__ void begin(const f32x16 (&acc)[TCO][TPX], int h0, int w0, int H, int W,
                                          int wpx, int lane, int tiles_w, int HWpx, int co_blk,
                                          bool prefetch) {
        const int l31 = lane & 31;
        int nvalid = 0;
#pragma unroll
        for (int j = 0; j < TPX; ++j) {
            const int t = wpx * TPX + j;
            const int tr = t / C::TPR, tc = t - tr * C::TPR;
            const int gh = h0 + tr, gw = w0 + tc * 32 + l31;
            const bool pok = gh < H && gw < W;
            voff[j] = pok ? (unsigned)(co_wave * HWpx + gh * W + gw) * 4u : OOB;
            if constexpr (EMIT) nvalid += __popcll(__ballot(pok) & 0xFFFFFFFFull);
#pragma unroll
            for (int i = 0; i < TCO; ++i) accp[i][j] = acc[i][j];
        }
        if constexpr (EMIT) {
            const int slot_id = ((h0 / C::TH_) * tiles_w + w0 / C::TW_) * C::WPX_ + wpx;
            if constexpr (!pairs) {
                nv8 = (float)(8 * nvalid);
                ent_off = lane >= 60 ? (unsigned)(co_blk >> 3) * oct_stride + (unsigned)slot_id * 16u + (unsigned)(lane & 3) * 4u : OOB;
            } else {
                nv8 = (float)(2 * nvalid);
                ent_off = l31 >= 28 ? (unsigned)((co_blk >> 1) + 2 * (lane >> 5)) * oct_stride + (unsigned)slot_id * 16u + (unsigned)(lane & 3) * 4u : OOB;
            }
        }
        if (prefetch) {                                 // (the last tile is drained with its own loads)
#pragma unroll
            for (int k = 0; k < LAG * VPT; ++k)
                if (k < NV) issue_res(k);
        }
    }
};

// the tall level-0 kernel (conv_f16x2_tall.hip, tile cfg 27)
bool tall_eligible(const ConvArgsH& a);
int launch_tall(ConvArgsH a, hipStream_t st);

}  // namespace lcconv
