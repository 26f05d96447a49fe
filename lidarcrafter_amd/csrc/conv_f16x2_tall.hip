// ---------------------------------------------------------------------------------------------
// TALL variant of the fp32-input 3x3 ring convolution (round 5, tile cfg 27): the level-0 kernel.
// Same arithmetic, accumulation order, deferred epilogue (DefEpi, conv_f16x2_common.h) and statistics
// partition as conv_f16x2_pipe_kernel's 64 co x (4 rows x 64 px) tile (conv_f16x2.hip), a different
// STAGING design.  What bounds a level-0 launch (profiles/r04_pp_kernel.txt, profiles/r05_level0.txt):
// the x loads (14 % of the launch), the GroupNorm + SiLU + split arithmetic applied to 1.55x the
// tile's own pixels because of the halo (15 %) and the matrix pipe (32 %), hardly overlapping.
//
//  * KEPT HALO ROWS.  A block walks `tpb` tiles of 4 rows x 64 columns DOWN the image.  Tile t+1's
//    first two staged rows are tile t's last two: they stay in LDS (per 16-channel chunk, two
//    parities: 66 KB), so in steady state a tile stages 4 rows, not 6 -- 1.16x its own pixels
//    instead of 1.55x: a third less GroupNorm / SiLU / split arithmetic and a third fewer loads.
//  * 64-BIT x LOADS BEHIND THE DMA.  A staging task is (2 adjacent pixels) x (4 channels): four
//    `buffer_load_dwordx2` per thread and chunk (pipelined kernel: sixteen 32-bit loads) into one of
//    two register sets (static parity); the two halo columns of the four new rows are one 32-bit load
//    per wave (16 lanes x one element).  The loads of chunk g + 2 go out in chunk g, BEHIND its last
//    weight-DMA piece, and are staged during chunk g + 1: the wait in front of the chunk barrier ("my
//    DMA pieces have landed", an exact vmcnt) leaves them in flight.
//  * WEIGHT DMA IN INLINE ASSEMBLY.  Through the builtin, hipcc orders every later ds_read behind
//    the LDS-DMA: an `s_waitcnt vmcnt(1..2)` in front of each tap's fragment reads, i.e. every tap
//    waited for the piece issued one tap earlier and -- VMEM returns in order -- for every deferred
//    store before it (22 such waits per chunk in the first build).  See dma_w.
//  The launch is power-limited (the same stream on zero operands runs 22-26 % faster): what pays is
//  removed work, not re-placed work -- the developer switches below (tap positions, priorities,
//  prefetch distance) all measure as noise (profiles/r05_level0.txt section 5).
//
// (First version of the round, measured and dropped -- profiles/r05_level0.txt: TRANSPOSED
// accumulators, i.e. MFMA operands swapped so that a lane holds one channel x 4 consecutive pixels
// and the epilogue is 16-byte stores / residual loads, a quarter of the epilogue's VMEM
// instructions.  Correct on its first run, but each such store touches 32 different 128-byte lines
// with 32 bytes each: the epilogue went from ~3 us hidden to 16-20 us exposed.  What the memory
// pipe prices is lines per instruction, not instructions.)
//
// LDS (units of 16 bytes = 8 channels of one pixel, one fp16 plane):
//    weights   2 x [plane hi/lo][tap 9][cb 2][co 64]            (LDS-DMA, double buffered by chunk)
//    x images are built from ROW-PAIR regions [plane][cb 2][row 2][col 66]:
//    main      2 x region   rows 2,3 of the staged tile          (double buffered by chunk)
//    keep      2 x NCH x region: keep[p][c] = rows 4,5 of a tile of parity p^1 for chunk c
//              = rows 0,1 of the next tile (parity p)
// A chunk's image is (keep[t&1][c], main[cur], keep[(t+1)&1][c]); the fragment reads of a wave touch
// three image rows (wpx + dy), i.e. three wave-uniform region bases.
//
// Constraints (the launcher falls back to the pipelined kernel otherwise): 3x3, Ci in {32, 64}
// (nchunk = Ci / 16 in {2, 4}: the kept rows of 4 chunks fill the LDS), H % 4 == 0, W % 64 == 0,
// Cgn <= 64.
// Reference semantics: ops.Conv2d + ops.Pad, lidargen/models/unets/ops.py:32-49,149-173;
// GroupNorm / AdaGN / SiLU in front: efficient_unet.py:61-115.
#include "conv_f16x2_common.h"

using namespace lcconv;

#ifndef LC_TALL_ABL
#define LC_TALL_ABL 0     // developer ablation (wrong results): 1 no x loads, 2 no staging arithmetic, 4 no MFMAs,
                          // 8 no deferred epilogue (nothing is stored), 16 no weight DMA in the K loop, 32 no drain of the last tile, 64 half of the fragment reads
#endif
#ifndef LC_TALL_DMA_ASM
#define LC_TALL_DMA_ASM 1 // weight LDS-DMA through inline assembly (see dma_w)
#endif
#ifndef LC_TALL_PRIO
#define LC_TALL_PRIO 0    // static wave priority for the K loop: 1 = s_setprio 1 for the younger half (waves 4-7), 2 = for the older half
#endif
#ifndef LC_TALL_T0
#define LC_TALL_T0 3      // taps that stage pixel 0 / pixel 1 / the edge element of the next chunk (the rows are read one tap before T0)
#define LC_TALL_T1 5
#define LC_TALL_T2 7
#endif
#ifndef LC_TALL_AHEAD
#define LC_TALL_AHEAD 2   // x loads run this many K chunks ahead of the chunk that stages them (1 or 2)
#endif

#ifndef LC_TALL_TIMING
#define LC_TALL_TIMING 0  // developer build: s_memtime phase totals per wave -> lc_dbg_tall (devtools/tall_phases.py)
#endif
#if LC_TALL_TIMING
__device__ unsigned long long lc_dbg_tall[32];
extern "C" int lc_debug_read_tall(unsigned long long* out32, int reset) {
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(out32, HIP_SYMBOL(lc_dbg_tall), sizeof(unsigned long long) * 32);
    if (reset) { unsigned long long z[32] = {0}; z[17] = z[19] = z[21] = ~0ull; hipMemcpyToSymbol(HIP_SYMBOL(lc_dbg_tall), z, sizeof(z)); }
    return 0;
}
#define LC_TT(var) { const unsigned long long t__ = __builtin_amdgcn_s_memtime(); var += t__ - t_mark; t_mark = t__; }
#else
#define LC_TT(var)
#endif

namespace {

typedef _Float16 h4_t __attribute__((ext_vector_type(4)));
typedef unsigned u2_t __attribute__((ext_vector_type(2)));

struct TLG {   // geometry
    static constexpr int TH = 4, TW = 64, BN = 64, CB = 2, NTAP = 9, NT = 512;
    static constexpr int RS = 66, CBS = 2 * RS, PLS = CB * CBS, RPU = 2 * PLS;      // row, channel block, plane, region (units)
    static constexpr int WU = NTAP * CB * BN, WB = 2 * WU;                          // units per weight plane / buffer
    static constexpr int MAXCH = 4;
    static constexpr int WB0 = 0, WDUMMY = 2 * WB, MAIN0 = WDUMMY + 64, KEEP0 = MAIN0 + 2 * RPU;
    static constexpr int LDS_UNITS = KEEP0 + 2 * MAXCH * RPU;                       // 9952 units = 159 232 bytes
    static constexpr int KW = 5;                                                    // weight DMA pieces per wave and chunk
    static constexpr int NWI = WU / 64;                                             // DMA instructions per weight plane (18)
    static constexpr int CTAB = 96;                                                 // fused-GroupNorm rows (<= 64 channels)
};
typedef HCfg<2, 4, 1, 2, 4, 64, 3> TLC;   // the pipelined kernel's tile: wave = 32 co x one image row (two 32-pixel blocks)

template <int NCH, int EMIT, int GNM>
__global__ __launch_bounds__(512, 2) void conv_f16x2_tall_kernel(ConvArgsH a) {
    typedef TLG G;
    typedef DefEpi<TLC, EMIT> DE;
    constexpr unsigned OOB = 0x80000000u, XOOB = 0xFFFFFFF0u;
    constexpr int RS = G::RS, CBS = G::CBS, PLS = G::PLS, RPU = G::RPU, WU = G::WU, KW = G::KW, NTAP = G::NTAP;
    constexpr int AH = LC_TALL_AHEAD;
    static_assert(NCH % 2 == 0 && (AH == 1 || AH == 2), "static register-set parity");
    __shared__ half8 lds[G::LDS_UNITS];
    __shared__ f32x4 ctab[G::CTAB];
    __shared__ float2 gtab[GN_MAX_G];
    __shared__ float bias_s[G::BN];
    char* const ldsb = reinterpret_cast<char*>(lds);
#if LC_TALL_TIMING
    const unsigned long long t_enter = __builtin_amdgcn_s_memtime();
    unsigned long long t_mark = t_enter, t_pro = 0, t_taps = 0, t_wait = 0, t_bar = 0, t_park = 0, t_drain = 0;
#endif

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kh = lane >> 5, l31 = lane & 31;
    // compute role: 32 output channels (wco) x image row wpx of the tile (two 32-pixel blocks)
    const int wco = wave >> 2, wpx = wave & 3;
    // staging role: row pair (0: tile rows 2,3 -> main, 1: rows 4,5 -> keep), 8-channel block, 4-channel half;
    // lane = (row of the pair, aligned pixel pair)
    const int rp = wave >> 2, scb = (wave >> 1) & 1, shalf = wave & 1;
    const int rsub = lane >> 5, pair = lane & 31;

    const int NTL = a.tpb;
    int bx = blockIdx.x;
    if (a.xcd) bx = (bx & 7) * (gridDim.x >> 3) + (bx >> 3);
    const int nseg = a.tiles_h / NTL;
    const int tw_i = bx % a.tiles_w; bx /= a.tiles_w;
    const int th0 = (bx % nseg) * NTL; bx /= nseg;
    const int b = bx;
    const int w0 = tw_i * G::TW;
    const int co0 = blockIdx.y * G::BN;
    const int H = a.H, W = a.W, HW = H * W;

    const float* xptr = a.x + (long long)b * a.x_bs;
    __amdgpu_buffer_rsrc_t rs_x =
        __builtin_amdgcn_make_buffer_rsrc((void*)xptr, 0, (unsigned)a.Ci * (unsigned)HW * 4u, 0x00020000);
    const float xs = a.range->x_scale;
    const float amax_seen = a.range->amax_scaled;
    const float out_unscale = a.range->x_unscale * a.wmeta[1];
    const float silu_c = -1.4426950408889634f * a.range->x_unscale;   // exp2(silu_c * (y * xs)) = exp(-y)
    float am = 0.0f;

    // ---- x staging state: byte offsets of this thread's task in chunk 0 of the tile whose loads are issued next
    unsigned xm_voff = XOOB, xe_voff = XOOB;      // main task (channel scb*8 + shalf*4, 2 pixels), edge element
    bool xm_ok = false, xe_ok = false;            // inside the image (else: exact zeros -- the reference pads AFTER the activation)
    const int e_rsub = (lane >> 3) & 1, e_side = (lane >> 2) & 1, e_chq = lane & 3;   // edge element of lanes 0..15
    auto set_tile = [&](int h0t, bool real) __attribute__((always_inline)) {
        const int gh = h0t + 1 + 2 * rp + rsub;                 // tile rows 2 .. 5 = image rows h0 + 1 .. h0 + 4
        xm_ok = real && gh < H;
        xm_voff = xm_ok ? (unsigned)((scb * 8 + shalf * 4) * HW + gh * W + w0 + 2 * pair) * 4u : XOOB;
        const int ghe = h0t + 1 + 2 * rp + e_rsub;
        int gwe = e_side ? w0 + G::TW : w0 - 1;
        gwe = gwe < 0 ? gwe + W : (gwe >= W ? gwe - W : gwe);
        xe_ok = real && lane < 16 && ghe < H;
        xe_voff = xe_ok ? (unsigned)((scb * 8 + shalf * 4 + e_chq) * HW + ghe * W + gwe) * 4u : XOOB;
    };
    // a loaded chunk: 4 channels x 2 pixels + one edge element, with the in-image flags of the tile it belongs to
    struct XSet { u2_t m[4]; float e; bool ok_m, ok_e; };
    XSet xset[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int k = 0; k < 4; ++k) xset[i].m[k] = u2_t{0u, 0u};
        xset[i].e = 0.f; xset[i].ok_m = xset[i].ok_e = false;
    }
    auto load_x = [&](XSet& xv, int ch) __attribute__((always_inline)) {
        xv.ok_m = xm_ok; xv.ok_e = xe_ok;
        if (LC_TALL_ABL & 1) return;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            xv.m[k] = __builtin_amdgcn_raw_buffer_load_b64(rs_x, xm_voff, (unsigned)(ch * 16 + k) * (unsigned)HW * 4u, 0);
        xv.e = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_x, xe_voff, (unsigned)(ch * 16) * (unsigned)HW * 4u, 0));
    };
    // destination bytes inside a row-pair region: this thread's first pixel (hi plane), its edge element
    const int dm_lane = (scb * CBS + rsub * RS + 1 + 2 * pair) * 16 + shalf * 8;
    const int de_lane = (scb * CBS + e_rsub * RS + (e_side ? RS - 1 : 0)) * 16 + (shalf * 4 + e_chq) * 2;

    auto stage_pair2 = [&](float x0, float x1, const f32x4 row, h2_t& ph, h2_t& pl) __attribute__((always_inline)) {
        if (LC_TALL_ABL & 2) { ph = h2_t{0, 0}; pl = h2_t{0, 0}; asm volatile("" ::"v"(x0), "v"(x1)); return; }
        if constexpr (GNM != 0) {
            f2_t v = {x0, x1};
            const f2_t A = {row.x, row.y}, Bv = {row.z, row.w};
            v = __builtin_elementwise_fma(v, A, Bv);
            if constexpr (GNM == 1) {
                const f2_t t = v * silu_c;
                f2_t e = {__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
                e = e + 1.0f;
                const f2_t rc = {__builtin_amdgcn_rcpf(e.x), __builtin_amdgcn_rcpf(e.y)};
                v = v * rc;
            }
            split_pair_scaled(v.x, v.y, ph, pl, am);
        } else {
            split_pair<false>(x0, x1, xs, ph, pl, am);
        }
    };
    // fused-GroupNorm rows of the main task (pair quads (A0, A1, B0, B1) x xs; 4 zero quads behind the table)
    f32x4 grow[2];
    auto stage_rows = [&](const XSet& xv, int ch) __attribute__((always_inline)) {
        if constexpr (GNM != 0) {
            const f32x4* g = xv.ok_m ? ctab + ch * 8 + scb * 4 + shalf * 2 : ctab + (a.Cgn >> 1);
            grow[0] = g[0]; grow[1] = g[1];
        } else {
            grow[0] = grow[1] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto stage_px = [&](const XSet& xv, int p, int dreg_b) __attribute__((always_inline)) {   // pixel p of the pair -> region at byte dreg_b
        const float v0 = __uint_as_float(xv.m[0][p]), v1 = __uint_as_float(xv.m[1][p]);
        const float v2 = __uint_as_float(xv.m[2][p]), v3 = __uint_as_float(xv.m[3][p]);
        h2_t h0, l0, h1, l1;
        stage_pair2(v0, v1, grow[0], h0, l0);
        stage_pair2(v2, v3, grow[1], h1, l1);
        const h4_t hv = {h0.x, h0.y, h1.x, h1.y}, lv = {l0.x, l0.y, l1.x, l1.y};
        char* d = ldsb + dreg_b + dm_lane + p * 16;
        *reinterpret_cast<h4_t*>(d) = hv;
        *reinterpret_cast<h4_t*>(d + PLS * 16) = lv;
    };
    auto stage_edge = [&](const XSet& xv, int ch, int dreg_b) __attribute__((always_inline)) {   // lanes 0..15: one halo-column element each
        h2_t ph, pl;
        if constexpr (GNM != 0) {
            const float* cf = reinterpret_cast<const float*>(ctab);
            const int qi = xv.ok_e ? (ch * 8 + scb * 4 + shalf * 2 + (e_chq >> 1)) * 4 + (e_chq & 1) : (a.Cgn >> 1) * 4;
            const f32x4 row = {cf[qi], 0.f, cf[qi + 2], 0.f};
            stage_pair2(xv.e, 0.0f, row, ph, pl);
        } else {
            stage_pair2(xv.e, 0.0f, f32x4{0.f, 0.f, 0.f, 0.f}, ph, pl);
        }
        const int dh = lane < 16 ? dreg_b + de_lane : G::WDUMMY * 16;
        const int dl = lane < 16 ? dreg_b + de_lane + PLS * 16 : G::WDUMMY * 16 + 16;
        *reinterpret_cast<_Float16*>(ldsb + dh) = ph.x;
        *reinterpret_cast<_Float16*>(ldsb + dl) = pl.x;
    };

    // ---- weight LDS-DMA: both planes through one descriptor (the lo plane follows the hi plane in ONE allocation);
    // instruction j = wave + 8 q of the chunk's 36 covers packed row j % 18 = (tap, cb): 64 channels x 16 bytes
    const unsigned wl_delta = (unsigned)((const char*)a.wl - (const char*)a.wh);
    const unsigned wplane_b = (unsigned)(NTAP * a.Cib) * (unsigned)a.Cop * 16u;
    __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)a.wh, 0, wl_delta + wplane_b, 0x00020000);
    const unsigned w_chunk = (unsigned)G::CB * (unsigned)a.Cop * 16u;
    unsigned voff_w[KW];
    int loff_w[KW];          // unit offset inside a weight buffer, -1 = the dummy block
#pragma unroll
    for (int q = 0; q < KW; ++q) {
        const int j = wave + q * 8;
        const int plane = j / G::NWI, jj = j - plane * G::NWI;
        const int tap = jj >> 1, cb = jj & 1;
        const bool real = j < 2 * G::NWI;
        loff_w[q] = real ? plane * WU + jj * 64 : -1;
        voff_w[q] = real ? (unsigned)((tap * a.Cib + cb) * a.Cop + co0 + lane) * 16u + (unsigned)plane * wl_delta : OOB;
    }
    // The DMA instruction is written in inline assembly (LC_TALL_DMA_ASM): through the builtin, hipcc treats it as a
    // store to LDS that every later ds_read may alias and puts an `s_waitcnt vmcnt(1..2)` in front of the NEXT tap's
    // fragment reads -- i.e. every tap waited for the piece issued one tap earlier AND (VMEM returns in order) for
    // the acknowledgement of every deferred-epilogue store before it (ISA of the first version: 22 such waits per
    // chunk).  The piece's landing is ordered by hand anyway: wait_vmcnt + s_barrier at the end of the chunk.
    const unsigned lds_base = (unsigned)(unsigned long long)(lds_vptr)lds;
    auto dma_w = [&](int wbuf_u, int q, int ch) __attribute__((always_inline)) {
        const int dst = loff_w[q] >= 0 ? wbuf_u + loff_w[q] : G::WDUMMY;
#if LC_TALL_DMA_ASM
        const unsigned m0v = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)dst * 16u);
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                     ::"s"(m0v), "v"(voff_w[q]), "s"(rs_w), "s"((unsigned)ch * w_chunk)
                     : "memory", "m0");
#else
        lds_dma16(rs_w, (lds_vptr)(lds + dst), voff_w[q], (unsigned)ch * w_chunk);
#endif
    };

    // ---- prologue ------------------------------------------------------------------------------------------------
    for (int i = tid; i < G::BN; i += G::NT)
        bias_s[i] = (a.bias && co0 + i < a.Co) ? a.bias[co0 + i] * (1.0f / out_unscale) : 0.0f;
#pragma unroll
    for (int q = 0; q < KW; ++q) dma_w(G::WB0, q, 0);
    const int h00 = th0 * G::TH;
    set_tile(h00, true);
    load_x(xset[0], 0);                                    // chunk n lives in register set n & 1
    if (AH == 2) load_x(xset[1], 1);
    // ... and the loads of the block's top rows (staged behind the GroupNorm fold below): NTR rounds of 4 x 64 bits
    constexpr int NTOP = NCH * 2 * 2 * 2 * 34, NTR = (NTOP + G::NT - 1) / G::NT;
    u2_t tv[NTR][4];
#pragma unroll
    for (int rd = 0; rd < NTR; ++rd) {
        const int t = tid + rd * G::NT;
        const int pp = t % 34;
        int u = t / 34;
        const int row = u & 1; u >>= 1;
        const int hf = u & 1; u >>= 1;
        const int cb = u & 1;
        const int c = u >> 1;
        const int gh = h00 - 1 + row;
        int gw = w0 - 2 + 2 * pp;
        gw = gw < 0 ? gw + W : (gw >= W ? gw - W : gw);
        const unsigned vo = (t < NTOP && gh >= 0) ? (unsigned)((c * 16 + cb * 8 + hf * 4) * HW + gh * W + gw) * 4u : XOOB;
#pragma unroll
        for (int k = 0; k < 4; ++k) tv[rd][k] = __builtin_amdgcn_raw_buffer_load_b64(rs_x, vo, (unsigned)k * (unsigned)HW * 4u, 0);
    }
    if constexpr (GNM != 0) {   // rows of the fused input norm (as conv_f16x2_pipe_kernel), repacked per channel pair
        if (a.gs.partials) {
            for (int i = tid; i < a.Cgn; i += G::NT) ctab[i] = gn_row_from_stats(a.gs, xptr, b, i, a.Ci, HW);
        } else if (a.seg[0].p) {
            gn_rows_from_ostats<G::NT>(a, b, tid, ctab, gtab);
        } else {
            const f32x4* g = a.gn + (long long)b * a.Cgn;
            for (int i = tid; i < a.Cgn; i += G::NT) ctab[i] = g[i];
        }
        __syncthreads();
        const int npair = a.Cgn >> 1;                      // <= 32
        f32x4 qd = {0.f, 0.f, 0.f, 0.f};
        if (tid < npair) {
            const f32x4 r0 = ctab[2 * tid], r1 = ctab[2 * tid + 1];
            qd = f32x4{r0.y * xs, r1.y * xs, fmaf(-r0.x, r0.y, r0.z) * xs, fmaf(-r1.x, r1.y, r1.z) * xs};
        }
        __syncthreads();
        if (tid < npair) ctab[tid] = qd;
        if (tid < 4) ctab[npair + tid] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();                                       // bias_s and the rows visible
    // the first tile's rows 0,1 (image rows h0 - 1, h0) of every chunk -> keep[0][c]: tasks (chunk, cb, half, row,
    // aligned pair p = columns w0 - 2 + 2 p, + 1 of which columns -1 .. 64 of the tile are used).  Once per block; their
    // loads were issued in front of the GroupNorm fold (top_load), all NTR rounds at once.
#pragma unroll
    for (int rd = 0; rd < NTR; ++rd) {
        const int t = tid + rd * G::NT;
        if (t < NTOP) {
            const int pp = t % 34;
            int u = t / 34;
            const int row = u & 1; u >>= 1;
            const int hf = u & 1; u >>= 1;
            const int cb = u & 1;
            const int c = u >> 1;
            const bool ok = h00 - 1 + row >= 0;
            f32x4 r0 = {0.f, 0.f, 0.f, 0.f}, r1 = r0;
            if constexpr (GNM != 0) {
                const f32x4* g = ok ? ctab + c * 8 + cb * 4 + hf * 2 : ctab + (a.Cgn >> 1);
                r0 = g[0]; r1 = g[1];
            }
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int uc = 2 * pp - 1 + e;
                h2_t h0, l0, h1, l1;
                stage_pair2(__uint_as_float(tv[rd][0][e]), __uint_as_float(tv[rd][1][e]), r0, h0, l0);
                stage_pair2(__uint_as_float(tv[rd][2][e]), __uint_as_float(tv[rd][3][e]), r1, h1, l1);
                if (uc >= 0 && uc < RS) {
                    char* d = ldsb + ((G::KEEP0 + c * RPU + cb * CBS + row * RS + uc) * 16 + hf * 8);
                    *reinterpret_cast<h4_t*>(d) = h4_t{h0.x, h0.y, h1.x, h1.y};
                    *reinterpret_cast<h4_t*>(d + PLS * 16) = h4_t{l0.x, l0.y, l1.x, l1.y};
                }
            }
        }
    }
    // rows 2 .. 5 of the first tile's chunk 0 -> main[0], keep[1][0]
    {
        const int dreg_b = (rp ? G::KEEP0 + (G::MAXCH + 0) * RPU : G::MAIN0) * 16;
        stage_rows(xset[0], 0);
        stage_px(xset[0], 0, dreg_b);
        stage_px(xset[0], 1, dreg_b);
        stage_edge(xset[0], 0, dreg_b);
    }

    // ---- compute state ---------------------------------------------------------------------------------------------
    const int fl = kh * CBS + l31;                         // lane part of an x fragment address (units)
    const int wfl = kh * G::BN + wco * 32 + l31;           // ... of a weight fragment
    f32x16 acc[1][2];
    auto acc_init = [&]() __attribute__((always_inline)) {  // accumulators start from bias / out_unscale (power of two: exact)
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const f32x4 bq = *reinterpret_cast<const f32x4*>(&bias_s[wco * 32 + 8 * m + 4 * kh]);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                acc[0][j][4 * m] = bq.x; acc[0][j][4 * m + 1] = bq.y;
                acc[0][j][4 * m + 2] = bq.z; acc[0][j][4 * m + 3] = bq.w;
            }
        }
    };
    DE de;
    const int co_wave = co0 + wco * 32 + 4 * kh;
    float* yb = a.y + (long long)b * a.y_bs;
    const float* rb = a.res ? a.res + (long long)b * a.res_bs : nullptr;
    de.init(yb, rb, a.Co, HW, out_unscale, a.out_scale, co_wave,
            a.ostats ? a.ostats + (long long)b * (a.Co >> (a.ounit == 2 ? 1 : 3)) * a.oslots : nullptr, a.oslots, a.ounit);
    acc_init();

    // ---- one K chunk: MFMAs of chunk C of tile t from (keep[t & 1][C], main[bp], keep[(t + 1) & 1][C]) and wbuf[bp];
    // riding in the stream: the weight DMA and the staging of the NEXT chunk (from the register set loaded AH chunks
    // ago), the loads of the chunk AH ahead, the parked tile's deferred slots C * 9 ..
    auto k_iter = [&](auto ctag, int t) __attribute__((always_inline)) {
        constexpr int C = decltype(ctag)::value;
        constexpr int NC = (C + 1) % NCH;                   // the chunk staged meanwhile
        constexpr int LC = (C + AH) % NCH;                  // the chunk whose loads are issued
        constexpr int bp = C & 1;                           // NCH is even: buffer parity = chunk parity
        const int nt = C + 1 < NCH ? t : t + 1;             // tile of the staged chunk
        const bool more = nt < NTL;
        if (C + AH == NCH) {                                // the loads enter the next tile
            const int lt = t + 1;
            set_tile((th0 + lt) * G::TH, lt < NTL);
        }
        XSet& xl = xset[(C + AH) & 1];                      // chunk n lives in set n & 1
        const XSet& xst = xset[(C + 1) & 1];
        // AH == 2: the loads go out BEHIND the chunk's last weight-DMA piece (tap KW), so that the wait in front of the
        // chunk barrier -- "my DMA pieces have landed" -- leaves them in flight; AH == 1: up front, staged below
        if (AH == 1) {
            load_x(xl, LC);
            __builtin_amdgcn_sched_barrier(0);
        }
        // destination region of this wave's staging: main[bp ^ 1] or keep[(nt + 1) & 1][NC] (past the block's last
        // chunk: both into the dead main buffer)
        const int dmain = G::MAIN0 + (bp ^ 1) * RPU;
        const int dreg_b = ((rp && more) ? G::KEEP0 + (((nt + 1) & 1) * G::MAXCH + NC) * RPU : dmain) * 16;
        const int wnext = G::WB0 + (bp ^ 1) * G::WB;
        // fragment sources
        const int r0u = G::KEEP0 + ((t & 1) * G::MAXCH + C) * RPU, r1u = G::MAIN0 + bp * RPU,
                  r2u = G::KEEP0 + (((t + 1) & 1) * G::MAXCH + C) * RPU;
        const half8* rowp[3];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int row = wpx + dy;
            const int ru = (row >> 1) == 0 ? r0u : ((row >> 1) == 1 ? r1u : r2u);
            rowp[dy] = lds + ru + (row & 1) * RS + fl;
        }
        const half8* cw = lds + G::WB0 + bp * G::WB + wfl;
        half8 wh_[2], wl_[2], xh_[2][2], xl_[2][2];
        auto fetch = [&](int tap, int s) __attribute__((always_inline)) {
            const int dy = tap / 3, dx = tap - 3 * dy;
            wh_[s] = cw[tap * (G::CB * G::BN)];
            wl_[s] = cw[WU + tap * (G::CB * G::BN)];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                xh_[s][j] = rowp[dy][j * 32 + dx];
                xl_[s][j] = rowp[dy][PLS + j * 32 + dx];
            }
        };
        fetch(0, 0);
#pragma unroll
        for (int tap = 0; tap < NTAP; ++tap) {
            const int s = tap & 1;
            if (tap == LC_TALL_T0 - 1) { stage_rows(xst, NC); __builtin_amdgcn_sched_barrier(0); }   // rows ahead of the fragment fetch (no lgkmcnt(0))
            if (tap + 1 < NTAP && !((LC_TALL_ABL & 64) && (tap & 1))) fetch(tap + 1, s ^ 1);   // (64: every second tap re-uses stale fragments)
            __builtin_amdgcn_sched_barrier(0);
            if (!(LC_TALL_ABL & 8) && C * NTAP + tap < DE::NUSED) de.slot(C * NTAP + tap);
            // this tap's weight DMA piece BEHIND the deferred slot: the wait in front of the chunk barrier then leaves
            // exactly the later taps' deferred operations in flight.  The count is by hand (the DMA is inline assembly,
            // invisible to hipcc's vmcnt tracking): it holds as long as every counted operation really is issued behind
            // the last piece -- an operation hipcc hoists in front of it, merges or drops would let a piece stay in
            // flight across the barrier.  tests/test_isa_audit.py::test_tall_kernel_chunk_barrier_waits_cover_the_weight_dma
            // checks exactly that on the disassembly of every instantiation.
            if (tap < KW && !(LC_TALL_ABL & 16)) dma_w(wnext, tap, NC);
            if (AH == 2 && tap == KW) load_x(xl, LC);
            if (tap == LC_TALL_T0) stage_px(xst, 0, dreg_b);
            if (tap == LC_TALL_T1) stage_px(xst, 1, dreg_b);
            if (tap == LC_TALL_T2) stage_edge(xst, NC, dreg_b);
            if (LC_TALL_ABL & 4) {
                asm volatile("" ::"v"(wh_[s]), "v"(wl_[s]), "v"(xh_[s][0]), "v"(xl_[s][0]), "v"(xh_[s][1]), "v"(xl_[s][1]));
                __builtin_amdgcn_sched_barrier(0);
                continue;
            }
            if (LC_F16X2_TERMS & 2) {
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl_[s], xh_[s][j], acc[0][j], 0, 0, 0);
            }
            if (LC_F16X2_TERMS & 4) {
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh_[s], xl_[s][j], acc[0][j], 0, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh_[s], xh_[s][j], acc[0][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        LC_TT(t_taps)
        // chunk barrier: this wave's ds_writes done, its weight DMA pieces landed = everything but the deferred
        // operations issued behind the last piece (VMEM returns in order); a bare s_barrier (no vmcnt(0) drain)
        {
            int later = (AH == 2 && !(LC_TALL_ABL & 1)) ? 5 : 0;
#pragma unroll
            for (int tap = KW; tap < NTAP; ++tap) {
                const int sl = C * NTAP + tap;
                if (!(LC_TALL_ABL & 8) && sl < DE::NUSED) later += DE::ops_of(sl);
            }
            wait_vmcnt(later);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        LC_TT(t_wait)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        LC_TT(t_bar)
    };

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the prologue's weight DMA has landed
    __syncthreads();
    if (LC_TALL_PRIO == 1 && wave >= 4) __builtin_amdgcn_s_setprio(1);
    if (LC_TALL_PRIO == 2 && wave < 4) __builtin_amdgcn_s_setprio(1);
    LC_TT(t_pro)
    for (int t = 0; t < NTL; ++t) {
        k_iter(std::integral_constant<int, 0>{}, t);
        k_iter(std::integral_constant<int, 1>{}, t);
        if constexpr (NCH > 2) {
            k_iter(std::integral_constant<int, 2>{}, t);
            k_iter(std::integral_constant<int, 3>{}, t);
        }
        if constexpr (NCH < DE::DCH) {                      // short K: the rest of the parked tile now
            if (!(LC_TALL_ABL & 8)) de.flush_from(NCH * NTAP);
        }
        // park this tile; the accumulators restart from the bias
        const bool more = t + 1 < NTL;
        de.begin(acc, (th0 + t) * G::TH, w0, H, W, wpx, lane, a.tiles_w, HW, co0 + wco * 32, more);
        if (more) acc_init();
        LC_TT(t_park)
    }
#if LC_TALL_ABL & 32   // developer timing build: what does the open drain cost?  (the last tile is never stored)
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" ::"v"(de.accp[0][0]), "v"(de.accp[0][1]));
#endif
#else
    if (!(LC_TALL_ABL & 8)) de.drain(acc);                  // the last tile drains in the open
#endif
#if LC_TALL_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // (instrumentation: the drain's stores acknowledged)
#endif
    LC_TT(t_drain)
    publish_amax(a.range, am, amax_seen);
#if LC_TALL_TIMING
    if (lane == 0) {
        unsigned long long* d = lc_dbg_tall + 8 * (wave >> 2);         // waves 0-3 / 4-7 (older / younger half)
        atomicAdd(&d[0], __builtin_amdgcn_s_memtime() - t_enter);
        atomicAdd(&d[1], t_pro); atomicAdd(&d[2], t_taps); atomicAdd(&d[3], t_wait); atomicAdd(&d[4], t_bar);
        atomicAdd(&d[5], t_park); atomicAdd(&d[6], t_drain); atomicAdd(&d[7], 1ull);
        // spread over the chip: max / min lifetime, max taps, latest end and earliest start (absolute ticks)
        atomicMax(&lc_dbg_tall[16], __builtin_amdgcn_s_memtime() - t_enter);
        atomicMin(&lc_dbg_tall[17], __builtin_amdgcn_s_memtime() - t_enter);
        atomicMax(&lc_dbg_tall[18], t_taps + t_wait + t_bar);
        atomicMin(&lc_dbg_tall[19], t_taps + t_wait + t_bar);
        atomicMax(&lc_dbg_tall[20], __builtin_amdgcn_s_memtime());
        atomicMin(&lc_dbg_tall[21], t_enter);
        atomicMax(&lc_dbg_tall[22], t_enter);
    }
#endif
}

}  // namespace

namespace lcconv {

bool tall_eligible(const ConvArgsH& a) {
    if (a.xsp || a.part || (a.Ci != 32 && a.Ci != 64) || a.H % TLG::TH || a.W % TLG::TW) return false;
    if (a.gn && a.Cgn > 64) return false;
    return true;
}

// Launch (after tall_eligible): tiles per block as launch_pipe chooses them (walking down H only).
int launch_tall(ConvArgsH a, hipStream_t st) {
    if (!tall_eligible(a)) return LC_EUNSUP;
    a.tiles_h = a.H / TLG::TH;
    a.tiles_w = a.W / TLG::TW;
    const int ncot = (a.Co + TLG::BN - 1) / TLG::BN;
    const long long n_tiles = (long long)a.B * a.tiles_h * a.tiles_w * ncot;
    int tpb = 1;
    while (tpb < 8 && a.tiles_h % (tpb * 2) == 0 && n_tiles / (tpb * 2) >= 256) tpb *= 2;
    if (a.tpb > 0 && a.tiles_h % a.tpb == 0) tpb = a.tpb;             // explicit override (tests): tpb * 100 + cfg
    a.tpb = tpb;
    a.vert = 1;
    dim3 grid(a.B * a.tiles_h * a.tiles_w / tpb, ncot);
    static const int xcd_env = [] { const char* e = getenv("LC_CONV_XCD"); return e ? atoi(e) : 1; }();
    a.xcd = (xcd_env && grid.x % 8 == 0 && grid.x >= 16) ? 1 : 0;
    {
        const long long d = (const char*)a.wl - (const char*)a.wh;
        if (d <= 0 || d >= (1ll << 31)) return LC_EINVAL;
        if (d + (long long)TLG::NTAP * a.Cib * a.Cop * 16 >= (1ll << 31)) return LC_EUNSUP;
    }
    const int gnm = a.gn ? (a.gn_silu ? 1 : 2) : 0;
    const int emit = a.ostats ? (a.ounit == 2 ? 2 : 1) : 0;
#define LC_TALL_LAUNCH(N, E, Gm) hipLaunchKernelGGL((conv_f16x2_tall_kernel<N, E, Gm>), grid, dim3(TLG::NT), 0, st, a)
#define LC_TALL_G(N, E) do { if (gnm == 1) LC_TALL_LAUNCH(N, E, 1); else if (gnm == 2) LC_TALL_LAUNCH(N, E, 2); else LC_TALL_LAUNCH(N, E, 0); } while (0)
#define LC_TALL_E(N) do { if (emit == 2) LC_TALL_G(N, 2); else if (emit == 1) LC_TALL_G(N, 1); else LC_TALL_G(N, 0); } while (0)
    if (a.Ci == 64) LC_TALL_E(4); else LC_TALL_E(2);
#undef LC_TALL_E
#undef LC_TALL_G
#undef LC_TALL_LAUNCH
    return lc_launch_status();
}

}  // namespace lcconv

LC_TOUCH_TU(conv_f16x2_tall, conv_f16x2_tall_kernel<4, 1, 1>)
