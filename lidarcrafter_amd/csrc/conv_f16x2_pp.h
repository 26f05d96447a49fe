// ---------------------------------------------------------------------------------------------
// PING-PONG variant of the fp32-input 3x3 ring convolution (round 4) -- included by conv_f16x2.hip
// inside its anonymous namespace (shares ConvArgsH, the split rule, the GroupNorm row derivation).
//
// Why.  In conv_f16x2_pipe_kernel all eight waves of a block run the SAME mixed stream (MFMAs +
// GroupNorm / SiLU / hi-lo split of the next chunk + the deferred epilogue), two waves per SIMD.  The
// matrix pipe is per SIMD and an in-order wave cannot slip an MFMA into a gap shorter than the 32 cycles
// it needs (MI355X_MICROARCH.md, "Two waves per SIMD"): PMC of the level-0 launch showed the pipe busy
// 34 % of the time -- ~35 us of matrix time next to ~49 us of staging and store traffic with hardly any
// overlap (profiles/r03_conv_phases.txt, r03_pmc_level0.txt).
//
// What.  The block's eight waves form two GROUPS of four (one wave of each group per SIMD).  Both work
// on the same strip of 4 rows x 64 columns; group g owns output channels 32 g .. 32 g + 31 of the
// block's 64 (a wave = 32 channels x one image row of 64 pixels = two 32x32 accumulators).  The groups
// alternate between two roles, one block-wide s_barrier per role change, group 1 one phase behind:
//     phase   group 0                                   group 1
//     2s      COMPUTE step s: 54 MFMAs, nothing else    STAGE: GroupNorm + SiLU + hi/lo split of ITS
//             but fragment reads (+ issues the x        8-channel half of chunk s+1 -> LDS, weight LDS-DMA
//             loads it will stage next phase)           of step s+1, one octet of the parked strip's epilogue
//     2s+1    STAGE: its half of chunk s+1, epilogue    COMPUTE step s
// (step = (strip, 16-channel chunk)).  At any time one wave per SIMD issues back-to-back MFMAs while its
// partner does all the VALU / memory work.  The x image of a step (hi + lo planes, 25 KB) and the packed
// weights of its chunk (36 KB) are double buffered by step parity: the image of step s+1 is written during
// phases 2s (channels 8-15 by group 1) and 2s+1 (channels 0-7 by group 0), after its last reader (group 1,
// step s-1, phase 2s-1) and before its first (group 0, phase 2s+2).
//
// Epilogue: a finished strip's accumulators are parked in a second register set and finalised one channel
// octet (8 values per lane) per staging phase of the next strip -- x out_unscale, + residual (loaded one
// phase ahead), x out_scale, GroupNorm statistics (octet or pair entries, one 32-bit store each), one 32-bit
// write-through store per value -- so the write traffic is spread over the launch; only the block's last
// strip drains in the open.
//
// Same arithmetic and the same accumulation order per output value as conv_f16x2_pipe_kernel, same
// statistics partition as its 64 co x 256 px tile (one entry per octet / pair and image row of the strip):
// outputs are bit-identical to it (tests/test_hip_parity.py::test_conv_pp_matches_pipe).
// Constraints (callers fall back to the pipelined kernel otherwise): 3x3, Ci % 16 == 0, 64 <= Ci <= 512,
// Co % 64 == 0, H % 4 == 0, W % 64 == 0.
#ifndef LC_PP_DEPTH
#define LC_PP_DEPTH 1     // fragment prefetch distance of the compute phase, in taps
#endif
#ifndef LC_PP_PRIO
#define LC_PP_PRIO 0      // 1: s_setprio 2 for the duration of a compute phase
#endif
#ifndef LC_PP_DMA0
#define LC_PP_DMA0 9      // weight-DMA pieces (of 9 per wave) issued by group 0's compute phase; the rest by group 1's stage
                          // slot of the same phase
#endif
#ifndef LC_PP_ABL
#define LC_PP_ABL 0   // developer ablation (wrong results): 1 no output stores, 2 no x loads, 4 no weight DMA, 8 no MFMAs,
                      // 16 no staging arithmetic / ds_write, 32 no residual loads
#endif
struct PPG {   // geometry
    static constexpr int TH = 4, TW = 64, XR = 6, XW = 66, PL = XR * XW, CB = 2, BN = 64, NTAP = 9;
    static constexpr int XU = CB * PL, XUP = XU + 1;       // 16-byte units per x plane (+ one dummy unit)
    static constexpr int WU = NTAP * CB * BN;              // units per weight plane and chunk (1152 = 18 x 64)
    static constexpr int NT = 512, GW = 4;                 // threads, waves per group
    static constexpr int XPR = XW / 2 + 1;                 // aligned pixel PAIRS a staged row is loaded as (34: columns -2 .. 65 of the tile)
    static constexpr int NXP = XR * XPR;                   // pairs per 8-channel block and step (204 <= 256 lanes of a group)
    static constexpr int NXU = 2;                          // staged positions per thread and step (the two pixels of its pair)
    static constexpr int NWD = 2 * WU / 64 / GW;           // weight DMA instructions per wave of group 1 (9)
    static constexpr int TPX = 2;                          // 32-pixel sub-tiles per wave (one image row)
    static constexpr int XB = 2 * XUP;                     // units per x image (hi + lo plane)
    static constexpr int WB = 2 * WU;                      // units per weight buffer (hi + lo plane)
    static constexpr int LDS_UNITS = 2 * XB + 2 * WB;
    static constexpr int MAX_C = 512;                      // fused-GroupNorm rows held in LDS
};

template <int EMIT, int GNM>
__global__ __launch_bounds__(512, 1) void conv_f16x2_pp_kernel(ConvArgsH a) {
    typedef PPG G;
    constexpr int XW = G::XW, PL = G::PL, XUP = G::XUP, WU = G::WU, BN = G::BN, NXU = G::NXU, TPX = G::TPX;
    constexpr bool pairs = EMIT == 2;
    constexpr unsigned OOB = 0x80000000u;
    __shared__ half8 lds[G::LDS_UNITS];
    __shared__ f32x4 ctab[G::MAX_C];
    __shared__ float2 gtab[GN_MAX_G];
    __shared__ float bias_s[BN];

#if LC_TIMING
    const unsigned long long t_enter = __builtin_amdgcn_s_memtime();
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wq = wave & 3;              // group = 32-channel half, wq = image row of the strip
    const int kh = lane >> 5, l31 = lane & 31;

    // block -> (sample, W tile, first strip row); NS consecutive strips down H
    const int NS = a.tpb;
    int bx = blockIdx.x;
    if (a.xcd) bx = (bx & 7) * (gridDim.x >> 3) + (bx >> 3);
    const int nseg = a.tiles_h / NS;
    const int tw_i = bx % a.tiles_w; bx /= a.tiles_w;
    const int th0 = (bx % nseg) * NS; bx /= nseg;
    const int b = bx;
    const int w0 = tw_i * G::TW;
    const int co0 = blockIdx.y * BN;
    const int H = a.H, W = a.W, HW = H * W;
    const int nchunk = a.Cib >> 1;
    const int NSTEP = NS * nchunk;

    const float* xptr = a.x + (long long)b * a.x_bs;
    __amdgpu_buffer_rsrc_t rs_x =
        __builtin_amdgcn_make_buffer_rsrc((void*)xptr, 0, (unsigned)a.Ci * (unsigned)HW * 4u, 0x00020000);
    const float xs = a.range->x_scale;
    const float amax_seen = a.range->amax_scaled;
    const float out_unscale = a.range->x_unscale * a.wmeta[1];
    const float out_scale = a.out_scale;
    float am = 0.0f;

    half8* xbuf = lds;                        // x images (step parity): hi plane, lo plane at + XUP
    half8* wbuf = lds + 2 * G::XB;            // weight buffers (step parity): hi plane, lo plane at + WU

    // ---- x staging: a thread owns ONE aligned pixel pair (columns w0 - 2 + 2 p, + 1) of one row of its group's
    // 8-channel block: eight 64-bit loads per step (one per channel; half the VMEM instructions of 32-bit loads -- the
    // launch is bound by VMEM instruction issue, profiles/r04_pp_kernel.txt) = the image positions 2 p - 1 and 2 p.
    // An aligned pair never straddles the ring seam (W is even).
    const int x_lin = wq * 64 + lane;                      // pair index r * XPR + p, valid below NXP
    const int x_r = x_lin / G::XPR, x_p = x_lin - x_r * G::XPR;
    const bool x_has = x_lin < G::NXP;
    int x_loc[NXU];                                        // position r * XW + column of the pair's pixels, -1 = outside the image
    x_loc[0] = (x_has && x_p >= 1) ? x_r * XW + 2 * x_p - 1 : -1;
    x_loc[1] = (x_has && 2 * x_p < XW) ? x_r * XW + 2 * x_p : -1;
    // The loads of a step are issued one stage slot (two phases) before the slot that stages it, into the registers
    // that slot has just consumed; the set carries the byte offset its loads used (the sentinel marks a row outside
    // the image: it stages exact zeros).  (A second set -- loads two slots ahead -- was tried: hipcc waits for it with
    // vmcnt(0) across the loop back-edge, which defeats it; profiles/r04_pp_kernel.txt.)
    struct XSet { float v[NXU][8]; unsigned voff; };       // v[pixel of the pair][channel]
    XSet xsa;
    auto x_offset = [&](int v) __attribute__((always_inline)) {   // byte offset of this thread's pair in channel grp * 8, strip of step v
        const int t = (v < NSTEP ? v : NSTEP - 1) / nchunk;
        const int gh = (th0 + t) * G::TH - 1 + x_r;
        int gw = w0 - 2 + 2 * x_p;
        gw %= W; if (gw < 0) gw += W;
        const bool ok = x_has && gh >= 0 && gh < H;
        return ok ? (unsigned)(grp * 8 * HW + gh * W + gw) * 4u : 0xFFFFFFF0u;
    };
    auto load_x = [&](XSet& xs_, int v) __attribute__((always_inline)) {   // (past the last step: a valid chunk, never staged)
        if (LC_PP_ABL & 2) return;
        typedef unsigned u2_t __attribute__((ext_vector_type(2)));
        const int ch = v % nchunk;
        xs_.voff = x_offset(v);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const u2_t d = __builtin_amdgcn_raw_buffer_load_b64(rs_x, xs_.voff, (unsigned)(ch * 16 + k) * HW * 4u, 0);
            xs_.v[0][k] = __uint_as_float(d.x);
            xs_.v[1][k] = __uint_as_float(d.y);
        }
    };

    // ---- weight LDS-DMA (issued by group 1): both planes through one descriptor.  Instruction j = wq + 4 q of the
    // chunk's 36 (18 per plane) covers packed row j % 18 = (tap, cb) completely (64 channels x 16 bytes): the per-lane
    // part of the address is lane * 16, the rest is wave-uniform and rides in the SGPR soffset
    const unsigned wl_delta = (unsigned)((const char*)a.wl - (const char*)a.wh);
    const unsigned wplane_b = (unsigned)(G::NTAP * a.Cib) * (unsigned)a.Cop * 16u;
    __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)a.wh, 0, wl_delta + wplane_b, 0x00020000);
    const unsigned w_chunk = (unsigned)G::CB * (unsigned)a.Cop * 16u;
    const unsigned voff_lane = (unsigned)lane * 16u;
    auto dma_w_piece = [&](int step, int q) __attribute__((always_inline)) {   // piece q of 9 of this wave's share
        if (LC_PP_ABL & 4) return;
        half8* dst = wbuf + (step & 1) * G::WB;
        const unsigned so = (unsigned)(step % nchunk) * w_chunk + (unsigned)co0 * 16u;
        const int j = wq + q * G::GW, plane = j / (WU / 64), jj = j - plane * (WU / 64);
        const int tap = jj >> 1, cb = jj & 1;
        const unsigned srow = (unsigned)((tap * a.Cib + cb) * a.Cop) * 16u + (unsigned)plane * wl_delta;
        lds_dma16(rs_w, (lds_vptr)(dst + plane * WU + jj * 64), voff_lane, so + srow);
    };
    auto dma_w = [&](int step) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < G::NWD; ++q) dma_w_piece(step, q);
    };

    // ---- bias and fused-GroupNorm rows (as conv_f16x2_pipe_kernel); the first loads are in flight meanwhile
    for (int i = tid; i < BN; i += G::NT)
        bias_s[i] = (a.bias && co0 + i < a.Co) ? a.bias[co0 + i] * (1.0f / out_unscale) : 0.0f;
    if (grp == 0) dma_w(0);
    load_x(xsa, 0);
    if constexpr (GNM != 0) {
        if (a.gs.partials) {
            for (int i = tid; i < a.Cgn; i += G::NT) ctab[i] = gn_row_from_stats(a.gs, xptr, b, i, a.Ci, HW);
        } else if (a.seg[0].p) {
            gn_rows_from_ostats<G::NT>(a, b, tid, ctab, gtab);
        } else {
            const f32x4* g = a.gn + (long long)b * a.Cgn;
            for (int i = tid; i < a.Cgn; i += G::NT) ctab[i] = g[i];
        }
        __syncthreads();
        // (mu, A, B, 0) rows repacked in place per channel PAIR: (A0, A1, B0, B1) pre-multiplied by x_scale
        const int npair = a.Cgn >> 1;                      // <= 256: one pair per thread
        f32x4 qd = {0.f, 0.f, 0.f, 0.f};
        if (tid < npair) {
            const f32x4 r0 = ctab[2 * tid], r1 = ctab[2 * tid + 1];
            qd = f32x4{r0.y * xs, r1.y * xs, fmaf(-r0.x, r0.y, r0.z) * xs, fmaf(-r1.x, r1.y, r1.z) * xs};
        }
        __syncthreads();
        if (tid < npair) ctab[tid] = qd;
        if (tid < 4) ctab[npair + tid] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();                                       // bias_s (and the rows) visible
    const float silu_c = -1.4426950408889634f * a.range->x_unscale;

    // ---- staging of one unit into the image of `step`: GroupNorm (+SiLU) + split, the arithmetic of the pipelined kernel
    auto stage_unit = [&](const XSet& xs_, int i, int step) __attribute__((always_inline)) {
        if (LC_PP_ABL & 16) { asm volatile("" ::"v"(xs_.v[i][0]), "v"(xs_.v[i][7])); return; }
        const int ch = step % nchunk;
        half8* img = xbuf + (step & 1) * G::XB;
        half8 hi, lo;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            h2_t ph, pl;
            if constexpr (GNM != 0) {
                const f32x4* g = xs_.voff != 0xFFFFFFF0u ? ctab + ch * 8 + grp * 4 : ctab + (a.Cgn >> 1);
                const f32x4 row = g[q];
                f2_t v = {xs_.v[i][2 * q], xs_.v[i][2 * q + 1]};
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
                split_pair<false>(xs_.v[i][2 * q], xs_.v[i][2 * q + 1], xs, ph, pl, am);
            }
            hi[2 * q] = ph.x; hi[2 * q + 1] = ph.y;
            lo[2 * q] = pl.x; lo[2 * q + 1] = pl.y;
        }
        const int d = x_loc[i] >= 0 ? grp * PL + x_loc[i] : G::XU;
        img[d] = hi;
        img[XUP + d] = lo;
    };

    // ---- accumulators, parked set, epilogue state
    f32x16 acc[TPX];
    float accp[TPX][16];          // parked strip (scalars: an octet's registers are free as soon as it is finalised)
    auto acc_init = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const f32x4 bq = *reinterpret_cast<const f32x4*>(&bias_s[grp * 32 + 8 * m + 4 * kh]);
#pragma unroll
            for (int j = 0; j < TPX; ++j) {
                acc[j][4 * m] = bq.x; acc[j][4 * m + 1] = bq.y; acc[j][4 * m + 2] = bq.z; acc[j][4 * m + 3] = bq.w;
            }
        }
    };
    acc_init();
#pragma unroll
    for (int j = 0; j < TPX; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) accp[j][r] = 0.f;
    const int co_wave = co0 + grp * 32 + 4 * kh;
    float* yb = a.y + (long long)b * a.y_bs;
    const float* rb = a.res ? a.res + (long long)b * a.res_bs : nullptr;
    const unsigned ybytes = (unsigned)a.Co * (unsigned)HW * 4u;
    __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc((void*)yb, 0, ybytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc((void*)rb, 0, rb ? ybytes : 0u, 0x00020000);
    f32x4* ostats_b = a.ostats ? a.ostats + (long long)b * (a.Co >> (pairs ? 1 : 3)) * a.oslots : nullptr;
    __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(
        (void*)ostats_b, 0, ostats_b ? (unsigned)(a.Co >> (pairs ? 1 : 3)) * (unsigned)a.oslots * 16u : 0u, 0x00020000);
    const unsigned HW4 = (unsigned)HW * 4u, oct_stride = (unsigned)a.oslots * 16u;
    // byte offset of (channel co_wave, image row wq of a strip at row 0, column w0 + l31) in the sample: per lane,
    // constant; the strip's row offset, the sub-tile and the channel are wave-uniform and ride in the SGPR soffset
    const unsigned pv_lane = (unsigned)(co_wave * HW + wq * W + w0 + l31) * 4u;
    auto strip_row = [&](int t) __attribute__((always_inline)) { return (unsigned)((th0 + t) * G::TH * W) * 4u; };
    unsigned p_row = 0;           // bytes: first image row of the PARKED strip
    unsigned ent_off = OOB;       // entry of the parked strip (lane 63, or lanes 31 / 63 for pair entries)
    auto park = [&](int t) __attribute__((always_inline)) {      // strip t is complete
        p_row = strip_row(t);
#pragma unroll
        for (int j = 0; j < TPX; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) accp[j][r] = acc[j][r];
        if constexpr (EMIT != 0) {
            const int slot_id = ((th0 + t) * a.tiles_w + tw_i) * 4 + wq;
            const int co_blk = co0 + grp * 32;
            if constexpr (!pairs)
                ent_off = lane >= 60 ? (unsigned)(co_blk >> 3) * oct_stride + (unsigned)slot_id * 16u + (unsigned)(lane & 3) * 4u : OOB;
            else
                ent_off = l31 >= 28 ? (unsigned)((co_blk >> 1) + 2 * kh) * oct_stride + (unsigned)slot_id * 16u + (unsigned)(lane & 3) * 4u : OOB;
        }
        acc_init();
    };
    // octet m of a strip: 8 values per lane (2 sub-tiles x 4 registers) + its statistics entry, in three pieces so that
    // a stage slot can order its VMEM operations freely: epi_math (VALU: final values and entry fields from the parked
    // accumulators and the residual values), epi_load (the residual values of the NEXT octet; for octet 0 before the
    // strip is even complete: `row` = byte offset of the strip's first image row), epi_store.  `real == false` turns
    // loads and stores into out-of-range operations: every slot issues the same VMEM sequence.
    constexpr int NENT = EMIT == 0 ? 0 : (pairs ? 2 : 1);       // statistics entries per octet and wave
    constexpr int NST = 4 * TPX + NENT;                         // stores per stage slot
    float rq[4 * TPX];
#pragma unroll
    for (int k = 0; k < 4 * TPX; ++k) rq[k] = 0.f;
    auto epi_load = [&](float (&dst)[4 * TPX], int m, unsigned row, bool real) __attribute__((always_inline)) {
        // no residual operand: no loads, and dst keeps the zeros it was initialised with (NOT re-zeroed here: a VALU
        // write of registers that loads may still target makes hipcc drain the VMEM queue first -- a vmcnt(0) in the
        // middle of the compute phase)
        if ((LC_PP_ABL & 32) || !rb) return;
        const unsigned vo = real ? pv_lane : OOB;
#pragma unroll
        for (int k = 0; k < 4 * TPX; ++k)
            dst[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                   rs_r, vo, row + (unsigned)(k >> 2) * 128u + (unsigned)((k & 3) + 8 * m) * HW4, 0));
    };
    auto epi_math = [&](auto mtag, const float (&res)[4 * TPX], float (&ov)[4 * TPX], float (&ev)[2]) __attribute__((always_inline)) {
        constexpr int m = decltype(mtag)::value;
        // (a distinct marker per octet: without it hipcc folds the four instantiations of the statistics-free variant back
        // into one body with a run-time index into the parked set -- which then lives in scratch memory, and every
        // scratch access in a stage slot is a vmcnt(0) that drains the loads just issued)
        asm volatile("; epilogue octet %0" ::"n"(m));
        float st_p = 0.f, st_s = 0.f, st_q = 0.f, st_s2 = 0.f, st_q2 = 0.f;
#pragma unroll
        for (int k = 0; k < 4 * TPX; ++k) {
            const int j = k >> 2, q = k & 3;
            const float v = fmaf(accp[j][4 * m + q], out_unscale, res[k]) * out_scale;
            ov[k] = v;
            if constexpr (EMIT != 0) {
                if (k == 0) st_p = __builtin_amdgcn_readlane(v, 0);
                const float d = v - st_p;
                if (pairs && (q & 2)) { st_s2 += d; st_q2 = fmaf(d, d, st_q2); }
                else { st_s += d; st_q = fmaf(d, d, st_q); }
            }
        }
        ev[0] = ev[1] = 0.f;
        if constexpr (EMIT != 0) {
            // one 32-bit store per entry: lanes R-3 .. R carry the four fields (see DefEpi::store_entry_lanes)
            auto entry = [&](float p_, float n_, float s_, float q_) __attribute__((always_inline)) {
                const float s1 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s_), 0x101, 0xF, 0xF, true));
                const unsigned f = lane & 3;
                return f == 0 ? p_ : (f == 1 ? n_ : (f == 2 ? s1 : q_));
            };
            if constexpr (!pairs) {
                ev[0] = entry(st_p, 8.0f * 32.0f * TPX, wave_sum_to_lane63(st_s), wave_sum_to_lane63(st_q));
            } else {
                ev[0] = entry(st_p, 2.0f * 32.0f * TPX, half_sum_to_lane31_63(st_s), half_sum_to_lane31_63(st_q));
                ev[1] = entry(st_p, 2.0f * 32.0f * TPX, half_sum_to_lane31_63(st_s2), half_sum_to_lane31_63(st_q2));
            }
        }
    };
    auto epi_math_dyn = [&](int m, const float (&res)[4 * TPX], float (&ov)[4 * TPX], float (&ev)[2]) __attribute__((always_inline)) {
        if (m == 0) epi_math(std::integral_constant<int, 0>{}, res, ov, ev);   // (a compile-time octet per call: the
        else if (m == 1) epi_math(std::integral_constant<int, 1>{}, res, ov, ev);   // parked registers are indexed statically)
        else if (m == 2) epi_math(std::integral_constant<int, 2>{}, res, ov, ev);
        else epi_math(std::integral_constant<int, 3>{}, res, ov, ev);
    };
    auto epi_store = [&](int m, const float (&ov)[4 * TPX], const float (&ev)[2], bool real) __attribute__((always_inline)) {
        const unsigned vo = real ? pv_lane : OOB, eo = real ? ent_off : OOB;
        if (LC_PP_ABL & 1) { asm volatile("" ::"v"(ov[0]), "v"(ov[7]), "v"(ev[0]), "v"(ev[1])); return; }
#pragma unroll
        for (int k = 0; k < 4 * TPX; ++k)
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, ov[k]), rs_y, vo,
                                                  p_row + (unsigned)(k >> 2) * 128u + (unsigned)((k & 3) + 8 * m) * HW4, LC_DEF_AUX);
        if constexpr (EMIT == 1) {
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(ev[0]), rs_o, eo, (unsigned)m * oct_stride, 0);
        } else if constexpr (EMIT == 2) {
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(ev[0]), rs_o, eo, (unsigned)(4 * m) * oct_stride, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(ev[1]), rs_o, eo, (unsigned)(4 * m + 1) * oct_stride, 0);
        }
    };

    // ---- one compute phase: 9 taps x 2 sub-tiles x 3 products from the step's x image and weights
    const int xbase = kh * PL + wq * XW + l31;             // sub-tile j: + 32 j
    const int wbase = kh * BN + grp * 32 + l31;
    // Besides its MFMAs a compute phase issues what its own next stage slot must not wait for: the residual values of
    // the octet that slot finalises (R) and -- group 0 only, one 1 KB piece per tap -- the weight DMA of the NEXT step:
    // its buffer was last read by group 1's compute phase of step - 1, which ended with the previous barrier, and
    // group 0 waits for it at the end of its stage slot, a full phase later.
    auto compute = [&](int step, int dma_step, int res_m, unsigned res_row, bool res_real) __attribute__((always_inline)) {
        const half8* cxh = xbuf + (step & 1) * G::XB;
        const half8* cxl = cxh + XUP;
        const half8* cwh = wbuf + (step & 1) * G::WB;
        const half8* cwl = cwh + WU;
        // fragments are pipelined LC_PP_DEPTH taps ahead (one tap = six MFMAs = 192 cycles of cover)
        constexpr int NF = LC_PP_DEPTH + 1;
        half8 ah[NF], al[NF], bh[NF][TPX], bl[NF][TPX];
        auto fetch = [&](int tap) __attribute__((always_inline)) {
            const int dy = tap / 3, dx = tap - dy * 3, s = tap % NF;
            ah[s] = cwh[tap * G::CB * BN + wbase];
            al[s] = cwl[tap * G::CB * BN + wbase];
#pragma unroll
            for (int j = 0; j < TPX; ++j) {
                bh[s][j] = cxh[xbase + 32 * j + dy * XW + dx];
                bl[s][j] = cxl[xbase + 32 * j + dy * XW + dx];
            }
        };
        if (LC_PP_PRIO) __builtin_amdgcn_s_setprio(2);
#pragma unroll
        for (int d = 0; d < LC_PP_DEPTH; ++d) fetch(d);
#pragma unroll
        for (int tap = 0; tap < G::NTAP; ++tap) {
            const int s = tap % NF;
            __builtin_amdgcn_sched_barrier(0);
            if (tap + LC_PP_DEPTH < G::NTAP) fetch(tap + LC_PP_DEPTH);
            if (tap == 1) epi_load(rq, res_m, res_row, res_real);
            if (dma_step >= 0 && tap < LC_PP_DMA0) dma_w_piece(dma_step, tap);
            __builtin_amdgcn_sched_barrier(0);
            if (LC_PP_ABL & 8) {
                asm volatile("" ::"v"(ah[s]), "v"(al[s]), "v"(bh[s][0]), "v"(bl[s][0]), "v"(bh[s][1]), "v"(bl[s][1]));
                continue;
            }
#pragma unroll
            for (int j = 0; j < TPX; ++j) {
                if (LC_F16X2_TERMS & 2) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[s], bh[s][j], acc[j], 0, 0, 0);
                if (LC_F16X2_TERMS & 4) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], bl[s][j], acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], bh[s][j], acc[j], 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (LC_PP_PRIO) __builtin_amdgcn_s_setprio(0);
    };

    // ---- the phase loop.  Both groups run the same straight-line loop  [compute(s); barrier; stage slot; barrier];
    // group 1 enters it one phase (= one barrier) later:
    //     phase      -1                  0                   1                   2
    //     group 0    stage(0)            compute(0)          stage(1) + epi      compute(1)
    //     group 1    stage(0)            stage(1) + DMA(1)   compute(0)          stage(2) + DMA(2) + epi
    // so the stage slot behind compute(s) handles step v = s + 1 + grp, whose x loads rode in compute(s).
    auto phase_barrier = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    int epi_m = 4;                // next octet of the parked strip to finalise (4 = nothing parked)
#if LC_TIMING
    unsigned long long t_comp = 0, t_cwait = 0, t_stage = 0, t_swait = 0, t_drain = 0;
    const unsigned long long t_loop = __builtin_amdgcn_s_memtime();
#define LC_T(var) { const unsigned long long t__ = __builtin_amdgcn_s_memtime(); var += t__ - t_mark; t_mark = t__; }
    unsigned long long t_mark = t_loop;
#else
#define LC_T(var)
#endif
    // VMEM order of a wave over one [compute; stage slot] pair:
    //   compute   R  4 * TPX residual loads of the octet its slot finalises (when there is a residual operand)
    //             D  9 weight-DMA pieces of the next step (group 0)
    //   slot      A  stage_unit: consumes the x loads of the previous slot's B (in flight for two phases)
    //             B  8 x loads (64 bit) of the step the NEXT slot stages
    //                (VALU: final values of this slot's octet)
    //             S  NST stores: the octet + its statistics entries
    // At the end of the slot, vmcnt <= 8 + NST means: R and D have landed (VMEM returns in order), only B and S may
    // still be in flight.
    auto stage_slot = [&](XSet& xs_, int v, int s_done) __attribute__((always_inline)) {     // s_done: the step this group just computed (-1: none yet)
        if (s_done >= 0 && (s_done + 1) % nchunk == 0) { park(s_done / nchunk); epi_m = 0; }
        if (v < NSTEP) {
#pragma unroll
            for (int i = 0; i < NXU; ++i) stage_unit(xs_, i, v);
        }
        __builtin_amdgcn_sched_barrier(0);
        load_x(xs_, v + 1);                                 // the step the next slot stages
        __builtin_amdgcn_sched_barrier(0);
        const bool real = epi_m < 4;
        const int m = epi_m & 3;
        float ov[4 * TPX], ev[2];
        epi_math_dyn(m, rq, ov, ev);
        // the values are final HERE (hipcc otherwise sinks this arithmetic next to the stores and its wait with it)
#pragma unroll
        for (int k = 0; k < 4 * TPX; ++k) asm volatile("" : "+v"(ov[k]));
        asm volatile("" : "+v"(ev[0]), "+v"(ev[1]));
        __builtin_amdgcn_sched_barrier(0);
        if (LC_PP_DMA0 < G::NWD && grp == 1 && v < NSTEP) {      // group 1's share of the pieces of step v (= its own next step)
#pragma unroll
            for (int q = LC_PP_DMA0; q < G::NWD; ++q) dma_w_piece(v, q);
            __builtin_amdgcn_sched_barrier(0);
        }
        epi_store(m, ov, ev, real);
        if (real) ++epi_m;
        // group 0: its pieces (issued in the compute phase before this slot) are older than B and S
        if (LC_PP_DMA0 == G::NWD || grp == 0) wait_vmcnt(8 + NST);
    };
    // phase -1: both halves of step 0's image, the x loads of step 1
#pragma unroll
    for (int i = 0; i < NXU; ++i) stage_unit(xsa, i, 0);
    load_x(xsa, 1);
    wait_vmcnt(8);                // group 0's weight DMA of step 0 (older than the x loads consumed above) has landed
    if (grp == 1) {
        phase_barrier();
        stage_slot(xsa, 1, -1);
    }
    phase_barrier();
    // one compute phase + stage slot; the octet the slot finalises: octet 0 of the strip that completes now, or the next
    // one of the parked strip
    auto iteration = [&](XSet& xs_, int s) __attribute__((always_inline)) {
        const int v = s + 1 + grp;
        const bool parks = (s + 1) % nchunk == 0;
        LC_T(t_swait)
        compute(s, (grp == 0 && s + 1 < NSTEP) ? s + 1 : -1, parks ? 0 : (epi_m & 3), parks ? strip_row(s / nchunk) : p_row,
                parks || epi_m < 4);
        LC_T(t_comp)
        if (LC_PP_DMA0 < G::NWD && grp == 1) {                  // group 1's pieces of its last slot: S and this phase's R behind them
            if (rb) wait_vmcnt(NST + 4 * TPX); else wait_vmcnt(NST);
        }
        phase_barrier();
        LC_T(t_cwait)
        if (grp == 0 || s + 1 < NSTEP) {
            stage_slot(xs_, v, s);
            LC_T(t_stage)
            phase_barrier();
        }
    };
    for (int s = 0; s < NSTEP; ++s) iteration(xsa, s);
    // The last strip drains in the open (group 0: its octet 0 went out in the last stage slot, under group 1's last
    // compute phase; group 1 parks here, its octet 0's residual values arrived during that compute phase).  All
    // residual loads first -- a load behind a store would wait for the store's acknowledgement -- then the stores.
    const bool rq_valid = grp == 1;
    if (grp == 1) { park(NS - 1); epi_m = 0; }
    {
        float rqd[4][4 * TPX];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            if (m == epi_m && rq_valid) {
#pragma unroll
                for (int k = 0; k < 4 * TPX; ++k) rqd[m][k] = rq[k];
            } else {
#pragma unroll
                for (int k = 0; k < 4 * TPX; ++k) rqd[m][k] = 0.f;
                epi_load(rqd[m], m, p_row, m >= epi_m);
            }
        }
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            float ov[4 * TPX], ev[2];
            if (m == 0) epi_math(std::integral_constant<int, 0>{}, rqd[m], ov, ev);
            else if (m == 1) epi_math(std::integral_constant<int, 1>{}, rqd[m], ov, ev);
            else if (m == 2) epi_math(std::integral_constant<int, 2>{}, rqd[m], ov, ev);
            else epi_math(std::integral_constant<int, 3>{}, rqd[m], ov, ev);
            epi_store(m, ov, ev, m >= epi_m);
        }
    }
    LC_T(t_drain)
    publish_amax(a.range, am, amax_seen);
#if LC_TIMING
    if (lane == 0) {
        unsigned long long* d = lc_dbg + 8 * grp;
        atomicAdd(&d[0], __builtin_amdgcn_s_memtime() - t_enter);
        atomicAdd(&d[1], t_loop - t_enter);
        atomicAdd(&d[2], t_comp);
        atomicAdd(&d[3], t_cwait);
        atomicAdd(&d[4], t_stage);
        atomicAdd(&d[5], t_swait);
        atomicAdd(&d[6], 1ull);
        atomicAdd(&d[7], t_drain);
    }
#endif
#undef LC_T
}
