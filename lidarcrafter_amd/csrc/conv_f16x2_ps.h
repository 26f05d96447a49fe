// The pre-split-input convolution kernel (conv_f16x2_ps_kernel), shared by conv_f16x2.hip (3x3 ring conv, stride 1) and
// conv_f16x2_s2.hip (the stride-2 form behind the FIR pre-filter: Block.downsample of EfficientUNet folded into one conv).
// Include inside an anonymous namespace after conv_f16x2_common.h (`using namespace lcconv`).
#pragma once

// ---------------------------------------------------------------------------------------------
// PRE-SPLIT INPUT variant: the activation arrives as two fp16 planes (hi, lo) in channel-octet
// innermost layout  xsp[b][plane][c/8][h][w][8]  -- written by the PRODUCER (the GroupNorm apply
// pass, lc_groupnorm_apply*_split: same 4 bytes per element as fp32 NCHW), already multiplied by
// the layer's x_scale.  Staging is then a pure global -> LDS copy: every wave issues a handful of
// `buffer_load_dwordx4 ... lds` (LDS-DMA, 1 KiB per wave-instruction, no VGPRs, no ds_write, no
// VALU) per K chunk for the x tile (ring / zero halo through per-lane source offsets and the
// descriptor's out-of-range zero fill) and for the packed weights.  The K loop of a wave shrinks
// from ~440 instructions per chunk (250 VALU of hi/lo split + selects, 38 loads, 10 ds_write) to
// the 54 MFMAs, their 54 fragment reads and ~8 DMA issues.  Same block shapes, persistent tiles
// and epilogue (bias / residual / scale / GroupNorm statistics of the output) as
// conv_f16x2_pipe_kernel.  LDS image of a plane = unit index e = (cb, row, col) exactly as there,
// padded to whole waves (the pad lanes read out of range -> zeros).
//
// S2 (round 6): the STRIDE-2 form, y[oy][ox] = sum_{ky,kx,c} w[ky][kx][c] * F[2 oy + ky - 1][2 ox + kx - 1 (ring)][c], over the
// FIR-pre-filtered input F that lc_fir_down2_prefilter_split wrote (conv_f16x2_s2.hip: rows F[-1 .. Hin-1] stored at row index
// + 1, then the two boundary variants; per row the ODD input columns first -- index i = column 2 i - 1 -- then the even ones).
// a.H / a.W are the OUTPUT extents.  The LDS image of a tile row is (ky = 0, 1, 2) x [TW + 1 odd-phase units | TW even-phase
// units]: each tap reads a unit-stride run (kx = 0: odd[p], 1: even[p], 2: odd[p + 1]), and because a tile is ONE output row
// (TH = 1) every staged input row belongs to exactly one ky -- so the two rows whose value depends on ky at the image's top /
// bottom (the FIR's zero padding of the CONV OUTPUT, see conv_f16x2_s2.hip) are simply different source rows.
template <class C, bool EMIT_STATS, bool S2 = false>
__global__ __launch_bounds__(C::NT, C::NT / 256) void conv_f16x2_ps_kernel(ConvArgsH a) {
    constexpr int CB = C::CB, HALO = C::HALO, NTAP = C::NTAP, BN = C::BN;
    constexpr int XR = S2 ? 3 * C::TH_ : C::XR, XW = S2 ? 2 * C::TW_ + 1 : C::XW, XU = CB * XR * XW, WU = C::WU;
    static_assert(!S2 || (C::TH_ == 1 && NTAP == 9), "stride-2 form: 3x3, one output row per tile");
    constexpr int KS = 2 * HALO + 1;
    constexpr int NT = C::NT, NWV = NT / 64;
    constexpr int NXI = (XU + 63) / 64, NWI = (WU + 63) / 64;     // DMA instructions per plane
    constexpr int XS = NXI * 64, WS = NWI * 64;                   // units per plane in LDS
    // DMA slots of a wave per chunk: slots [0, KX) move x (instruction wave + k * NWV of the 2 * NXI
    // x instructions), slots [KX, KX + KW) move weights -- the KIND of a slot is static, so the
    // issue code has no branch; a slot whose instruction index runs past the end reads out of
    // range and lands in a 1 KiB dummy block behind the buffer.
    constexpr int KX = (2 * NXI + NWV - 1) / NWV, KW = (2 * NWI + NWV - 1) / NWV;
    constexpr int IPW = KX + KW;
#ifndef LC_PS_SPT
#define LC_PS_SPT 0   // developer switch: DMA slots issued per tap (0 = spread over all taps: 1 with 8 waves)
#endif
    constexpr int SPT = LC_PS_SPT ? LC_PS_SPT : (IPW + NTAP - 1) / NTAP;   // slots issued per tap
    constexpr int BUF = 2 * XS + 2 * WS + 64;                     // + the dummy block
    constexpr unsigned OOB = 0x80000000u;
    __shared__ half8 lds[2 * BUF];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wco = wave / C::WPX_, wpx = wave % C::WPX_;

    const int tpb = a.tpb;
    int bx = blockIdx.x;
    if (a.xcd) bx = (bx & 7) * (gridDim.x >> 3) + (bx >> 3);
    int tw_i, th_i;
    if (a.vert) {
        const int gh_ = a.tiles_h / tpb;
        tw_i = bx % a.tiles_w; bx /= a.tiles_w;
        th_i = (bx % gh_) * tpb; bx /= gh_;
    } else {
        const int gw_ = a.tiles_w / tpb;
        tw_i = (bx % gw_) * tpb; bx /= gw_;
        th_i = bx % a.tiles_h; bx /= a.tiles_h;
    }
    const int b = bx;
    int h0 = th_i * C::TH_, w0 = tw_i * C::TW_;
    const int dh = a.vert ? C::TH_ : 0, dw = a.vert ? 0 : C::TW_;
    const int co0 = blockIdx.y * BN;
    const int H = a.H, W = a.W;
    const int HW = H * W;
    const int C8 = a.xsp_c8;
    const float out_unscale = a.range->x_unscale * a.wmeta[1];

    // descriptors: both planes of sample b; both weight planes (lo plane follows the hi plane)
    // (S2: a plane of F has 2 H + 3 rows of 2 W units per channel octet)
    const int XROWS = S2 ? 2 * H + 3 : H, XCOLS = S2 ? 2 * W : W;
    const unsigned xbytes = 2u * (unsigned)C8 * (unsigned)(XROWS * XCOLS) * 16u;
    __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.xsp + (long long)b * a.xsp_bs), 0, xbytes, 0x00020000);
    const unsigned wplane = (unsigned)(NTAP * a.Cib) * (unsigned)a.Cop;       // units per weight plane
    __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)a.wh, 0, 2u * wplane * 16u,
                                                                    0x00020000);

    // per-lane source byte offsets (VGPR) and the LDS block of every slot (SGPR, units)
    unsigned voff[IPW];
    int ldsoff[IPW];
#pragma unroll
    for (int k = 0; k < IPW; ++k) {
        if (k < KX) {
            const int j = wave + k * NWV;
            const int plane = j / NXI;
            ldsoff[k] = j < 2 * NXI ? plane * XS + (j - plane * NXI) * 64 : 2 * XS + 2 * WS;
        } else {
            const int j = wave + (k - KX) * NWV;
            const int plane = j / NWI;
            ldsoff[k] = j < 2 * NWI ? 2 * XS + plane * WS + (j - plane * NWI) * 64 : 2 * XS + 2 * WS;
        }
    }
    auto set_offsets = [&](int h0t, int w0t, bool weights_too) {
#pragma unroll
        for (int k = 0; k < IPW; ++k) {
            if (k < KX) {
                const int j = wave + k * NWV;
                const int plane = j / NXI, e = (j - plane * NXI) * 64 + lane;
                const int cb = e / (XR * XW), rem = e - cb * (XR * XW);
                const int r = rem / XW, c = rem - r * XW;
                if constexpr (S2) {
                    const int ky = r;                              // (TH = 1: the tile's output row is h0t)
                    int row = 2 * h0t + ky;                        // F[2 oy + ky - 1] is stored at row 2 oy + ky
                    if (h0t == 0 && ky == 2) row = 2 * H + 1;      // top variant:    F[1] without its f0 * XW[0] term
                    if (h0t == H - 1 && ky == 0) row = 2 * H + 2;  // bottom variant: F[Hin-3] without f3 * XW[Hin-1]
                    int col;
                    if (c <= C::TW_) { col = w0t + c; col = col >= W ? col - W : col; }   // odd phase, ring
                    else col = W + w0t + (c - C::TW_ - 1);                                // even phase
                    const bool ok = j < 2 * NXI && e < XU && h0t < H;
                    voff[k] = ok ? (unsigned)(((plane * C8 + cb) * XROWS + row) * XCOLS + col) * 16u : OOB;
                    continue;
                }
                const int gh = h0t - HALO + r;
                int gw = w0t - HALO + c;
                gw %= W; if (gw < 0) gw += W;
                const bool ok = j < 2 * NXI && e < XU && gh >= 0 && gh < H;
                voff[k] = ok ? (unsigned)(((plane * C8 + cb) * H + gh) * W + gw) * 16u : OOB;
            } else if (weights_too) {
                const int j = wave + (k - KX) * NWV;
                const int plane = j / NWI, e = (j - plane * NWI) * 64 + lane;
                const int row = e / BN, cu = e - row * BN;
                const int tap = row / CB, cb = row - tap * CB;
                voff[k] = (j < 2 * NWI && e < WU)
                              ? ((unsigned)((tap * a.Cib + cb) * a.Cop + co0 + cu) + plane * wplane) * 16u
                              : OOB;
            }
        }
    };
    set_offsets(h0, w0, true);
    const unsigned x_chunk = (unsigned)CB * (unsigned)(XROWS * XCOLS) * 16u;      // bytes between K chunks (x)
    const unsigned w_chunk = (unsigned)CB * (unsigned)a.Cop * 16u;   // ... (weights)
    auto issue_slot = [&](half8* buf, int k, unsigned xso, unsigned wso) {
        if (LC_PS_ABL & 1) return;
        if (k < KX) {
            if (!(LC_PS_ABL & 4)) lds_dma16(rs_x, (lds_vptr)(buf + ldsoff[k]), voff[k], xso);
        } else {
            if (!(LC_PS_ABL & 8)) lds_dma16(rs_w, (lds_vptr)(buf + ldsoff[k]), voff[k], wso);
        }
    };
    auto issue = [&](half8* buf, int ch) {
#pragma unroll
        for (int k = 0; k < IPW; ++k) issue_slot(buf, k, (unsigned)ch * x_chunk, (unsigned)ch * w_chunk);
    };

    f32x16 acc[C::TCO_][C::TPX_];
#pragma unroll
    for (int i = 0; i < C::TCO_; ++i)
#pragma unroll
        for (int j = 0; j < C::TPX_; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int kh = lane >> 5, l31 = lane & 31;
    int xbase[C::TPX_];
#pragma unroll
    for (int j = 0; j < C::TPX_; ++j) {
        const int t = wpx * C::TPX_ + j;
        const int tr = t / C::TPR, tc = t - tr * C::TPR;
        xbase[j] = kh * (XR * XW) + (S2 ? 3 * tr : tr) * XW + tc * 32 + l31;
    }
    const int wbase = kh * BN + wco * C::TCO_ * 32 + l31;

    // One chunk of MFMAs from `cur`; the DMA of the NEXT chunk into `nxt` is issued one slot per tap
    // inside the same stream, so that it lands in the shadow of the matrix pipe (all waves of a
    // block run between the same barriers: whatever they do up front, they do it together and the
    // pipe idles).
    auto compute = [&](const half8* cur, half8* nxt, int nxt_ch) {
        const unsigned xso = (unsigned)nxt_ch * x_chunk, wso = (unsigned)nxt_ch * w_chunk;
        const half8* cxh = cur;
        const half8* cxl = cur + XS;
        const half8* cwh = cur + 2 * XS;
        const half8* cwl = cwh + WS;
        half8 ah[2][C::TCO_], al[2][C::TCO_], bh[2][C::TPX_], bl[2][C::TPX_];
        auto fetch = [&](int tap, int s) {
            const int dy = tap / KS, dx = tap - dy * KS;
#pragma unroll
            for (int i = 0; i < C::TCO_; ++i) {
                ah[s][i] = cwh[tap * CB * BN + wbase + i * 32];
                al[s][i] = cwl[tap * CB * BN + wbase + i * 32];
            }
#pragma unroll
            for (int j = 0; j < C::TPX_; ++j) {
                // (S2: row class ky = dy; kx = 0 / 2 -> odd-phase units p / p + 1, kx = 1 -> even-phase unit p)
                const int o = S2 ? dy * XW + (dx == 1 ? C::TW_ + 1 : (dx == 2 ? 1 : 0)) : dy * XW + dx;
                bh[s][j] = cxh[xbase[j] + o];
                bl[s][j] = cxl[xbase[j] + o];
            }
        };
        // The fragments of tap t+1 are requested BEFORE the MFMAs of tap t and consumed after them
        // (two register sets): hipcc, left alone, sinks every ds_read next to its first use and
        // exposes the LDS latency three times per tap (measured: waves parked 48 % of their
        // lifetime) -- the scheduling fences pin the software pipeline.
        fetch(0, 0);
#pragma unroll
        for (int tap = 0; tap < NTAP; ++tap) {
            const int s = tap & 1;
            __builtin_amdgcn_sched_barrier(0);
            if (tap + 1 < NTAP) fetch(tap + 1, s ^ 1);
#pragma unroll
            for (int q = 0; q < SPT; ++q)
                if (tap * SPT + q < IPW) issue_slot(nxt, tap * SPT + q, xso, wso);
            if (LC_PS_SCHED == 0) __builtin_amdgcn_sched_barrier(0);
            if (LC_F16X2_TERMS & 2) {
#pragma unroll
                for (int i = 0; i < C::TCO_; ++i)
#pragma unroll
                    for (int j = 0; j < C::TPX_; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[s][i], bh[s][j], acc[i][j], 0, 0, 0);
            }
            if (LC_F16X2_TERMS & 4) {
#pragma unroll
                for (int i = 0; i < C::TCO_; ++i)
#pragma unroll
                    for (int j = 0; j < C::TPX_; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s][i], bl[s][j], acc[i][j], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < C::TCO_; ++i)
#pragma unroll
                for (int j = 0; j < C::TPX_; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s][i], bh[s][j], acc[i][j], 0, 0, 0);
            if (LC_PS_SCHED == 1) {
                // one fragment read in the shadow of each MFMA (mask 0x008 MFMA, 0x100 DS read)
#pragma unroll
                for (int q = 0; q < 3 * C::TCO_ * C::TPX_; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };

    half8* cur = lds;
    half8* nxt = lds + BUF;
    const int nchunk = a.Cib / CB;
    // split-K: this block's chunk range [ch_lo, ch_hi)
    const int ksp = a.part ? a.ksplit : 1;
    const int ch_lo = (int)blockIdx.z * nchunk / ksp, ch_hi = ((int)blockIdx.z + 1) * nchunk / ksp;
    const int last = ch_hi - 1;
    issue(cur, ch_lo);
    const int co_wave = co0 + wco * C::TCO_ * 32 + 4 * kh;
    float bias_r[C::TCO_][16];
#pragma unroll
    for (int i = 0; i < C::TCO_; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co_wave + i * 32 + (r & 3) + 8 * (r >> 2);
            bias_r[i][r] = (a.bias && co < a.Co) ? a.bias[co] : 0.0f;
        }
    float res_r[C::TCO_][C::TPX_][16];
    const float* rb = a.res ? a.res + (long long)b * a.res_bs : nullptr;
    // Residual loads / output stores are raw buffer operations (round 5; before: predicated global loads / stores, a branch and
    // a 64-bit address per value): descriptors of sample b (residual: no records when there is none -> zeros); per pixel column j ONE
    // byte offset of (channel co_wave, pixel), the value's channel rides in the scalar offset; an out-of-image pixel
    // or a channel past Co is an out-of-range offset.  (Host side: Co * H * W * 4 < 2^31.)
    const unsigned HW4 = (unsigned)HW * 4u;
    const __amdgpu_buffer_rsrc_t rs_yb = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.y + (long long)b * a.y_bs), 0, (unsigned)a.Co * HW4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_rb = __builtin_amdgcn_make_buffer_rsrc((void*)rb, 0, rb ? (unsigned)a.Co * HW4 : 0u,
                                                                          0x00020000);
    auto px_off = [&](int j) -> unsigned {       // byte offset of (co_wave, pixel j of this wave's tile), or OOB
        const int t = wpx * C::TPX_ + j;
        const int tr = t / C::TPR, tc = t - tr * C::TPR;
        const int gh = h0 + tr, gw = w0 + tc * 32 + l31;
        return (gh < H && gw < W) ? (unsigned)(co_wave * HW + gh * W + gw) * 4u : OOB;
    };
    auto prefetch_res = [&]() {
#pragma unroll
        for (int j = 0; j < C::TPX_; ++j) {
            const unsigned vo = px_off(j);
#pragma unroll
            for (int i = 0; i < C::TCO_; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int cor = i * 32 + (r & 3) + 8 * (r >> 2);
                    res_r[i][j][r] = (LC_PS_ABL & 16) ? 0.0f : __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                        rs_rb, co_wave + cor < a.Co ? vo : OOB, (unsigned)cor * HW4, 0));
                }
        }
    };
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    auto k_iter = [&](int nxt_ch) {
        if (LC_PS_ABL & 2) issue(nxt, nxt_ch);
        else compute(cur, nxt, nxt_ch);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        half8* t = cur; cur = nxt; nxt = t;
    };
    for (int tile = 0; tile < tpb; ++tile) {
        for (int ch = ch_lo; ch < last; ++ch) k_iter(ch + 1);
        if (!a.part) prefetch_res();
        const bool more = tile + 1 < tpb;
        if (more) set_offsets(h0 + dh, w0 + dw, false);
        k_iter(more ? ch_lo : last);               // first chunk of the next tile (or a harmless refill)
        if (a.part) {                              // split-K: raw partial sums, finished by the reduce pass
            float* pb = a.part + ((long long)blockIdx.z * a.B + b) * a.Co * HW;
#pragma unroll
            for (int j = 0; j < C::TPX_; ++j) {
                const int t = wpx * C::TPX_ + j;
                const int tr = t / C::TPR, tc = t - tr * C::TPR;
                const int gh = h0 + tr, gw = w0 + tc * 32 + l31;
                const bool pok = gh < H && gw < W;
                const long long poff = (long long)gh * W + gw;
#pragma unroll
                for (int i = 0; i < C::TCO_; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int co = co_wave + i * 32 + (r & 3) + 8 * (r >> 2);
                        if (pok && co < a.Co) epi_store(&pb[(long long)co * HW + poff], acc[i][j][r] * out_unscale);
                        acc[i][j][r] = 0.0f;
                    }
            }
            h0 += dh; w0 += dw;
            continue;
        }
        // ---- epilogue: as conv_f16x2_pipe_kernel --------------------------------------------
        float st_p[C::TCO_][4], st_s[C::TCO_][4], st_q[C::TCO_][4];
        int nvalid = 0;
#pragma unroll
        for (int j = 0; j < C::TPX_; ++j) {
            const int t = wpx * C::TPX_ + j;
            const int tr = t / C::TPR, tc = t - tr * C::TPR;
            const int gh = h0 + tr, gw = w0 + tc * 32 + l31;
            const bool pok = gh < H && gw < W;
            const unsigned vo_j = pok ? (unsigned)(co_wave * HW + gh * W + gw) * 4u : OOB;
            if constexpr (EMIT_STATS) nvalid += __popcll(__ballot(pok) & 0xFFFFFFFFull);
#pragma unroll
            for (int i = 0; i < C::TCO_; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = co_wave + i * 32 + (r & 3) + 8 * (r >> 2);
                    // (S2: the FIR behind the conv zero-pads the conv's OUTPUT, so the bias reaches the image's first / last
                    //  row through three of the four vertical taps only: 1 - f0 = 1 - f3 = 7/8)
                    float bia = bias_r[i][r];
                    if constexpr (S2) bia *= (h0 == 0 ? 0.875f : 1.0f) + (h0 == H - 1 ? 0.875f : 1.0f) - 1.0f;
                    const float v = ((acc[i][j][r] * out_unscale + bia) + res_r[i][j][r]) *
                                    a.out_scale;
                    if (!(LC_PS_ABL & 16))
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs_yb, co < a.Co ? vo_j : OOB,
                                                              (unsigned)(i * 32 + (r & 3) + 8 * (r >> 2)) * HW4, LC_DEF_AUX);
                    if constexpr (EMIT_STATS && !(LC_EMIT_ABL & 1)) {
                        const int m = r >> 2;
                        if (j == 0 && (r & 3) == 0) {
                            st_p[i][m] = __builtin_amdgcn_readlane(pok ? v : 0.0f, 0);
                            st_s[i][m] = 0.f; st_q[i][m] = 0.f;
                        }
                        const float d = pok ? v - st_p[i][m] : 0.0f;
                        st_s[i][m] += d;
                        st_q[i][m] = fmaf(d, d, st_q[i][m]);
                    }
                    acc[i][j][r] = 0.0f;
                }
            }
        }
        if constexpr (EMIT_STATS && !(LC_EMIT_ABL & 2)) {
            // One 32-bit buffer store per entry, fields spread over the four lanes below the reducing lane (round 5: the
            // first form -- four volatile 32-bit stores from that lane -- compiled to sc0 sc1 stores with a vmcnt(0) behind
            // each, i.e. every entry waited for all of the tile's output stores: +6 ... +11 us per launch,
            // profiles/r05_level0.txt section 7).  Octet entries: lane 63 holds the sums; quad entries (a consumer
            // GroupNorm with 4 / 12 channels per group): lanes 0-31 hold channels 8m .. 8m+3, lanes 32-63 8m+4 .. 8m+7
            // -- the two half-wave sums in lanes 31 / 63, one pivot.
            const int slot = ((h0 / C::TH_) * a.tiles_w + w0 / C::TW_) * C::WPX_ + wpx;
            const int co_blk = co0 + wco * C::TCO_ * 32;
            const bool quads = a.ounit == 4;
            const int ush = quads ? 2 : 3;
            const unsigned ebytes = (unsigned)(a.Co >> ush) * (unsigned)a.oslots * 16u;
            const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(
                (void*)(a.ostats + (long long)b * (a.Co >> ush) * a.oslots), 0, ebytes, 0x00020000);
            const bool mine = quads ? (lane & 31) >= 28 : lane >= 60;
            const float nv = (float)((quads ? 4 : 8) * nvalid);
#pragma unroll
            for (int i = 0; i < C::TCO_; ++i) {
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const int co_oct = co_blk + i * 32 + 8 * m;
                    const float sh = half_sum_to_lane31_63(st_s[i][m]), qh = half_sum_to_lane31_63(st_q[i][m]);
                    const float sf = dpp_add<0x143, 0xC>(sh), qf = dpp_add<0x143, 0xC>(qh);   // lane 63: the wave's sum
                    const int ent = quads ? (co_oct >> 2) + (lane >> 5) : (co_oct >> 3);
                    const unsigned vo = (mine && co_oct < a.Co)
                                            ? ((unsigned)ent * (unsigned)a.oslots + (unsigned)slot) * 16u + 4u * (lane & 3)
                                            : 0x80000000u;
                    store_entry_4lanes(rs_o, st_p[i][m], nv, quads ? sh : sf, quads ? qh : qf, vo);
                }
            }
        }
        h0 += dh; w0 += dw;
    }
}
