// Ring-padded 3x3 / 1x1 convolution on the f16 matrix cores with fp32-class accuracy
// ("f16x2 split"): every fp32 operand is split on the fly into hi + lo halves,
//     x*16 = xh + xl,   w*256 = wh + wl      (xh = fp16(x*16), xl = fp16(x*16 - xh), ...)
// and the product is accumulated in fp32 as  xh*wh + xh*wl + xl*wh  (three
// v_mfma_f32_32x32x16_f16 per 32x32x16 block; the dropped xl*wl term is 2^-22 relative).
// The pre-scales are powers of two held in DEVICE memory (lc_conv_range of the layer for x, the
// header of the packed weights for w): 16 / 256 unless the tensor's magnitude asks for another
// exponent (RANGE SAFETY below); they are undone exactly in the epilogue.
//
// RANGE SAFETY.  fp16 holds |v| < 65504.  Every block tracks the largest |x * x_scale| it stages and
// publishes it with one atomicMax per wave into lc_conv_range::amax_scaled; the host (ops.range_poll)
// reads it after the forward / sampling run: a value >= 2^15 (or so small that the lo halves are
// all subnormal) re-derives the layer's x_scale (target 2^12, sixteen-fold headroom) and the
// forward is re-run -- a result computed from saturated operands is never handed out silently.
// Both kernels use ONE split rule (split_pair below): hi = s truncated to 11 significant bits and
// packed round-toward-zero (saturates at 65504, never inf), lo = fp16(s - hi).
// Per-product relative error ~5e-7 (fp32 rounding: 6e-8) at 3/16 of the fp32-MFMA instruction
// cost: 16x rate / 3 passes = 5.3x the fp32 matrix peak.
//
// Same implicit-GEMM mapping as conv.hip (A rows = 32 output channels, B cols = 32 consecutive W
// pixels) but K = 16 input channels per MFMA, which needs channel-innermost operands: lane l
// supplies A[i=l&31][k=8*(l>>5)..+7], B[k=8*(l>>5)..+7][j=l&31] as one 16-byte vector.
//   weights : packed once per weight version as wh/wl[tap][Ci/8][Co^64][8] (lc_pack_conv_weight_f16x2)
//   input   : stays fp32 NCHW in HBM; the staging pass loads 8 channel planes per pixel
//             (coalesced along W), splits, and writes [cb][row][col] 16-byte units into LDS.
// Reference semantics: ops.Conv2d + ops.Pad, lidargen/models/unets/ops.py:32-49,149-173.
#include "conv_f16x2_common.h"

using namespace lcconv;

namespace {

#if LC_TIMING
__device__ unsigned long long lc_dbg[32];
#endif

#ifndef LC_NP_AUX
#define LC_NP_AUX LC_DEF_AUX     // cache policy of this kernel's output stores (16 = sc1 write-through, as the other epilogues)
#endif
template <class C, bool WIDE = false, bool EMIT = false>   // EMIT: octet statistics entries of the output (1x1 launches, round 5)
__global__ __launch_bounds__(256, (C::BN > 64 && C::NTAP == 9) ? 1 : 2) void conv_f16x2_kernel(ConvArgsH a) {
    constexpr int HALO = C::HALO, NTAP = C::NTAP, BN = C::BN;
    constexpr int XR = C::XR, XW = C::XW;
    constexpr int KS = 2 * HALO + 1;
    // 8-channel blocks per K chunk.  A 1x1 conv has 6 MFMAs per wave and 16 channels: with at most
    // one round of blocks on the chip (token projections, batch 1) its time is the chunks' global-load
    // latency and two barriers each, and WIDE stages 64 (or 32) channels per chunk -- four times
    // fewer barriers, 32 loads per thread in flight (512->512 @8x4x128: 28 -> 21 us, batch 1
    // 512->256: 24 -> 17 us); with several rounds the narrow chunk's higher occupancy wins
    // (128->64 @8x32x1024: 43 vs 65 us), r02y / r02z.
    constexpr int CB = (NTAP == 1 && WIDE) ? ((BN <= 64 && C::TH_ * C::TW_ <= 128) ? 8 : 4) : C::CB;
    constexpr int XU = CB * XR * XW, NXU = (XU + 255) / 256;
    constexpr int WU = NTAP * CB * BN, NWU = (WU + 255) / 256;
    __shared__ half8 lds[2 * XU + 2 * WU];
    __shared__ f32x4 ctab[GN_MAX_C];   // fused input GroupNorm rows (mu, A, B, 0) of sample b
    __shared__ float2 gtab[GN_MAX_G];  // (mean, rstd) per group while the rows are derived
    half8* xh = lds;
    half8* xl = lds + XU;
    half8* wh = lds + 2 * XU;
    half8* wl = lds + 2 * XU + WU;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wco = wave / C::WPX_, wpx = wave % C::WPX_;

    int bx = blockIdx.x;
    const int tw_i = bx % a.tiles_w; bx /= a.tiles_w;
    const int th_i = bx % a.tiles_h; bx /= a.tiles_h;
    const int b = bx;
    const int h0 = th_i * C::TH_, w0 = tw_i * C::TW_;
    const int co0 = blockIdx.y * BN;
    const int H = a.H, W = a.W;
    const long long HW = (long long)H * W;
    const float* xb = a.x + (long long)b * a.x_bs;
    const float xs = (LC_RANGE_ABL & 2) ? 16.0f : a.range->x_scale;
    const float amax_seen = (LC_RANGE_ABL & 2) ? 0.0f : a.range->amax_scaled;
    const float out_unscale = (LC_RANGE_ABL & 2) ? 1.0f / 4096.0f : a.range->x_unscale * a.wmeta[1];
    float am = 0.0f;

    int x_off[NXU];  // (cb << 24 | plane offset) or -1 for padding
#pragma unroll
    for (int i = 0; i < NXU; ++i) {
        const int e = tid + i * 256;
        const int cb = e / (XR * XW);
        const int rem = e - cb * (XR * XW);
        const int r = rem / XW, c = rem - r * XW;
        const int gh = h0 - HALO + r;
        int gw = w0 - HALO + c;
        gw %= W; if (gw < 0) gw += W;
        const bool ok = (e < XU) && gh >= 0 && gh < H;
        x_off[i] = ok ? ((cb << 24) | (gh * W + gw)) : -1;
    }

    float xr[NXU][8];
    half8 whr[NWU], wlr[NWU];
    auto load_chunk = [&](int c0) {
#pragma unroll
        for (int i = 0; i < NXU; ++i) {
            const int cbase = c0 + 8 * (x_off[i] >> 24);
            const float* p = xb + (long long)cbase * HW + (x_off[i] & 0xFFFFFF);
#pragma unroll
            for (int k = 0; k < 8; ++k)
                xr[i][k] = (x_off[i] >= 0 && cbase + k < a.Ci) ? p[(long long)k * HW] : 0.0f;
        }
#pragma unroll
        for (int i = 0; i < NWU; ++i) {
            const int e = tid + i * 256;
            if (e < WU) {
                const int row = e / BN;                    // tap*CB + cb
                const int cu = e - row * BN;
                const int tap = row / CB, cb = row - tap * CB;
                // the last chunk may be partial; a block wider than the packed rows (BN = 128 with
                // Cop = 64) must not read past them -- past the END of the allocation for the last row
                // (a latent out-of-bounds read that faulted once the test order put the weights at the
                // end of a mapped segment, round 3)
                const bool in = (c0 >> 3) + cb < a.Cib && co0 + cu < a.Cop;
                const long long idx = ((long long)tap * a.Cib + (in ? (c0 >> 3) + cb : 0)) * a.Cop + (in ? co0 + cu : 0);
                const half8 z = {0, 0, 0, 0, 0, 0, 0, 0};
                whr[i] = in ? a.wh[idx] : z;
                wlr[i] = in ? a.wl[idx] : z;
            }
        }
    };
    int cur_c0 = 0;   // first channel of the chunk held in xr
    auto store_chunk = [&]() {
#pragma unroll
        for (int i = 0; i < NXU; ++i) {
            const int e = tid + i * 256;
            if (e < XU) {
                if (a.gn && x_off[i] >= 0) {   // padding stays exactly 0 (the reference pads
                                               // AFTER the activation); tails have zero rows
                    const f32x4* g = ctab + cur_c0 + 8 * (x_off[i] >> 24);
#pragma unroll
                    for (int k = 0; k < 8; ++k) xr[i][k] = gn_act(xr[i][k], g[k], a.gn_silu);
                }
                half8 hi, lo;
                split8<true>(xr[i], xs, hi, lo, am);
                xh[e] = hi;
                xl[e] = lo;
            }
        }
#pragma unroll
        for (int i = 0; i < NWU; ++i) {
            const int e = tid + i * 256;
            if (e < WU) { wh[e] = whr[i]; wl[e] = wlr[i]; }
        }
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
        xbase[j] = kh * (XR * XW) + tr * XW + tc * 32 + l31;
    }
    const int wbase = kh * BN + wco * C::TCO_ * 32 + l31;

    const int nchunk = (a.Cib + CB - 1) / CB;
    if (a.gn) {
        if (a.gs.partials) {
            for (int i = tid; i < a.Cgn; i += 256) ctab[i] = gn_row_from_stats(a.gs, xb, b, i, a.Ci, HW);
        } else if (a.seg[0].p) {
            gn_rows_from_ostats<256>(a, b, tid, ctab, gtab);
        } else {
            const f32x4* g = a.gn + (long long)b * a.Cgn;
            for (int i = tid; i < a.Cgn; i += 256) ctab[i] = g[i];
        }
        // zero rows for the channels a partial last chunk stages past the table (x reads 0 there)
        for (int i = a.Cgn + tid; i < nchunk * 8 * CB && i < GN_MAX_C; i += 256) ctab[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        __syncthreads();
    }
    load_chunk(0);
    store_chunk();
    __syncthreads();
    for (int ch = 0; ch < nchunk; ++ch) {
        cur_c0 = (ch + 1) * 8 * CB;
        if (ch + 1 < nchunk) load_chunk((ch + 1) * 8 * CB);
#pragma unroll
        for (int tap = 0; tap < NTAP; ++tap) {
            const int dy = tap / KS, dx = tap - dy * KS;
#pragma unroll
            for (int ks = 0; ks < CB / 2; ++ks) {       // 16 input channels per MFMA K step
                half8 ah[C::TCO_], al[C::TCO_], bh[C::TPX_], bl[C::TPX_];
#pragma unroll
                for (int i = 0; i < C::TCO_; ++i) {
                    ah[i] = wh[(tap * CB + 2 * ks) * BN + wbase + i * 32];
                    al[i] = wl[(tap * CB + 2 * ks) * BN + wbase + i * 32];
                }
#pragma unroll
                for (int j = 0; j < C::TPX_; ++j) {
                    bh[j] = xh[2 * ks * (XR * XW) + xbase[j] + dy * XW + dx];
                    bl[j] = xl[2 * ks * (XR * XW) + xbase[j] + dy * XW + dx];
                }
#pragma unroll
                for (int i = 0; i < C::TCO_; ++i)
#pragma unroll
                    for (int j = 0; j < C::TPX_; ++j) {
                        if (LC_F16X2_TERMS & 2)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
                        if (LC_F16X2_TERMS & 4)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                    }
            }
        }
        __syncthreads();
        if (ch + 1 < nchunk) {
            store_chunk();
            __syncthreads();
        }
    }

    publish_amax(a.range, am, amax_seen);
    // Epilogue on raw buffer operations (round 5): bias and residual loads of a pixel column all go out first, then the
    // values are finished and stored -- no branch, no 64-bit address per value.  (First form: per value a predicated block
    // "bias load, wait, residual load, wait, store" -- 2 x 16 serial memory round trips per thread; the projections of the
    // attention blocks spent most of their 25 us there.)  Out-of-image pixels, channels past Co, "no bias" and "no
    // residual" are out-of-range offsets / empty descriptors (loads return 0).  Same arithmetic order as before.
    constexpr unsigned OOB = 0x80000000u;
    const unsigned HW4 = (unsigned)HW * 4u;
    const float* rb = a.res ? a.res + (long long)b * a.res_bs : nullptr;
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc((void*)(a.y + (long long)b * a.y_bs), 0,
                                                                         (unsigned)a.Co * HW4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc((void*)rb, 0, rb ? (unsigned)a.Co * HW4 : 0u,
                                                                         0x00020000);
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)a.bias, 0,
                                                                         a.bias ? (unsigned)a.Co * 4u : 0u, 0x00020000);
    const int co_wave = co0 + wco * C::TCO_ * 32 + 4 * kh;          // (per lane half)
    float bias_r[C::TCO_][16];
#pragma unroll
    for (int i = 0; i < C::TCO_; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co_wave + i * 32 + (r & 3) + 8 * (r >> 2);
            bias_r[i][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_b, (unsigned)co * 4u, 0, 0));
        }
    float st_p[C::TCO_][4], st_s[C::TCO_][4], st_q[C::TCO_][4];
    int nvalid = 0;
#pragma unroll
    for (int j = 0; j < C::TPX_; ++j) {
        const int t = wpx * C::TPX_ + j;
        const int tr = t / C::TPR, tc = t - tr * C::TPR;
        const int gh = h0 + tr, gw = w0 + tc * 32 + l31;
        const bool pok = gh < H && gw < W;
        const unsigned vo = pok ? (unsigned)(co_wave * (int)HW + gh * W + gw) * 4u : OOB;
        if constexpr (EMIT) nvalid += __popcll(__ballot(pok) & 0xFFFFFFFFull);
        float res_r[C::TCO_][16];
#pragma unroll
        for (int i = 0; i < C::TCO_; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cor = i * 32 + (r & 3) + 8 * (r >> 2);
                res_r[i][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                    rs_r, co_wave + cor < a.Co ? vo : OOB, (unsigned)cor * HW4, 0));
            }
#pragma unroll
        for (int i = 0; i < C::TCO_; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cor = i * 32 + (r & 3) + 8 * (r >> 2);
                float v = acc[i][j][r] * out_unscale;
                v += bias_r[i][r];
                v += res_r[i][r];
                v *= a.out_scale;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs_y,
                                                      co_wave + cor < a.Co ? vo : OOB, (unsigned)cor * HW4, LC_NP_AUX);
                if constexpr (EMIT) {      // per channel octet (m; both lane halves) of this wave's pixels, as the pipelined kernels
                    const int m = r >> 2;
                    if (j == 0 && (r & 3) == 0) {
                        st_p[i][m] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pok ? v : 0.0f), 0));
                        st_s[i][m] = 0.f; st_q[i][m] = 0.f;
                    }
                    const float d = pok ? v - st_p[i][m] : 0.0f;
                    st_s[i][m] += d;
                    st_q[i][m] = fmaf(d, d, st_q[i][m]);
                }
            }
        }
    }
    if constexpr (EMIT) {
        const int slot = (th_i * a.tiles_w + tw_i) * C::WPX_ + wpx;
        const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(a.ostats + (long long)b * (a.Co >> 3) * a.oslots), 0, (unsigned)(a.Co >> 3) * (unsigned)a.oslots * 16u,
            0x00020000);
        const float nv = (float)(8 * nvalid);
        const int co_blk = co0 + wco * C::TCO_ * 32;
#pragma unroll
        for (int i = 0; i < C::TCO_; ++i)
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int co_oct = co_blk + i * 32 + 8 * m;
                const float sf = wave_sum_to_lane63(st_s[i][m]), qf = wave_sum_to_lane63(st_q[i][m]);
                const unsigned vo = (lane >= 60 && co_oct < a.Co)
                                        ? ((unsigned)(co_oct >> 3) * (unsigned)a.oslots + (unsigned)slot) * 16u + 4u * (lane & 3)
                                        : OOB;
                store_entry_4lanes(rs_o, st_p[i][m], nv, sf, qf, vo);
            }
    }
}

// ---------------------------------------------------------------------------------------------
// Pipelined variant (one 4-wave block per CU, LDS double buffered, one barrier per K chunk):
//   iteration ch:  [LDS-DMA weights(ch+1) -> buf^1]  [buffer loads x(ch+2) -> register set B]
//                  MFMAs of chunk ch from buf  ||  split+store of register set A (x(ch+1)) -> buf^1
//   The VALU split / ds_write work sits in the same basic block as the MFMAs, so it issues in
//   the shadow of the matrix pipe instead of between barriers (PMC of the first version: 33 %
//   MFMA busy, 35 % of wave time issuing VALU/LDS, 33 % waiting).
// x loads are raw buffer loads: the descriptor covers exactly the Ci*H*W floats of sample b, so
// channel tails and H padding (offset forced out of range) read as 0 with no per-load predication
// and no 64-bit address arithmetic.

// GNM: 0 = plain input, 1 = GroupNorm(+AdaGN) + SiLU of the input fused into staging, 2 = GroupNorm
// only.  A template parameter, not a branch: the staging arithmetic must sit in the same basic block
// as the MFMAs (as a runtime branch it formed its own block, with every LDS latency of the row
// reads exposed and no MFMA issued meanwhile -- measured r02r: 64->64 @8x32x1024 103 us, 76 us
// with the staging arithmetic removed).
template <class C, int EMIT_STATS, int GNM>
__global__ __launch_bounds__(C::NT, C::NT / 256) void conv_f16x2_pipe_kernel(ConvArgsH a) {
    constexpr int CB = C::CB, HALO = C::HALO, NTAP = C::NTAP, BN = C::BN;
    constexpr int XR = C::XR, XW = C::XW, XU = C::XU, WU = C::WU, NWU = C::NWU;
    constexpr int KS = 2 * HALO + 1;
    constexpr int NT = C::NT, NWV = NT / 64;
#if LC_TIMING
    const unsigned long long t_enter = __builtin_amdgcn_s_memtime();
#endif
    // x staging units: a wave-instruction covers 64 consecutive positions of ONE 8-channel block
    // (wave-unit j = wave + NWV * i  ->  block j / WPC), so the GroupNorm rows of a unit are the same
    // for all lanes and the channel block is a scalar.
    constexpr int PL = XR * XW;                        // positions per 8-channel block
    constexpr int WPC = (PL + 63) / 64;                // wave-units per block
    constexpr int NXU = (CB * WPC + NWV - 1) / NWV;    // units per thread
    // every thread always loads NXU / NWU units (no exec-mask branches inside the K loop: a whole
    // chunk is one basic block); units past the end of a plane are stored to one dummy slot.
    // Weights travel global -> LDS by LDS-DMA (round 3; as in the pre-split kernel): no weight
    // VGPRs, no ds_write, and the ~36 registers this frees are what the deferred epilogue parks a
    // tile's accumulators in.  Each wave issues KW DMA instructions per chunk (1 KiB each, one per
    // tap, taps 0 .. KW-1); instruction index past the end -> out of range -> zeros into a dummy block.
    constexpr int NWI = (WU + 63) / 64, WS = NWI * 64;            // DMA instructions / units per weight plane
    constexpr int KW = (2 * NWI + NWV - 1) / NWV;
    static_assert(KW <= NTAP || NTAP == 1, "one weight DMA slot per tap");
    constexpr int XUP = XU + 1;
    constexpr int BUF = 2 * XUP + 2 * WS + 64;         // half8 units per LDS buffer (+ the dummy block)
    constexpr unsigned OOB_W = 0x80000000u;
    __shared__ half8 lds[2 * BUF];
    __shared__ f32x4 ctab[GN_MAX_C];                   // fused input GroupNorm rows of sample b
    __shared__ float2 gtab[GN_MAX_G];                  // (mean, rstd) per group while they are derived
    // bias / (x_unscale * w_unscale) of the block's channels: the accumulators START from it (the
    // scale is a power of two: exact), so no bias registers are held across the K loop
    __shared__ float bias_s[BN];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wco = wave / C::WPX_, wpx = wave % C::WPX_;

    // Persistent over `tpb` consecutive tiles of one sample: the next tile's first chunk is
    // prefetched under the last chunk of the current tile and the epilogue stores drain under the
    // next tile's MFMAs, so the per-block prologue / epilogue cost (~10 us of the 85 us of a
    // 64->64 @32x1024 launch) is paid once per `tpb` tiles.
    // HBM-side reads of x are what bounds the wide levels (PMC: 185 MB fetched for a 67 MB input
    // with W-walking blocks dealt round-robin to the XCDs: each 66-float row segment touches 4
    // 128-byte lines, each 4-row tile re-reads 2 halo rows, and neighbours sit in other L2s).
    // With `vert` the block walks DOWN H and with `xcd` each XCD owns a contiguous range of
    // (sample, row group, W tile) ids, so at any time an XCD works on complete image rows of one
    // sample: the line overlaps of W-neighbours hit in its L2.
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
    int h0 = th_i * C::TH_, w0 = tw_i * C::TW_;        // first tile of the block
    const int dh = a.vert ? C::TH_ : 0, dw = a.vert ? 0 : C::TW_;
    const int co0 = blockIdx.y * BN;
    const int H = a.H, W = a.W;
    const int HW = H * W;
    const float* xb = a.x + (long long)b * a.x_bs;
    const unsigned nbytes = (unsigned)a.Ci * (unsigned)HW * 4u;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)xb, 0, nbytes, 0x00020000);
    const float xs = a.range->x_scale;
    const float amax_seen = a.range->amax_scaled;
    const float out_unscale = a.range->x_unscale * a.wmeta[1];
    float am = 0.0f;

    int x_rc[NXU];         // (row << 16 | column) of the unit inside the staged tile, -1 = no unit
    int x_cb[NXU];         // 8-channel block of the unit (wave-uniform)
    int x_ok[NXU];         // CURRENT tile: the unit lies inside the image (else it stages zeros)
    unsigned x_voff[NXU];  // byte offset of the unit's first channel in the sample, OOB marker = pad
#pragma unroll
    for (int i = 0; i < NXU; ++i) {
        const int j = wave + NWV * i;
        const int cb = j / WPC;
        const int loc = (j - cb * WPC) * 64 + lane;
        const int r = loc / XW, c = loc - r * XW;
        x_cb[i] = cb < CB ? cb : 0;
        x_rc[i] = (cb < CB && loc < PL) ? ((r << 16) | c) : -1;
    }
    auto set_tile = [&](int h0t, int w0t) {
#pragma unroll
        for (int i = 0; i < NXU; ++i) {
            const int r = (x_rc[i] >> 16) & 0xFFF, c = x_rc[i] & 0xFFFF;
            const int gh = h0t - HALO + r;
            int gw = w0t - HALO + c;
            gw %= W; if (gw < 0) gw += W;
            const bool ok = x_rc[i] >= 0 && gh >= 0 && gh < H;
            x_voff[i] = ok ? (unsigned)(x_cb[i] * 8 * HW + gh * W + gw) * 4u : 0xFFFFFFF0u;
            x_ok[i] = ok;
        }
    };
    set_tile(h0, w0);
    // weight DMA: descriptor over both planes (the lo plane lies behind the hi plane in ONE allocation,
    // checked by the host entry), per-lane source offsets and the LDS block of every slot
    const unsigned wl_delta = (unsigned)((const char*)a.wl - (const char*)a.wh);
    const unsigned wplane_b = (unsigned)(NTAP * a.Cib) * (unsigned)a.Cop * 16u;
    __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)a.wh, 0, wl_delta + wplane_b,
                                                                    0x00020000);
    unsigned voff_w[KW];
    int loff_w[KW];
#pragma unroll
    for (int q = 0; q < KW; ++q) {
        const int j = wave + q * NWV;
        const int plane = j / NWI, e = (j - plane * NWI) * 64 + lane;
        const int row = e / BN, cu = e - row * BN;
        const int tap = row / CB, cb = row - tap * CB;
        loff_w[q] = j < 2 * NWI ? 2 * XUP + plane * WS + (j - plane * NWI) * 64 : 2 * XUP + 2 * WS;
        voff_w[q] = (j < 2 * NWI && e < WU)
                        ? (unsigned)((tap * a.Cib + cb) * a.Cop + co0 + cu) * 16u + plane * wl_delta
                        : OOB_W;
    }
    const unsigned w_chunk = (unsigned)CB * (unsigned)a.Cop * 16u;   // bytes between K chunks
    auto dma_w = [&](half8* buf, int q, int ch) {
        if (!(LC_ABLATE & 2)) lds_dma16(rs_w, (lds_vptr)(buf + loff_w[q]), voff_w[q], (unsigned)ch * w_chunk);
    };

    auto load_x = [&](float (&xr)[NXU][8], int ch) {
#pragma unroll
        for (int i = 0; i < NXU; ++i)
#pragma unroll
            for (int k = 0; k < 8; ++k)
                xr[i][k] = __builtin_bit_cast(
                    float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, x_voff[i],
                                                                (unsigned)(ch * 16 + k) * HW * 4u, 0));
    };

    for (int i = tid; i < BN; i += NT)
        bias_s[i] = (a.bias && co0 + i < a.Co) ? a.bias[co0 + i] * (1.0f / out_unscale) : 0.0f;
    const int kh = lane >> 5, l31 = lane & 31;
    f32x16 acc[C::TCO_][C::TPX_];
    auto acc_init = [&]() {                            // (first call: behind the prologue's barriers)
#pragma unroll
        for (int i = 0; i < C::TCO_; ++i)
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const f32x4 bq = *reinterpret_cast<const f32x4*>(&bias_s[(wco * C::TCO_ + i) * 32 + 8 * m + 4 * kh]);
#pragma unroll
                for (int j = 0; j < C::TPX_; ++j) {
                    acc[i][j][4 * m] = bq.x; acc[i][j][4 * m + 1] = bq.y;
                    acc[i][j][4 * m + 2] = bq.z; acc[i][j][4 * m + 3] = bq.w;
                }
            }
    };

    int xbase[C::TPX_];
#pragma unroll
    for (int j = 0; j < C::TPX_; ++j) {
        const int t = wpx * C::TPX_ + j;
        const int tr = t / C::TPR, tc = t - tr * C::TPR;
        xbase[j] = kh * (XR * XW) + tr * XW + tc * 32 + l31;
    }
    const int wbase = kh * BN + wco * C::TCO_ * 32 + l31;

    // fused GroupNorm rows in LDS, per channel PAIR: (A0, A1, B0, B1) with A = rstd * gamma' * xs,
    // B = (beta' - mu * A') * xs  (x_scale is a power of two: folding it here is exact), so that a
    // pair of channels costs one broadcast ds_read_b128 and one v_pk_fma_f32;  4 zero quads behind
    // the table serve the units outside the image (they must stage exact zeros).
    const float silu_c = -1.4426950408889634f * a.range->x_unscale;   // exp2(silu_c * (y * xs)) = exp(-y)
    // A unit (8 channels of one position) is staged in four channel-pair steps spread over the taps
    // of the chunk (2 steps per tap: ~25 VALU in the shadow of 6 MFMAs), then written.
    half8 st_hi[NXU], st_lo[NXU];
    auto read_row = [&](int i, int q, int ch) -> f32x4 {
        if constexpr (GNM != 0) {
            const f32x4* g = x_ok[i] ? ctab + ch * 8 + x_cb[i] * 4 : ctab + (a.Cgn >> 1);
            return g[q];
        } else {
            return f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto stage_pair = [&](float (&xr)[NXU][8], int i, int q, const f32x4 row) {
        if (LC_ABLATE & 1) { asm volatile("" ::"v"(xr[i][2 * q]), "v"(xr[i][2 * q + 1])); return; }
        h2_t ph, pl;
        if constexpr (GNM != 0) {
            f2_t v = {xr[i][2 * q], xr[i][2 * q + 1]};
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
            split_pair<false>(xr[i][2 * q], xr[i][2 * q + 1], xs, ph, pl, am);
        }
        st_hi[i][2 * q] = ph.x; st_hi[i][2 * q + 1] = ph.y;
        st_lo[i][2 * q] = pl.x; st_lo[i][2 * q + 1] = pl.y;
    };
    auto commit_x = [&](half8* buf, int i) {
        if (LC_ABLATE & 1) return;
        const int d = x_rc[i] >= 0 ? x_cb[i] * PL + ((x_rc[i] >> 16) & 0xFFF) * XW + (x_rc[i] & 0xFFFF) : XU;
        buf[d] = st_hi[i];
        buf[XUP + d] = st_lo[i];
    };
    constexpr int NSTEP = 4 * NXU;
    constexpr int SPT = NTAP == 1 ? NSTEP : ((NSTEP + NTAP - 2) / (NTAP - 1) > 2 ? (NSTEP + NTAP - 2) / (NTAP - 1) : 2);
    auto step_tap = [](int st) {   // constexpr-foldable: the steps end one tap before the last
        if (NTAP == 1) return 0;
        const int ntaps = (NSTEP + SPT - 1) / SPT;
        const int first = NTAP - 1 - ntaps > 0 ? NTAP - 1 - ntaps : 0;
        return first + st / SPT;
    };
    // one chunk of MFMAs from `cur`; the split+store of the NEXT chunk's registers into `nxt`
    // is spread over the taps (same basic block as the MFMAs)
    typedef DefEpi<C, EMIT_STATS> DE;
    DE de;
    // dslot0 >= 0: the taps of this chunk also carry slots dslot0 .. dslot0 + NTAP - 1 of the parked
    // tile's deferred epilogue (a compile-time constant at every call site)
    auto compute = [&](const half8* cur, half8* nxt, float (&xr)[NXU][8], int chn, int dslot0) {
        const half8* cxh = cur;
        const half8* cxl = cur + XUP;
        const half8* cwh = cur + 2 * XUP;
        const half8* cwl = cwh + WS;
        // operand fragments are software pipelined ONE TAP AHEAD (two register sets, static
        // parity after unrolling): with one wave per SIMD nothing else hides the LDS latency.
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
                bh[s][j] = cxh[xbase[j] + dy * XW + dx];
                bl[s][j] = cxl[xbase[j] + dy * XW + dx];
            }
        };
        fetch(0, 0);
#if LC_DMA_FIRST
        // All weight-DMA pieces of the chunk go out BEFORE its first deferred-epilogue slot: VMEM returns in order, so a
        // piece issued behind a tap's stores (one per tap, the round-3 placement) makes the wait in front of the chunk
        // barrier a wait for those stores' acknowledgements -- in the fourth peeled chunk of a tile a vmcnt(0).
        if (NTAP != 1) {
#pragma unroll
            for (int q = 0; q < KW; ++q) dma_w(nxt, q, chn);
        }
#endif
#pragma unroll
        for (int tap = 0; tap < NTAP; ++tap) {
            const int s = (LC_ABLATE & 4) ? 0 : (tap & 1);
            // rows of the fused norm for THIS tap's staging steps: requested ahead of the fragment
            // fetch, so that their first use waits with lgkmcnt(#fragment reads), not lgkmcnt(0)
            // (which drained the prefetched fragments eight times per chunk; ISA of r02)
            f32x4 rows[SPT];
            if (LC_PIPE_ROWS && GNM != 0) {
#pragma unroll
                for (int st = 0; st < NSTEP; ++st)
                    if (tap == step_tap(st)) rows[st % SPT] = read_row(st >> 2, st & 3, chn);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (tap + 1 < NTAP && !(LC_ABLATE & 4)) fetch(tap + 1, s ^ 1);
            __builtin_amdgcn_sched_barrier(0);
            if (dslot0 >= 0 && dslot0 + tap < DE::NUSED) de.slot(dslot0 + tap);
            // this tap's weight DMA of chunk chn (behind the deferred slot: the wait in front of the
            // chunk barrier then leaves exactly the later taps' deferred loads / stores in flight)
            if (NTAP == 1) {
#pragma unroll
                for (int q = 0; q < KW; ++q) dma_w(nxt, q, chn);
            } else if (tap < KW && !LC_DMA_FIRST) {
                dma_w(nxt, tap, chn);
            }
            // loads of this chunk's successor were issued before tap 0; consume them as late as
            // possible: x units over taps [T0, T0+NXU), weight units over the last taps.
            // (A finer sched_group_barrier interleave was measured 3-10 % slower than letting
            // hipcc schedule each tap region on its own.)
#pragma unroll
            for (int st = 0; st < NSTEP; ++st)
                if (tap == step_tap(st)) {
                    stage_pair(xr, st >> 2, st & 3,
                               (LC_PIPE_ROWS && GNM != 0) ? rows[st % SPT] : read_row(st >> 2, st & 3, chn));
                    if ((st & 3) == 3) commit_x(nxt, st >> 2);
                }
            if (LC_ABLATE & 32) {   // no MFMAs: keep the fragments alive
#pragma unroll
                for (int i = 0; i < C::TCO_; ++i) asm volatile("" ::"v"(ah[s][i]), "v"(al[s][i]));
#pragma unroll
                for (int j = 0; j < C::TPX_; ++j) asm volatile("" ::"v"(bh[s][j]), "v"(bl[s][j]));
                __builtin_amdgcn_sched_barrier(0);
                continue;
            }
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
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    half8* cur = lds;
    half8* nxt = lds + BUF;
    const int nchunk = a.Cib / CB;
    const int last = nchunk - 1;
    float xr[NXU][8];
    // prologue: chunk 0 -> cur
    load_x(xr, 0);
#pragma unroll
    for (int q = 0; q < KW; ++q) dma_w(cur, q, 0);
    if constexpr (GNM != 0) {   // rows of the fused input norm, derived while the chunk-0 loads are in flight
        if (a.gs.partials) {
            for (int i = tid; i < a.Cgn; i += NT) ctab[i] = gn_row_from_stats(a.gs, xb, b, i, a.Ci, HW);
        } else if (a.seg[0].p) {
            gn_rows_from_ostats<NT>(a, b, tid, ctab, gtab);
        } else {
            const f32x4* g = a.gn + (long long)b * a.Cgn;
            for (int i = tid; i < a.Cgn; i += NT) ctab[i] = g[i];
        }
        __syncthreads();                                // (mu, A, B, 0) rows visible
        // ... repacked in place into the pair quads the staging code reads
        constexpr int NQ = (GN_MAX_C / 2 + NT - 1) / NT;
        const int npair = a.Cgn >> 1;
        f32x4 qd[NQ];
#pragma unroll
        for (int k = 0; k < NQ; ++k) {
            const int p = tid + k * NT;
            if (p < npair) {
                const f32x4 r0 = ctab[2 * p], r1 = ctab[2 * p + 1];
                qd[k] = f32x4{r0.y * xs, r1.y * xs, fmaf(-r0.x, r0.y, r0.z) * xs, fmaf(-r1.x, r1.y, r1.z) * xs};
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < NQ; ++k) {
            const int p = tid + k * NT;
            if (p < npair) ctab[p] = qd[k];
        }
        if (tid < 4) ctab[npair + tid] = f32x4{0.f, 0.f, 0.f, 0.f};
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < NXU; ++i) {
#pragma unroll
        for (int q = 0; q < 4; ++q) stage_pair(xr, i, q, read_row(i, q, 0));
        commit_x(cur, i);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the prologue's weight DMA has landed
    __syncthreads();
    const int co_wave = co0 + wco * C::TCO_ * 32 + 4 * kh;
    float* yb = a.y + (long long)b * a.y_bs;
    const float* rb = (a.res && !(LC_ABLATE & 16)) ? a.res + (long long)b * a.res_bs : nullptr;
    de.init(yb, rb, a.Co, HW, out_unscale, a.out_scale, co_wave,
            a.ostats ? a.ostats + (long long)b * (a.Co >> (a.ounit == 2 ? 1 : 3)) * a.oslots : nullptr, a.oslots,
            a.ounit);
    acc_init();
#if LC_TIMING
    unsigned long long t_comp = 0, t_bar = 0, t_epi = 0, n_chunk = 0;
    const unsigned long long t_start = __builtin_amdgcn_s_memtime();
#endif
    auto k_iter = [&](int nxt_ch, int dslot0) {
        // issue the loads of the next chunk (unconditional), run the MFMAs of the chunk in `cur`
        // (+ the deferred epilogue slots of the parked tile) and stage the loaded chunk into the
        // other buffer during the last taps; one barrier.
        if (!(LC_ABLATE & 2)) load_x(xr, nxt_ch);
        __builtin_amdgcn_sched_barrier(0);   // keep the prefetch ahead of the MFMAs (hipcc would
                                             // otherwise sink every load next to its use)
#if LC_TIMING
        const unsigned long long ta = __builtin_amdgcn_s_memtime();
#endif
        compute(cur, nxt, xr, nxt_ch, dslot0);
#if LC_TIMING
        const unsigned long long tb = __builtin_amdgcn_s_memtime();
#endif
        // Chunk barrier.  NOT __syncthreads(): its workgroup release fence makes hipcc drain every
        // pending LDS-DMA with vmcnt(0), which would also wait for the deferred epilogue's
        // write-through stores to be acknowledged -- the latency this kernel is built to hide.
        // Needed: this wave's ds_writes done (lgkmcnt), its weight DMAs landed = everything but the
        // deferred loads / stores issued in the taps after the last DMA slot (VMEM returns in order).
        {
            int later = 0;
#pragma unroll
            for (int tap = (NTAP == 1 ? 1 : (LC_DMA_FIRST ? 0 : KW)); tap < NTAP; ++tap) {
                const int sl = dslot0 + tap;
                if (dslot0 >= 0 && sl < DE::NUSED) later += DE::ops_of(sl);
            }
            wait_vmcnt(later);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#if LC_TIMING
        t_comp += tb - ta; t_bar += __builtin_amdgcn_s_memtime() - tb; ++n_chunk;
#endif
        half8* t = cur; cur = nxt; nxt = t;
    };
    for (int tile = 0; tile < tpb; ++tile) {
        const bool more = tile + 1 < tpb;
        // the chunk that is staged while chunk ch runs: ch + 1, or the first chunk of the next tile
        // (whose x offsets are set right before; uniform branches, between two chunks)
        auto iter = [&](int ch, int dslot0) {
            if (ch == last && more) set_tile(h0 + dh, w0 + dw);
            k_iter(ch == last ? (more ? 0 : last) : ch + 1, dslot0);
        };
        // the first DCH chunks are peeled: their taps carry the parked tile's deferred epilogue
#pragma unroll
        for (int c = 0; c < DE::DCH; ++c)
            if (c < nchunk) iter(c, c * NTAP);
        if (nchunk < DE::DCH) {                        // short K (Ci = 32): the rest of the parked tile now
            if (nchunk == 1) de.flush_from(NTAP);
            else if (nchunk == 2) de.flush_from(2 * NTAP);
            else de.flush_from(3 * NTAP);
        }
        for (int ch = DE::DCH; ch < nchunk; ++ch) iter(ch, -1);
        // park this tile; the accumulators restart from the bias
        de.begin(acc, h0, w0, H, W, wpx, lane, a.tiles_w, HW, co0 + wco * C::TCO_ * 32, more);
        if (more) acc_init();
        h0 += dh; w0 += dw;
    }
#if LC_TIMING
    const unsigned long long te0 = __builtin_amdgcn_s_memtime();
#endif
    de.drain(acc);                                     // the last tile drains in the open
#if LC_TIMING
    t_epi = __builtin_amdgcn_s_memtime() - te0;
#endif
    publish_amax(a.range, am, amax_seen);
#if LC_TIMING
    if (lane == 0) {
        atomicAdd(&lc_dbg[0], __builtin_amdgcn_s_memtime() - t_start);   // wave lifetime from the prologue's end
        atomicAdd(&lc_dbg[1], t_start - t_enter);
        atomicAdd(&lc_dbg[2], t_comp);
        atomicAdd(&lc_dbg[3], t_bar);
        atomicAdd(&lc_dbg[4], t_epi);
        atomicAdd(&lc_dbg[5], n_chunk);
        atomicAdd(&lc_dbg[6], 1ull);
    }
#endif
}


#include "conv_f16x2_pp.h"

#include "conv_f16x2_ps.h"

template <class C>
int launch_pipe(ConvArgsH a, hipStream_t st) {
    a.tiles_h = (a.H + C::TH_ - 1) / C::TH_;
    a.tiles_w = (a.W + C::TW_ - 1) / C::TW_;
    const int ncot = (a.Co + C::BN - 1) / C::BN;
    // up to 4 consecutive tiles per block while every CU still gets a block; walk down H when
    // that allows as many tiles as walking along W (see the kernel header comment)
    const long long n_tiles = (long long)a.B * a.tiles_h * a.tiles_w * ncot;
    auto max_tpb = [&](int extent) {
        int t = 1;
        while (t < 4 && extent % (t * 2) == 0 && n_tiles / (t * 2) >= 256) t *= 2;
        return t;
    };
    static const int vert_env = [] { const char* e = getenv("LC_CONV_VERT"); return e ? atoi(e) : 1; }();
    static const int xcd_env = [] { const char* e = getenv("LC_CONV_XCD"); return e ? atoi(e) : 1; }();
    const int tpb_h = max_tpb(a.tiles_h), tpb_w = max_tpb(a.tiles_w);
    int vert = vert_env && tpb_h >= tpb_w;
    int tpb = vert ? tpb_h : tpb_w;
    if (C::NTAP == 1) tpb = 1;   // 1x1: a chunk is 6 MFMAs per wave, block-level parallelism wins
    if (a.tpb > 0) {                                     // explicit override (tests): tpb*100 + cfg
        tpb = a.tpb;
        vert = vert_env && a.tiles_h % tpb == 0;
        if (!vert && a.tiles_w % tpb) tpb = 1;
    }
    if (a.part) tpb = 1;                                 // split-K blocks own one tile
    a.tpb = tpb;
    a.vert = vert;
    dim3 grid(a.B * a.tiles_h * a.tiles_w / tpb, ncot);
    a.xcd = (xcd_env && grid.x % 8 == 0 && grid.x >= 16) ? 1 : 0;

    if (a.xsp) {
        if constexpr (C::NTAP == 9) {
            if (a.part) grid.z = a.ksplit;
            if (a.ostats && !a.part) hipLaunchKernelGGL((conv_f16x2_ps_kernel<C, true>), grid, dim3(C::NT), 0, st, a);
            else hipLaunchKernelGGL((conv_f16x2_ps_kernel<C, false>), grid, dim3(C::NT), 0, st, a);
            return lc_launch_status();
        } else {
            return LC_EUNSUP;
        }
    }
    {   // the weight DMA addresses the lo plane through the hi plane's descriptor: one allocation
        const long long d = (const char*)a.wl - (const char*)a.wh;
        if (d <= 0 || d >= (1ll << 31)) return LC_EINVAL;
        if (d + (long long)C::NTAP * a.Cib * a.Cop * 16 >= (1ll << 31)) return LC_EUNSUP;   // 32-bit offsets in the descriptor
    }
    const int gnm = a.gn ? (a.gn_silu ? 1 : 2) : 0;
#define LC_PIPE_LAUNCH(E, G) hipLaunchKernelGGL((conv_f16x2_pipe_kernel<C, E, G>), grid, dim3(C::NT), 0, st, a)
    if (a.ostats && a.ounit == 2) {                    // pair entries: 3x3 only
        if constexpr (C::NTAP == 9) {
            if (gnm == 1) LC_PIPE_LAUNCH(2, 1); else if (gnm == 2) LC_PIPE_LAUNCH(2, 2); else LC_PIPE_LAUNCH(2, 0);
        } else {
            return LC_EUNSUP;
        }
    } else if (a.ostats) {
        if (gnm == 1) LC_PIPE_LAUNCH(1, 1); else if (gnm == 2) LC_PIPE_LAUNCH(1, 2); else LC_PIPE_LAUNCH(1, 0);
    } else {
        if (gnm == 1) LC_PIPE_LAUNCH(0, 1); else if (gnm == 2) LC_PIPE_LAUNCH(0, 2); else LC_PIPE_LAUNCH(0, 0);
    }
#undef LC_PIPE_LAUNCH
    return lc_launch_status();
}

template <class C>
int launch_h(ConvArgsH a, hipStream_t st) {
    a.tiles_h = (a.H + C::TH_ - 1) / C::TH_;
    a.tiles_w = (a.W + C::TW_ - 1) / C::TW_;
    dim3 grid(a.B * a.tiles_h * a.tiles_w, (a.Co + C::BN - 1) / C::BN);
    if constexpr (C::NTAP == 1) {
        const bool wide = (long long)grid.x * grid.y <= 512;
        if (a.ostats) {                        // octet entries (1x1 launches only: lc_conv2d_ring_f16x2_stats_slots)
            if (a.ounit != 8) return LC_EUNSUP;
            if (wide) hipLaunchKernelGGL((conv_f16x2_kernel<C, true, true>), grid, dim3(256), 0, st, a);
            else hipLaunchKernelGGL((conv_f16x2_kernel<C, false, true>), grid, dim3(256), 0, st, a);
            return lc_launch_status();
        }
        if (wide) {
            hipLaunchKernelGGL((conv_f16x2_kernel<C, true>), grid, dim3(256), 0, st, a);
            return lc_launch_status();
        }
    } else if (a.ostats) {
        return LC_EUNSUP;
    }
    hipLaunchKernelGGL((conv_f16x2_kernel<C, false>), grid, dim3(256), 0, st, a);
    return lc_launch_status();
}

// Ping-pong kernel (conv_f16x2_pp.h): shapes it takes, strips per block, launch.
bool pp_eligible(int Ci, int Co, int H, int W, int ks) {
    return ks == 3 && Ci % 16 == 0 && Ci >= 64 && Ci <= PPG::MAX_C && Co % 64 == 0 && H % 4 == 0 && W % 64 == 0;
}
int launch_pp(ConvArgsH a, hipStream_t st) {
    if (!pp_eligible(a.Ci, a.Co, a.H, a.W, 3) || a.xsp || a.part) return LC_EUNSUP;
    if (a.gn && a.Cgn > PPG::MAX_C) return LC_EUNSUP;
    a.tiles_h = a.H / PPG::TH;
    a.tiles_w = a.W / PPG::TW;
    const int ncot = a.Co / PPG::BN;
    // strips per block (walking down H): as many as leave one block per CU -- the epilogue of every strip but a
    // block's last is hidden under the next strip
    const long long strips = (long long)a.B * a.tiles_h * a.tiles_w * ncot;
    int ns = 1;
    while (ns < 8 && a.tiles_h % (2 * ns) == 0 && strips / (2 * ns) >= 256) ns *= 2;
    if (a.tpb > 0 && a.tiles_h % a.tpb == 0) ns = a.tpb;           // explicit override (tests): tpb * 100 + cfg
    a.tpb = ns;
    dim3 grid(a.B * a.tiles_h * a.tiles_w / ns, ncot);
    static const int xcd_env = [] { const char* e = getenv("LC_CONV_XCD"); return e ? atoi(e) : 1; }();
    a.xcd = (xcd_env && grid.x % 8 == 0 && grid.x >= 16) ? 1 : 0;
    a.vert = 1;
    {
        const long long d = (const char*)a.wl - (const char*)a.wh;
        if (d <= 0 || d >= (1ll << 31)) return LC_EINVAL;
        if (d + (long long)PPG::NTAP * a.Cib * a.Cop * 16 >= (1ll << 31)) return LC_EUNSUP;
    }
    const int gnm = a.gn ? (a.gn_silu ? 1 : 2) : 0;
#define LC_PP_LAUNCH(E, G) hipLaunchKernelGGL((conv_f16x2_pp_kernel<E, G>), grid, dim3(PPG::NT), 0, st, a)
    if (a.ostats && a.ounit == 2) {
        if (gnm == 1) LC_PP_LAUNCH(2, 1); else if (gnm == 2) LC_PP_LAUNCH(2, 2); else LC_PP_LAUNCH(2, 0);
    } else if (a.ostats) {
        if (gnm == 1) LC_PP_LAUNCH(1, 1); else if (gnm == 2) LC_PP_LAUNCH(1, 2); else LC_PP_LAUNCH(1, 0);
    } else {
        if (gnm == 1) LC_PP_LAUNCH(0, 1); else if (gnm == 2) LC_PP_LAUNCH(0, 2); else LC_PP_LAUNCH(0, 0);
    }
#undef LC_PP_LAUNCH
    return lc_launch_status();
}

template <int KS>
int dispatch_h(int cfg, const ConvArgsH& a, hipStream_t st) {
    switch (cfg) {
        case 1: return launch_h<HCfg<2, 2, 2, 2, 2, 64, KS>>(a, st);   // 128 co x 128 px
        case 2: return launch_h<HCfg<1, 4, 2, 2, 4, 64, KS>>(a, st);   //  64 co x 256 px
        case 3: return launch_h<HCfg<2, 2, 1, 1, 2, 32, KS>>(a, st);   //  64 co x  64 px
        case 4: return launch_h<HCfg<2, 2, 2, 2, 4, 32, KS>>(a, st);   // 128 co x 128 px (4x32)
        case 5: return launch_h<HCfg<1, 4, 2, 1, 2, 64, KS>>(a, st);   //  64 co x 128 px
        case 12: return launch_pipe<HCfg<1, 4, 2, 2, 4, 64, KS>>(a, st);  // pipelined 64 co x 256 px
        case 13: return launch_pipe<HCfg<2, 2, 1, 1, 2, 32, KS>>(a, st);  // pipelined 64 co x  64 px
        case 15: return launch_pipe<HCfg<1, 4, 2, 1, 2, 64, KS>>(a, st);  // pipelined 64 co x 128 px
        case 22: return launch_pipe<HCfg<1, 8, 2, 1, 4, 64, KS>>(a, st);  // 8 waves, 64 co x 256 px
        case 23: return launch_pipe<HCfg<2, 4, 1, 2, 4, 64, KS>>(a, st);  // 8 waves, 64 co x 256 px (1x2)
        case 25: return launch_pipe<HCfg<2, 4, 1, 1, 2, 64, KS>>(a, st);  // 8 waves, 64 co x 128 px
        case 27:   // tall kernel (kept halo rows, transposed accumulators); shapes it does not take: the pipelined tile
            if (KS == 3 && tall_eligible(a)) return launch_tall(a, st);
            return launch_pipe<HCfg<2, 4, 1, 2, 4, 64, KS>>(a, st);
        case 28: return launch_pipe<HCfg<1, 8, 1, 1, 4, 64, KS>>(a, st);  // 8 waves, 32 co x 256 px (Co <= 32)
        case 33: return KS == 3 ? launch_pp(a, st) : LC_EUNSUP;           // ping-pong wave groups, 64 co x 256 px
        default: return LC_EUNSUP;
    }
}

// wave tiles per (sample, channel) plane of the pipelined configurations = statistics entries per
// channel octet; 0 for the configurations of the other kernel (they emit no statistics)
int pipe_stat_slots(int cfg, int H, int W, int ks = 3) {
    int th, tw, wpx;
    switch (cfg) {
        // the non-pipelined kernel writes entries for 1x1 launches only (round 5): its tiles 1 ... 5
        case 1: if (ks != 1) return 0; th = 2; tw = 64; wpx = 2; break;
        case 2: if (ks != 1) return 0; th = 4; tw = 64; wpx = 4; break;
        case 3: if (ks != 1) return 0; th = 2; tw = 32; wpx = 2; break;
        case 4: if (ks != 1) return 0; th = 4; tw = 32; wpx = 2; break;
        case 5: if (ks != 1) return 0; th = 2; tw = 64; wpx = 4; break;
        case 12: th = 4; tw = 64; wpx = 4; break;
        case 13: th = 2; tw = 32; wpx = 2; break;
        case 15: th = 2; tw = 64; wpx = 4; break;
        case 22: th = 4; tw = 64; wpx = 8; break;
        case 23: th = 4; tw = 64; wpx = 4; break;
        case 25: th = 2; tw = 64; wpx = 4; break;
        case 27: th = 4; tw = 64; wpx = 4; break;
        case 28: th = 4; tw = 64; wpx = 8; break;
        case 33: th = 4; tw = 64; wpx = 4; break;
        default: return 0;
    }
    return ((H + th - 1) / th) * ((W + tw - 1) / tw) * wpx;
}

int auto_cfg_h(int B, int Ci, int Co, int H, int W, int ks) {
    // Measured on MI355X at batch 8 (profiles/r01_c_conv_sweep_f16x2_b8.txt).  Two kernels:
    //  * conv_f16x2_kernel (cfg 2/5/3): 2 blocks per CU, single LDS buffer -- best for short K
    //    (Ci <= 128): 200-278 TF effective;
    //  * conv_f16x2_pipe_kernel (cfg 12/15/13): 1 block per CU, LDS double buffered, prefetch of
    //    the next K chunk overlapped with the MFMAs -- best for long K (Ci >= 256): 260-310 TF,
    //    and the only one that keeps the 4x128-pixel level above 200 TF.
    const long long px = (long long)B * H * W;
    auto blocks = [&](int bn, int pxb) { return ((Co + bn - 1) / bn) * ((px + pxb - 1) / pxb); };
    const bool t256 = H % 4 == 0 && W % 64 == 0, t128 = H % 2 == 0 && W % 64 == 0;
    if (ks == 1) {  // one tap per chunk: nothing to pipeline, two resident blocks per CU win (1.3-2x) ...
        // ... except when the 256-pixel tiles are exactly one round of blocks: then the pipelined kernel
        // with the deferred epilogue is ahead (256 -> 256 @ 8 x 8 x 256 + residual: 23.0 vs 37.5 us,
        // 512 -> 256: 34.3 vs 38.1, profiles/r03_conv_phases.txt r03m) and can emit GroupNorm statistics
        if (t256 && blocks(64, 256) > 192 && blocks(64, 256) <= 256) return 23;
        return (t128 && blocks(64, 128) >= 512) ? 5 : 3;
    }
    if (Ci >= 24) {    // 8-wave pipelined, persistent blocks: 230-333 TF when >= 1 block per CU
                       // exists (also the 32-channel input layer: 41 vs 70 us on the other kernel)
        if (Co <= 32 && t256 && blocks(64, 256) >= 256) return 28;   // output head (Co = 2)
        // >= 2 strips per block of the ping-pong kernel: level-0 layers from batch 4
        // (opt-in while it only matches the pipelined kernel: LC_PP_MIN_STRIPS=512; profiles/r04_pp_kernel.txt)
        static const long long pp_min = [] { const char* e = getenv("LC_PP_MIN_STRIPS"); return e ? atoll(e) : 0ll; }();
        if (pp_min > 0 && pp_eligible(Ci, Co, H, W, ks) && blocks(64, 256) >= pp_min) return 33;
        if (t256 && blocks(64, 256) >= 256) {
            // 32 / 64 input channels: the tall kernel (conv_f16x2_tall.hip: kept halo rows, 64-bit x loads, weight DMA
            // the compiler does not wait for): level-0 64->64 -9 %, 64->128 -11 % (profiles/r05_level0.txt);
            // LC_TALL=0: the pipelined tile everywhere.  (Its launcher falls back to cfg 23 for what it does not take.)
            static const int tall_env = [] { const char* e = getenv("LC_TALL"); return e ? atoi(e) : 1; }();
            if (tall_env && ks == 3 && (Ci == 32 || Ci == 64)) return 27;
            return 23;
        }
        if (t128 && blocks(64, 128) >= 256) return 25;
        return 13;
    }
    if (t256 && blocks(64, 256) >= 256 && Ci >= 64) return 2;
    if (t128 && blocks(64, 128) >= 512) return 5;
    return 3;
}

// max |w| of the tensor -> wmeta[2] (atomicMax on the bit pattern; wmeta zeroed before)
__global__ void weight_amax_kernel(const float* __restrict__ w, long long n, float* wmeta) {
    float am = 0.0f;
    for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < n;
         e += (long long)gridDim.x * blockDim.x)
        am = fmaxf(am, fabsf(w[e]));
    lc_block_amax_publish(am, reinterpret_cast<unsigned*>(wmeta + 2));
}

// the weight pre-scale from max|w| (see lc_pack_conv_weight_f16x2 in the header)
__device__ __forceinline__ float weight_scale_for(float amax) {
    if (!(amax > 0.0f) || !(amax < 3.0e38f)) return W_PRESCALE_DEFAULT;
    const float A = amax * W_PRESCALE_DEFAULT;
    if (A >= 32.0f && A <= 32768.0f) return W_PRESCALE_DEFAULT;
    int e;
    frexpf(amax, &e);                                   // amax = m * 2^e, m in [0.5, 1)
    int k = 13 - e;
    k = k < -120 ? -120 : (k > 120 ? 120 : k);
    return ldexpf(1.0f, k);                             // amax * scale in [2^12, 2^13)
}

// DX = false: w is the conv's own OIHW weight [Co][Ci][ntap].  DX = true: w is the FORWARD weight [Ci][Co][ntap] of the
// layer whose input gradient this conv computes -- the packed weight is its transpose with both kernel axes flipped
// (wt[co][ci][tap] = w[ci][co][ntap-1-tap]), read in place: no flipped / transposed copy exists; amax_src (may be
// NULL) = the wmeta of the forward pack of the same weight version, whose max|w| is this one's too.
template <bool DX>
__device__ __forceinline__ void pack_weight_elems(const float* __restrict__ w, _Float16* __restrict__ ph,
                                                  _Float16* __restrict__ pl, int Co, int Ci, int ntap, int Cib,
                                                  int Cop, float ws, long long first, long long stride) {
    const long long n = (long long)ntap * Cib * Cop * 8;
    for (long long e = first; e < n; e += stride) {
        const int k = e & 7;
        long long r = e >> 3;
        const int co = r % Cop; r /= Cop;
        const int cb = r % Cib;
        const int tap = r / Cib;
        const int ci = cb * 8 + k;
        float v = 0.0f;
        if (co < Co && ci < Ci)
            v = (DX ? w[((long long)ci * Co + co) * ntap + (ntap - 1 - tap)]
                    : w[((long long)co * Ci + ci) * ntap + tap]) * ws;
        // the split rule of the activations: hi = 11 significant bits (truncated, exact), lo = rest
        const float hf = __uint_as_float(__float_as_uint(v) & 0xFFFFE000u);
        ph[e] = (_Float16)hf;
        pl[e] = (_Float16)(v - hf);
    }
}

template <bool DX>
__global__ void pack_weight_h_kernel(const float* __restrict__ w, _Float16* __restrict__ ph,
                                     _Float16* __restrict__ pl, int Co, int Ci, int ntap, int Cib,
                                     int Cop, float* wmeta, const float* __restrict__ amax_src) {
    const float amax = amax_src ? amax_src[2] : wmeta[2];
    const float ws = weight_scale_for(amax);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        wmeta[0] = ws; wmeta[1] = 1.0f / ws;
        if (amax_src) { wmeta[2] = amax; wmeta[3] = 0.0f; }
    }
    pack_weight_elems<DX>(w, ph, pl, Co, Ci, ntap, Cib, Cop, ws, blockIdx.x * (long long)blockDim.x + threadIdx.x,
                          (long long)gridDim.x * blockDim.x);
}

// ---- all conv weights of a model in three launches (training: every weight changes at every optimizer step) ----------
__global__ void multi_weight_zero_kernel(const lc_weight_pack_job* __restrict__ jobs, int n) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) jobs[j].fwd_meta[2] = 0.0f;
}
__global__ void multi_weight_amax_kernel(const lc_weight_pack_job* __restrict__ jobs) {
    const lc_weight_pack_job jb = jobs[blockIdx.y];
    const long long n = (long long)jb.Co * jb.Ci * jb.ks * jb.ks;
    float am = 0.0f;
    for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x)
        am = fmaxf(am, fabsf(jb.w[e]));
    lc_block_amax_publish(am, reinterpret_cast<unsigned*>(jb.fwd_meta + 2));
}
// blockIdx.z = 0: the layer's own packed weight; 1: the packed weight of its input-gradient conv (Co / Ci swapped)
__global__ void multi_weight_pack_kernel(const lc_weight_pack_job* __restrict__ jobs) {
    const lc_weight_pack_job jb = jobs[blockIdx.y];
    const float amax = jb.fwd_meta[2];
    const float ws = weight_scale_for(amax);
    const int ntap = jb.ks * jb.ks;
    const long long first = blockIdx.x * (long long)blockDim.x + threadIdx.x, stride = (long long)gridDim.x * blockDim.x;
    if (blockIdx.z == 0) {
        if (blockIdx.x == 0 && threadIdx.x == 0) { jb.fwd_meta[0] = ws; jb.fwd_meta[1] = 1.0f / ws; }
        pack_weight_elems<false>(jb.w, (_Float16*)jb.fwd_hi, (_Float16*)jb.fwd_lo, jb.Co, jb.Ci, ntap,
                                 (jb.Ci + 15) / 16 * 2, (jb.Co + 63) / 64 * 64, ws, first, stride);
    } else {
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            jb.dx_meta[0] = ws; jb.dx_meta[1] = 1.0f / ws; jb.dx_meta[2] = amax; jb.dx_meta[3] = 0.0f;
        }
        pack_weight_elems<true>(jb.w, (_Float16*)jb.dx_hi, (_Float16*)jb.dx_lo, jb.Ci, jb.Co, ntap,
                                (jb.Co + 15) / 16 * 2, (jb.Ci + 63) / 64 * 64, ws, first, stride);
    }
}

}  // namespace

extern "C" int64_t lc_packed_conv_weight_f16x2_elems(int Co, int Ci, int ks) {
    const int64_t Cip = (Ci + 15) / 16 * 16, Cop = (Co + 63) / 64 * 64;
    return (int64_t)ks * ks * Cip * Cop;  // halves per plane (hi and lo planes each this size)
}

namespace {
int pack_weight_h(const float* w, void* wp_hi, void* wp_lo, int Co, int Ci, int ks, float* wmeta,
                  const float* amax_src, bool dx, lc_stream_t s) {
    if (!w || !wp_hi || !wp_lo || !wmeta || Co <= 0 || Ci <= 0 || (ks != 1 && ks != 3)) return LC_EINVAL;
    const int Cib = (Ci + 15) / 16 * 2, Cop = (Co + 63) / 64 * 64;
    const long long n = (long long)ks * ks * Cib * Cop * 8;
    const long long nw = (long long)Co * Ci * ks * ks;
    const int blocks = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    const int ablocks = (int)((nw + 255) / 256 > 1024 ? 1024 : (nw + 255) / 256);
    if (!amax_src) {
        if (hipMemsetAsync(wmeta, 0, 4 * sizeof(float), lc_s(s)) != hipSuccess) return lc_launch_status();
        hipLaunchKernelGGL(weight_amax_kernel, dim3(ablocks), dim3(256), 0, lc_s(s), w, nw, wmeta);
    }
    if (dx)
        hipLaunchKernelGGL(pack_weight_h_kernel<true>, dim3(blocks), dim3(256), 0, lc_s(s), w, (_Float16*)wp_hi,
                           (_Float16*)wp_lo, Co, Ci, ks * ks, Cib, Cop, wmeta, amax_src);
    else
        hipLaunchKernelGGL(pack_weight_h_kernel<false>, dim3(blocks), dim3(256), 0, lc_s(s), w, (_Float16*)wp_hi,
                           (_Float16*)wp_lo, Co, Ci, ks * ks, Cib, Cop, wmeta, amax_src);
    return lc_launch_status();
}
}  // namespace

extern "C" int lc_pack_conv_weight_f16x2(const float* w, void* wp_hi, void* wp_lo, int Co, int Ci,
                                         int ks, float* wmeta, lc_stream_t s) {
    return pack_weight_h(w, wp_hi, wp_lo, Co, Ci, ks, wmeta, nullptr, false, s);
}

// Every job's forward pack and (with_dx) input-gradient pack in three launches; `jobs` is a DEVICE array.
extern "C" int lc_pack_conv_weights_f16x2_multi(const lc_weight_pack_job* jobs, int n, int with_dx, lc_stream_t s) {
    if (!jobs || n <= 0) return LC_EINVAL;
    hipLaunchKernelGGL(multi_weight_zero_kernel, dim3((n + 255) / 256), dim3(256), 0, lc_s(s), jobs, n);
    hipLaunchKernelGGL(multi_weight_amax_kernel, dim3(16, n), dim3(256), 0, lc_s(s), jobs);
    hipLaunchKernelGGL(multi_weight_pack_kernel, dim3(48, n, with_dx ? 2 : 1), dim3(256), 0, lc_s(s), jobs);
    return lc_launch_status();
}

// Packed weight of the INPUT-GRADIENT conv of a layer, straight from the layer's forward weight w_fwd [Cf_o][Cf_i][ks][ks]:
// the dX conv has Co = Cf_i output and Ci = Cf_o input channels (pass THOSE as Co / Ci).
extern "C" int lc_pack_conv_weight_f16x2_dx(const float* w_fwd, void* wp_hi, void* wp_lo, int Co, int Ci, int ks,
                                            float* wmeta, const float* wmeta_fwd, lc_stream_t s) {
    return pack_weight_h(w_fwd, wp_hi, wp_lo, Co, Ci, ks, wmeta, wmeta_fwd, true, s);
}

// ---------------------------------------------------------------------------------------------
// Range record of a tensor derived ON THE DEVICE (training: activations and gradients change
// magnitude from step to step, and a backward pass cannot be repeated from inside autograd, so
// the scale is measured just before the conv that consumes the tensor -- one extra read of it,
// no host synchronisation, the record is exact for the very tensor the kernel will split).
__global__ void tensor_amax_kernel(const float* __restrict__ x, long long x_bs, long long n, int B,
                                   lc_conv_range* rg) {
    float am = 0.0f;
    const long long n4 = n >> 2;
    for (int b = blockIdx.y; b < B; b += gridDim.y) {
        const float* p = x + b * x_bs;
        if ((reinterpret_cast<unsigned long long>(p) & 15) == 0) {
            const f32x4* p4 = reinterpret_cast<const f32x4*>(p);
            for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < n4;
                 e += (long long)gridDim.x * blockDim.x) {
                const f32x4 v = p4[e];
                am = fmaxf(fmaxf(am, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
            }
            for (long long e = (n4 << 2) + blockIdx.x * (long long)blockDim.x + threadIdx.x; e < n;
                 e += (long long)gridDim.x * blockDim.x)
                am = fmaxf(am, fabsf(p[e]));
        } else {
            for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < n;
                 e += (long long)gridDim.x * blockDim.x)
                am = fmaxf(am, fabsf(p[e]));
        }
    }
    lc_block_amax_publish(am, reinterpret_cast<unsigned*>(&rg->reserved));   // NaN / 0 never win; inf is handled below
}
__device__ __forceinline__ void range_record_set(lc_conv_range* rg, float amax) {
    float sc = X_PRESCALE_DEFAULT;
    if (amax > 0.0f && amax < 3.0e38f) {
        int e;
        frexpf(amax, &e);                                   // amax = m * 2^e, m in [0.5, 1)
        int k = 13 - e;
        k = k < -120 ? -120 : (k > 120 ? 120 : k);
        sc = ldexpf(1.0f, k);                               // amax * scale in [2^12, 2^13)
    }
    rg->x_scale = sc;
    rg->x_unscale = 1.0f / sc;
    rg->amax_scaled = 0.0f;
    rg->reserved = 0.0f;
}
__global__ void range_set_kernel(lc_conv_range* rg) { range_record_set(rg, rg->reserved); }
__global__ void range_from_amax_kernel(lc_conv_range* rg, const float* __restrict__ amax, long long n, float bound_mult) {
    float m = 0.0f;
    for (long long i = threadIdx.x; i < n; i += blockDim.x) m = fmaxf(m, amax[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    __shared__ float wm[4];
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) range_record_set(rg, fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3])) * bound_mult);
}

extern "C" int lc_range_from_tensor(const float* x, int64_t x_bs, int B, int64_t n, lc_conv_range* range,
                                    lc_stream_t s) {
    if (!x || !range || B <= 0 || n <= 0) return LC_EINVAL;
    long long bx = (n / 4 + 255) / 256;
    bx = bx < 1 ? 1 : (bx > 512 ? 512 : bx);
    const int by = B < 8 ? B : 8;
    hipLaunchKernelGGL(tensor_amax_kernel, dim3((unsigned)bx, by), dim3(256), 0, lc_s(s), x, (long long)x_bs,
                       (long long)n, B, range);
    hipLaunchKernelGGL(range_set_kernel, dim3(1), dim3(1), 0, lc_s(s), range);
    return lc_launch_status();
}

// The record from the partial maxima some PRODUCER of x left while writing it (lc_groupnorm_apply_train /
// lc_groupnorm_bwd_train: one float per block of the pass) -- the extra read of x by lc_range_from_tensor disappears.
// bound_mult >= 1: x is a known elementwise contraction / rescale of the measured tensor (dropout: 1 / (1 - p)).
extern "C" int lc_range_from_amax(const float* amax, int64_t n, float bound_mult, lc_conv_range* range, lc_stream_t s) {
    if (!amax || n <= 0 || !range || !(bound_mult >= 1.0f)) return LC_EINVAL;
    hipLaunchKernelGGL(range_from_amax_kernel, dim3(1), dim3(256), 0, lc_s(s), range, amax, (long long)n, bound_mult);
    return lc_launch_status();
}

extern "C" int lc_conv2d_ring_f16x2_fwd(const float* x, int64_t x_bs, const void* wp_hi,
                                        const void* wp_lo, const float* bias, const float* res,
                                        int64_t res_bs, float* y, int64_t y_bs, int B, int Ci,
                                        int Co, int H, int W, int ks, float out_scale, int tile_cfg,
                                        const float* gn_coeffs, int gn_cpad, int gn_silu,
                                        const lc_gn_stats_input* gn_stats, float* gn_ostats_out, int gn_ostats_unit,
                                        const float* wmeta, lc_conv_range* range, lc_stream_t s) {
    if (!x || !wp_hi || !wp_lo || !y || !wmeta || !range || B <= 0 || Ci <= 0 || Co <= 0 || H <= 0 ||
        W <= 0)
        return LC_EINVAL;
    if (ks != 1 && ks != 3) return LC_EUNSUP;
    if ((long long)H * W >= (1 << 24)) return LC_EUNSUP;
    // the kernels address a sample with 32-bit byte offsets (0x80000000 is their out-of-range marker)
    if ((long long)Ci * H * W * 4 >= (1ll << 31) || (long long)Co * H * W * 4 >= (1ll << 31)) return LC_EUNSUP;
    ConvArgsH a;
    a.x = x; a.wh = (const half8*)wp_hi; a.wl = (const half8*)wp_lo; a.bias = bias; a.res = res;
    a.y = y; a.x_bs = x_bs; a.res_bs = res_bs; a.y_bs = y_bs;
    a.range = range; a.wmeta = wmeta;
    a.xsp = nullptr; a.xsp_bs = 0; a.xsp_c8 = 0; a.part = nullptr; a.ksplit = 0;
    a.B = B; a.Ci = Ci; a.Co = Co; a.H = H; a.W = W;
    a.Cib = (Ci + 15) / 16 * 2; a.Cop = (Co + 63) / 64 * 64;
    a.out_scale = out_scale;
    a.tiles_h = a.tiles_w = 0;
    a.gn = reinterpret_cast<const f32x4*>(gn_coeffs);
    a.Cgn = gn_cpad; a.gn_silu = gn_silu;
    a.gs = lc_gn_stats_input{nullptr, 0, 0, 0.f, nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr};
    a.seg[0] = a.seg[1] = ConvArgsH::OctSeg{nullptr, 0, 0, 3};
    if (gn_stats) {
        if (gn_coeffs || gn_stats->G <= 0 || Ci % gn_stats->G) return LC_EINVAL;
        if (gn_stats->partials) {
            if (gn_stats->nch <= 0) return LC_EINVAL;
        } else {
            const lc_oct_stats *s0 = gn_stats->os0, *s1 = gn_stats->os1;
            if (!s0 || !s0->p || s0->channels <= 0 || s0->slots <= 0) return LC_EINVAL;
            if (s1 && (!s1->p || s1->channels <= 0 || s1->slots <= 0)) return LC_EINVAL;
            const int c0 = s0->channels, c1 = s1 ? s1->channels : 0, cpg = Ci / gn_stats->G;
            const int u0 = s0->unit, u1 = s1 ? s1->unit : u0;
            auto ush_of = [](int u) { return u == 8 ? 3 : (u == 4 ? 2 : (u == 2 ? 1 : (u == 1 ? 0 : -1))); };
            if (ush_of(u0) < 0 || ush_of(u1) < 0) return LC_EINVAL;      // entries per octet / quad / pair / channel
            if (c0 + c1 != Ci || cpg % u0 || cpg % u1 || c0 % cpg || gn_stats->G > GN_MAX_G) return LC_EUNSUP;
            a.seg[0] = ConvArgsH::OctSeg{reinterpret_cast<const f32x4*>(s0->p), c0, s0->slots, ush_of(u0)};
            if (s1) a.seg[1] = ConvArgsH::OctSeg{reinterpret_cast<const f32x4*>(s1->p), c1, s1->slots, ush_of(u1)};
        }
        a.gs = *gn_stats;
        a.gs.os0 = a.gs.os1 = nullptr;
        a.gn = reinterpret_cast<const f32x4*>(x);   // non-null marker: "normalise the input"
        a.Cgn = gn_cpad > 0 ? gn_cpad : (Ci + 15) / 16 * 16;
    }
    a.tpb = 0;
    if (tile_cfg >= 100) { a.tpb = tile_cfg / 100; tile_cfg %= 100; }   // cfg = tpb*100 + tile id
    if (a.gn && (a.Cgn < (Ci + 15) / 16 * 16 || a.Cgn > GN_MAX_C || (a.Cgn & 1))) return LC_EINVAL;
    // A 1x1 conv has no spatial structure: fold the contiguous H*W plane into rows of 64 pixels so
    // that the 2-row tiles are fully used whatever the caller's aspect ratio is (a Conv1d over L
    // tokens arrives as H = 1, W = L and would leave every second tile row empty).
    if (ks == 1 && ((long long)H * W) % 128 == 0) {
        a.H = H = (int)(((long long)H * W) / 64);
        a.W = W = 64;
    }
    if (tile_cfg == 0) tile_cfg = auto_cfg_h(B, Ci, Co, H, W, ks);
    a.ostats = nullptr; a.oslots = 0; a.ounit = 8;
    if (gn_ostats_out) {
        if (gn_ostats_unit != 8 && gn_ostats_unit != 2) return LC_EINVAL;
        a.oslots = pipe_stat_slots(tile_cfg, H, W, ks);
        if (a.oslots <= 0 || Co % 8) return LC_EUNSUP;     // ask lc_conv2d_ring_f16x2_stats_slots first
        a.ostats = reinterpret_cast<f32x4*>(gn_ostats_out);
        a.ounit = gn_ostats_unit;
    }
    return ks == 3 ? dispatch_h<3>(tile_cfg, a, lc_s(s)) : dispatch_h<1>(tile_cfg, a, lc_s(s));
}

#if LC_TIMING
extern "C" int lc_debug_read(unsigned long long* out16, int reset) {
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(out16, HIP_SYMBOL(lc_dbg), sizeof(unsigned long long) * 32);
    if (reset) { unsigned long long z[32] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(lc_dbg), z, sizeof(z)); }
    return 0;
}
#endif

extern "C" int64_t lc_conv2d_ring_f16x2_stats_slots(int B, int Ci, int Co, int H, int W, int ks,
                                                    int tile_cfg) {
    if (B <= 0 || Ci <= 0 || Co <= 0 || Co % 8 || H <= 0 || W <= 0 || (ks != 1 && ks != 3)) return 0;
    if (tile_cfg >= 100) tile_cfg %= 100;
    if (ks == 1 && ((long long)H * W) % 128 == 0) { H = (int)(((long long)H * W) / 64); W = 64; }
    if (tile_cfg == 0) tile_cfg = auto_cfg_h(B, Ci, Co, H, W, ks);
    return pipe_stat_slots(tile_cfg, H, W, ks);
}

// Pre-split input: see conv_f16x2_ps_kernel.  3x3 ring convolution only.
extern "C" int lc_conv2d_ring_f16x2_ps_fwd(const void* x_split, const void* wp_hi, const void* wp_lo,
                                           const float* bias, const float* res, int64_t res_bs,
                                           float* y, int64_t y_bs, int B, int Ci, int Co, int H, int W,
                                           float out_scale, int tile_cfg, float* gn_ostats_out, int gn_ostats_unit,
                                           float* splitk_part, int ksplit, const float* wmeta,
                                           lc_conv_range* range, lc_stream_t s) {
    if (!x_split || !wp_hi || !wp_lo || (!y && !splitk_part) || !wmeta || !range || B <= 0 || Ci <= 0 ||
        Co <= 0 || H <= 0 || W <= 0)
        return LC_EINVAL;
    if (Ci % 16) return LC_EUNSUP;
    if (splitk_part && (ksplit < 2 || ksplit > Ci / 16)) return LC_EINVAL;
    if ((long long)H * W >= (1 << 24) || (long long)2 * (Ci / 8) * H * W * 16 >= (1ll << 31)) return LC_EUNSUP;
    if ((long long)Co * H * W * 4 >= (1ll << 31)) return LC_EUNSUP;       // 32-bit offsets into one output sample
    ConvArgsH a;
    a.x = nullptr; a.wh = (const half8*)wp_hi; a.wl = (const half8*)wp_lo; a.bias = bias; a.res = res;
    a.y = y; a.x_bs = 0; a.res_bs = res_bs; a.y_bs = y_bs;
    a.range = range; a.wmeta = wmeta;
    a.B = B; a.Ci = Ci; a.Co = Co; a.H = H; a.W = W;
    a.Cib = Ci / 8; a.Cop = (Co + 63) / 64 * 64;
    // the kernel addresses the lo plane as hi + one plane: the two must be one allocation
    if (a.wl != a.wh + (long long)9 * a.Cib * a.Cop) return LC_EINVAL;
    a.xsp = (const half8*)x_split; a.xsp_c8 = Ci / 8; a.xsp_bs = (long long)2 * (Ci / 8) * H * W;
    a.part = splitk_part; a.ksplit = splitk_part ? ksplit : 0;
    a.out_scale = out_scale;
    a.tiles_h = a.tiles_w = 0;
    a.gn = nullptr; a.Cgn = 0; a.gn_silu = 0;
    a.gs = lc_gn_stats_input{nullptr, 0, 0, 0.f, nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr};
    a.seg[0] = a.seg[1] = ConvArgsH::OctSeg{nullptr, 0, 0, 3};
    a.tpb = 0;
    if (tile_cfg >= 100) { a.tpb = tile_cfg / 100; tile_cfg %= 100; }
    // (the pipelined tile shapes only: the heuristic's Ci >= 24 branch)
    if (tile_cfg == 0) tile_cfg = auto_cfg_h(B, Ci < 24 ? 24 : Ci, Co, H, W, 3);
    if (pipe_stat_slots(tile_cfg, H, W) <= 0) return LC_EUNSUP;
    a.ostats = nullptr; a.oslots = 0; a.ounit = 8;
    if (gn_ostats_out && !splitk_part) {
        a.oslots = pipe_stat_slots(tile_cfg, H, W);
        if (Co % 8 || (gn_ostats_unit != 8 && gn_ostats_unit != 4)) return LC_EUNSUP;   // octet or quad entries
        a.ostats = reinterpret_cast<f32x4*>(gn_ostats_out);
        a.ounit = gn_ostats_unit;
    }
    return dispatch_h<3>(tile_cfg, a, lc_s(s));
}

#include "conv_f16x2_ps1.h"

// ---- split-K finish ------------------------------------------------------------------------------
namespace {
// y = (sum_z part[z] + bias [+ res]) * out_scale, one thread per (pixel, channel octet): the ksplit
// partial planes are summed in index order (deterministic), optional GroupNorm statistics of what
// is stored in the entry format of the conv epilogue (one entry per octet and wave of 64 pixels).
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, int ks,
                                                           const float* __restrict__ bias,
                                                           const float* __restrict__ res,
                                                           long long res_bs, float* __restrict__ y,
                                                           long long y_bs, int B, int Co, int HW,
                                                           float out_scale, f32x4* __restrict__ ostats,
                                                           int oslots) {
    const int b = blockIdx.z, oct = blockIdx.y, c0 = oct * 8;
    const int p = blockIdx.x * 256 + threadIdx.x;
    const bool pok = p < HW;
    const long long plane = (long long)B * Co * HW;
    // all 8 channel planes of a partial in flight together, two partials per round (the first
    // version walked channel by channel: 8 * ksplit dependent L2 round trips, 14 us per launch)
    float v[8];
    bool ok[8];
    const float* pp[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        ok[k] = pok && c0 + k < Co;
        pp[k] = part + ((long long)b * Co + (ok[k] ? c0 + k : c0)) * HW + (pok ? p : 0);
        v[k] = 0.0f;
    }
    int z = 0;
    for (; z + 1 < ks; z += 2) {
        float t0[8], t1[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { t0[k] = pp[k][z * plane]; t1[k] = pp[k][(z + 1) * plane]; }
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = (v[k] + t0[k]) + t1[k];
    }
    if (z < ks) {
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] += pp[k][z * plane];
    }
    float bv[8], rv[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        bv[k] = (bias && ok[k]) ? bias[c0 + k] : 0.0f;
        rv[k] = (res && ok[k]) ? res[b * res_bs + (long long)(c0 + k) * HW + p] : 0.0f;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        v[k] = ok[k] ? ((v[k] + bv[k]) + rv[k]) * out_scale : 0.0f;
        if (ok[k]) y[b * y_bs + (long long)(c0 + k) * HW + p] = v[k];
    }
    if (ostats) {                                  // Co % 8 == 0 (checked by the launcher)
        const float piv = __builtin_amdgcn_readfirstlane(v[0]);
        float s_ = 0.f, q_ = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float d = pok ? v[k] - piv : 0.0f;
            s_ += d; q_ = fmaf(d, d, q_);
        }
        const int nvalid = __popcll(__ballot(pok));
        s_ = wave_sum_to_lane63(s_);
        q_ = wave_sum_to_lane63(q_);
        const int slot = blockIdx.x * 4 + (threadIdx.x >> 6);
        if ((threadIdx.x & 63) == 63 && slot < oslots)
            ostats[((long long)b * (Co >> 3) + oct) * oslots + slot] =
                f32x4{piv, (float)(8 * nvalid), s_, q_};
    }
}
}  // namespace

extern "C" int64_t lc_splitk_stats_slots(int H, int W) {
    return H > 0 && W > 0 ? ((int64_t)H * W + 255) / 256 * 4 : 0;
}

extern "C" int lc_splitk_reduce(const float* part, int ksplit, const float* bias, const float* res,
                                int64_t res_bs, float* y, int64_t y_bs, int B, int Co, int H, int W,
                                float out_scale, float* gn_ostats_out, lc_stream_t s) {
    if (!part || !y || ksplit < 1 || B <= 0 || Co <= 0 || H <= 0 || W <= 0) return LC_EINVAL;
    if (gn_ostats_out && Co % 8) return LC_EUNSUP;
    const int HW = H * W;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((HW + 255) / 256, (Co + 7) / 8, B), dim3(256), 0,
                       lc_s(s), part, ksplit, bias, res, (long long)res_bs, y, (long long)y_bs, B, Co, HW,
                       out_scale, reinterpret_cast<f32x4*>(gn_ostats_out),
                       (int)lc_splitk_stats_slots(H, W));
    return lc_launch_status();
}

LC_TOUCH_TU(conv_f16x2, tensor_amax_kernel)
