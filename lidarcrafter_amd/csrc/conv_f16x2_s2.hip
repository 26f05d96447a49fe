// Block.downsample of EfficientUNet -- `Conv2d(3x3, ring) -> Resample(down=2)` (lidargen/models/unets/efficient_unet.py:132-135,
// ops.py:52-146,149-173) -- as ONE stride-2 convolution behind a FIR pre-filter (round 6).
//
// The reference evaluates the 3x3 conv at full resolution and then filters ([1 3 3 1] / 8 along W with ring padding, along H
// with ZERO padding of the conv's output) and keeps every second row / column: three quarters of the conv's MFMA work and of
// its output bytes are thrown away.  Both operators are linear and, away from the H border, shift-invariant, so they
// commute:
//     y[oy][ox] = sum_j f_j sum_i f_i c[2 oy - 1 + j][2 ox - 1 + i],   c = b + w * x  (rows of c outside the image are 0)
//               = b * sum_{j valid} f_j + sum_{ky,kx,ci} w[ky][kx][ci] * G_ky[oy][2 ox + kx - 1]
//     G_ky[oy][s] = sum_{j valid} f_j XW[2 oy + j + ky - 2][s],   XW[r][s] = sum_i f_i x[r][s + i - 1 (ring)]  (0 for r outside)
// With F[m] = sum_j f_j XW[m + j - 1] (m = -1 .. Hin - 1) the inner operand is F[2 oy + ky - 1] for every (oy, ky) EXCEPT the
// two where a vertical FIR tap falls on the zero padding of c while its x row exists:
//     (oy = 0,      ky = 2):  F[1]       minus f0 * XW[0]        ("top variant")
//     (oy = Ho - 1, ky = 0):  F[Hin - 3] minus f3 * XW[Hin - 1]  ("bottom variant")
// and the bias reaches the first / last output row through 1 - f0 (1 - f3) of the vertical taps.  So:
//   1. fir_down2_prefilter_split_kernel: x (fp32 NCHW) -> F, already multiplied by the consumer conv's x_scale and split into
//      the fp16 hi / lo planes of the pre-split layout (16-byte units of 8 channels), rows F[-1 .. Hin-1] + the two variants,
//      each row stored column-phase separated (odd input columns, then even ones): one read of x, one write of the same
//      number of bytes (+ 3 rows), no MFMA.
//   2. conv_f16x2_ps_kernel<.., S2 = true> (conv_f16x2_ps.h): the stride-2 3x3 conv over F by LDS-DMA staging -- a quarter
//      of the reference conv's MFMA work, a quarter of its output bytes, and no resampling pass behind it.
// Same f16x2 arithmetic as every convolution here (wh*xh + wh*xl + wl*xh, fp32 accumulate); the result differs from the
// reference's by summation order only (tests/test_fold_down.py: <= 2e-6 rel-L2 against oracle conv + resample, and the
// reference's own `down_y` of tests/golden/ops.npz).
#include "conv_f16x2_common.h"

using namespace lcconv;

namespace {

#include "conv_f16x2_ps.h"

typedef _Float16 half8_s2 __attribute__((ext_vector_type(8)));

// thread = (channel octet, column pair (2 i, 2 i + 1), row strip); it walks its strip of F rows with a sliding window of four
// W-filtered rows (8 channels x 2 columns each) in registers, so x is read once (plus 3 halo rows per strip).  The raw
// values of the NEXT input row are requested before the current row's filter / split / stores (the walk is a serial chain of
// HBM round trips otherwise: first version 42.6 us for 67 MB in + 69 MB out at 8 x 64 x 32 x 1024, profiles/r06_fold_down.txt).
__global__ __launch_bounds__(256) void fir_down2_prefilter_split_kernel(
    const float* __restrict__ x, long long x_bs, half8_s2* __restrict__ ysp, int C8, int H, int W, int nstrip, int strip_len,
    lc_conv_range* range) {
    const int Wh = W >> 1;
    const int i = blockIdx.x * 256 + threadIdx.x;            // column pair
    const int c8 = blockIdx.y;
    const int b = blockIdx.z / nstrip, strip = blockIdx.z - b * nstrip;
    const float xs = range->x_scale, seen = range->amax_scaled;
    float am = 0.f;
    if (i < Wh) {
        const long long HW = (long long)H * W;
        const float* xb = x + b * x_bs + (long long)c8 * 8 * HW;
        const int R = H + 3;
        half8_s2* yh = ysp + ((long long)b * 2 * C8 + c8) * R * W;          // hi plane of this octet: rows of W units
        half8_s2* yl = yh + (long long)C8 * R * W;
        const int cm1 = (2 * i - 1 + W) % W, cp2 = (2 * i + 2) % W;         // (cp2 is even: columns cp2, cp2 + 1 are one float2)
        constexpr float f0 = 0.125f, f1 = 0.375f, f2 = 0.375f, f3 = 0.125f;
        struct Raw { float a[8]; float2 m[8], c[8]; };
        // raw values of input row r at columns 2 i - 1 .. 2 i + 3 (a clamped row: the caller zeroes what lies outside)
        auto load_raw = [&](int r, Raw& w) {
            const int rc = r < 0 ? 0 : (r >= H ? H - 1 : r);
            const float* p = xb + (long long)rc * W;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float* q = p + (long long)k * HW;
                w.a[k] = q[cm1];
                w.m[k] = *reinterpret_cast<const float2*>(q + 2 * i);
                w.c[k] = *reinterpret_cast<const float2*>(q + cp2);
            }
        };
        // W-filtered row of the 8 channels at columns 2 i / 2 i + 1 (zeros outside the image)
        auto xw_row = [&](int r, const Raw& w, float (&e)[8], float (&o)[8]) {
            const float in = (r >= 0 && r < H) ? 1.0f : 0.0f;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                // (explicit fma chains: left to -ffp-contract the prologue and loop copies of this code were contracted
                //  differently and the result depended on the strip length in its last bit)
                e[k] = in * fmaf(f3, w.c[k].x, fmaf(f2, w.m[k].y, fmaf(f1, w.m[k].x, f0 * w.a[k])));        // column 2 i
                o[k] = in * fmaf(f3, w.c[k].y, fmaf(f2, w.c[k].x, fmaf(f1, w.m[k].y, f0 * w.m[k].x)));      // column 2 i + 1
            }
        };
        auto put = [&](int row, const float (&e)[8], const float (&o)[8]) {
            half8 h, l;
            split8(e, xs, h, l, am);
            yh[(long long)row * W + Wh + i] = h;                        // even column 2 i: even-phase index i
            yl[(long long)row * W + Wh + i] = l;
            const int jo = i + 1 == Wh ? 0 : i + 1;                     // odd column 2 i + 1 = 2 (i + 1) - 1: odd-phase index i + 1
            split8(o, xs, h, l, am);
            yh[(long long)row * W + jo] = h;
            yl[(long long)row * W + jo] = l;
        };
        // F rows m0 .. m1 - 1 of this strip (m = -1 .. H - 1)
        const int m0 = -1 + strip * strip_len;
        const int m1 = (m0 + strip_len < H) ? m0 + strip_len : H;
        float we[4][8], wo[4][8];                                       // XW[m - 1 .. m + 2]
        Raw raw;
        load_raw(m0 - 1, raw); xw_row(m0 - 1, raw, we[0], wo[0]);
        load_raw(m0, raw); xw_row(m0, raw, we[1], wo[1]);
        load_raw(m0 + 1, raw); xw_row(m0 + 1, raw, we[2], wo[2]);
        load_raw(m0 + 2, raw);
        for (int m = m0; m < m1; ++m) {
            xw_row(m + 2, raw, we[3], wo[3]);
            if (m + 1 < m1) load_raw(m + 3, raw);                       // next row's loads fly under this row's arithmetic + stores
            float fe[8], fo[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                fe[k] = fmaf(f3, we[3][k], fmaf(f2, we[2][k], fmaf(f1, we[1][k], f0 * we[0][k])));
                fo[k] = fmaf(f3, wo[3][k], fmaf(f2, wo[2][k], fmaf(f1, wo[1][k], f0 * wo[0][k])));
            }
            put(m + 1, fe, fo);
            if (m == 1) {                                               // top variant: F[1] without f0 * XW[0]
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    fe[k] = fmaf(f3, we[3][k], fmaf(f2, we[2][k], f1 * we[1][k]));
                    fo[k] = fmaf(f3, wo[3][k], fmaf(f2, wo[2][k], f1 * wo[1][k]));
                }
                put(H + 1, fe, fo);
            }
            if (m == H - 3) {                                           // bottom variant: F[H - 3] without f3 * XW[H - 1]
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    fe[k] = fmaf(f2, we[2][k], fmaf(f1, we[1][k], f0 * we[0][k]));
                    fo[k] = fmaf(f2, wo[2][k], fmaf(f1, wo[1][k], f0 * wo[0][k]));
                }
                put(H + 2, fe, fo);
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                we[0][k] = we[1][k]; we[1][k] = we[2][k]; we[2][k] = we[3][k];
                wo[0][k] = wo[1][k]; wo[1][k] = wo[2][k]; wo[2][k] = wo[3][k];
            }
        }
    }
    publish_amax(range, am, seen);
}

using S2Cfg_ = HCfg<2, 2, 1, 1, 1, 64, 3>;

// ---------------------------------------------------------------------------------------------------------------------
// The stride-2 conv, second form (round 6, second half): WEIGHTS SHARED ACROSS THE BLOCK'S TILES.
// The first form (conv_f16x2_ps_kernel<S2Cfg, EMIT, S2 = true>) is bound by its LDS-DMA traffic: 62 KB per 16-channel chunk
// of a 64 co x 64 px tile, 58 % of it weights -- and a persistent block re-loads the SAME weight chunk for every one of its
// tiles.  Here the block keeps the accumulators of all its TPB tiles (16 registers each) and runs the chunk loop OUTSIDE the
// tile loop: weight chunk c travels once per block, the x image of (tile t, chunk c) once per (t, c) as before.  Per
// (tile, chunk) iteration 26 KB of x + 37 / TPB KB of weights instead of 63.5 KB: -44 % at TPB = 4.  Same tile (64 co x
// 1 x 64 px, 4 waves), same LDS image, fragment addressing, border variants and epilogue as the first form; the x buffers
// ping-pong per iteration, the weight buffers per chunk.  Results are bit-identical to the first form (same products in the
// same order per accumulator).
template <int TPB, bool EMIT_STATS>
__global__ __launch_bounds__(256, 1) void conv_s2_shared_w_kernel(ConvArgsH a) {
    using C = S2Cfg_;
    constexpr int CB = 2, NTAP = 9, BN = 64, TW = 64;
    constexpr int XW = 2 * TW + 1, XR = 3, XU = CB * XR * XW, WU = NTAP * CB * BN;
    constexpr int NWV = 4;
    constexpr int NXI = (XU + 63) / 64, NWI = (WU + 63) / 64;
    constexpr int XS = NXI * 64, WS = NWI * 64;
    constexpr int KX = (2 * NXI + NWV - 1) / NWV, KW = (2 * NWI + NWV - 1) / NWV;     // DMA slots per wave: x 7, weights 9
    constexpr unsigned OOB = 0x80000000u;
    __shared__ half8 xlds[2][2 * XS];
    __shared__ half8 wlds[2][2 * WS];
    __shared__ half8 dummy[64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wco = wave >> 1, wpx = wave & 1;
    const int kh = lane >> 5, l31 = lane & 31;

    int bx = blockIdx.x;
    if (a.xcd) bx = (bx & 7) * (gridDim.x >> 3) + (bx >> 3);
    const int gh_ = a.tiles_h / TPB;
    const int tw_i = bx % a.tiles_w; bx /= a.tiles_w;
    const int h0 = (bx % gh_) * TPB; bx /= gh_;
    const int b = bx;
    const int w0 = tw_i * TW;
    const int co0 = blockIdx.y * BN;
    const int H = a.H, W = a.W, HW = H * W;
    const int C8 = a.xsp_c8;
    const int XROWS = 2 * H + 3, XCOLS = 2 * W;
    const float out_unscale = a.range->x_unscale * a.wmeta[1];

    const unsigned xbytes = 2u * (unsigned)C8 * (unsigned)(XROWS * XCOLS) * 16u;
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)(a.xsp + (long long)b * a.xsp_bs), 0, xbytes, 0x00020000);
    const unsigned wplane = (unsigned)(NTAP * a.Cib) * (unsigned)a.Cop;
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)a.wh, 0, 2u * wplane * 16u, 0x00020000);

    // x slots: per-lane source offsets of every tile (the row part differs per tile; the variants only at the image border)
    unsigned vx[TPB][KX];
    int lx[KX];
#pragma unroll
    for (int k = 0; k < KX; ++k) {
        const int j = wave + k * NWV;
        const int plane = j / NXI, e = (j - plane * NXI) * 64 + lane;
        const int cb = e / (XR * XW), rem = e - cb * (XR * XW);
        const int ky = rem / XW, c = rem - ky * XW;
        int col;
        if (c <= TW) { col = w0 + c; col = col >= W ? col - W : col; }        // odd phase, ring
        else col = W + w0 + (c - TW - 1);                                      // even phase
        lx[k] = j < 2 * NXI ? plane * XS + (j - plane * NXI) * 64 : -1;
#pragma unroll
        for (int t = 0; t < TPB; ++t) {
            const int oy = h0 + t;
            int row = 2 * oy + ky;
            if (oy == 0 && ky == 2) row = 2 * H + 1;
            if (oy == H - 1 && ky == 0) row = 2 * H + 2;
            const bool ok = j < 2 * NXI && e < XU && oy < H;
            vx[t][k] = ok ? (unsigned)(((plane * C8 + cb) * XROWS + row) * XCOLS + col) * 16u : OOB;
        }
    }
    unsigned vw[KW];
    int lw[KW];
#pragma unroll
    for (int k = 0; k < KW; ++k) {
        const int j = wave + k * NWV;
        const int plane = j / NWI, e = (j - plane * NWI) * 64 + lane;
        const int row = e / BN, cu = e - row * BN;
        const int tap = row / CB, cb = row - tap * CB;
        lw[k] = j < 2 * NWI ? plane * WS + (j - plane * NWI) * 64 : -1;
        vw[k] = (j < 2 * NWI && e < WU) ? ((unsigned)((tap * a.Cib + cb) * a.Cop + co0 + cu) + plane * wplane) * 16u : OOB;
    }
    const unsigned x_chunk = (unsigned)CB * (unsigned)(XROWS * XCOLS) * 16u;
    const unsigned w_chunk = (unsigned)CB * (unsigned)a.Cop * 16u;
    auto dma_x = [&](int buf, int t, int k, int ch) {
        half8* dst = lx[k] >= 0 ? &xlds[buf][lx[k]] : dummy;
        lds_dma16(rs_x, (lds_vptr)dst, vx[t][k], (unsigned)ch * x_chunk);
    };
    auto dma_w = [&](int buf, int k, int ch) {
        half8* dst = lw[k] >= 0 ? &wlds[buf][lw[k]] : dummy;
        lds_dma16(rs_w, (lds_vptr)dst, vw[k], (unsigned)ch * w_chunk);
    };

    f32x16 acc[TPB];
#pragma unroll
    for (int t = 0; t < TPB; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
    const int xbase = kh * (XR * XW) + wpx * 32 + l31;
    const int wbase = kh * BN + wco * 32 + l31;

    // one (tile, chunk) iteration: MFMAs of tile T from xlds[xb] / wlds[wb]; the NEXT iteration's x image (tile NT, chunk nch)
    // is DMA'd into xlds[xb ^ 1] under them and, when NEXTW, weight chunk nch into wlds[wb ^ 1]
    auto iteration = [&](auto T_, auto NT_, auto NEXTW_, int xb, int wb, int nch, bool issue_next) {
        constexpr int T = decltype(T_)::value, NT = decltype(NT_)::value;
        constexpr bool NEXTW = decltype(NEXTW_)::value;
        constexpr int NSLOT = KX + (NEXTW ? KW : 0);
        constexpr int SPT = (NSLOT + NTAP - 1) / NTAP;
        const half8* cxh = &xlds[xb][0];
        const half8* cxl = cxh + XS;
        const half8* cwh = &wlds[wb][0];
        const half8* cwl = cwh + WS;
        half8 ah[2], al[2], bh[2], bl[2];
        auto fetch = [&](int tap, int s) {
            const int dy = tap / 3, dx = tap - dy * 3;
            const int o = dy * XW + (dx == 1 ? TW + 1 : (dx == 2 ? 1 : 0));
            ah[s] = cwh[tap * CB * BN + wbase];
            al[s] = cwl[tap * CB * BN + wbase];
            bh[s] = cxh[xbase + o];
            bl[s] = cxl[xbase + o];
        };
        fetch(0, 0);
#pragma unroll
        for (int tap = 0; tap < NTAP; ++tap) {
            const int s = tap & 1;
            __builtin_amdgcn_sched_barrier(0);
            if (tap + 1 < NTAP) fetch(tap + 1, s ^ 1);
            if (issue_next) {
#pragma unroll
                for (int q = 0; q < SPT; ++q) {
                    const int k = tap * SPT + q;
                    if (k < KX) dma_x(xb ^ 1, NT, k, nch);
                    else if (NEXTW && k < NSLOT) dma_w(wb ^ 1, k - KX, nch);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (LC_F16X2_TERMS & 2) acc[T] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[s], bh[s], acc[T], 0, 0, 0);
            if (LC_F16X2_TERMS & 4) acc[T] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], bl[s], acc[T], 0, 0, 0);
            acc[T] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], bh[s], acc[T], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };

    // prologue: x(tile 0, chunk 0) and weight chunk 0
#pragma unroll
    for (int k = 0; k < KX; ++k) dma_x(0, 0, k, 0);
#pragma unroll
    for (int k = 0; k < KW; ++k) dma_w(0, k, 0);
    const int co_wave = co0 + wco * 32 + 4 * kh;
    float bias_r[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int co = co_wave + (r & 3) + 8 * (r >> 2);
        bias_r[r] = (a.bias && co < a.Co) ? a.bias[co] : 0.0f;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const int nchunk = a.Cib / CB;
    int xb = 0;
    for (int ch = 0; ch < nchunk; ++ch) {
        const int wb = ch & 1;
        const bool more = ch + 1 < nchunk;
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>;
        using I3 = std::integral_constant<int, 3>;
        using NO = std::false_type;
        using YES = std::true_type;
        if constexpr (TPB == 1) {
            iteration(I0{}, I0{}, YES{}, xb, wb, ch + 1, more); xb ^= 1;
        } else if constexpr (TPB == 2) {
            iteration(I0{}, I1{}, NO{}, xb, wb, ch, true); xb ^= 1;
            iteration(I1{}, I0{}, YES{}, xb, wb, ch + 1, more); xb ^= 1;
        } else {
            iteration(I0{}, I1{}, NO{}, xb, wb, ch, true); xb ^= 1;
            iteration(I1{}, I2{}, NO{}, xb, wb, ch, true); xb ^= 1;
            iteration(I2{}, I3{}, NO{}, xb, wb, ch, true); xb ^= 1;
            iteration(I3{}, I0{}, YES{}, xb, wb, ch + 1, more); xb ^= 1;
        }
    }

    // ---- epilogue of every tile: bias (x 7/8 on the image's first / last row), scale, store, statistics entries
    const unsigned HW4 = (unsigned)HW * 4u;
    const __amdgpu_buffer_rsrc_t rs_yb = __builtin_amdgcn_make_buffer_rsrc((void*)(a.y + (long long)b * a.y_bs), 0, (unsigned)a.Co * HW4, 0x00020000);
    const bool quads = a.ounit == 4;
    const int ush = quads ? 2 : 3;
    __amdgpu_buffer_rsrc_t rs_o = rs_yb;
    if constexpr (EMIT_STATS)
        rs_o = __builtin_amdgcn_make_buffer_rsrc((void*)(a.ostats + (long long)b * (a.Co >> ush) * a.oslots), 0,
                                                 (unsigned)(a.Co >> ush) * (unsigned)a.oslots * 16u, 0x00020000);
#pragma unroll
    for (int t = 0; t < TPB; ++t) {
        const int gh = h0 + t, gw = w0 + wpx * 32 + l31;
        const bool pok = gh < H && gw < W;
        const unsigned vo = pok ? (unsigned)(co_wave * HW + gh * W + gw) * 4u : OOB;
        const float brow = (gh == 0 ? 0.875f : 1.0f) + (gh == H - 1 ? 0.875f : 1.0f) - 1.0f;
        float st_p[4], st_s[4], st_q[4];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int cor = (r & 3) + 8 * (r >> 2);
            const float v = (acc[t][r] * out_unscale + bias_r[r] * brow) * a.out_scale;
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs_yb, co_wave + cor < a.Co ? vo : OOB,
                                                  (unsigned)cor * HW4, LC_DEF_AUX);
            if constexpr (EMIT_STATS) {
                const int m = r >> 2;
                if ((r & 3) == 0) { st_p[m] = __builtin_amdgcn_readlane(pok ? v : 0.0f, 0); st_s[m] = 0.f; st_q[m] = 0.f; }
                const float d = pok ? v - st_p[m] : 0.0f;
                st_s[m] += d;
                st_q[m] = fmaf(d, d, st_q[m]);
            }
        }
        if constexpr (EMIT_STATS) {
            const int nvalid = __popcll(__ballot(pok) & 0xFFFFFFFFull);
            const int slot = (gh * a.tiles_w + tw_i) * 2 + wpx;
            const bool mine = quads ? (lane & 31) >= 28 : lane >= 60;
            const float nv = (float)((quads ? 4 : 8) * nvalid);
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int co_oct = co0 + wco * 32 + 8 * m;
                const float sh = half_sum_to_lane31_63(st_s[m]), qh = half_sum_to_lane31_63(st_q[m]);
                const float sf = dpp_add<0x143, 0xC>(sh), qf = dpp_add<0x143, 0xC>(qh);
                const int ent = quads ? (co_oct >> 2) + (lane >> 5) : (co_oct >> 3);
                const unsigned eo = (mine && co_oct < a.Co && gh < H)
                                        ? ((unsigned)ent * (unsigned)a.oslots + (unsigned)slot) * 16u + 4u * (lane & 3)
                                        : OOB;
                store_entry_4lanes(rs_o, st_p[m], nv, quads ? sh : sf, quads ? qh : qf, eo);
            }
        }
    }
}

using S2Cfg = HCfg<2, 2, 1, 1, 1, 64, 3>;      // 4 waves, 64 co x (1 x 64) output pixels; LDS 129 KB: one block per CU

}  // namespace

// slots of the stride-2 conv's statistics entries per (sample, channel unit): one per wave tile of the OUTPUT image
extern "C" int64_t lc_conv2d_ring_s2_stats_slots(int Ho, int Wo) {
    if (Ho <= 0 || Wo <= 0 || Wo % S2Cfg::TW_) return 0;
    return (int64_t)Ho * (Wo / S2Cfg::TW_) * S2Cfg::WPX_;
}

extern "C" int64_t lc_fir_down2_split_units(int B, int C, int H, int W) {
    if (B <= 0 || C <= 0 || C % 8 || H <= 0 || W <= 0) return 0;
    return (int64_t)B * 2 * (C / 8) * (H + 3) * W;
}

extern "C" int lc_fir_down2_prefilter_split(const float* x, int64_t x_bs, void* y_split, int B, int C, int H, int W,
                                            lc_conv_range* range, lc_stream_t s) {
    if (!x || !y_split || !range || B <= 0 || C <= 0 || H <= 0 || W <= 0) return LC_EINVAL;
    if (C % 16 || H % 2 || H < 4 || W % 128) return LC_EUNSUP;      // 16-channel K chunks; whole 64-pixel output tiles
    if ((reinterpret_cast<uintptr_t>(x) & 7) || (x_bs & 1)) return LC_EUNSUP;   // float2 loads
    // rows per strip: 8 (3 halo rows are re-read per strip).  Shorter strips for more waves were measured and lose on the
    // small planes: 8 x 256 x 8 x 256: 20.6 us at 8 rows per strip, 21.1 at 2, 25.2 at 1 (profiles/r06_fold_down.txt section 2)
    int strip_len = 8;
    static const int strip_env = [] { const char* e = getenv("LC_PF_STRIP"); return e ? atoi(e) : 0; }();
    if (strip_env > 0) strip_len = strip_env;
    const int nstrip = (H + 1 + strip_len - 1) / strip_len;
    if ((long long)B * nstrip > 65535) return LC_EUNSUP;
    dim3 grid((W / 2 + 255) / 256, C / 8, B * nstrip);
    hipLaunchKernelGGL(fir_down2_prefilter_split_kernel, grid, dim3(256), 0, lc_s(s), x, (long long)x_bs,
                       reinterpret_cast<half8_s2*>(y_split), C / 8, H, W, nstrip, strip_len, range);
    return lc_launch_status();
}

extern "C" int lc_conv2d_ring_s2_f16x2_ps_fwd(const void* x_split, const void* wp_hi, const void* wp_lo, const float* bias,
                                              float* y, int64_t y_bs, int B, int Ci, int Co, int Ho, int Wo, float out_scale,
                                              float* gn_ostats_out, int gn_ostats_unit, const float* wmeta,
                                              lc_conv_range* range, lc_stream_t s) {
    using C = S2Cfg;
    if (!x_split || !wp_hi || !wp_lo || !y || !wmeta || !range || B <= 0 || Ci <= 0 || Co <= 0 || Ho <= 0 || Wo <= 0)
        return LC_EINVAL;
    if (Ci % 16 || Ho < 2 || Wo % C::TW_) return LC_EUNSUP;
    const long long plane_units = (long long)(Ci / 8) * (2 * Ho + 3) * (2 * Wo);
    if (2 * plane_units * 16 >= (1ll << 31) || (long long)Co * Ho * Wo * 4 >= (1ll << 31)) return LC_EUNSUP;   // 32-bit offsets
    ConvArgsH a;
    a.x = nullptr; a.wh = (const half8*)wp_hi; a.wl = (const half8*)wp_lo; a.bias = bias; a.res = nullptr;
    a.y = y; a.x_bs = 0; a.res_bs = 0; a.y_bs = y_bs;
    a.range = range; a.wmeta = wmeta;
    a.B = B; a.Ci = Ci; a.Co = Co; a.H = Ho; a.W = Wo;
    a.Cib = Ci / 8; a.Cop = (Co + 63) / 64 * 64;
    if (a.wl != a.wh + (long long)9 * a.Cib * a.Cop) return LC_EINVAL;     // one allocation: lo plane behind the hi plane
    a.xsp = (const half8*)x_split; a.xsp_c8 = Ci / 8; a.xsp_bs = 2 * plane_units;
    a.part = nullptr; a.ksplit = 0;
    a.out_scale = out_scale;
    a.gn = nullptr; a.Cgn = 0; a.gn_silu = 0;
    a.gs = lc_gn_stats_input{nullptr, 0, 0, 0.f, nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr};
    a.seg[0] = a.seg[1] = ConvArgsH::OctSeg{nullptr, 0, 0, 3};
    a.ostats = nullptr; a.oslots = 0; a.ounit = 8;
    if (gn_ostats_out) {
        if (Co % 8 || (gn_ostats_unit != 8 && gn_ostats_unit != 4)) return LC_EUNSUP;
        a.oslots = (int)lc_conv2d_ring_s2_stats_slots(Ho, Wo);
        a.ostats = reinterpret_cast<f32x4*>(gn_ostats_out);
        a.ounit = gn_ostats_unit;
    }
    a.tiles_h = Ho; a.tiles_w = Wo / C::TW_;
    const int ncot = (Co + C::BN - 1) / C::BN;
    // persistent blocks: up to 4 consecutive output rows per block while every CU still gets a block
    const long long n_tiles = (long long)B * a.tiles_h * a.tiles_w * ncot;
    int tpb = 1;
    while (tpb < 4 && a.tiles_h % (tpb * 2) == 0 && n_tiles / (tpb * 2) >= 256) tpb *= 2;
    a.tpb = tpb; a.vert = 1;
    dim3 grid(B * a.tiles_h * a.tiles_w / tpb, ncot);
    a.xcd = (grid.x % 8 == 0 && grid.x >= 16) ? 1 : 0;
    // LC_S2_FORM=1: the first form (weights re-loaded per tile; conv_f16x2_ps_kernel<.., S2>) -- developer A/B, same bits
    static const int form_env = [] { const char* e = getenv("LC_S2_FORM"); return e ? atoi(e) : 2; }();
    if (form_env == 1) {
        if (a.ostats) hipLaunchKernelGGL((conv_f16x2_ps_kernel<C, true, true>), grid, dim3(C::NT), 0, lc_s(s), a);
        else hipLaunchKernelGGL((conv_f16x2_ps_kernel<C, false, true>), grid, dim3(C::NT), 0, lc_s(s), a);
        return lc_launch_status();
    }
#define LC_S2_LAUNCH(T)                                                                                          \
    do {                                                                                                         \
        if (a.ostats) hipLaunchKernelGGL((conv_s2_shared_w_kernel<T, true>), grid, dim3(256), 0, lc_s(s), a);    \
        else hipLaunchKernelGGL((conv_s2_shared_w_kernel<T, false>), grid, dim3(256), 0, lc_s(s), a);            \
    } while (0)
    if (tpb == 4) LC_S2_LAUNCH(4);
    else if (tpb == 2) LC_S2_LAUNCH(2);
    else LC_S2_LAUNCH(1);
#undef LC_S2_LAUNCH
    return lc_launch_status();
}

LC_TOUCH_TU(conv_f16x2_s2, fir_down2_prefilter_split_kernel)
