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
                e[k] = in * (f0 * w.a[k] + f1 * w.m[k].x + f2 * w.m[k].y + f3 * w.c[k].x);          // column 2 i
                o[k] = in * (f0 * w.m[k].x + f1 * w.m[k].y + f2 * w.c[k].x + f3 * w.c[k].y);        // column 2 i + 1
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
                fe[k] = f0 * we[0][k] + f1 * we[1][k] + f2 * we[2][k] + f3 * we[3][k];
                fo[k] = f0 * wo[0][k] + f1 * wo[1][k] + f2 * wo[2][k] + f3 * wo[3][k];
            }
            put(m + 1, fe, fo);
            if (m == 1) {                                               // top variant: F[1] without f0 * XW[0]
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    fe[k] = f1 * we[1][k] + f2 * we[2][k] + f3 * we[3][k];
                    fo[k] = f1 * wo[1][k] + f2 * wo[2][k] + f3 * wo[3][k];
                }
                put(H + 1, fe, fo);
            }
            if (m == H - 3) {                                           // bottom variant: F[H - 3] without f3 * XW[H - 1]
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    fe[k] = f0 * we[0][k] + f1 * we[1][k] + f2 * we[2][k];
                    fo[k] = f0 * wo[0][k] + f1 * wo[1][k] + f2 * wo[2][k];
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
    // rows per strip: as long as possible (3 halo rows are re-read per strip) while the launch still has ~16 waves per CU
    const long long per_row_strip = (long long)B * (C / 8) * ((W / 2 + 63) / 64);   // waves per strip index
    int strip_len = 8;
    while (strip_len > 1 && per_row_strip * ((H + strip_len) / strip_len) < 4096) strip_len >>= 1;
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
    if (a.ostats) hipLaunchKernelGGL((conv_f16x2_ps_kernel<C, true, true>), grid, dim3(C::NT), 0, lc_s(s), a);
    else hipLaunchKernelGGL((conv_f16x2_ps_kernel<C, false, true>), grid, dim3(C::NT), 0, lc_s(s), a);
    return lc_launch_status();
}

LC_TOUCH_TU(conv_f16x2_s2, fir_down2_prefilter_split_kernel)
