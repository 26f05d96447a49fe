// Ring-padded 3x3 / plain 1x1 convolution as an implicit GEMM on the fp32 matrix cores.
//
// Replaces ops.Conv2d + ops.Pad of the reference (lidargen/models/unets/ops.py:32-49,149-173):
// the reference materialises two padded copies of the input (F.pad circular in W, F.pad zeros
// in H) and calls cuDNN; here the halo is built while staging the input tile into LDS, so the
// padded tensor never exists in HBM.
//
// GEMM view:  D[co][px] = sum_{tap,ci} Wp[tap][ci][co] * X[ci][px + tap]
//   A operand (32 rows i)  = 32 output channels  (weights, packed so co is contiguous)
//   B operand (32 cols j)  = 32 consecutive W positions of one image row
//   v_mfma_f32_32x32x2_f32: lane l supplies A[i=l&31][k=l>>5], B[k=l>>5][j=l&31]; k = ci pair.
//   D layout: lane l holds column j=l&31 (pixel), rows (r&3)+8*(r>>2)+4*(l>>5) (co) -> every
//   store instruction writes two full 128-B row segments of the NCHW output.
// fp32 MFMA is bit-for-bit an fmaf chain (exact fp32), peak 157.3 TFLOP/s = the roofline.
//
// Block = 256 threads = 4 waves arranged WCO x WPX; each wave owns TCO x TPX MFMA tiles.
// K loop: chunks of CK input channels; chunk c+1 is prefetched global->registers while the
// MFMAs of chunk c run from LDS (register-staged double buffering, one LDS buffer).
#include "common.h"

namespace {

struct ConvArgs {
    const float* x;
    const float* wp;
    const float* bias;
    const float* res;
    float* y;
    long long x_bs, res_bs, y_bs;
    int B, Ci, Co, H, W, Cip, Cop;
    int tiles_h, tiles_w;
    float out_scale;
};

template <int WCO, int WPX, int TCO, int TPX, int TH, int TW, int KS>
struct ConvCfg {
    static constexpr int CK = 8;                        // input channels per K chunk
    static constexpr int HALO = KS / 2;
    static constexpr int NTAP = KS * KS;
    static constexpr int BN = WCO * TCO * 32;           // output channels per block
    static constexpr int NPT = WPX * TPX;               // 32-pixel tiles per block
    static constexpr int TPR = TW / 32;                 // pixel tiles per tile row
    static constexpr int XR = TH + 2 * HALO;            // staged rows
    static constexpr int XW = TW + 2 * HALO;            // staged cols
    static constexpr int EX = CK * XR * XW;             // staged x elements per chunk
    static constexpr int NXL = (EX + 255) / 256;        // x elements per thread
    static constexpr int EW4 = NTAP * CK * BN / 4;      // staged weight float4s per chunk
    static constexpr int NWL = (EW4 + 255) / 256;
    static constexpr int LDS_FLOATS = EX + NTAP * CK * BN;
    static_assert(NPT * 32 == TH * TW, "tile shape");
    static_assert(WCO * WPX == 4, "4 waves");
};

template <class C>
__global__ __launch_bounds__(256, 2) void conv_ring_kernel(ConvArgs a) {
    constexpr int CK = C::CK, HALO = C::HALO, NTAP = C::NTAP, BN = C::BN;
    constexpr int XR = C::XR, XW = C::XW, EX = C::EX, NXL = C::NXL, EW4 = C::EW4, NWL = C::NWL;
    constexpr int KS = 2 * HALO + 1;
    __shared__ __attribute__((aligned(16))) float lds[C::LDS_FLOATS];
    float* xs = lds;            // [CK][XR][XW]
    float* ws = lds + EX;       // [NTAP][CK][BN]

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

    // ---- per-thread staging coordinates (chunk independent) ---------------------------------
    int x_off[NXL];      // (ci << 24 | offset inside one channel plane), or -1 when padding
#pragma unroll
    for (int i = 0; i < NXL; ++i) {
        const int e = tid + i * 256;
        const int ci = e / (XR * XW);
        const int rem = e - ci * (XR * XW);
        const int r = rem / XW, c = rem - r * XW;
        const int gh = h0 - HALO + r;
        int gw = w0 - HALO + c;
        gw %= W; if (gw < 0) gw += W;                     // ring in W
        const bool ok = (e < EX) && gh >= 0 && gh < H;     // zeros in H
        x_off[i] = ok ? ((ci << 24) | (gh * W + gw)) : -1;
    }

    float xr[NXL];
    f32x4 wr[NWL];
    auto load_chunk = [&](int c0) {
#pragma unroll
        for (int i = 0; i < NXL; ++i) {
            const int cg = c0 + (x_off[i] >> 24);
            const bool ok = x_off[i] >= 0 && cg < a.Ci;
            xr[i] = ok ? xb[(long long)cg * HW + (x_off[i] & 0xFFFFFF)] : 0.0f;
        }
#pragma unroll
        for (int i = 0; i < NWL; ++i) {
            const int e = tid + i * 256;
            if (e < EW4) {
                const int row = e / (BN / 4);                 // tap*CK + ci
                const int c4 = e - row * (BN / 4);
                const int tap = row / CK, ci = row - tap * CK;
                const float* p = a.wp + ((long long)tap * a.Cip + (c0 + ci)) * a.Cop + co0 + c4 * 4;
                // packed rows are padded to 64 output channels: the upper half of a 128-channel
                // block may lie past the row (Co % 128 in (0, 64]) -- its outputs are never stored
                wr[i] = co0 + c4 * 4 < a.Cop ? *reinterpret_cast<const f32x4*>(p) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int i = 0; i < NXL; ++i) {
            const int e = tid + i * 256;
            if (e < EX) xs[e] = xr[i];
        }
#pragma unroll
        for (int i = 0; i < NWL; ++i) {
            const int e = tid + i * 256;
            if (e < EW4) *reinterpret_cast<f32x4*>(ws + e * 4) = wr[i];
        }
    };

    f32x16 acc[C::TCO_][C::TPX_];
#pragma unroll
    for (int i = 0; i < C::TCO_; ++i)
#pragma unroll
        for (int j = 0; j < C::TPX_; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // per-wave LDS read bases
    const int kh = lane >> 5, l31 = lane & 31;
    int xbase[C::TPX_];
#pragma unroll
    for (int j = 0; j < C::TPX_; ++j) {
        const int t = wpx * C::TPX_ + j;
        const int tr = t / C::TPR, tc = t - tr * C::TPR;
        xbase[j] = kh * (XR * XW) + tr * XW + tc * 32 + l31;
    }
    const int wbase = kh * BN + wco * C::TCO_ * 32 + l31;

    const int nchunk = a.Cip / CK;
    load_chunk(0);
    store_chunk();
    __syncthreads();
    for (int ch = 0; ch < nchunk; ++ch) {
        if (ch + 1 < nchunk) load_chunk((ch + 1) * CK);
#pragma unroll
        for (int tap = 0; tap < NTAP; ++tap) {
            const int dy = tap / KS, dx = tap - dy * KS;
#pragma unroll
            for (int kk = 0; kk < CK / 2; ++kk) {
                float av[C::TCO_], bv[C::TPX_];
#pragma unroll
                for (int i = 0; i < C::TCO_; ++i)
                    av[i] = ws[(tap * CK + 2 * kk) * BN + wbase + i * 32];
#pragma unroll
                for (int j = 0; j < C::TPX_; ++j)
                    bv[j] = xs[2 * kk * (XR * XW) + xbase[j] + dy * XW + dx];
#pragma unroll
                for (int i = 0; i < C::TCO_; ++i)
#pragma unroll
                    for (int j = 0; j < C::TPX_; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
            }
        }
        __syncthreads();
        if (ch + 1 < nchunk) {
            store_chunk();
            __syncthreads();
        }
    }

    // ---- epilogue: + bias (+ residual), * scale, 128-B row-segment stores --------------------
    float* yb = a.y + (long long)b * a.y_bs;
    const float* rb = a.res ? a.res + (long long)b * a.res_bs : nullptr;
#pragma unroll
    for (int j = 0; j < C::TPX_; ++j) {
        const int t = wpx * C::TPX_ + j;
        const int tr = t / C::TPR, tc = t - tr * C::TPR;
        const int gh = h0 + tr, gw = w0 + tc * 32 + l31;
        const bool pok = gh < H && gw < W;
        const long long poff = (long long)gh * W + gw;
#pragma unroll
        for (int i = 0; i < C::TCO_; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + (wco * C::TCO_ + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (pok && co < a.Co) {
                    float v = acc[i][j][r];
                    if (a.bias) v += a.bias[co];
                    if (rb) v += rb[(long long)co * HW + poff];
                    yb[(long long)co * HW + poff] = v * a.out_scale;
                }
            }
        }
    }
}

template <int WCO, int WPX, int TCO, int TPX, int TH, int TW, int KS>
struct Cfg : ConvCfg<WCO, WPX, TCO, TPX, TH, TW, KS> {
    static constexpr int WCO_ = WCO, WPX_ = WPX, TCO_ = TCO, TPX_ = TPX, TH_ = TH, TW_ = TW;
};

template <class C>
int launch_conv(ConvArgs a, hipStream_t st) {
    a.tiles_h = (a.H + C::TH_ - 1) / C::TH_;
    a.tiles_w = (a.W + C::TW_ - 1) / C::TW_;
    dim3 grid(a.B * a.tiles_h * a.tiles_w, (a.Co + C::BN - 1) / C::BN);
    hipLaunchKernelGGL(conv_ring_kernel<C>, grid, dim3(256), 0, st, a);
    return lc_launch_status();
}

// tile configurations (cfg id -> shape); see DESIGN.md "conv tile selection"
//  1: 128 co x 128 px (2x64)   2: 64 co x 256 px (4x64)   3: 64 co x 64 px (2x32)
//  4: 128 co x 128 px (4x32)   5: 64 co x 128 px (1x128)... kept small on purpose.
template <int KS>
int dispatch(int cfg, const ConvArgs& a, hipStream_t st) {
    switch (cfg) {
        case 1: return launch_conv<Cfg<2, 2, 2, 2, 2, 64, KS>>(a, st);
        case 2: return launch_conv<Cfg<1, 4, 2, 2, 4, 64, KS>>(a, st);
        case 3: return launch_conv<Cfg<2, 2, 1, 1, 2, 32, KS>>(a, st);
        case 4: return launch_conv<Cfg<2, 2, 2, 2, 4, 32, KS>>(a, st);
        case 5: return launch_conv<Cfg<1, 4, 2, 1, 2, 64, KS>>(a, st);
        default: return LC_EUNSUP;
    }
}

int auto_cfg(int B, int Co, int H, int W) {
    // Measured on MI355X at batch 8 (devtools/conv_sweep.py, profiles/r01_conv_sweep.txt):
    //   Co <= 64  : 64co x 128px blocks (cfg 5) reach 106-114 TF, 64x64 (cfg 3) 100-110 TF
    //   Co >= 128 : 128co x 128px (cfg 4 / 1) reach 108-122 TF when they still give >= 2 blocks
    //               per CU; otherwise the 64x64 tile (cfg 3) keeps all 256 CUs busy (L3 512->512:
    //               100 TF vs 56 TF).
    const long long px = (long long)B * H * W;
    auto blocks = [&](int bn, int pxb) { return ((Co + bn - 1) / bn) * ((px + pxb - 1) / pxb); };
    if (Co > 64 && blocks(128, 128) >= 512) {
        if (H % 4 == 0 && W % 32 == 0) return 4;
        if (H % 2 == 0 && W % 64 == 0) return 1;
    }
    if (Co <= 64 && H % 2 == 0 && W % 64 == 0 && blocks(64, 128) >= 512) return 5;
    return 3;
}

__global__ void pack_weight_kernel(const float* __restrict__ w, float* __restrict__ wp, int Co,
                                   int Ci, int ntap, int Cip, int Cop) {
    const long long n = (long long)ntap * Cip * Cop;
    for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < n;
         e += (long long)gridDim.x * blockDim.x) {
        const int co = e % Cop;
        const long long r = e / Cop;
        const int ci = r % Cip;
        const int tap = r / Cip;
        wp[e] = (co < Co && ci < Ci) ? w[((long long)co * Ci + ci) * ntap + tap] : 0.0f;
    }
}

}  // namespace

extern "C" int64_t lc_packed_conv_weight_elems(int Co, int Ci, int ks) {
    const int64_t Cip = (Ci + 7) / 8 * 8, Cop = (Co + 63) / 64 * 64;
    return (int64_t)ks * ks * Cip * Cop;
}

extern "C" int lc_pack_conv_weight(const float* w, float* wp, int Co, int Ci, int ks,
                                   lc_stream_t s) {
    if (!w || !wp || Co <= 0 || Ci <= 0 || (ks != 1 && ks != 3)) return LC_EINVAL;
    const int Cip = (Ci + 7) / 8 * 8, Cop = (Co + 63) / 64 * 64;
    const long long n = (long long)ks * ks * Cip * Cop;
    const int blocks = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    hipLaunchKernelGGL(pack_weight_kernel, dim3(blocks), dim3(256), 0, lc_s(s), w, wp, Co, Ci,
                       ks * ks, Cip, Cop);
    return lc_launch_status();
}

extern "C" int lc_conv2d_ring_fwd(const float* x, int64_t x_bs, const float* wp, const float* bias,
                                  const float* res, int64_t res_bs, float* y, int64_t y_bs, int B,
                                  int Ci, int Co, int H, int W, int ks, float out_scale,
                                  int tile_cfg, lc_stream_t s) {
    if (!x || !wp || !y || B <= 0 || Ci <= 0 || Co <= 0 || H <= 0 || W <= 0) return LC_EINVAL;
    if (ks != 1 && ks != 3) return LC_EUNSUP;
    ConvArgs a;
    a.x = x; a.wp = wp; a.bias = bias; a.res = res; a.y = y;
    a.x_bs = x_bs; a.res_bs = res_bs; a.y_bs = y_bs;
    a.B = B; a.Ci = Ci; a.Co = Co; a.H = H; a.W = W;
    a.Cip = (Ci + 7) / 8 * 8; a.Cop = (Co + 63) / 64 * 64;
    a.out_scale = out_scale;
    a.tiles_h = a.tiles_w = 0;
    if (tile_cfg == 0) tile_cfg = auto_cfg(B, Co, H, W);
    return ks == 3 ? dispatch<3>(tile_cfg, a, lc_s(s)) : dispatch<1>(tile_cfg, a, lc_s(s));
}

LC_TOUCH_TU(conv, pack_weight_kernel)
