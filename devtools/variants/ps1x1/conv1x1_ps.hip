// CANDIDATE (not part of the product library; see README.md in this directory): 1x1 convolution of a PRE-SPLIT activation.
//
// Why: the 1x1 projections of the layout model (qkv 256 -> 768, 512 -> 1536, ...) run on conv_f16x2_kernel<KS = 1>, where
// every 64-channel output block stages and SPLITS the same fp32 input tile again (no tap reuse in a 1x1 conv): 57 us per
// launch, 1.03 ms of the 9.05 ms C3 step (profiles/r04_cond_kernel_stats.csv).  The GroupNorm in
// front of these projections can already write fp16 hi / lo planes (lc_groupnorm_apply_split: xsp[b][plane][c/8][p][8],
// pre-multiplied by the layer's x_scale); this kernel consumes them with LDS-DMA only -- no VALU in the K loop:
//
//   block  = 8 waves, 128 output channels x 256 pixels (a 1x1 conv has no spatial structure: pixels = the H*W plane)
//   wave   = 64 channels x 64 pixels (2 x 2 accumulators of 32 x 32), waves 2 (channels) x 4 (pixels)
//   chunk  = 32 input channels (4 units of 8): 2 k-steps x 12 MFMAs per wave between two barriers, double-buffered LDS
//            (48 KB per buffer: x 2 planes x 4 x 256 units, w 2 planes x 4 x 128 units), 48 DMA wave-instructions per chunk
//
// Same arithmetic as every f16x2 kernel of the library: wh*xh + wh*xl + wl*xh in fp32 (v_mfma_f32_32x32x16_f16),
// result * (1 / (x_scale * w_scale)) + bias (+ res), * out_scale.  Needs Ci % 32 == 0.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void* lds_vptr;

struct lc_conv_range { float x_scale, x_unscale, amax_scaled, reserved; };

namespace {

#ifndef CAND_NBUF
#define CAND_NBUF 2     // LDS chunk buffers: 2 = the next chunk lands under this chunk's MFMAs (measured 2-6 % ahead), 3 = prefetch distance of two chunks
#endif
constexpr int CBK = 4, NBUF = CAND_NBUF;
constexpr unsigned OOB = 0x80000000u;

// block = WCO x WPX waves, each 64 output channels x 64 pixels
template <int WCO_, int WPX_>
struct Cfg {
    static constexpr int WCO = WCO_, WPX = WPX_, NWV = WCO * WPX, NT = 64 * NWV;
    static constexpr int BN = 64 * WCO, BP = 64 * WPX;
    static constexpr int XS = CBK * BP, WS = CBK * BN;            // units per plane per buffer
    static constexpr int BUF = 2 * XS + 2 * WS;
    static constexpr int NX = 2 * XS / 64, NW = 2 * WS / 64;      // DMA wave-instructions per chunk
    static constexpr int IPW = (NX + NW) / NWV;                   // per wave
    static constexpr int BLOCKS_PER_CU = (NBUF * BUF * 16 <= 80 * 1024 && NT <= 256) ? 2 : 1;
    static_assert(NX % NWV == 0 && NW % NWV == 0, "slot kinds must not depend on the wave");
    static_assert(NBUF * BUF * 16 <= 160 * 1024, "LDS");
};

struct P1Args {
    const half8* xsp; long long xsp_bs; int C8, P;
    const half8* wh; int Cib, Cop;
    const float* bias; const float* res; long long res_bs;
    float* y; long long y_bs;
    int B, Co; float out_scale;
    const lc_conv_range* range; const float* wmeta;
};

__device__ __forceinline__ void lds_dma16(__amdgpu_buffer_rsrc_t rs, lds_vptr dst, unsigned voff, unsigned soff) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, dst, 16, voff, soff, 0, 0);
#endif
}

template <class C>
__global__ __launch_bounds__(C::NT, C::BLOCKS_PER_CU) void conv1x1_ps_kernel(P1Args a) {
    constexpr int BN = C::BN, BP = C::BP, XS = C::XS, WS = C::WS, BUF = C::BUF, NX = C::NX, NW = C::NW, IPW = C::IPW;
    constexpr int NWV = C::NWV;
    __shared__ half8 lds[NBUF * BUF];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, kh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wco = wave / C::WPX, wpx = wave % C::WPX;
    const int tiles_p = (a.P + BP - 1) / BP;
    const int b = blockIdx.x / tiles_p, p0 = (blockIdx.x - b * tiles_p) * BP;
    const int co0 = blockIdx.y * BN;

    const unsigned xbytes = 2u * (unsigned)a.C8 * (unsigned)a.P * 16u;
    __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)(a.xsp + (long long)b * a.xsp_bs), 0, xbytes, 0x00020000);
    const unsigned wplane = (unsigned)a.Cib * (unsigned)a.Cop;                 // units per weight plane (1 tap)
    __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)a.wh, 0, 2u * wplane * 16u, 0x00020000);

    // this wave's DMA instructions of a chunk: j = wave + NWV k; j < NX moves x, the rest weights
    // (NX is a multiple of the wave count: the KIND of slot k is the same for every wave -- static, no branch in the issue code)
    unsigned voff[IPW];
    int ldsoff[IPW];
#pragma unroll
    for (int k = 0; k < IPW; ++k) {
        const int j = wave + NWV * k;
        if (k < NX / NWV) {
            const int plane = j / (NX / 2), rem = j - plane * (NX / 2);
            const int cb = rem / (BP / 64), q = rem - cb * (BP / 64);
            const int p = p0 + q * 64 + lane;
            ldsoff[k] = plane * XS + cb * BP + q * 64;
            voff[k] = p < a.P ? (unsigned)((plane * a.C8 + cb) * a.P + p) * 16u : OOB;
        } else {
            const int jw = j - NX;
            const int plane = jw / (NW / 2), rem = jw - plane * (NW / 2);
            const int cb = rem / (BN / 64), q = rem - cb * (BN / 64);
            const int cu = co0 + q * 64 + lane;
            ldsoff[k] = 2 * XS + plane * WS + cb * BN + q * 64;
            voff[k] = cu < a.Cop ? ((unsigned)(cb * a.Cop + cu) + plane * wplane) * 16u : OOB;
        }
    }
    const unsigned x_chunk = (unsigned)CBK * (unsigned)a.P * 16u;      // bytes between K chunks (x)
    const unsigned w_chunk = (unsigned)CBK * (unsigned)a.Cop * 16u;    // ... (weights)
    auto issue = [&](half8* buf, int ch) {
#pragma unroll
        for (int k = 0; k < IPW; ++k)
            if (k < NX / NWV) lds_dma16(rs_x, (lds_vptr)(buf + ldsoff[k]), voff[k], (unsigned)ch * x_chunk);
            else lds_dma16(rs_w, (lds_vptr)(buf + ldsoff[k]), voff[k], (unsigned)ch * w_chunk);
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int nchunk = a.C8 / CBK;
    // prologue: NBUF - 1 chunks in flight; steady state: issue chunk ch + NBUF - 1, compute chunk ch, then wait until
    // chunk ch + 1 has landed (everything but the newest NBUF - 2 chunks' instructions) and meet at the barrier
#pragma unroll
    for (int c = 0; c < NBUF - 1; ++c)
        if (c < nchunk) issue(lds + c * BUF, c);
    if (NBUF == 3 && nchunk > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(IPW) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int ib = 0;                                            // buffer of chunk ch
    for (int ch = 0; ch < nchunk; ++ch) {
        const int nb = ib + NBUF - 1 >= NBUF ? ib - 1 : ib + NBUF - 1;      // (ib + NBUF - 1) % NBUF
        if (ch + NBUF - 1 < nchunk) issue(lds + nb * BUF, ch + NBUF - 1);
        const half8* xh = lds + ib * BUF;
        const half8* xl = xh + XS;
        const half8* wh = xh + 2 * XS;
        const half8* wl = wh + WS;
#pragma unroll
        for (int ks = 0; ks < CBK / 2; ++ks) {
            half8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                ah[i] = wh[(2 * ks + kh) * BN + wco * 64 + i * 32 + l31];
                al[i] = wl[(2 * ks + kh) * BN + wco * 64 + i * 32 + l31];
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                bh[j] = xh[(2 * ks + kh) * BP + wpx * 64 + j * 32 + l31];
                bl[j] = xl[(2 * ks + kh) * BP + wpx * 64 + j * 32 + l31];
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                }
        }
        // chunk ch + 1 must have landed; with three buffers the chunk issued in this iteration may still be in flight
        if (NBUF == 3 && ch + 2 < nchunk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(IPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        ib = ib + 1 == NBUF ? 0 : ib + 1;
    }

    // ---- epilogue: accumulator register r of a lane = channel (r & 3) + 8 (r >> 2) + 4 kh of the 32, pixel l31 ----------
    const float out_unscale = a.range->x_unscale * a.wmeta[1];
    float* yb = a.y + (long long)b * a.y_bs;
    const float* rb = a.res ? a.res + (long long)b * a.res_bs : nullptr;
    const int co_lane = co0 + wco * 64 + 4 * kh;
    float bias_r[2][16];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co_lane + i * 32 + (r & 3) + 8 * (r >> 2);
            bias_r[i][r] = (a.bias && co < a.Co) ? a.bias[co] : 0.0f;
        }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int p = p0 + wpx * 64 + j * 32 + l31;
        const bool pok = p < a.P;
        float res_r[2][16];                                 // all residual loads of this pixel column in flight together
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co_lane + i * 32 + (r & 3) + 8 * (r >> 2);
                res_r[i][r] = (rb && pok && co < a.Co) ? rb[(long long)co * a.P + p] : 0.0f;
            }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co_lane + i * 32 + (r & 3) + 8 * (r >> 2);
                if (pok && co < a.Co)
                    yb[(long long)co * a.P + p] = ((acc[i][j][r] * out_unscale + bias_r[i][r]) + res_r[i][r]) * a.out_scale;
            }
    }
}

}  // namespace

// x_split: [B][2][Ci/8][P][8] halves as lc_groupnorm_apply_split writes them (P = H * W); wp_hi / wp_lo, wmeta: the ks = 1
// pack of lc_pack_conv_weight_f16x2 (lo plane directly behind the hi plane); y, res: fp32 [B][Co][P] with batch strides.
extern "C" int cand_conv1x1_ps_fwd(const void* x_split, const void* wp_hi, const void* wp_lo, const float* bias,
                                   const float* res, int64_t res_bs, float* y, int64_t y_bs, int B, int Ci, int Co,
                                   int P, float out_scale, const float* wmeta, const lc_conv_range* range,
                                   int cfg, void* stream) {
    if (!x_split || !wp_hi || !wp_lo || !y || !wmeta || !range || B <= 0 || Ci <= 0 || Co <= 0 || P <= 0) return -1;
    if (Ci % 32) return -2;
    if ((long long)2 * (Ci / 8) * P * 16 >= (1ll << 31)) return -2;
    P1Args a;
    a.xsp = (const half8*)x_split; a.C8 = Ci / 8; a.P = P; a.xsp_bs = (long long)2 * (Ci / 8) * P;
    a.wh = (const half8*)wp_hi; a.Cib = (Ci + 15) / 16 * 2; a.Cop = (Co + 63) / 64 * 64;
    if ((const half8*)wp_lo != a.wh + (long long)a.Cib * a.Cop) return -1;
    a.bias = bias; a.res = res; a.res_bs = res_bs; a.y = y; a.y_bs = y_bs;
    a.B = B; a.Co = Co; a.out_scale = out_scale; a.range = range; a.wmeta = wmeta;
    auto launch = [&](auto cfg) {
        using C = decltype(cfg);
        dim3 grid((unsigned)(B * ((P + C::BP - 1) / C::BP)), (unsigned)((Co + C::BN - 1) / C::BN));
        hipLaunchKernelGGL(conv1x1_ps_kernel<C>, grid, dim3(C::NT), 0, (hipStream_t)stream, a);
    };
    switch (cfg) {
        case 0: launch(Cfg<2, 4>{}); break;      // 128 co x 256 px, 8 waves, 1 block per CU  (the first measurement)
        case 1: launch(Cfg<2, 2>{}); break;      // 128 co x 128 px, 4 waves, 2 blocks per CU
        case 2: launch(Cfg<4, 2>{}); break;      // 256 co x 128 px, 8 waves, 1 block per CU  (half the x re-reads)
        case 3: launch(Cfg<1, 4>{}); break;      //  64 co x 256 px, 4 waves, 2 blocks per CU (the narrow projections)
        default: return -2;
    }
    return (int)hipGetLastError();
}
