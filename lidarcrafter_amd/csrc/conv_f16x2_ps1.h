// 1x1 convolution of a PRE-SPLIT activation (included by conv_f16x2.hip behind the pre-split 3x3 kernel).
//
// A 1x1 conv has no tap reuse: on conv_f16x2_kernel<KS = 1> every 64-channel output block stages and SPLITS the same fp32
// input tile again (the qkv projections of the layout model have 12 / 24 such blocks per pixel tile: 57 us per launch,
// 1.03 ms of the 9.05 ms C3 step in round 4).  When the GroupNorm in front writes fp16 hi / lo planes anyway
// (lc_groupnorm_apply*_split: xsp[b][plane][c/8][p][8], pre-multiplied by the layer's x_scale), this kernel consumes
// them with LDS-DMA only -- no VALU in the K loop:
//
//   block  = 8 waves, 128 output channels x 256 pixels (no spatial structure: pixels = the H*W plane)
//   wave   = 64 channels x 64 pixels (2 x 2 accumulators of 32 x 32), waves 2 (channels) x 4 (pixels)
//   chunk  = 32 input channels (4 units of 8): 2 k-steps x 12 MFMAs per wave between two barriers, two LDS buffers of
//            48 KB (x: 2 planes x 4 x 256 units, w: 2 planes x 4 x 128 units), 48 DMA wave-instructions per chunk
//
// Same arithmetic as every f16x2 kernel: wh*xh + wh*xl + wl*xh in fp32, * 1 / (x_scale * w_scale), + bias (+ res),
// * out_scale.  Needs Ci % 32 == 0.  First measurement (the candidate's harness of round 5): GroupNorm + projection
// 256 -> 768 @ 8 x 256, batch 8: 74.9 -> 54.4 us; 512 -> 1536 @ 4 x 128: 64.8 -> 39.9 us.
#pragma once
#include <type_traits>

namespace {

// BPV = pixels per block: 256 (8 waves, 2 x 48 KB of LDS: one block per CU) or 128 (4 waves, 2 x 32 KB: two blocks per CU --
// the shapes whose 256-pixel grid leaves CUs idle or runs a half-empty second round, e.g. 512 -> 1536 @ 4 x 128 at batch 8:
// 192 blocks on 256 CUs)
// BNV / MIV (round 6, third part): output channels per block and 32-channel accumulator rows per wave.  128 / 2 = the form
// above.  288 / 3 (6 waves of 96 channels x 64 pixels, 128-pixel blocks, 2 x 52 KB of LDS): the projection to the NINE tap
// planes of the up-path fold (csrc/upfold.hip: Co' = 9 Co, a multiple of 288 for every Co % 32 == 0) -- 9 Co x pixels
// then divides into blocks whose number is a multiple of the chip's 256 CUs at every shape of the denoisers (2304 x 4096:
// 256 blocks, where the 128 x 256 grid had 288 = 1.125 rounds), and a wave issues 36 MFMAs between two barriers instead of 24.
template <int BPV, int BNV = 128, int MIV = 2>
struct P1T {
    static constexpr int BN = BNV, BP = BPV, CBK = 4, MI = MIV, NWV = (BNV / (32 * MIV)) * (BPV / 64), NT = 64 * NWV;
    static constexpr int XS = CBK * BP, WS = CBK * BN;            // units per plane per buffer
    static constexpr int BUF = 2 * XS + 2 * WS;                   // 48 KB / 32 KB / 52 KB
    static constexpr int NX = 2 * XS / 64, NW = 2 * WS / 64;      // DMA wave-instructions per chunk: 32 + 16 / 16 + 16 / 16 + 36
    static constexpr int XPW = (NX + NWV - 1) / NWV;              // x slots per wave (the last may be absent: NX % NWV != 0)
    static constexpr int IPW = XPW + NW / NWV;                    // per wave: 6 / 8 / 9
    static_assert(BNV % (32 * MIV) == 0 && NW % NWV == 0 && WS % 64 == 0 && XS % 64 == 0, "the kind of a DMA slot must not depend on the wave");
};
using P1 = P1T<256>;

struct P1Args {
    const half8* xsp; long long xsp_bs; int C8, P;
    const half8* wh; int Cib, Cop;
    const float* bias; const float* res; long long res_bs;
    float* y; long long y_bs;
    int Co; float out_scale;
    const lc_conv_range* range; const float* wmeta;
    // QKV form (conv1x1_ps_kernel<true>): Co = 3 C; rows [0, C) are written to y as fp32, rows [C, 2C) and [2C, 3C) --
    // the keys and values of ObjectAwareCrossAttention, 32 channels per head -- go straight into the unit form the
    // attention kernel stages by LDS-DMA (csrc/attention_units.hip; kv: [B][heads][kv_tiles][768] units of 8 halves)
    half8* kv; int C, kv_heads, kv_tiles;
    // store form of the plain epilogue: 0 = one write-through dword per lane and register (two 128-byte runs per wave
    // instruction); 2 = each 32 x 32 accumulator tile transposed through the wave's own LDS strip and stored as 16 bytes per
    // lane (eight 128-byte rows per wave instruction), write-through -- measured on the nine-plane projections of the
    // up-path fold (output-heavy: 9 Co channels from one pass over K): 47 -> 23 us at 256 -> 2304 @ 8 x 4 x 128, 100 -> 68 at
    // 512 -> 4608, 184 -> 108 at 128 -> 1152 @ 8 x 16 x 512 (profiles/r06_fold_up.txt); write-back instead of write-through
    // changes neither form.  Form 2 needs P % 4 == 0, a 16-byte aligned y and no residual (the launcher falls back to 0).
    int st_mode;
};

// The 8-byte halves of a K unit / the 16-byte V units a lane of the QKV epilogue owns; the split is the attention kernels'
// own (attention_split.h: pre-scale 16, hi = 11 leading bits, lo = fp16(rest)).
__device__ __forceinline__ void qkv_split2(float a, float b, unsigned& hi, unsigned& lo) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const float s0 = a * 16.0f, s1 = b * 16.0f;
    const float h0 = __uint_as_float(__float_as_uint(s0) & 0xFFFFE000u);
    const float h1 = __uint_as_float(__float_as_uint(s1) & 0xFFFFE000u);
    hi = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(h0, h1));
    f2 r; r.x = s0 - h0; r.y = s1 - h1;
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, h2));
}

template <bool QKV, int BPV = 256, int BNV = 128, int MIV = 2>
__global__ __launch_bounds__((P1T<BPV, BNV, MIV>::NT), (BPV == 256 || BNV != 128 ? 1 : 2)) void conv1x1_ps_kernel(P1Args a) {
    using P1 = P1T<BPV, BNV, MIV>;
    constexpr int BN = P1::BN, BP = P1::BP, CBK = P1::CBK, XS = P1::XS, WS = P1::WS, BUF = P1::BUF, MI = P1::MI, WCO = 32 * MI;
    constexpr int NX = P1::NX, NW = P1::NW, IPW = P1::IPW, NWV = P1::NWV, WPX = BP / 64, XPW = P1::XPW;
    constexpr unsigned OOB = 0x80000000u;
    static_assert(!QKV || (BNV == 128 && MIV == 2), "the QKV epilogue is written for the 128 x 2 form");
    __shared__ half8 lds[2 * BUF];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, kh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wco = wave / WPX, wpx = wave % WPX;
    const int tiles_p = (a.P + BP - 1) / BP;
    const int b = __builtin_amdgcn_readfirstlane((int)(blockIdx.x / tiles_p)), p0 = (blockIdx.x - b * tiles_p) * BP;
    const int co0 = blockIdx.y * BN;

    const unsigned xbytes = 2u * (unsigned)a.C8 * (unsigned)a.P * 16u;
    // (the sample's base address through readfirstlane: the 64-bit product runs on the vector unit, and a descriptor in VGPRs
    //  makes hipcc wrap every x DMA of the K loop in a waterfall loop -- 4-5 per chunk and wave; without them the nine-plane
    //  projections run 3-7 % faster: 67.6 -> 63.4 us at 512 -> 4608, 71.9 -> 66.6 at 256 -> 2304 @ 8 x 8 x 256, profiles/r06_fold_up.txt)
    const unsigned long long xaddr = reinterpret_cast<unsigned long long>(a.xsp + (long long)b * a.xsp_bs);
    const unsigned long long xaddr_u = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(xaddr >> 32)) << 32) |
                                       (unsigned)__builtin_amdgcn_readfirstlane((int)xaddr);
    __amdgpu_buffer_rsrc_t rs_x =
        __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(xaddr_u), 0, xbytes, 0x00020000);
    const unsigned wplane = (unsigned)a.Cib * (unsigned)a.Cop;                 // units per weight plane (one tap)
    __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)a.wh, 0, 2u * wplane * 16u, 0x00020000);

    // this wave's DMA instructions of a chunk: j = wave + 8 k; slots k < NX / 8 move x, the rest weights (static kinds)
    unsigned voff[IPW];
    int ldsoff[IPW];
#pragma unroll
    for (int k = 0; k < IPW; ++k) {
        if (k < XPW) {
            const int j = wave + NWV * k;
            const int plane = j / (NX / 2), rem = j - plane * (NX / 2);
            const int cb = rem / (BP / 64), q = rem - cb * (BP / 64);
            const int p = p0 + q * 64 + lane;
            const bool slot = j < NX;                          // (wave-uniform; false only where NX % NWV != 0)
            ldsoff[k] = slot ? plane * XS + cb * BP + q * 64 : 0;
            voff[k] = slot && p < a.P ? (unsigned)((plane * a.C8 + cb) * a.P + p) * 16u : OOB;
        } else {
            // (a plane's image [cb 4][BN] is one run of WS units: a wave-instruction moves 64 consecutive ones, which may
            //  straddle two channel blocks when BN % 64 != 0 -- the source offset is per lane anyway)
            const int jw = wave + NWV * (k - XPW);
            const int plane = jw / (NW / 2), q = jw - plane * (NW / 2);
            const int u = q * 64 + lane;
            const int cb = u / BN, cu = co0 + (u - cb * BN);
            ldsoff[k] = 2 * XS + plane * WS + q * 64;
            voff[k] = cu < a.Cop ? ((unsigned)(cb * a.Cop + cu) + plane * wplane) * 16u : OOB;
        }
    }
    const unsigned x_chunk = (unsigned)CBK * (unsigned)a.P * 16u;      // bytes between K chunks (x)
    const unsigned w_chunk = (unsigned)CBK * (unsigned)a.Cop * 16u;    // ... (weights)
    auto issue = [&](half8* buf, int ch) {
#pragma unroll
        for (int k = 0; k < IPW; ++k) {
            if (k < XPW) {
                // (the last x slot of a wave does not exist where NX % NWV != 0: a wave-uniform branch, and every wait
                //  of this kernel is vmcnt(0))
                if (NX % NWV == 0 || k + 1 < XPW || wave + NWV * k < NX)
                    lds_dma16(rs_x, (lds_vptr)(buf + ldsoff[k]), voff[k], (unsigned)ch * x_chunk);
            } else {
                lds_dma16(rs_w, (lds_vptr)(buf + ldsoff[k]), voff[k], (unsigned)ch * w_chunk);
            }
        }
    };

    f32x16 acc[MI][2];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int nchunk = a.C8 / CBK;
    half8* cur = lds;
    half8* nxt = lds + BUF;
    // QKV: the blocks of the value rows run the MFMAs with the operands exchanged -- the accumulator of a lane is then
    // one CHANNEL x 16 pixels in the order the attention kernel contracts keys in, i.e. two whole V units
    const bool vt = QKV && co0 >= 2 * a.C;                   // (block-uniform)
    issue(cur, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // (the whole K loop exists twice in a QKV kernel: a branch around each k-step's MFMAs keeps hipcc from interleaving
    //  the fragment reads of one k-step with the MFMAs of the other -- measured +11 us at 512 -> 1536 @ 4 x 128)
    auto k_loop = [&](auto transposed) {
        constexpr bool VT = decltype(transposed)::value;
        for (int ch = 0; ch < nchunk; ++ch) {
            if (ch + 1 < nchunk) issue(nxt, ch + 1);          // lands while this chunk's MFMAs run
            const half8* xh = cur;
            const half8* xl = cur + XS;
            const half8* wh = cur + 2 * XS;
            const half8* wl = wh + WS;
#pragma unroll
            for (int ks = 0; ks < CBK / 2; ++ks) {
                half8 ah[MI], al[MI], bh[2], bl[2];
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    ah[i] = wh[(2 * ks + kh) * BN + wco * WCO + i * 32 + l31];
                    al[i] = wl[(2 * ks + kh) * BN + wco * WCO + i * 32 + l31];
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    bh[j] = xh[(2 * ks + kh) * BP + wpx * 64 + j * 32 + l31];
                    bl[j] = xl[(2 * ks + kh) * BP + wpx * 64 + j * 32 + l31];
                }
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        if (VT) {
                            if (LC_F16X2_TERMS & 2) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[j], al[i], acc[i][j], 0, 0, 0);
                            if (LC_F16X2_TERMS & 4) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[j], ah[i], acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[j], ah[i], acc[i][j], 0, 0, 0);
                        } else {
                            if (LC_F16X2_TERMS & 2) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
                            if (LC_F16X2_TERMS & 4) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                        }
                    }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            half8* t = cur; cur = nxt; nxt = t;
        }
    };
    if (QKV && vt) k_loop(std::true_type{});
    else k_loop(std::false_type{});

    // ---- epilogue: accumulator register r of a lane = channel (r & 3) + 8 (r >> 2) + 4 kh of the 32, pixel l31 ----------
    const float out_unscale = a.range->x_unscale * a.wmeta[1];
    float* yb = a.y + (long long)b * a.y_bs;
    const float* rb = a.res ? a.res + (long long)b * a.res_bs : nullptr;
    const int co_lane = co0 + wco * WCO + 4 * kh;
    float bias_r[MI][16];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co_lane + i * 32 + (r & 3) + 8 * (r >> 2);
            // (QKV value blocks: the accumulators are transposed, a lane owns ONE channel per 32-block)
            const int cb_ = (QKV && vt) ? co0 + wco * WCO + i * 32 + l31 : co;
            bias_r[i][r] = (a.bias && cb_ < a.Co) ? a.bias[cb_] : 0.0f;
        }
    if (QKV && co0 >= a.C) {
        // ---- keys / values -> unit form.  Tile image (768 units): K hi [cb 8][key 32] | K lo | V hi [step 2][half 2][c 32] | V lo
        const float us = out_unscale, os = a.out_scale;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int cpart = (vt ? co0 - 2 * a.C : co0 - a.C) + wco * 64 + i * 32;      // first channel of this 32-block
            const int head = cpart >> 5;                                                  // = one head's 32 channels
            half8* hb = a.kv + ((long long)(b * a.kv_heads + head) * a.kv_tiles) * 768;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int pg = p0 + wpx * 64 + j * 32;                                    // 32 pixels = one key tile
                if (pg >= a.P) continue;
                half8* tile = hb + (long long)(pg >> 5) * 768;
                if (!vt) {
                    // lane = key l31, registers = channels (r & 3) + 8 (r >> 2) + 4 kh: the kh-th 8-byte half of units cb = r >> 2
#pragma unroll
                    for (int cb = 0; cb < 4; ++cb) {
                        float v[4];
#pragma unroll
                        for (int m = 0; m < 4; ++m) v[m] = (acc[i][j][4 * cb + m] * us + bias_r[i][4 * cb + m]) * os;
                        uint2 hi, lo;
                        qkv_split2(v[0], v[1], hi.x, lo.x);
                        qkv_split2(v[2], v[3], hi.y, lo.y);
                        *(reinterpret_cast<uint2*>(&tile[cb * 32 + l31]) + kh) = hi;
                        *(reinterpret_cast<uint2*>(&tile[256 + cb * 32 + l31]) + kh) = lo;
                    }
                } else {
                    // lane = channel l31, registers = keys (r & 3) + 8 (r >> 2) + 4 kh: registers 8 s .. 8 s + 7 = unit (s, kh, l31)
                    const float bv = bias_r[i][0];
#pragma unroll
                    for (int st = 0; st < 2; ++st) {
                        uint4 hi, lo;
                        float v[8];
#pragma unroll
                        for (int m = 0; m < 8; ++m) v[m] = (acc[i][j][8 * st + m] * us + bv) * os;
                        qkv_split2(v[0], v[1], hi.x, lo.x);
                        qkv_split2(v[2], v[3], hi.y, lo.y);
                        qkv_split2(v[4], v[5], hi.z, lo.z);
                        qkv_split2(v[6], v[7], hi.w, lo.w);
                        *reinterpret_cast<uint4*>(&tile[512 + (st * 2 + kh) * 32 + l31]) = hi;
                        *reinterpret_cast<uint4*>(&tile[640 + (st * 2 + kh) * 32 + l31]) = lo;
                    }
                }
            }
        }
        return;
    }
    if (a.st_mode >= 2 && !rb) {
        // (behind the K loop's last barrier: the operand buffers are free.  A strip = 32 rows x 36 floats, rows 16-byte
        //  aligned; the wave writes its tile in the accumulator layout and reads it back as rows -- LDS operations of one
        //  wave execute in order, the wave barriers only keep the compiler from moving them across each other)
        float* strip = reinterpret_cast<float*>(lds) + wave * (32 * 36);
        const unsigned long long ya = reinterpret_cast<unsigned long long>(yb);            // (uniform: keeps the descriptor in SGPRs)
        const unsigned long long ya_u = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(ya >> 32)) << 32) |
                                        (unsigned)__builtin_amdgcn_readfirstlane((int)ya);
        const __amdgpu_buffer_rsrc_t rs_yb = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(ya_u), 0, 0x7FFFFFFFu, 0x00020000);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    strip[((r & 3) + 8 * (r >> 2) + 4 * kh) * 36 + l31] = (acc[i][j][r] * out_unscale + bias_r[i][r]) * a.out_scale;
                __builtin_amdgcn_wave_barrier();
                const int cob = co0 + wco * WCO + i * 32, pb_ = p0 + wpx * 64 + j * 32 + 4 * (lane & 7);
                // (all four reads first, into four register quads: no instruction may write the data registers of a
                //  > 64-bit store directly behind it -- tests/test_isa_audit.py; the whole offset rides in the VGPR)
                f32x4 v[4];
#pragma unroll
                for (int it = 0; it < 4; ++it)
                    v[it] = *reinterpret_cast<const f32x4*>(strip + (it * 8 + (lane >> 3)) * 36 + 4 * (lane & 7));
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int row = it * 8 + (lane >> 3);
                    const unsigned off = cob + row < a.Co && pb_ < a.P ? ((unsigned)(cob + row) * (unsigned)a.P + (unsigned)pb_) * 4u : 0x80000000u;
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(lc_u32x4, v[it]), rs_yb, off, 0, 16);   // sc1
                }
            }
        return;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int p = p0 + wpx * 64 + j * 32 + l31;
        const bool pok = p < a.P;
        float res_r[MI][16];                                // all residual loads of this pixel column in flight together
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co_lane + i * 32 + (r & 3) + 8 * (r >> 2);
                res_r[i][r] = (rb && pok && co < a.Co) ? rb[(long long)co * a.P + p] : 0.0f;
            }
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co_lane + i * 32 + (r & 3) + 8 * (r >> 2);
                if (pok && co < a.Co)
                    epi_store(&yb[(long long)co * a.P + p],
                              ((acc[i][j][r] * out_unscale + bias_r[i][r]) + res_r[i][r]) * a.out_scale);
            }
    }
}

}  // namespace

// 128-pixel blocks (two per CU) where the 256-pixel grid does not fill the chip once (512 -> 1536 @ 4 x 128 at batch 8: 192
// blocks, 48 -> 43 us inside a step); LC_P1_BP=128 / 256 forces one
static bool p1_small_tile(int B, long long P, int Co) {
    static const int env = [] { const char* e = getenv("LC_P1_BP"); return e ? atoi(e) : 0; }();
    if (env == 128) return true;
    if (env == 256) return false;
    const long long blocks256 = (long long)B * ((P + 255) / 256) * ((Co + 127) / 128);
    return blocks256 <= 256;      // (384 blocks -- 256 -> 768 @ 8 x 256 at batch 8 -- measure 42 against 46.5 us inside a step)
}
// ... and for the plain kernel when the launch is output-heavy (Co >= 4 Ci: the nine-plane projections of the up-path fold --
// 16-byte stores of one block under the other block's K loop: 80.4 -> 72.7 us at 256 -> 2304 @ 8 x 8 x 256, 109.6 -> 105.0 at
// 128 -> 1152 @ 8 x 16 x 512, 43.7 -> 39.7 at 64 -> 576)
static bool p1_small_tile_plain(int B, long long P, int Ci, int Co) {
    static const int env = [] { const char* e = getenv("LC_P1_BP"); return e ? atoi(e) : 0; }();
    if (env == 128) return true;
    if (env == 256) return false;
    return Co >= 4 * Ci || p1_small_tile(B, P, Co);
}

// 288-channel blocks when the output channels divide into them (9 Co' of the up-path fold) and the grid is ONE round of the
// chip (256 blocks where the 128-channel forms run 1.125 rounds: 30.5 -> 22.8 us at 256 -> 2304 @ 8 x 4 x 128); on longer grids
// the 128-channel x 128-pixel form (two blocks per CU: one block's epilogue under the other's K loop) is level or ahead:
// 512 -> 4608 @ 8 x 4 x 128 68.9 (wide) / 68.7, 128 -> 1152 @ 8 x 8 x 256 31.1 / 28.9, 256 -> 2304 @ 8 x 8 x 256 84.9 / 72.7
// (profiles/r06_fold_up.txt section 6); LC_P1_BN=128 / 288 forces one
static bool p1_wide_tile(int B, long long P, int Co) {
    static const int env = [] { const char* e = getenv("LC_P1_BN"); return e ? atoi(e) : 0; }();
    if (Co % 288 || env == 128) return false;
    return env == 288 || (long long)B * ((P + 127) / 128) * (Co / 288) <= 256;
}

// LC_P1_ST=0: the dword form of the plain epilogue everywhere (developer A/B)
static int p1_store_mode() {
    static const int env = [] { const char* e = getenv("LC_P1_ST"); return e ? atoi(e) : 2; }();
    return env;
}

// x_split: [B][2][Ci/8][P][8] halves as lc_groupnorm_apply*_split write them (P = H * W); wp_hi / wp_lo / wmeta: the ks = 1
// pack of lc_pack_conv_weight_f16x2 (lo plane directly behind the hi plane); y, res: fp32 [B][Co][P] with batch strides.
extern "C" int lc_conv1x1_f16x2_ps_fwd(const void* x_split, const void* wp_hi, const void* wp_lo, const float* bias,
                                       const float* res, int64_t res_bs, float* y, int64_t y_bs, int B, int Ci,
                                       int Co, int H, int W, float out_scale, const float* wmeta,
                                       lc_conv_range* range, lc_stream_t s) {
    if (!x_split || !wp_hi || !wp_lo || !y || !wmeta || !range || B <= 0 || Ci <= 0 || Co <= 0 || H <= 0 || W <= 0)
        return LC_EINVAL;
    if (Ci % 32) return LC_EUNSUP;
    const long long P = (long long)H * W;
    if (P >= (1 << 24) || 2 * (Ci / 8) * P * 16 >= (1ll << 31) || (long long)Co * P * 4 >= (1ll << 31)) return LC_EUNSUP;
    P1Args a;
    a.xsp = (const half8*)x_split; a.C8 = Ci / 8; a.P = (int)P; a.xsp_bs = 2ll * (Ci / 8) * P;
    a.wh = (const half8*)wp_hi; a.Cib = Ci / 8; a.Cop = (Co + 63) / 64 * 64;
    if ((const half8*)wp_lo != a.wh + (long long)a.Cib * a.Cop) return LC_EINVAL;    // one allocation, lo behind hi
    a.bias = bias; a.res = res; a.res_bs = res_bs; a.y = y; a.y_bs = y_bs;
    a.Co = Co; a.out_scale = out_scale; a.range = range; a.wmeta = wmeta;
    a.kv = nullptr; a.C = a.kv_heads = a.kv_tiles = 0;
    a.st_mode = (p1_store_mode() == 2 && !res && P % 4 == 0 && (y_bs & 3) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0) ? 2 : 0;
    if (p1_wide_tile(B, P, Co)) {       // 288 x 128 blocks (the nine tap planes of the up-path fold: Co = 9 Co')
        using PW = P1T<128, 288, 3>;
        dim3 grid((unsigned)(B * ((P + 127) / 128)), (unsigned)(Co / PW::BN));
        hipLaunchKernelGGL((conv1x1_ps_kernel<false, 128, 288, 3>), grid, dim3(PW::NT), 0, lc_s(s), a);
        return lc_launch_status();
    }
    if (p1_small_tile_plain(B, P, Ci, Co)) {
        dim3 grid((unsigned)(B * ((P + 127) / 128)), (unsigned)((Co + P1::BN - 1) / P1::BN));
        hipLaunchKernelGGL((conv1x1_ps_kernel<false, 128>), grid, dim3(P1T<128>::NT), 0, lc_s(s), a);
        return lc_launch_status();
    }
    dim3 grid((unsigned)(B * ((P + P1::BP - 1) / P1::BP)), (unsigned)((Co + P1::BN - 1) / P1::BN));
    hipLaunchKernelGGL(conv1x1_ps_kernel<false>, grid, dim3(P1::NT), 0, lc_s(s), a);
    return lc_launch_status();
}

// The qkv projection of ObjectAwareCrossAttention (layout_unet_v1.py:416-430: conv_nd(1, C, 3 C, 1) on the normalised
// tokens) with its keys and values written in the attention kernel's unit form: q [B][C][P] fp32 as before; rows C .. 3 C
// never reach memory as fp32.  C % 128 == 0, 32 channels per head (heads = C / 32), P % 32 == 0; kv as
// lc_attention_units_elems(B, C / 32, P, Lk1) describes it (the image keys are tiles 0 .. P / 32 - 1).
extern "C" int lc_conv1x1_f16x2_ps_qkv_fwd(const void* x_split, const void* wp_hi, const void* wp_lo, const float* bias,
                                           float* q, int64_t q_bs, void* kv, int B, int Ci, int C, int H, int W, int Lk1,
                                           const float* wmeta, lc_conv_range* range, lc_stream_t s) {
    if (!x_split || !wp_hi || !wp_lo || !q || !kv || !wmeta || !range || B <= 0 || Ci <= 0 || C <= 0 || H <= 0 || W <= 0)
        return LC_EINVAL;
    const long long P = (long long)H * W;
    if (Ci % 32 || C % 128 || P % 32 || Lk1 < 0 || Lk1 > 32) return LC_EUNSUP;
    const int Co = 3 * C;
    if (P >= (1 << 24) || 2 * (Ci / 8) * P * 16 >= (1ll << 31) || (long long)C * P * 4 >= (1ll << 31)) return LC_EUNSUP;
    P1Args a;
    a.xsp = (const half8*)x_split; a.C8 = Ci / 8; a.P = (int)P; a.xsp_bs = 2ll * (Ci / 8) * P;
    a.wh = (const half8*)wp_hi; a.Cib = Ci / 8; a.Cop = (Co + 63) / 64 * 64;
    if ((const half8*)wp_lo != a.wh + (long long)a.Cib * a.Cop) return LC_EINVAL;
    a.bias = bias; a.res = nullptr; a.res_bs = 0; a.y = q; a.y_bs = q_bs;
    a.Co = Co; a.out_scale = 1.0f; a.range = range; a.wmeta = wmeta;
    a.kv = (half8*)kv; a.C = C; a.kv_heads = C / 32; a.kv_tiles = (int)(P / 32) + (Lk1 > 0);
    a.st_mode = (p1_store_mode() == 2 && (q_bs & 3) == 0 && (reinterpret_cast<uintptr_t>(q) & 15) == 0) ? 2 : 0;   // (P % 32 == 0 here)
    if (p1_small_tile(B, P, Co)) {
        dim3 grid((unsigned)(B * ((P + 127) / 128)), (unsigned)(Co / P1::BN));
        hipLaunchKernelGGL((conv1x1_ps_kernel<true, 128>), grid, dim3(P1T<128>::NT), 0, lc_s(s), a);
        return lc_launch_status();
    }
    dim3 grid((unsigned)(B * ((P + P1::BP - 1) / P1::BP)), (unsigned)(Co / P1::BN));
    hipLaunchKernelGGL(conv1x1_ps_kernel<true>, grid, dim3(P1::NT), 0, lc_s(s), a);
    return lc_launch_status();
}
