// Shared helpers for the gfx950 kernels.  Wave = 64 lanes, hard-coded.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/lidarcrafter_hip.h"

#define LC_WAVE 64

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

static inline hipStream_t lc_s(lc_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

static inline int lc_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? LC_OK : (int)e;
}

// Output stores of the streaming kernels (resampling, attention output, fp32 GroupNorm apply): write-THROUGH (sc1), like the
// convolution epilogues (conv_f16x2_common.h epi_store / LC_DEF_AUX): lines left dirty in the per-XCD L2s are written back at
// the kernel boundary, in front of the next launch; written through, they stream out while the kernel runs.  Clean same-box
// A/B of these conversions together (r05z20): C2 262.2 -> 262.7 steps/s, C3 8.90 -> 8.88 ms -- marginal.  The pre-split
// GroupNorm apply pass is NOT written through (norm.hip LC_GNS_STORE: the pass alone 20.3 -> 29.4 us, the step 3-4 % slower).
// A first reading of that experiment said "+5 % on the step": the library that measured it wrote its 128-bit planes with
// inline-assembly stores that lacked the ISA's wait states, the planes were partly garbage (55 of 136 pre-split parity tests
// failed once they were run against it), and the power-limited convolutions run faster on degenerate operands
// (profiles/r05_level0.txt section 8).  LC_ST_WT=0: plain stores (developer A/B).  Only for data the storing kernel does
// not read back.
#ifndef LC_ST_WT
#define LC_ST_WT 1
#endif
__device__ __forceinline__ void lc_st(float* p, float v) {
#if LC_ST_WT
    asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
#else
    *p = v;
#endif
}
__device__ __forceinline__ void lc_st2(float* p, float2 v) {
#if LC_ST_WT
    asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
#else
    *reinterpret_cast<float2*>(p) = v;
#endif
}
// 128-bit form: a raw buffer store the compiler knows -- it keeps the ISA's wait states between a > 64-bit store and a write
// of its data registers; behind INLINE assembly it cannot (gn_apply_kernel's next instruction, a v_max into the first data
// register, corrupted the stored quad: test_groupnorm[affine-2-64-32-1024-8] 0.25 off).  `rs`: lc_wt_buf(base) of a
// wave-uniform base, byte offsets < 4 GiB.
typedef unsigned lc_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t lc_wt_buf(void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(base, 0, 0xFFFFFFFFu, 0x00020000);
}
__device__ __forceinline__ void lc_st4(__amdgpu_buffer_rsrc_t rs, unsigned byte_off, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(lc_u32x4, v), rs, byte_off, 0, LC_ST_WT ? 16 : 0);
}

// x * sigmoid(x) with the hardware reciprocal (1 ulp), as the convolutions' fused input norm computes it
// (conv_f16x2_common.h gn_act).  Round 5: the IEEE division here was ~10 of the ~24 VALU instructions per element of the
// GroupNorm apply passes, which are VALU-bound, not HBM-bound (profiles/r05_level0.txt section 7).
__device__ __forceinline__ float lc_silu(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

__device__ __forceinline__ double lc_wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float lc_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// max |.| of a block -> *slot (a float seen as its bit pattern; non-negative floats order like unsigned).  One atomicMax
// per BLOCK, and only when the block's maximum beats what is already published (thousands of same-address atomics
// serialise in L2: 150 us per call in the first version).  NaN / 0 never win.  Every thread of the block must call it.
__device__ __forceinline__ void lc_block_amax_publish(float am, unsigned* slot) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) am = fmaxf(am, __shfl_xor(am, o, 64));
    __shared__ float wmax[16];
    const int nw = (blockDim.x + 63) >> 6;
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = am;
    __syncthreads();
    if (threadIdx.x == 0) {
        float m = wmax[0];
        for (int i = 1; i < nw; ++i) m = fmaxf(m, wmax[i]);
        const unsigned bits = __float_as_uint(m);
        if (m > 0.0f && bits > __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            atomicMax(slot, bits);
    }
}

// max |.| of a block -> *out as a plain store (no atomics, nothing to wait for: a pass with thousands of short blocks
// leaves one partial maximum per block and the consumer reduces them).  Every thread of the block must call it.
__device__ __forceinline__ void lc_block_amax_store(float am, float* out) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) am = fmaxf(am, __shfl_xor(am, o, 64));
    __shared__ float wmax_s[16];
    const int nw = (blockDim.x + 63) >> 6;
    if ((threadIdx.x & 63) == 0) wmax_s[threadIdx.x >> 6] = am;
    __syncthreads();
    if (threadIdx.x == 0) {
        float m = wmax_s[0];
        for (int i = 1; i < nw; ++i) m = fmaxf(m, wmax_s[i]);
        *out = m;
    }
}

// Code-object warm-up (misc.hip: lc_load_code_objects): every translation unit names one of its kernels; asking the
// runtime for that kernel's attributes makes it load the unit's code object for the current device now instead of
// at the unit's first launch.
#define LC_TOUCH_TU(name, ...)                                                                        \
    extern "C" __attribute__((visibility("hidden"))) int lc_touch_##name() {                          \
        hipFuncAttributes fa;                                                                         \
        return (int)hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&__VA_ARGS__));           \
    }
