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
