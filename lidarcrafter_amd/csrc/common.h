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

__device__ __forceinline__ float lc_silu(float v) { return v / (1.0f + __expf(-v)); }

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
