// The fp16 hi / lo split of the attention operands (attn_h_kernel of attention.hip, the unit-form kernels of
// attention_units.hip): constants of the inference kernels and the 8-element split.
#pragma once

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

constexpr float QK_PRE = 16.0f;      // q (after the softmax scale) and k are pre-scaled by 16
constexpr float P_PRE = 2048.0f;     // p in [0,1]
constexpr float P_LOG2 = 11.0f;      // log2(P_PRE)
constexpr float V_PRE = 16.0f;

__device__ __forceinline__ void split8(const float (&v)[8], float scale, half8& hi, half8& lo) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int k = 0; k < 8; k += 2) {
        const float s0 = scale == 1.0f ? v[k] : v[k] * scale;
        const float s1 = scale == 1.0f ? v[k + 1] : v[k + 1] * scale;
        const float h0 = __uint_as_float(__float_as_uint(s0) & 0xFFFFE000u);
        const float h1 = __uint_as_float(__float_as_uint(s1) & 0xFFFFE000u);
        const h2 ph = __builtin_bit_cast(h2, __builtin_amdgcn_cvt_pkrtz(h0, h1));
        f2 r; r.x = s0 - h0; r.y = s1 - h1;
        const h2 pl = __builtin_convertvector(r, h2);
        hi[k] = ph.x; hi[k + 1] = ph.y; lo[k] = pl.x; lo[k + 1] = pl.y;
    }
}
