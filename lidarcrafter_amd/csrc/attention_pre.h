// Range-safe pre-scales of the TRAINING attention's f16x2 operands (round 6).
// q, k, v are multiplied by a power of two before their fp16 hi / lo split.  Rounds 1-5 used the constant 16: exact for
// activations of a normalised network, but |x| >= 4094 saturates the hi half silently (cvt_pkrtz saturates, it does not
// produce inf) and |x| << 1 loses the lo half.  The training entry points now measure max |q|, max |k|, max |v| on the
// device (attn_amax3_kernel: no host synchronisation -- a backward pass cannot be repeated after a host-side poll) and
// every kernel derives its pre-scales from those three words with the rule below: 16 while max |x| * 16 lies in
// [2^2, 2^15) -- results of typical networks stay bit-identical -- otherwise the power of two that puts max |x| * pre into
// [2^11, 2^12).  The scales cancel exactly in the epilogues (powers of two).
#pragma once

__device__ __forceinline__ float attn_pre_from(float amax) {
    if (!(amax > 0.0f) || !(amax < 3.0e38f)) return 16.0f;
    const float s = amax * 16.0f;
    if (s >= 4.0f && s < 32768.0f) return 16.0f;
    int e;
    frexpf(amax, &e);                       // amax = m * 2^e, m in [0.5, 1)
    int kx = 12 - e;
    kx = kx < -100 ? -100 : (kx > 100 ? 100 : kx);
    return ldexpf(1.0f, kx);
}

// amax[0 .. 2] = bit patterns of max |q|, max |k|, max |v| (zeroed by the caller); n* = element counts (multiples of 4 are
// read as float4, the tail as scalars)
__global__ __launch_bounds__(256) void attn_amax3_kernel(const float* __restrict__ q, long long nq, const float* __restrict__ k,
                                                        long long nk, const float* __restrict__ v, long long nv,
                                                        unsigned* __restrict__ amax) {
    const int which = blockIdx.y;
    const float* p = which == 0 ? q : (which == 1 ? k : v);
    const long long n = which == 0 ? nq : (which == 1 ? nk : nv);
    float am = 0.f;
    const long long n4 = ((reinterpret_cast<uintptr_t>(p) & 15) == 0) ? n / 4 : 0;
    const f32x4* p4 = reinterpret_cast<const f32x4*>(p);
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const f32x4 w = p4[i];
        am = fmaxf(am, fmaxf(fmaxf(fabsf(w.x), fabsf(w.y)), fmaxf(fabsf(w.z), fabsf(w.w))));
    }
    for (long long i = n4 * 4 + blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) am = fmaxf(am, fabsf(p[i]));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) am = fmaxf(am, __shfl_xor(am, off, 64));
    if ((threadIdx.x & 63) == 0 && am > 0.f) atomicMax(amax + which, __float_as_uint(am));
}
