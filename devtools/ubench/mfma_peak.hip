// Developer micro-benchmark: sustained v_mfma_f32_32x32x16_f16 rate vs number of independent
// accumulators per wave and waves per SIMD (to interpret the conv kernel's MFMA utilisation).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ void k(float* out, int iters) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    half8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(0.5f + i * 0.01f); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 8; ++rep)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC> void run(int threads, int blocks_per_cu) {
    float* out; hipMalloc(&out, 256 * 8 * 1024 * 4);
    const int iters = 2000, grid = 256 * blocks_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC><<<grid, threads>>>(out, 10); hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NACC><<<grid, threads>>>(out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = 2.0 * 32 * 32 * 16 * 8.0 * NACC * iters * (threads / 64) * grid;
    printf("NACC=%d threads=%d blocks/CU=%d: %.1f TF  (%.1f us)\n", NACC, threads, blocks_per_cu, flops / ms / 1e9, ms * 1e3);
    hipFree(out);
}
int main() {
    run<1>(256, 1); run<2>(256, 1); run<4>(256, 1); run<8>(256, 1);
    run<1>(512, 1); run<2>(512, 1); run<4>(512, 1);
    run<2>(256, 2); run<4>(256, 2);
    return 0;
}
