// Developer micro-benchmark: does the sustained 32x32x16 f16 MFMA rate depend on (a) how many
// distinct A/B operand registers the stream cycles through, (b) the operand DATA (DVFS)?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NOPS, int PAT>
__global__ __launch_bounds__(512) void k(float* out, const float* in, int iters) {
    f32x16 acc0, acc1;
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    half8 a[NOPS], b[NOPS];
    for (int n = 0; n < NOPS; ++n)
        for (int i = 0; i < 8; ++i) {
            a[n][i] = (_Float16)in[(threadIdx.x * 8 + i + n * 17) & 4095];
            b[n][i] = (_Float16)in[(threadIdx.x * 8 + i + n * 29 + 7) & 4095];
        }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 24; ++rep) {
            const int ia = PAT == 0 ? rep % NOPS : (rep / 3) % NOPS;
            const int ib = PAT == 0 ? (rep / 2) % NOPS : (rep * 5 / 3) % NOPS;
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ia], b[ib], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ia], b[(ib + 1) % NOPS], acc1, 0, 0, 0);
        }
    }
    float s = 0;
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NOPS, int PAT> void run(int threads, const float* in, const char* data) {
    float* out; hipMalloc(&out, 256 * 512 * 4);
    const int iters = 1000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NOPS, PAT><<<256, threads>>>(out, in, 10); hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NOPS, PAT><<<256, threads>>>(out, in, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mf = 48.0 * iters * (threads / 64) * 256;
    printf("NOPS=%d PAT=%d threads=%d data=%s: %.0f TF\n", NOPS, PAT, threads, data, mf * 32768 / ms / 1e9);
    hipFree(out);
}
int main() {
    float h[4096]; float *dz, *dr, *ds;
    hipMalloc(&dz, 4096 * 4); hipMalloc(&dr, 4096 * 4); hipMalloc(&ds, 4096 * 4);
    hipMemset(dz, 0, 4096 * 4);
    unsigned s = 12345;
    for (int i = 0; i < 4096; ++i) { s = s * 1664525u + 1013904223u; h[i] = ((s >> 8) & 0xFFFF) / 65536.0f * 2.f - 1.f; }
    hipMemcpy(dr, h, 4096 * 4, hipMemcpyHostToDevice);
    for (int i = 0; i < 4096; ++i) h[i] = 0.001f * (i & 63);
    hipMemcpy(ds, h, 4096 * 4, hipMemcpyHostToDevice);
    for (int t = 256; t <= 512; t += 256) {
        run<1, 0>(t, dz, "zero"); run<1, 0>(t, ds, "small"); run<1, 0>(t, dr, "rand");
        run<4, 0>(t, dz, "zero"); run<4, 0>(t, ds, "small"); run<4, 0>(t, dr, "rand");
        run<4, 1>(t, dr, "rand");
        run<8, 0>(t, dr, "rand");
    }
    return 0;
}
