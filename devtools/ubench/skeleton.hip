// Developer micro-benchmark: the K-chunk skeleton of conv_f16x2_pipe_kernel -- 54 MFMAs per wave per
// chunk on 2 accumulators (3 per fragment pair), optional barrier per chunk, optional one-tap-ahead
// ds_read_b128 fragment fetches (2 A + 4 B per tap) -- to separate barrier / LDS-read cost from
// matrix-pipe time.  hipcc --offload-arch=gfx950 -O3 skeleton.hip -o skeleton && ./skeleton
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int BAR, int LDSR, int SB>
__global__ __launch_bounds__(512) void k(float* out, int chunks) {
    __shared__ half8 lds[4096];   // 64 KB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 4096; i += blockDim.x) {
        half8 v;
        for (int j = 0; j < 8; ++j) v[j] = (_Float16)(0.001f * ((i + j) & 63));
        lds[i] = v;
    }
    __syncthreads();
    f32x16 acc0, acc1;
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    half8 ah[2], al[2], bh[2][2], bl[2][2];
    const int abase = (wave >> 2) * 64 + lane, bbase = 1024 + (wave & 3) * 128 + lane;
    auto fetch = [&](int tap, int s) {
        if (LDSR) {
            ah[s] = lds[abase + tap * 128]; al[s] = lds[abase + tap * 128 + 2048];
            bh[s][0] = lds[bbase + tap * 66]; bh[s][1] = lds[bbase + tap * 66 + 64];
            bl[s][0] = lds[bbase + tap * 66 + 2048]; bl[s][1] = lds[bbase + tap * 66 + 2048 + 64];
        }
    };
    if (!LDSR) {
        for (int s = 0; s < 2; ++s) {
            ah[s] = lds[abase]; al[s] = lds[abase + 1]; bh[s][0] = lds[bbase]; bh[s][1] = lds[bbase + 1];
            bl[s][0] = lds[bbase + 2]; bl[s][1] = lds[bbase + 3];
        }
    }
    for (int ch = 0; ch < chunks; ++ch) {
        fetch(0, 0);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int s = tap & 1;
            if (tap + 1 < 9) fetch(tap + 1, s ^ 1);
            if (SB) __builtin_amdgcn_sched_barrier(0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[s], bh[s][0], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[s], bh[s][1], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], bl[s][0], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], bl[s][1], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], bh[s][0], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], bh[s][1], acc1, 0, 0, 0);
            if (SB) __builtin_amdgcn_sched_barrier(0);
        }
        if (BAR) __syncthreads();
    }
    float s = 0;
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
    out[blockIdx.x * blockDim.x + tid] = s;
}

template <int BAR, int LDSR, int SB> void run(int threads, const char* name) {
    float* out; hipMalloc(&out, 256 * 512 * 4);
    const int chunks = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<BAR, LDSR, SB><<<256, threads>>>(out, 10); hipDeviceSynchronize();
    hipEventRecord(e0);
    k<BAR, LDSR, SB><<<256, threads>>>(out, chunks);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mf = 54.0 * chunks * (threads / 64) * 256;
    printf("%-34s threads=%d: %.0f TF  %.3f us/chunk  (ideal 32 clk/MFMA @2.4GHz: %.3f)\n", name, threads,
           mf * 32768 / ms / 1e9, ms * 1e3 / chunks, 54.0 * (threads / 256) * 32 / 2400.0);
    hipFree(out);
}
int main() {
    run<0, 0, 0>(512, "mfma only");
    run<1, 0, 0>(512, "mfma + barrier/chunk");
    run<0, 1, 0>(512, "mfma + ds_reads");
    run<1, 1, 0>(512, "mfma + ds_reads + barrier");
    run<1, 1, 1>(512, "mfma + ds_reads + barrier + sb");
    run<0, 0, 0>(256, "mfma only");
    run<1, 1, 1>(256, "mfma + ds_reads + barrier + sb");
    return 0;
}
