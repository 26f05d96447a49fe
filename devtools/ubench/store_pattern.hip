// Store-pattern microbenchmark (round 3): how fast does the chip take the conv epilogue's writes?
//   hipcc --offload-arch=gfx950 -O3 -o devtools/ubench/store_pattern devtools/ubench/store_pattern.hip
// Output tensor [B=8][C=64][H=32][W=1024] fp32 (67 MB), written once per launch by 256 blocks x 512 threads,
// each block owning 4 tiles of 4 rows x 64 columns x 64 channels like conv cfg 23:
//   mode 0: the epilogue's pattern -- 32 dword stores per thread (lane -> column, lane half -> channel + 4), write-back
//   mode 1: the same with write-through (sc1)
//   mode 2: float4 per lane (16 lanes cover a 64-column row segment; 8 stores per thread), write-back
//   mode 3: float4 + sc1
//   mode 4: fully streaming float4 (each block a contiguous 256 KB), write-back
//   mode 5: streaming float4 + sc1
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int B = 8, C = 64, H = 32, W = 1024, HW = H * W;

template <int SC1> __device__ __forceinline__ void st1(float* p, float v) {
    if (SC1) asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dword %0, %1, off" ::"v"(p), "v"(v) : "memory");
}
template <int SC1> __device__ __forceinline__ void st4(float* p, f4 v) {
    if (SC1) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory");
}

template <int MODE> __global__ __launch_bounds__(512) void k(float* y, float val) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int bx = blockIdx.x;
    bx = (bx & 7) * (gridDim.x >> 3) + (bx >> 3);          // XCD-contiguous like the conv
    const int tw = bx % 16, th0 = (bx / 16 % 2) * 4, b = bx / 32;   // 16 W tiles, 2 groups of 4 tiles down H, 8 samples
    float* yb = y + (long long)b * C * HW;
    if (MODE >= 4) {
        float* p = y + (long long)blockIdx.x * 65536 + tid * 4;       // 256 KB per block
        for (int i = 0; i < 32; ++i) st4<MODE & 1>(p + i * 2048, f4{val, val, val, val});
        return;
    }
    const int wco = wave >> 2, wpx = wave & 3;             // 2 channel waves x 4 pixel waves
    for (int t = 0; t < 4; ++t) {
        const int h0 = (th0 + t) * 4, w0 = tw * 64;
        if (MODE < 2) {
            const int kh = lane >> 5, l31 = lane & 31;
            for (int j = 0; j < 2; ++j) {
                const int tt = wpx * 2 + j, tr = tt / 2, tc = tt % 2;
                const int gh = h0 + tr, gw = w0 + tc * 32 + l31;
                for (int r = 0; r < 16; ++r) {
                    const int co = wco * 32 + 4 * kh + (r & 3) + 8 * (r >> 2);
                    st1<MODE & 1>(yb + (long long)co * HW + gh * W + gw, val + r);
                }
            }
        } else {
            // wave: 32 channels x 1 row (wpx) x 64 columns as float4: lane -> (channel sub-index lane / 16, 4 columns)
            const int c4 = lane >> 4, col = (lane & 15) * 4;
            for (int r = 0; r < 8; ++r) {
                const int co = wco * 32 + r * 4 + c4;
                st4<MODE & 1>(yb + (long long)co * HW + (h0 + wpx) * W + w0 + col, f4{val, val, val, val});
            }
        }
    }
}

int main() {
    float* y; hipMalloc(&y, (size_t)B * C * HW * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](int mode) {
        float best = 1e9;
        for (int rep = 0; rep < 6; ++rep) {
            hipEventRecord(e0);
            for (int i = 0; i < 10; ++i) {
                switch (mode) {
                    case 0: hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 0, 0, y, 1.f); break;
                    case 1: hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 0, 0, y, 1.f); break;
                    case 2: hipLaunchKernelGGL(k<2>, dim3(256), dim3(512), 0, 0, y, 1.f); break;
                    case 3: hipLaunchKernelGGL(k<3>, dim3(256), dim3(512), 0, 0, y, 1.f); break;
                    case 4: hipLaunchKernelGGL(k<4>, dim3(256), dim3(512), 0, 0, y, 1.f); break;
                    case 5: hipLaunchKernelGGL(k<5>, dim3(256), dim3(512), 0, 0, y, 1.f); break;
                }
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        const double us = best * 100.0, gb = (double)B * C * HW * 4 / 1e9;
        printf("mode %d: %7.1f us per 67 MB  -> %6.2f TB/s\n", mode, us, gb / us * 1e3);
    };
    for (int m = 0; m < 6; ++m) run(m);
    return 0;
}
