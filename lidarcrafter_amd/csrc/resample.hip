// FIR x2 resampling, window [1,3,3,1], ring in W / zeros in H -- ops.Resample ops.py:52-146.
// The reference builds it from pad + zero-insertion view + two depthwise convs + strided slice
// (~6 memory passes); closed forms, horizontal pass first like the reference:
//   down: y[i,j] = sum_a k[a] * ( sum_b k[b] x[2i+a-1, 2j+b-1] ),  k = [1,3,3,1]/8
//   up:   per axis y[2i] = .25 x[i-1] + .75 x[i],  y[2i+1] = .75 x[i] + .25 x[i+1]
// HBM-bound: one read of x (neighbours hit L1/L2), one write of y.
#include "common.h"

namespace {

#pragma clang fp contract(off)

__global__ __launch_bounds__(256) void down2_kernel(const float* __restrict__ x, long long x_bs,
                                                   float* __restrict__ y, long long y_bs, int C,
                                                   int H, int W) {
    const int Ho = H / 2, Wo = W / 2;
    const long long total = (long long)C * Ho * Wo;
    const int b = blockIdx.y;
    const float* xb = x + b * x_bs;
    float* yb = y + b * y_bs;
    for (long long e = blockIdx.x * 256ll + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int j = e % Wo;
        const long long r = e / Wo;
        const int i = r % Ho;
        const int c = r / Ho;
        const float* xc = xb + (long long)c * H * W;
        int cols[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            int w = 2 * j + q - 1;
            cols[q] = w < 0 ? w + W : (w >= W ? w - W : w);
        }
        float acc = 0.f;
        const float k[4] = {0.125f, 0.375f, 0.375f, 0.125f};
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int h = 2 * i + a - 1;
            float hz = 0.f;
            if (h >= 0 && h < H) {
                const float* row = xc + (long long)h * W;
                hz = ((k[0] * row[cols[0]] + k[1] * row[cols[1]]) + k[2] * row[cols[2]]) + k[3] * row[cols[3]];
            }
            acc += k[a] * hz;
        }
        yb[e] = acc;
    }
}

// one thread per INPUT pixel -> 2x2 outputs
__global__ __launch_bounds__(256) void up2_kernel(const float* __restrict__ x, long long x_bs,
                                                 float* __restrict__ y, long long y_bs, int C,
                                                 int H, int W) {
    const long long total = (long long)C * H * W;
    const int b = blockIdx.y;
    const float* xb = x + b * x_bs;
    float* yb = y + b * y_bs;
    const int W2 = 2 * W;
    for (long long e = blockIdx.x * 256ll + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int j = e % W;
        const long long r = e / W;
        const int i = r % H;
        const int c = r / H;
        const float* xc = xb + (long long)c * H * W;
        const int jm = j == 0 ? W - 1 : j - 1, jp = j == W - 1 ? 0 : j + 1;
        float ev[3], od[3];  // horizontal results for rows i-1, i, i+1
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const int h = i + d - 1;
            if (h >= 0 && h < H) {
                const float* row = xc + (long long)h * W;
                const float xm = row[jm], x0 = row[j], xp = row[jp];
                ev[d] = 0.25f * xm + 0.75f * x0;
                od[d] = 0.75f * x0 + 0.25f * xp;
            } else {
                ev[d] = 0.f; od[d] = 0.f;
            }
        }
        float* yc = yb + (long long)c * 4 * H * W + (long long)(2 * i) * W2 + 2 * j;
        float2 top, bot;
        top.x = 0.25f * ev[0] + 0.75f * ev[1];
        top.y = 0.25f * od[0] + 0.75f * od[1];
        bot.x = 0.75f * ev[1] + 0.25f * ev[2];
        bot.y = 0.75f * od[1] + 0.25f * od[2];
        *reinterpret_cast<float2*>(yc) = top;
        *reinterpret_cast<float2*>(yc + W2) = bot;
    }
}

}  // namespace

extern "C" int lc_resample2x_fwd(const float* x, int64_t x_bs, float* y, int64_t y_bs, int B, int C,
                                 int H, int W, int dir, lc_stream_t s) {
    if (!x || !y || B <= 0 || C <= 0 || H <= 0 || W <= 0) return LC_EINVAL;
    if (dir < 0) {
        if ((H & 1) || (W & 1)) return LC_EUNSUP;
        const long long total = (long long)C * (H / 2) * (W / 2);
        int blocks = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
        hipLaunchKernelGGL(down2_kernel, dim3(blocks, B), dim3(256), 0, lc_s(s), x, (long long)x_bs, y,
                           (long long)y_bs, C, H, W);
    } else {
        if (((y_bs | (int64_t)(2 * W)) & 1) || (reinterpret_cast<uintptr_t>(y) & 7)) return LC_EUNSUP;
        const long long total = (long long)C * H * W;
        int blocks = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
        hipLaunchKernelGGL(up2_kernel, dim3(blocks, B), dim3(256), 0, lc_s(s), x, (long long)x_bs, y,
                           (long long)y_bs, C, H, W);
    }
    return lc_launch_status();
}
