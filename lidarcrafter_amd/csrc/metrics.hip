// Weight-free BEV metrics front-end (SURVEY.md section 8f-3 ii): lidargen/metrics/bev.py.
//   point_cloud_to_histogram :5-24  -- torch.histogramdd of (x, y) on the CPU in the reference;
//     here: one pass over the points, integer atomics into the bins x bins grid (exact counts).
//   cdist_rbf / compute_mmd_2d :27-34,47-55 -- sum_ij exp(-gamma |p_i - q_j|^2) over [M, D] rows,
//     tiled pairwise differences (no |p|^2+|q|^2-2pq cancellation), fp64 block partials.
// HBM / atomics bound.
#include "common.h"

namespace {

#pragma clang fp contract(off)

__global__ void hist_clear_kernel(int* h, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) h[i] = 0;
}

// bin such that e[bin] <= v < e[bin+1], last bin right-inclusive (ATen HistogramKernel.cpp,
// LINEAR_INTERPOLATION_WITH_LOCAL_SEARCH); -1 outside [e[0], e[bins]]
__device__ __forceinline__ int bin_of(float v, const float* __restrict__ e, int bins) {
    const float lo = e[0], hi = e[bins];
    if (!(v >= lo) || !(v <= hi)) return -1;
    int g = (int)((v - lo) * (float)bins / (hi - lo));
    g = g < 0 ? 0 : (g > bins - 1 ? bins - 1 : g);
    while (g > 0 && v < e[g]) --g;
    while (g < bins - 1 && v >= e[g + 1]) ++g;
    return g;
}

__global__ __launch_bounds__(256) void bev_hist_kernel(const float* __restrict__ pts, int stride, int N,
                                                      const float* __restrict__ edges, int bins,
                                                      float min_d, float max_d, int* __restrict__ h) {
    extern __shared__ float se[];
    for (int i = threadIdx.x; i <= bins; i += 256) se[i] = edges[i];
    __syncthreads();
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const float x = pts[(long long)i * stride], y = pts[(long long)i * stride + 1],
                z = pts[(long long)i * stride + 2];
    const float d = sqrtf((x * x + y * y) + z * z);
    if (!(d > min_d && d < max_d)) return;
    const int bx = bin_of(x, se, bins), by = bin_of(y, se, bins);
    if (bx < 0 || by < 0) return;
    atomicAdd(h + bx * bins + by, 1);
}

__global__ void hist_to_float_kernel(const int* __restrict__ h, float* __restrict__ o, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) o[i] = (float)h[i];
}

// 16 x 16 pairs per block, D streamed through LDS in slabs of 64
constexpr int PT = 16, PD = 64;

__global__ __launch_bounds__(256) void rbf_sum_kernel(const float* __restrict__ p, const float* __restrict__ q,
                                                     int M, int Mq, int D, float gamma,
                                                     double* __restrict__ partial) {
    __shared__ float sp[PT][PD + 1], sq[PT][PD + 1];
    __shared__ double red[4];
    const int ti = threadIdx.x >> 4, tj = threadIdx.x & 15;
    const int i = blockIdx.y * PT + ti, j = blockIdx.x * PT + tj;
    float acc = 0.f;
    for (int d0 = 0; d0 < D; d0 += PD) {
        for (int e = threadIdx.x; e < PT * PD; e += 256) {
            const int r = e / PD, c = e - r * PD;
            const int gi = blockIdx.y * PT + r, gj = blockIdx.x * PT + r;
            sp[r][c] = (gi < M && d0 + c < D) ? p[(long long)gi * D + d0 + c] : 0.f;
            sq[r][c] = (gj < Mq && d0 + c < D) ? q[(long long)gj * D + d0 + c] : 0.f;
        }
        __syncthreads();
#pragma unroll 8
        for (int c = 0; c < PD; ++c) {
            const float df = sp[ti][c] - sq[tj][c];
            acc += df * df;
        }
        __syncthreads();
    }
    double v = (i < M && j < Mq) ? (double)expf(-gamma * acc) : 0.0;
    v = lc_wave_sum(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0)
        partial[blockIdx.y * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

}  // namespace

extern "C" int lc_bev_histogram(const float* pts, int pt_stride, int N, const float* edges, int bins,
                                float min_depth, float max_depth, float* hist, int32_t* scratch,
                                lc_stream_t s) {
    if (N < 0 || (N > 0 && !pts) || !edges || !hist || !scratch || bins <= 0 || bins > 4096 ||
        pt_stride < 3)
        return LC_EINVAL;
    const int nb = bins * bins;
    hipLaunchKernelGGL(hist_clear_kernel, dim3((nb + 255) / 256), dim3(256), 0, lc_s(s), scratch, nb);
    if (N > 0)
        hipLaunchKernelGGL(bev_hist_kernel, dim3((N + 255) / 256), dim3(256),
                           (bins + 1) * sizeof(float), lc_s(s), pts, pt_stride, N, edges, bins,
                           min_depth, max_depth, scratch);
    hipLaunchKernelGGL(hist_to_float_kernel, dim3((nb + 255) / 256), dim3(256), 0, lc_s(s), scratch,
                       hist, nb);
    return lc_launch_status();
}

extern "C" int64_t lc_rbf_partials_elems(int M, int Mq) {
    if (M <= 0 || Mq <= 0) return 0;
    return (int64_t)((M + PT - 1) / PT) * ((Mq + PT - 1) / PT);
}

extern "C" int lc_rbf_kernel_sum(const float* p, const float* q, int M, int Mq, int D, float gamma,
                                 double* partials, lc_stream_t s) {
    if (!p || !q || !partials || M <= 0 || Mq <= 0 || D <= 0) return LC_EINVAL;
    dim3 grid((Mq + PT - 1) / PT, (M + PT - 1) / PT);
    if (grid.y > 65535) return LC_EUNSUP;
    hipLaunchKernelGGL(rbf_sum_kernel, grid, dim3(256), 0, lc_s(s), p, q, M, Mq, D, gamma, partials);
    return lc_launch_status();
}

// ---------------------------------------------------------------------------------------------
// Chamfer distance, forward: lidargen/metrics/modules/chamfer3D/chamfer3D.cu:12-120
// (NmDistanceKernel, launched once per direction): for every point of xyz1 the squared distance to
// and the index of its nearest point of xyz2 (first minimum wins).  One thread per query point,
// targets streamed through LDS in tiles of 512; d = (dx*dx + dy*dy) + dz*dz with separate
// roundings (no contraction) so that a float32 CPU restatement reproduces it bit for bit.
namespace {

#pragma clang fp contract(off)

constexpr int NN_TILE = 512;

__global__ __launch_bounds__(256) void nn_kernel(const float* __restrict__ q, int n,
                                                const float* __restrict__ t, int m,
                                                float* __restrict__ dist, int* __restrict__ idx) {
    __shared__ float buf[NN_TILE * 3];
    const int b = blockIdx.y;
    const int j = blockIdx.x * 256 + threadIdx.x;
    const float* qb = q + (long long)b * n * 3;
    const float* tb = t + (long long)b * m * 3;
    float x1 = 0.f, y1 = 0.f, z1 = 0.f;
    if (j < n) { x1 = qb[3 * j]; y1 = qb[3 * j + 1]; z1 = qb[3 * j + 2]; }
    float best = 0.f;
    int best_i = 0;
    for (int k2 = 0; k2 < m; k2 += NN_TILE) {
        const int cnt = (m - k2 < NN_TILE ? m - k2 : NN_TILE);
        __syncthreads();
        for (int e = threadIdx.x; e < cnt * 3; e += 256) buf[e] = tb[(long long)k2 * 3 + e];
        __syncthreads();
        for (int k = 0; k < cnt; ++k) {
            const float dx = buf[3 * k] - x1, dy = buf[3 * k + 1] - y1, dz = buf[3 * k + 2] - z1;
            const float d = (dx * dx + dy * dy) + dz * dz;
            if ((k2 == 0 && k == 0) || d < best) { best = d; best_i = k2 + k; }
        }
    }
    if (j < n) { dist[(long long)b * n + j] = best; idx[(long long)b * n + j] = best_i; }
}

}  // namespace

extern "C" int lc_chamfer3d_fwd(const float* xyz1, const float* xyz2, int B, int N, int M, float* dist1,
                                int32_t* idx1, float* dist2, int32_t* idx2, lc_stream_t s) {
    if (!xyz1 || !xyz2 || !dist1 || !idx1 || !dist2 || !idx2 || B <= 0 || N <= 0 || M <= 0)
        return LC_EINVAL;
    hipLaunchKernelGGL(nn_kernel, dim3((N + 255) / 256, B), dim3(256), 0, lc_s(s), xyz1, N, xyz2, M,
                       dist1, idx1);
    hipLaunchKernelGGL(nn_kernel, dim3((M + 255) / 256, B), dim3(256), 0, lc_s(s), xyz2, M, xyz1, N,
                       dist2, idx2);
    return lc_launch_status();
}

LC_TOUCH_TU(metrics, hist_clear_kernel)
