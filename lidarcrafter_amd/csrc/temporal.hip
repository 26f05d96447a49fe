// Point-set glue of the autoregressive (temporal) loop, resident on the device:
//   rigid / affine transforms of [N,4] point sets  (get_next_frame_points pipe_related.py:243-249,
//     warp_lidar_future common.py:59-112, object <-> box frame pipe_related.py:56-66,263-268),
//   range image -> point list with the condition mask applied and degenerate points dropped
//     (get_temporal_boxes_3d :68-75, refine_next_frame_points :274-283, remove_ego_points :11-13),
//   order-preserving stream compaction (numpy boolean indexing points[mask]).
// HBM-bound elementwise / scan work: 16-byte point loads and stores, one pass each.
#include "common.h"

namespace {

#pragma clang fp contract(off)

// out.xyz = R p + t with the 3x4 top of a row-major 4x4 (float64 entries, evaluated in fp64 and
// rounded once -- the reference does this product in float64, pipe_related.py:248); out.w = p.w
struct Affine { double m[12]; };

__global__ __launch_bounds__(256) void transform_kernel(const float* __restrict__ pts, int N, Affine T,
                                                       float* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const f32x4 p = *reinterpret_cast<const f32x4*>(pts + 4ll * i);
    const double x = p.x, y = p.y, z = p.z;
    f32x4 o;
    o.x = (float)(((T.m[0] * x + T.m[1] * y) + T.m[2] * z) + T.m[3]);
    o.y = (float)(((T.m[4] * x + T.m[5] * y) + T.m[6] * z) + T.m[7]);
    o.z = (float)(((T.m[8] * x + T.m[9] * y) + T.m[10] * z) + T.m[11]);
    o.w = p.w;
    *reinterpret_cast<f32x4*>(out + 4ll * i) = o;
}

// float64 rows out (round 3): the reference keeps `(Ts @ homo_pts.T).T` as a float64 array and
// projects it in float64 (pipe_related.py:245-257 -> common.py:41-91), so the moved background is
// NOT rounded here.  rot_f32 = 1: out.xyz = (double)(float)(R p) + t -- the re-posed object points
// `rotate_points_along_z(p, yaw)` (a float32 matmul, dataset/utils.py:37-59; its library-defined
// summation order is replaced by the correctly rounded value) `+ np.array([x, y, z])` (float64,
// pipe_related.py:263-266).
struct f64x4 { double x, y, z, w; };
__global__ __launch_bounds__(256) void transform64_kernel(const float* __restrict__ pts, int N, Affine T,
                                                         int rot_f32, double* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const f32x4 p = *reinterpret_cast<const f32x4*>(pts + 4ll * i);
    const double x = p.x, y = p.y, z = p.z;
    f64x4 o;
    if (rot_f32) {
        o.x = (double)(float)((T.m[0] * x + T.m[1] * y) + T.m[2] * z) + T.m[3];
        o.y = (double)(float)((T.m[4] * x + T.m[5] * y) + T.m[6] * z) + T.m[7];
        o.z = (double)(float)((T.m[8] * x + T.m[9] * y) + T.m[10] * z) + T.m[11];
    } else {
        o.x = ((T.m[0] * x + T.m[1] * y) + T.m[2] * z) + T.m[3];
        o.y = ((T.m[4] * x + T.m[5] * y) + T.m[6] * z) + T.m[7];
        o.z = ((T.m[8] * x + T.m[9] * y) + T.m[10] * z) + T.m[11];
    }
    o.w = (double)p.w;
    *reinterpret_cast<f64x4*>(out + 4ll * i) = o;
}

// xyz [3,H,W] planes (+ reflectance plane) -> rows (x, y, z, refl * refl_scale), multiplied by
// the background mask !(cond[h,w] > 0) when `cond` is given; keep[i] = 1 unless
//   |p.xyz| <= min_norm            (min_norm >= 0; the reference drops the zeroed pixels), or
//   |x| < ego_r and |y| < ego_r    (ego_r > 0; remove_ego_points)
__global__ __launch_bounds__(256) void image_points_kernel(
    const float* __restrict__ xyz, long long plane, const float* __restrict__ refl,
    const float* __restrict__ cond, int HW, float refl_scale, float min_norm, float ego_r,
    float* __restrict__ pts, int* __restrict__ keep) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= HW) return;
    const float m = (cond && cond[i] > 0.f) ? 0.f : 1.f;
    f32x4 o;
    o.x = xyz[i] * m; o.y = xyz[plane + i] * m; o.z = xyz[2 * plane + i] * m;
    o.w = refl ? (refl[i] * refl_scale) * m : 0.f;
    *reinterpret_cast<f32x4*>(pts + 4ll * i) = o;
    int k = 1;
    if (min_norm >= 0.f) {
        const float nrm = sqrtf((o.x * o.x + o.y * o.y) + o.z * o.z);
        k = nrm > min_norm;
    }
    if (ego_r > 0.f && fabsf(o.x) < ego_r && fabsf(o.y) < ego_r) k = 0;
    keep[i] = k;
}

// ---- order-preserving compaction of 16-byte rows: count / scan / scatter -------------------------
constexpr int CP_ITEMS = 4, CP_BLOCK = 256 * CP_ITEMS;

__device__ __forceinline__ int cp_pred(const int* __restrict__ keep, int i, int N, int mode) {
    if (i >= N) return 0;
    const int v = keep[i];
    return mode == 0 ? (v != 0) : (v == 0);      // mode 1: keep rows whose flag is ZERO
}

__global__ __launch_bounds__(256) void cp_count_kernel(const int* __restrict__ keep, int N, int mode,
                                                      int* __restrict__ block_counts) {
    const int base = blockIdx.x * CP_BLOCK + threadIdx.x * CP_ITEMS;
    int c = 0;
#pragma unroll
    for (int k = 0; k < CP_ITEMS; ++k) c += cp_pred(keep, base + k, N, mode);
    c = (int)lc_wave_sum((float)c);                // <= 256: exact in fp32
    __shared__ int sh[4];
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) block_counts[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// one block: exclusive scan of the block counts in place, total -> count[0]
__global__ __launch_bounds__(256) void cp_scan_kernel(int* __restrict__ block_counts, int nb,
                                                     int* __restrict__ count) {
    __shared__ int sh[256];
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < nb; base += 256) {
        const int i = base + threadIdx.x;
        const int v = i < nb ? block_counts[i] : 0;
        sh[threadIdx.x] = v;
        __syncthreads();
        for (int o = 1; o < 256; o <<= 1) {        // Hillis-Steele inclusive scan
            const int t = threadIdx.x >= o ? sh[threadIdx.x - o] : 0;
            __syncthreads();
            sh[threadIdx.x] += t;
            __syncthreads();
        }
        const int incl = sh[threadIdx.x];
        const int c0 = carry;
        if (i < nb) block_counts[i] = c0 + incl - v;
        __syncthreads();
        if (threadIdx.x == 255) carry = c0 + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) count[0] = carry;
}

__global__ __launch_bounds__(256) void cp_scatter_kernel(const float* __restrict__ rows,
                                                        const int* __restrict__ keep, int N, int mode,
                                                        const int* __restrict__ block_offsets,
                                                        float* __restrict__ out,
                                                        int* __restrict__ src_index) {
    const int base = blockIdx.x * CP_BLOCK + threadIdx.x * CP_ITEMS;
    int pr[CP_ITEMS], c = 0;
#pragma unroll
    for (int k = 0; k < CP_ITEMS; ++k) { pr[k] = cp_pred(keep, base + k, N, mode); c += pr[k]; }
    __shared__ int sh[256];
    sh[threadIdx.x] = c;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
        const int t = threadIdx.x >= o ? sh[threadIdx.x - o] : 0;
        __syncthreads();
        sh[threadIdx.x] += t;
        __syncthreads();
    }
    int pos = block_offsets[blockIdx.x] + sh[threadIdx.x] - c;
#pragma unroll
    for (int k = 0; k < CP_ITEMS; ++k) {
        if (pr[k]) {
            *reinterpret_cast<f32x4*>(out + 4ll * pos) =
                *reinterpret_cast<const f32x4*>(rows + 4ll * (base + k));
            if (src_index) src_index[pos] = base + k;
            ++pos;
        }
    }
}

}  // namespace

extern "C" int lc_transform_points(const float* pts, int N, const double* T16, float* out,
                                   lc_stream_t s) {
    if (N < 0 || !T16 || (N > 0 && (!pts || !out))) return LC_EINVAL;
    if ((reinterpret_cast<uintptr_t>(pts) | reinterpret_cast<uintptr_t>(out)) & 15) return LC_EINVAL;
    if (N == 0) return LC_OK;
    Affine T;
    for (int i = 0; i < 12; ++i) T.m[i] = T16[i];
    hipLaunchKernelGGL(transform_kernel, dim3((N + 255) / 256), dim3(256), 0, lc_s(s), pts, N, T, out);
    return lc_launch_status();
}

extern "C" int lc_transform_points_f64(const float* pts, int N, const double* T16, int rot_f32,
                                       double* out, lc_stream_t s) {
    if (N < 0 || !T16 || (N > 0 && (!pts || !out))) return LC_EINVAL;
    if ((reinterpret_cast<uintptr_t>(pts) & 15) || (reinterpret_cast<uintptr_t>(out) & 31)) return LC_EINVAL;
    if (N == 0) return LC_OK;
    Affine T;
    for (int i = 0; i < 12; ++i) T.m[i] = T16[i];
    hipLaunchKernelGGL(transform64_kernel, dim3((N + 255) / 256), dim3(256), 0, lc_s(s), pts, N, T,
                       rot_f32 ? 1 : 0, out);
    return lc_launch_status();
}

extern "C" int lc_image_to_points(const float* xyz, int64_t plane_stride, const float* refl,
                                  const float* cond, int H, int W, float refl_scale, float min_norm,
                                  float ego_radius, float* pts, int32_t* keep, lc_stream_t s) {
    if (!xyz || !pts || !keep || H <= 0 || W <= 0) return LC_EINVAL;
    if (reinterpret_cast<uintptr_t>(pts) & 15) return LC_EINVAL;
    const int HW = H * W;
    hipLaunchKernelGGL(image_points_kernel, dim3((HW + 255) / 256), dim3(256), 0, lc_s(s), xyz,
                       (long long)plane_stride, refl, cond, HW, refl_scale, min_norm, ego_radius, pts,
                       keep);
    return lc_launch_status();
}

extern "C" int64_t lc_compact_scratch_elems(int N) { return N <= 0 ? 1 : (N + CP_BLOCK - 1) / CP_BLOCK; }

extern "C" int lc_compact_points(const float* rows, const int32_t* keep, int N, int keep_if_zero,
                                 float* out, int32_t* src_index, int32_t* count, int32_t* scratch,
                                 lc_stream_t s) {
    if (N < 0 || !count || !scratch || (N > 0 && (!rows || !keep || !out))) return LC_EINVAL;
    if ((reinterpret_cast<uintptr_t>(rows) | reinterpret_cast<uintptr_t>(out)) & 15) return LC_EINVAL;
    const int nb = (int)lc_compact_scratch_elems(N);
    const int mode = keep_if_zero ? 1 : 0;
    if (N > 0)
        hipLaunchKernelGGL(cp_count_kernel, dim3(nb), dim3(256), 0, lc_s(s), keep, N, mode, scratch);
    hipLaunchKernelGGL(cp_scan_kernel, dim3(1), dim3(256), 0, lc_s(s), scratch, N > 0 ? nb : 0, count);
    if (N > 0)
        hipLaunchKernelGGL(cp_scatter_kernel, dim3(nb), dim3(256), 0, lc_s(s), rows, keep, N, mode,
                           scratch, out, src_index);
    return lc_launch_status();
}

LC_TOUCH_TU(temporal, transform_kernel)
