// Voxel scatter of the weight-free metrics front-end (SURVEY.md section 8f-3 ii):
//   pcd2bev_sum      lidargen/metrics/metric_utils.py:233-258  per sweep: mask to the x/y range,
//                    floor(xy / voxel) , UNIQUE voxels (sparse_quantize), grid[voxel] += 1.
//     Here: one scatter pass per sweep with two atomics -- atomicExch of the sweep's stamp on the
//     voxel decides who is first (exactly one thread per (sweep, voxel) sees a foreign stamp),
//     that thread atomicAdds 1 into the float grid.  No sort, no hash table.  HBM / atomics bound.
//   sparse_quantize  metric_utils.py:28-66 (ravel_hash + np.unique): floor(coords / voxel) -> int32,
//     unique voxels in hash (= lexicographic) order, index of the first occurrence, inverse map.
//     Here: u64 hash exactly as ravel_hash builds it, rocPRIM stable radix sort of (hash, index)
//     pairs -- equal hashes keep ascending index, so a run's head IS np.unique's first occurrence
//     --, head flags + exclusive scan -> unique ids, scatter of the inverse map.
#include <climits>
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "common.h"

namespace {

#pragma clang fp contract(off)

__global__ __launch_bounds__(256) void bev_occupancy_kernel(const float* __restrict__ pts, int stride,
                                                           int N, float x0, float x1, float y0,
                                                           float y1, float voxel, int minbx,
                                                           int minby, int nx, int ny, int stamp,
                                                           int* __restrict__ stamps,
                                                           float* __restrict__ grid) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const float x = pts[(long long)i * stride], y = pts[(long long)i * stride + 1];
    if (!(x > x0 && x < x1 && y > y0 && y < y1)) return;              // metric_utils.py:243-246
    const int ix = (int)floorf(x / voxel) - minbx, iy = (int)floorf(y / voxel) - minby;
    if (ix < 0 || ix >= nx || iy < 0 || iy >= ny) return;              // (numpy would raise)
    const int v = ix * ny + iy;
    if (atomicExch(stamps + v, stamp) != stamp) atomicAdd(grid + v, 1.0f);
}

// ---- sparse_quantize ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sq_floor_minmax_kernel(const float* __restrict__ c, int N, int D,
                                                             float vx, float vy, float vz,
                                                             int* __restrict__ q, int* __restrict__ mm) {
    // q = floor(coords / voxel) as int32; mm[0..2] = per-dim min, mm[3..5] = per-dim max
    const int i = blockIdx.x * 256 + threadIdx.x;
    int lo[3] = {INT_MAX, INT_MAX, INT_MAX}, hi[3] = {INT_MIN, INT_MIN, INT_MIN};
    if (i < N) {
        const float vs[3] = {vx, vy, vz};
        for (int d = 0; d < D; ++d) {
            // numpy: float32 coords / float64 voxel array -> float64 quotient, floor, astype(int32)
            const int v = (int)floor((double)c[(long long)i * D + d] / (double)vs[d]);
            q[(long long)i * D + d] = v;
            lo[d] = hi[d] = v;
        }
    }
    for (int d = 0; d < D; ++d) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            lo[d] = min(lo[d], __shfl_xor(lo[d], o, 64));
            hi[d] = max(hi[d], __shfl_xor(hi[d], o, 64));
        }
        if ((threadIdx.x & 63) == 0) { atomicMin(mm + d, lo[d]); atomicMax(mm + 3 + d, hi[d]); }
    }
}

__global__ void sq_init_kernel(int* mm, unsigned long long* count) {
    if (threadIdx.x < 3) { mm[threadIdx.x] = INT_MAX; mm[3 + threadIdx.x] = INT_MIN; }
    if (threadIdx.x == 0) *count = 0ull;
}

__global__ __launch_bounds__(256) void sq_hash_kernel(const int* __restrict__ q, int N, int D,
                                                     const int* __restrict__ mm,
                                                     unsigned long long* __restrict__ key,
                                                     int* __restrict__ idx) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    unsigned long long h = 0ull;                      // ravel_hash, metric_utils.py:28-40
    for (int d = 0; d < D - 1; ++d) {
        h += (unsigned long long)(long long)(q[(long long)i * D + d] - mm[d]);
        h *= (unsigned long long)(long long)(mm[3 + d + 1] - mm[d + 1]) + 1ull;
    }
    h += (unsigned long long)(long long)(q[(long long)i * D + D - 1] - mm[D - 1]);
    key[i] = h;
    idx[i] = i;
}

__global__ __launch_bounds__(256) void sq_heads_kernel(const unsigned long long* __restrict__ key, int N,
                                                      int* __restrict__ head) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < N) head[i] = (i == 0 || key[i] != key[i - 1]) ? 1 : 0;
}

__global__ __launch_bounds__(256) void sq_emit_kernel(const int* __restrict__ q, int N, int D,
                                                     const int* __restrict__ sidx,
                                                     const int* __restrict__ head,
                                                     const int* __restrict__ uid_excl,
                                                     int* __restrict__ out_coords,
                                                     long long* __restrict__ out_index,
                                                     long long* __restrict__ out_inverse,
                                                     unsigned long long* __restrict__ count) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const int src = sidx[i];
    const int u = uid_excl[i] + head[i] - 1;          // id of the run this sorted element is in
    if (out_inverse) out_inverse[src] = u;
    if (head[i]) {
        for (int d = 0; d < D; ++d) out_coords[(long long)u * D + d] = q[(long long)src * D + d];
        if (out_index) out_index[u] = src;
    }
    if (i == N - 1) *count = (unsigned long long)(u + 1);
}

struct SqScratch {
    int* mm; unsigned long long* count; int* q; unsigned long long* key_a; unsigned long long* key_b;
    int* idx_a; int* idx_b; int* head; int* uid; void* tmp; size_t tmp_bytes;
};

size_t align256(size_t v) { return (v + 255) / 256 * 256; }

size_t sq_tmp_bytes(int N) {
    size_t a = 0, b = 0;
    (void)rocprim::radix_sort_pairs(nullptr, a, (unsigned long long*)nullptr, (unsigned long long*)nullptr,
                              (int*)nullptr, (int*)nullptr, (size_t)N);
    (void)rocprim::exclusive_scan(nullptr, b, (int*)nullptr, (int*)nullptr, 0, (size_t)N,
                            rocprim::plus<int>());
    return a > b ? a : b;
}

SqScratch sq_carve(void* base, int N, int D) {
    char* p = static_cast<char*>(base);
    SqScratch s;
    auto take = [&](size_t bytes) { void* r = p; p += align256(bytes); return r; };
    s.mm = (int*)take(6 * sizeof(int));
    s.count = (unsigned long long*)take(sizeof(unsigned long long));
    s.q = (int*)take((size_t)N * D * sizeof(int));
    s.key_a = (unsigned long long*)take((size_t)N * 8);
    s.key_b = (unsigned long long*)take((size_t)N * 8);
    s.idx_a = (int*)take((size_t)N * 4);
    s.idx_b = (int*)take((size_t)N * 4);
    s.head = (int*)take((size_t)N * 4);
    s.uid = (int*)take((size_t)N * 4);
    s.tmp_bytes = sq_tmp_bytes(N);
    s.tmp = take(s.tmp_bytes);
    return s;
}

}  // namespace

extern "C" int lc_bev_occupancy_accumulate(const float* pts, int pt_stride, int N, float x0, float x1,
                                           float y0, float y1, float voxel, int min_bound_x,
                                           int min_bound_y, int nx, int ny, int stamp,
                                           int32_t* stamps, float* grid, lc_stream_t s) {
    if ((!pts && N > 0) || !stamps || !grid || N < 0 || pt_stride < 2 || nx <= 0 || ny <= 0 ||
        !(voxel > 0.f) || stamp == 0)
        return LC_EINVAL;
    if (N > 0)
        hipLaunchKernelGGL(bev_occupancy_kernel, dim3((N + 255) / 256), dim3(256), 0, lc_s(s), pts,
                           pt_stride, N, x0, x1, y0, y1, voxel, min_bound_x, min_bound_y, nx, ny,
                           stamp, stamps, grid);
    return lc_launch_status();
}

extern "C" int64_t lc_sparse_quantize_scratch_bytes(int N, int D) {
    if (N <= 0 || (D != 2 && D != 3)) return 0;
    return (int64_t)(align256(24) + align256(8) + align256((size_t)N * D * 4) + 2 * align256((size_t)N * 8) +
                     4 * align256((size_t)N * 4) + align256(sq_tmp_bytes(N)));
}

extern "C" int lc_sparse_quantize(const float* coords, int N, int D, float vx, float vy, float vz,
                                  void* scratch, int32_t* out_coords, int64_t* out_index,
                                  int64_t* out_inverse, uint64_t* out_count, lc_stream_t s) {
    if (!coords || !scratch || !out_coords || !out_count || N <= 0 || (D != 2 && D != 3) ||
        !(vx > 0.f) || !(vy > 0.f) || (D == 3 && !(vz > 0.f)))
        return LC_EINVAL;
    hipStream_t st = lc_s(s);
    SqScratch w = sq_carve(scratch, N, D);
    const dim3 g((N + 255) / 256), b(256);
    hipLaunchKernelGGL(sq_init_kernel, dim3(1), dim3(64), 0, st, w.mm, w.count);
    hipLaunchKernelGGL(sq_floor_minmax_kernel, g, b, 0, st, coords, N, D, vx, vy, vz, w.q, w.mm);
    hipLaunchKernelGGL(sq_hash_kernel, g, b, 0, st, w.q, N, D, w.mm, w.key_a, w.idx_a);
    size_t tb = w.tmp_bytes;
    if (rocprim::radix_sort_pairs(w.tmp, tb, w.key_a, w.key_b, w.idx_a, w.idx_b, (size_t)N, 0, 64,
                                  st) != hipSuccess)
        return lc_launch_status() ? lc_launch_status() : LC_EUNSUP;
    hipLaunchKernelGGL(sq_heads_kernel, g, b, 0, st, w.key_b, N, w.head);
    tb = w.tmp_bytes;
    if (rocprim::exclusive_scan(w.tmp, tb, w.head, w.uid, 0, (size_t)N, rocprim::plus<int>(), st) !=
        hipSuccess)
        return lc_launch_status() ? lc_launch_status() : LC_EUNSUP;
    hipLaunchKernelGGL(sq_emit_kernel, g, b, 0, st, w.q, N, D, w.idx_b, w.head, w.uid, out_coords,
                       (long long*)out_index, (long long*)out_inverse, w.count);
    if (hipMemcpyAsync(out_count, w.count, 8, hipMemcpyDeviceToDevice, st) != hipSuccess)
        return lc_launch_status();
    return lc_launch_status();
}

LC_TOUCH_TU(voxel, bev_occupancy_kernel)
