// Point-cloud side of the path: spherical projection z-buffer and points-in-boxes.
// HBM-bound gather/scatter work: coalesced point reads, 64-bit atomics for the z-buffer.
#include "common.h"

namespace {

#pragma clang fp contract(off)

// lidargen/dataset/transforms_3d/common.py:44-45,72-81 -- every float32 op correctly rounded
// (asin/atan2 in fp64, rounded once).  ELEV64 = true: the elevation -> row arithmetic runs in
// float64 on the float32 asin result, as the reference does under numpy >= 2 (np.deg2rad of a
// python float is a float64 scalar and promotes the expression; oracle/lidar.py "native_cr");
// false: all-float32, the reference under its pinned numpy 1.23.5 (oracle "f32").
template <bool ELEV64>
__device__ __forceinline__ int row_of(float a32, int H, double h_up64, double h_down64) {
    if constexpr (ELEV64) {
        const double elev = (double)a32 + fabs(h_down64);
        double fh = 1.0 - elev / (h_up64 - h_down64);
        fh = floor(fh * (double)H);
        return (int)fmin(fmax(fh, 0.0), (double)(H - 1));
    } else {
        const float h_up = (float)h_up64, h_down = (float)h_down64;
        const float elev = a32 + fabsf(h_down);
        float fh = 1.0f - elev / (h_up - h_down);
        fh = floorf(fh * (float)H);
        return (int)fminf(fmaxf(fh, 0.f), (float)(H - 1));
    }
}
// azimuth -> (v before np.mod, column of v): v = (-atan2 / pi32 + 1) / 2
__device__ __forceinline__ float col_v(float atan32) { return (-atan32 / 3.14159274101257324f + 1.0f) / 2.0f; }
__device__ __forceinline__ int col_of(float fw, int W) {
    fw = fw - floorf(fw);              // np.mod(v, 1) for v in [0, 1]
    fw = floorf(fw * (float)W);
    return (int)fminf(fmaxf(fw, 0.f), (float)(W - 1));
}

// The cell of a point.  DEFINITION (exact path): a32 = float32(asin(double t)), atan32 = float32(atan2(double y, double x)),
// then the reference's float32 / float64 arithmetic.  Round 5: the two fp64 libm calls were the launch (4 M points: ~80 of
// 120 us); only the CELL depends on them, so a point first brackets both angles with the fp32 functions (OpenCL accuracy:
// asin <= 4 ulp, atan2 <= 6 ulp; bracket = +-16 ulp) and runs the reference's rounded chain on both ends: every step of
// it is monotone (rounding is), so equal cells at the two ends ARE the exact path's cell.  The exact path runs only
// for a point whose bracket straddles a cell boundary (~3e-5 of the points on 32 x 1024 cells), touches the wrap of
// np.mod at v = 1, or holds a NaN.
template <bool ELEV64>
__device__ __forceinline__ void cell_of(float x, float y, float z, int H, int W, double h_up64,
                                        double h_down64, float& depth, int& gh, int& gw) {
    depth = sqrtf((x * x + y * y) + z * z);
    const float t = z / (depth + 1e-6f);
#ifndef LC_PROJ_EXACT_ONLY
    {
        const float af = asinf(t), zf = atan2f(y, x);
        const float ma = fabsf(af) * 1.9073486e-6f + 1e-30f, mz = fabsf(zf) * 1.9073486e-6f + 1e-30f;   // 16 ulp
        const int g0 = row_of<ELEV64>(af - ma, H, h_up64, h_down64), g1 = row_of<ELEV64>(af + ma, H, h_up64, h_down64);
        const float v0 = col_v(zf - mz), v1 = col_v(zf + mz);
        const int w0 = col_of(v0, W), w1 = col_of(v1, W);
        const double dlt = h_up64 - h_down64;
        const bool sane = af == af && zf == zf && dlt == dlt && dlt != 0.0 && fabs(dlt) < 1e30;
        if (sane && g0 == g1 && w0 == w1 && v0 >= 0.f && v0 < 1.f && v1 >= 0.f && v1 < 1.f) {
            gh = g0; gw = w0;
            return;
        }
    }
#endif
    gh = row_of<ELEV64>((float)asin((double)t), H, h_up64, h_down64);
    gw = col_of(col_v((float)atan2((double)y, (double)x)), W);
}

__global__ void zbuf_clear_kernel(unsigned long long* zb, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) zb[i] = ~0ull;
}

template <bool ELEV64>
__global__ __launch_bounds__(256) void project_scatter_kernel(const float* __restrict__ pts, int N,
                                                             int H, int W, double h_up,
                                                             double h_down,
                                                             unsigned long long* __restrict__ zb,
                                                             int* __restrict__ cells) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const f32x4 p = *reinterpret_cast<const f32x4*>(pts + 4ll * i);
    float depth; int gh, gw;
    cell_of<ELEV64>(p.x, p.y, p.z, H, W, h_up, h_down, depth, gh, gw);
    if (cells) { cells[2 * i] = gh; cells[2 * i + 1] = gw; }
    if (depth != depth) return;  // NaN never wins (numpy argsort puts NaN last -> overwritten)
    const unsigned long long key = ((unsigned long long)__float_as_uint(depth) << 32) | (unsigned)i;
    // Dense clouds (4 M points on 32 x 1024 cells: ~128 candidates per cell) serialise on same-address 64-bit atomics in
    // L2.  A cell's key only ever decreases, so a candidate that does not beat the value read a moment ago (an L2 read:
    // agent-scope relaxed load, never a stale L1 line that would merely filter less) can be dropped without the atomic.
    unsigned long long* cell = zb + (long long)gh * W + gw;
    if (key < __hip_atomic_load(cell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(cell, key);
}

// RESET: the cell is handed back empty (~0), so a caller-owned z-buffer needs no clear launch before the
// next projection (lc_project_points_ws)
template <bool RESET>
__global__ __launch_bounds__(256) void project_gather_kernel(const float* __restrict__ pts, int HW,
                                                            unsigned long long* __restrict__ zb,
                                                            float min_d, float max_d,
                                                            float* __restrict__ img,
                                                            int* __restrict__ winner) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= HW) return;
    const unsigned long long key = zb[c];
    if (RESET) zb[c] = ~0ull;
    float o[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int win = -1;
    if (key != ~0ull) {
        win = (int)(key & 0xFFFFFFFFull);
        const f32x4 p = *reinterpret_cast<const f32x4*>(pts + 4ll * win);
        const float depth = __uint_as_float((unsigned)(key >> 32));
        o[0] = p.x; o[1] = p.y; o[2] = p.z; o[3] = p.w; o[4] = depth;
        o[5] = (depth >= min_d && depth <= max_d) ? 1.f : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) img[6ll * c + k] = o[k];
    if (winner) winner[c] = win;
}

// ---- float64 point sets (round 3) -----------------------------------------------------------------
// The temporal glue hands `load_points_as_images` FLOAT64 points (pipe_related.py:245-258: the
// float64 product `Ts @ homo`, and the float64 concatenation [background | re-posed objects]); every
// line of common.py:41-84 then runs in float64 and the image is rounded to float32 once (:87-91).
// A 64-bit depth leaves no room for the index in one atomic key: pass 1 = atomicMin of the depth
// bits per cell, pass 2 = atomicMin of the index among the points AT that depth (lowest index wins,
// the same tie rule as the float32 path), pass 3 = gather.
__device__ __forceinline__ void cell_of64(double x, double y, double z, int H, int W, double h_up,
                                          double h_down, double& depth, int& gh, int& gw) {
    depth = sqrt((x * x + y * y) + z * z);
    const double elev = asin(z / (depth + 1e-6)) + fabs(h_down);
    double fh = 1.0 - elev / (h_up - h_down);
    fh = floor(fh * (double)H);
    gh = (int)fmin(fmax(fh, 0.0), (double)(H - 1));
    double fw = (-atan2(y, x) / 3.141592653589793 + 1.0) / 2.0;
    fw = fw - floor(fw);               // np.mod(v, 1)
    fw = floor(fw * (double)W);
    gw = (int)fmin(fmax(fw, 0.0), (double)(W - 1));
}

__global__ void zbuf64_clear_kernel(unsigned long long* zb, int* win, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { zb[i] = ~0ull; win[i] = 0x7fffffff; }
}

template <int PASS>
__global__ __launch_bounds__(256) void project64_scatter_kernel(const double* __restrict__ pts, int N,
                                                               int H, int W, double h_up,
                                                               double h_down,
                                                               unsigned long long* __restrict__ zb,
                                                               int* __restrict__ win) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const double x = pts[4ll * i], y = pts[4ll * i + 1], z = pts[4ll * i + 2];
    double depth; int gh, gw;
    cell_of64(x, y, z, H, W, h_up, h_down, depth, gh, gw);
    if (depth != depth) return;
    const unsigned long long key = (unsigned long long)__double_as_longlong(depth);  // depth >= 0
    const long long c = (long long)gh * W + gw;
    if (PASS == 0) atomicMin(zb + c, key);
    else if (zb[c] == key) atomicMin(win + c, i);
}

__global__ __launch_bounds__(256) void project64_gather_kernel(const double* __restrict__ pts, int HW,
                                                              const unsigned long long* __restrict__ zb,
                                                              double min_d, double max_d,
                                                              float* __restrict__ img,
                                                              int* __restrict__ winner) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= HW) return;
    const unsigned long long key = zb[c];
    float o[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int win = -1;
    if (key != ~0ull) {
        win = winner[c];
        const double depth = __longlong_as_double((long long)key);
        o[0] = (float)pts[4ll * win]; o[1] = (float)pts[4ll * win + 1];
        o[2] = (float)pts[4ll * win + 2]; o[3] = (float)pts[4ll * win + 3];
        o[4] = (float)depth;
        o[5] = (depth >= min_d && depth <= max_d) ? 1.f : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) img[6ll * c + k] = o[k];
    winner[c] = win;
}

// roiaware_pool3d.cpp:121-140 / roiaware_pool3d_kernel.cu:16-36: float rotation, double compares.
// Per-box constants live in LDS, 12 floats per box: (cx, cy, cz, cosa, sina, -, hz, hx, hy as three doubles from
// float 6): the reference evaluates cos(-rz) / sin(-rz) in double and rounds to float, and compares |.| against
// (double)d / 2.0 (+ (double)margin) -- all of that depends on the box only, so it is computed ONCE per block here
// (round 5; before, every (point, box) pair paid two fp64 transcendentals: 13 boxes x 4 M points = 1.37 TB/s
// "algorithmic").  Same expressions, same roundings: bit-exact with the per-pair form.
constexpr int PIB_BOX = 12;
__device__ __forceinline__ void pib_stage_boxes(const float* __restrict__ boxes, int nb, float margin, float* sb) {
    for (int k = threadIdx.x; k < nb; k += 256) {
        const float* bx = boxes + 7 * k;
        float* o = sb + PIB_BOX * k;
        const float rz = bx[6];
        o[0] = bx[0]; o[1] = bx[1]; o[2] = bx[2];
        o[3] = (float)cos((double)(-rz)); o[4] = (float)sin((double)(-rz)); o[5] = 0.f;
        double* d = reinterpret_cast<double*>(o + 6);
        d[0] = (double)bx[5] / 2.0;                         // |z - cz| >  hz  -> outside
        d[1] = (double)bx[3] / 2.0 + (double)margin;        // |lx|     <  hx
        d[2] = (double)bx[4] / 2.0 + (double)margin;        // |ly|     <  hy
    }
    __syncthreads();
}
__device__ __forceinline__ int pt_in_box(float x, float y, float z, const float* o) {
    const double* d = reinterpret_cast<const double*>(o + 6);
    if ((double)fabsf(z - o[2]) > d[0]) return 0;
    const float cosa = o[3], sina = o[4];
    const float sx = x - o[0], sy = y - o[1];
    const float lx = sx * cosa + sy * (-sina);
    const float ly = sx * sina + sy * cosa;
    return ((double)fabsf(lx) < d[1]) & ((double)fabsf(ly) < d[2]);
}

__global__ __launch_bounds__(256) void pib_mask_kernel(const float* __restrict__ boxes, int nb,
                                                      const float* __restrict__ pts, int np,
                                                      float margin, int* __restrict__ out) {
    extern __shared__ double sb_d[];
    float* sb = reinterpret_cast<float*>(sb_d);
    pib_stage_boxes(boxes, nb, margin, sb);
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= np) return;
    const float x = pts[3ll * j], y = pts[3ll * j + 1], z = pts[3ll * j + 2];
    for (int k = 0; k < nb; ++k) out[(long long)k * np + j] = pt_in_box(x, y, z, sb + PIB_BOX * k);
}

// [N,4] points (x, y, z, intensity): mask rows like points_in_boxes_cpu and, per point, the number
// of boxes containing it (delete_fg_points / get_temporal_boxes_3d, pipe_related.py:52-62,266-272)
__global__ __launch_bounds__(256) void pib_mask4_kernel(const float* __restrict__ boxes, int nb,
                                                       const float* __restrict__ pts, int np,
                                                       float margin, int* __restrict__ out_mask,
                                                       int* __restrict__ out_count) {
    extern __shared__ double sb_d[];
    float* sb = reinterpret_cast<float*>(sb_d);
    pib_stage_boxes(boxes, nb, margin, sb);
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= np) return;
    const f32x4 p = *reinterpret_cast<const f32x4*>(pts + 4ll * j);
    int cnt = 0;
    for (int k = 0; k < nb; ++k) {
        const int in = pt_in_box(p.x, p.y, p.z, sb + PIB_BOX * k);
        if (out_mask) out_mask[(long long)k * np + j] = in;
        cnt += in;
    }
    if (out_count) out_count[j] = cnt;
}

__global__ __launch_bounds__(256) void pib_index_kernel(const float* __restrict__ boxes, int nb,
                                                       const float* __restrict__ pts, int np,
                                                       float margin, int* __restrict__ out) {
    extern __shared__ double sb_d[];
    float* sb = reinterpret_cast<float*>(sb_d);
    const int b = blockIdx.y;
    pib_stage_boxes(boxes + (long long)b * nb * 7, nb, margin, sb);
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= np) return;
    const float* p = pts + ((long long)b * np + j) * 3;
    const float x = p[0], y = p[1], z = p[2];
    int idx = -1;
    for (int k = 0; k < nb; ++k)
        if (pt_in_box(x, y, z, sb + PIB_BOX * k)) { idx = k; break; }
    out[(long long)b * np + j] = idx;
}

}  // namespace

extern "C" int lc_project_points(const float* points, int N, int H, int W, double fov_up_deg,
                                 double fov_down_deg, float min_depth, float max_depth,
                                 uint64_t* zbuf, float* image, int32_t* winner, int32_t* cells,
                                 int elev_f64, lc_stream_t s) {
    if ((!points && N > 0) || !zbuf || !image || N < 0 || H <= 0 || W <= 0) return LC_EINVAL;
    if (reinterpret_cast<uintptr_t>(points) & 15) return LC_EINVAL;
    const int HW = H * W;
    // np.deg2rad in float64 (rounded to float32 inside the kernel for the all-float32 mode)
    const double h_up = fov_up_deg * 0.017453292519943295;
    const double h_down = fov_down_deg * 0.017453292519943295;
    auto zb = reinterpret_cast<unsigned long long*>(zbuf);
    hipLaunchKernelGGL(zbuf_clear_kernel, dim3((HW + 255) / 256), dim3(256), 0, lc_s(s), zb, HW);
    if (N > 0) {
        if (elev_f64)
            hipLaunchKernelGGL(project_scatter_kernel<true>, dim3((N + 255) / 256), dim3(256), 0,
                               lc_s(s), points, N, H, W, h_up, h_down, zb, cells);
        else
            hipLaunchKernelGGL(project_scatter_kernel<false>, dim3((N + 255) / 256), dim3(256), 0,
                               lc_s(s), points, N, H, W, h_up, h_down, zb, cells);
    }
    hipLaunchKernelGGL(project_gather_kernel<false>, dim3((HW + 255) / 256), dim3(256), 0, lc_s(s), points,
                       HW, zb, min_depth, max_depth, image, winner);
    return lc_launch_status();
}

extern "C" int lc_project_workspace_init(uint64_t* zbuf, int n_cells, lc_stream_t s) {
    if (!zbuf || n_cells <= 0) return LC_EINVAL;
    hipLaunchKernelGGL(zbuf_clear_kernel, dim3((n_cells + 255) / 256), dim3(256), 0, lc_s(s),
                       reinterpret_cast<unsigned long long*>(zbuf), n_cells);
    return lc_launch_status();
}

extern "C" int lc_project_points_ws(const float* points, int N, int H, int W, double fov_up_deg,
                                    double fov_down_deg, float min_depth, float max_depth,
                                    uint64_t* zbuf, float* image, int32_t* winner, int32_t* cells,
                                    int elev_f64, lc_stream_t s) {
    if ((!points && N > 0) || !zbuf || !image || N < 0 || H <= 0 || W <= 0) return LC_EINVAL;
    if (reinterpret_cast<uintptr_t>(points) & 15) return LC_EINVAL;
    const int HW = H * W;
    const double h_up = fov_up_deg * 0.017453292519943295;
    const double h_down = fov_down_deg * 0.017453292519943295;
    auto zb = reinterpret_cast<unsigned long long*>(zbuf);
    if (N > 0) {
        if (elev_f64)
            hipLaunchKernelGGL(project_scatter_kernel<true>, dim3((N + 255) / 256), dim3(256), 0,
                               lc_s(s), points, N, H, W, h_up, h_down, zb, cells);
        else
            hipLaunchKernelGGL(project_scatter_kernel<false>, dim3((N + 255) / 256), dim3(256), 0,
                               lc_s(s), points, N, H, W, h_up, h_down, zb, cells);
    }
    hipLaunchKernelGGL(project_gather_kernel<true>, dim3((HW + 255) / 256), dim3(256), 0, lc_s(s), points,
                       HW, zb, min_depth, max_depth, image, winner);
    return lc_launch_status();
}

extern "C" int lc_project_points_f64(const double* points, int N, int H, int W, double fov_up_deg,
                                     double fov_down_deg, double min_depth, double max_depth,
                                     uint64_t* zbuf, int32_t* winner, float* image, lc_stream_t s) {
    if ((!points && N > 0) || !zbuf || !winner || !image || N < 0 || H <= 0 || W <= 0) return LC_EINVAL;
    const int HW = H * W;
    const double h_up = fov_up_deg * 0.017453292519943295;
    const double h_down = fov_down_deg * 0.017453292519943295;
    auto zb = reinterpret_cast<unsigned long long*>(zbuf);
    hipLaunchKernelGGL(zbuf64_clear_kernel, dim3((HW + 255) / 256), dim3(256), 0, lc_s(s), zb, winner, HW);
    if (N > 0) {
        hipLaunchKernelGGL(project64_scatter_kernel<0>, dim3((N + 255) / 256), dim3(256), 0, lc_s(s),
                           points, N, H, W, h_up, h_down, zb, winner);
        hipLaunchKernelGGL(project64_scatter_kernel<1>, dim3((N + 255) / 256), dim3(256), 0, lc_s(s),
                           points, N, H, W, h_up, h_down, zb, winner);
    }
    hipLaunchKernelGGL(project64_gather_kernel, dim3((HW + 255) / 256), dim3(256), 0, lc_s(s), points,
                       HW, zb, min_depth, max_depth, image, winner);
    return lc_launch_status();
}

extern "C" int lc_points_in_boxes_mask(const float* boxes, int n_boxes, const float* pts, int n_pts,
                                       float margin, int32_t* out_mask, lc_stream_t s) {
    if (!boxes || !pts || !out_mask || n_boxes <= 0 || n_pts <= 0) return LC_EINVAL;
    if (n_boxes * PIB_BOX * sizeof(float) > 60000) return LC_EUNSUP;
    hipLaunchKernelGGL(pib_mask_kernel, dim3((n_pts + 255) / 256), dim3(256),
                       n_boxes * PIB_BOX * sizeof(float), lc_s(s), boxes, n_boxes, pts, n_pts, margin,
                       out_mask);
    return lc_launch_status();
}

extern "C" int lc_points_in_boxes_index(const float* boxes, const float* pts, int B, int n_boxes,
                                        int n_pts, float margin, int32_t* out_idx, lc_stream_t s) {
    if (!boxes || !pts || !out_idx || B <= 0 || n_boxes <= 0 || n_pts <= 0) return LC_EINVAL;
    if (n_boxes * PIB_BOX * sizeof(float) > 60000) return LC_EUNSUP;
    hipLaunchKernelGGL(pib_index_kernel, dim3((n_pts + 255) / 256, B), dim3(256),
                       n_boxes * PIB_BOX * sizeof(float), lc_s(s), boxes, n_boxes, pts, n_pts, margin,
                       out_idx);
    return lc_launch_status();
}

extern "C" int lc_points_in_boxes_mask4(const float* boxes, int n_boxes, const float* pts4, int n_pts,
                                        float margin, int32_t* out_mask, int32_t* out_count,
                                        lc_stream_t s) {
    if (!boxes || !pts4 || (!out_mask && !out_count) || n_boxes <= 0 || n_pts <= 0) return LC_EINVAL;
    if (reinterpret_cast<uintptr_t>(pts4) & 15) return LC_EINVAL;
    if (n_boxes * PIB_BOX * sizeof(float) > 60000) return LC_EUNSUP;
    hipLaunchKernelGGL(pib_mask4_kernel, dim3((n_pts + 255) / 256), dim3(256),
                       n_boxes * PIB_BOX * sizeof(float), lc_s(s), boxes, n_boxes, pts4, n_pts, margin,
                       out_mask, out_count);
    return lc_launch_status();
}

LC_TOUCH_TU(geometry, zbuf_clear_kernel)
