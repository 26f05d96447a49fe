// Layout condition rasteriser on the device: 3-D boxes -> 2-D range-image
// rectangles -> class / depth condition mask, painter's order.
//   /root/reference/lidargen/dataset/transforms_3d/common.py: rotz :93-97,
//   convert_boxes_to_2d :99-181 (numpy, Python loop over boxes), convert_points_to_2d :184-215.
// Kernel A (one thread per box): 8 corners -> spherical cells -> integer rectangle (float64
// geometry like the reference; for float32 boxes yaw cos/sin and the centre depth in float32 like
// np.cos(np.float32); for FLOAT64 boxes -- what NuscDataset.pre_process hands over,
// nuscenes_dataset.py:384-397 -- everything in float64 and the centre depth rounded to float32
// when it is painted, pinned on the reference's CustomDataset item, tests/golden/pipe_next.npz).
// Kernel B (one thread per pixel): the LAST box covering the pixel wins (== sequential overwrite),
// plus the training loss-weight map exp(sum_k cover_k * (3 - area_k / max area)).
#include "common.h"

namespace {

#pragma clang fp contract(off)

struct Rect { int x1, y1, x2, y2, wrap; float cls, depth, area; };

template <typename BT>
__global__ void box_rect_kernel(const BT* __restrict__ boxes, int stride, const int* __restrict__ nvalid,
                                int T, int H, int W, double h_up, double h_down,
                                Rect* __restrict__ rects, float* __restrict__ corners2d) {
    const int b = blockIdx.x, k = threadIdx.x;
    if (k >= T) return;
    Rect r = {0, 0, 0, 0, 0, 0.f, 0.f, 0.f};
    float c2[4] = {0.f, 0.f, 0.f, 0.f};
    if (k < nvalid[b]) {
        const BT* bx = boxes + ((long long)b * T + k) * stride;
        const double l = bx[3], w = bx[4], h = bx[5];
        const double c = (double)(BT)cos((double)bx[6]), s = (double)(BT)sin((double)bx[6]);
        const double cx = bx[0], cy = bx[1], cz = bx[2];
        const double sxs[8] = {.5, .5, -.5, -.5, .5, .5, -.5, -.5};
        const double sys[8] = {.5, -.5, -.5, .5, .5, -.5, -.5, .5};
        const double szs[8] = {.5, .5, .5, .5, -.5, -.5, -.5, -.5};
        double gwmin = 2, gwmax = -1, ghmin = 2, ghmax = -1;
        for (int q = 0; q < 8; ++q) {
            const double X = l * sxs[q], Y = w * sys[q], Z = h * szs[q];
            const double px = c * X - s * Y + cx, py = s * X + c * Y + cy, pz = Z + cz;
            const double depth = sqrt(px * px + py * py + pz * pz) + 1e-6;
            double gh = 1 - (asin(pz / depth) + fabs(h_down)) / (h_up - h_down);
            gh = fmin(fmax(floor(gh * H), 0.0), (double)(H - 1)) / H;
            double gw = (-atan2(py, px) / 3.141592653589793 + 1) / 2;
            gw = gw - floor(gw);                                    // % 1
            gw = fmin(fmax(floor(gw * W), 0.0), (double)(W - 1)) / W;
            gwmin = fmin(gwmin, gw); gwmax = fmax(gwmax, gw);
            ghmin = fmin(ghmin, gh); ghmax = fmax(ghmax, gh);
        }
        r.x1 = (int)(gwmin * W); r.x2 = (int)(gwmax * W);
        r.y1 = (int)(ghmin * H); r.y2 = (int)(ghmax * H);
        r.wrap = ((double)(r.x2 - r.x1) / W > 0.6) ? 1 : 0;
        r.area = r.wrap ? (float)((W - r.x2 + r.x1) * (r.y2 - r.y1))
                        : (float)((r.x2 - r.x1) * (r.y2 - r.y1));
        r.cls = (float)bx[7];
        if constexpr (sizeof(BT) == 8)
            r.depth = (float)(sqrt((cx * cx + cy * cy) + cz * cz) + 1e-6);
        else
            r.depth = sqrtf((bx[0] * bx[0] + bx[1] * bx[1]) + bx[2] * bx[2]) + 1e-6f;
        c2[0] = (float)gwmin; c2[1] = (float)ghmin; c2[2] = (float)gwmax; c2[3] = (float)ghmax;
    }
    rects[(long long)b * T + k] = r;
    if (corners2d)
        for (int q = 0; q < 4; ++q) corners2d[((long long)b * T + k) * 4 + q] = c2[q];
}

__global__ __launch_bounds__(256) void paint_kernel(const Rect* __restrict__ rects,
                                                   const int* __restrict__ nvalid, int T, int H,
                                                   int W, float* __restrict__ mask,
                                                   float* __restrict__ wmap) {
    extern __shared__ Rect sr[];
    const int b = blockIdx.y;
    const int n = min(nvalid[b], T);
    for (int i = threadIdx.x; i < n; i += 256) sr[i] = rects[(long long)b * T + i];
    __syncthreads();
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= H * W) return;
    const int y = p / W, x = p - y * W;
    float amax = 0.f;
    for (int k = 0; k < n; ++k) amax = fmaxf(amax, sr[k].area);
    float cls = 0.f, dep = 0.f, wsum = 0.f;
    for (int k = 0; k < n; ++k) {
        const Rect r = sr[k];
        const bool in = y >= r.y1 && y < r.y2 && (r.wrap ? (x < r.x1 || x >= r.x2) : (x >= r.x1 && x < r.x2));
        if (in) { cls = r.cls; dep = r.depth; wsum += 3.0f - r.area / amax; }
    }
    float* m = mask + (long long)b * 2 * H * W;
    m[p] = cls;
    m[(long long)H * W + p] = dep;
    if (wmap) wmap[(long long)b * H * W + p] = expf(wsum);
}

}  // namespace

extern "C" int64_t lc_layout_scratch_bytes(int B, int T) { return (int64_t)B * T * sizeof(Rect); }

extern "C" int lc_layout_condition(const void* boxes, int boxes_f64, int box_stride,
                                   const int32_t* n_valid, int B, int T, int H, int W,
                                   double fov_up_deg, double fov_down_deg, void* scratch,
                                   float* corners_2d, float* condition_mask, float* loss_weight_map,
                                   lc_stream_t s) {
    if (!boxes || !n_valid || !scratch || !condition_mask || B <= 0 || T <= 0 || T > 1024 ||
        box_stride < 8 || H <= 0 || W <= 0)
        return LC_EINVAL;
    const double h_up = fov_up_deg * 0.017453292519943295;
    const double h_down = fov_down_deg * 0.017453292519943295;
    Rect* rects = reinterpret_cast<Rect*>(scratch);
    if (boxes_f64)
        hipLaunchKernelGGL(box_rect_kernel<double>, dim3(B), dim3((T + 63) / 64 * 64), 0, lc_s(s),
                           reinterpret_cast<const double*>(boxes), box_stride, n_valid, T, H, W, h_up,
                           h_down, rects, corners_2d);
    else
        hipLaunchKernelGGL(box_rect_kernel<float>, dim3(B), dim3((T + 63) / 64 * 64), 0, lc_s(s),
                           reinterpret_cast<const float*>(boxes), box_stride, n_valid, T, H, W, h_up,
                           h_down, rects, corners_2d);
    hipLaunchKernelGGL(paint_kernel, dim3((H * W + 255) / 256, B), dim3(256), T * sizeof(Rect),
                       lc_s(s), rects, n_valid, T, H, W, condition_mask, loss_weight_map);
    return lc_launch_status();
}

LC_TOUCH_TU(layout, paint_kernel)
