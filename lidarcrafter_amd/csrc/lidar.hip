// Once-per-sample pre/post-processing of range images, fused into single passes.
//   postprocess : tools/evaluation/sample_and_save_cond.py:119-124 =
//                 LiDARUtility.denormalize (utils/lidar.py:61-64) -> revert_depth (:109-128)
//                 -> to_xyz (:71-82) -> cat[depth, xyz, reflectance]   (5 torch passes -> 1)
//   condition   : preprocess_condition_mask sample_and_save_cond.py:106-117 =
//                 one_hot(class) ++ LiDARUtility.convert_depth (utils/lidar.py:84-107)
#include "common.h"

namespace {

__device__ __forceinline__ float revert(float n, int fmt, float min_d, float max_d, float l2m) {
    float m;
    if (fmt == 0) m = exp2f(n * l2m) - 1.0f;            // log_depth
    else if (fmt == 1) m = min_d / (n + 1e-8f);         // inverse_depth
    else m = n * max_d;                                 // depth
    return (m > min_d && m < max_d) ? m : 0.0f;
}

__global__ __launch_bounds__(256) void postprocess_kernel(const float* __restrict__ x, long long x_bs,
                                                         const float* __restrict__ ang,
                                                         float* __restrict__ y, int HW, int fmt,
                                                         float min_d, float max_d, float l2m) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= HW) return;
    const float* xb = x + b * x_bs;
    const float d = (xb[i] + 1.0f) / 2.0f, r = (xb[HW + i] + 1.0f) / 2.0f;
    const float m = revert(d, fmt, min_d, max_d, l2m);
    const float phi = ang[i], th = ang[HW + i];
    const float cp = cosf(phi), sp = sinf(phi), ct = cosf(th), st = sinf(th);
    const float keep = (m > min_d && m < max_d) ? 1.0f : 0.0f;
    float* yb = y + (long long)b * 5 * HW;
    yb[i] = m;
    yb[HW + i] = m * cp * ct * keep;
    yb[2 * HW + i] = m * cp * st * keep;
    yb[3 * HW + i] = m * sp * keep;
    yb[4 * HW + i] = r;
}

__global__ __launch_bounds__(256) void condition_kernel(const float* __restrict__ cm, long long cm_bs,
                                                       float* __restrict__ y, long long y_bs, int HW,
                                                       int ncls, int fmt, float min_d, float max_d,
                                                       float inv_l2m) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= HW) return;
    const float* c = cm + b * cm_bs;
    const long long cls = (long long)c[i];              // .long(): truncation
    const float d = c[HW + i];
    float* yb = y + b * y_bs;
    for (int k = 0; k < ncls; ++k) yb[(long long)k * HW + i] = (cls == k) ? 1.0f : 0.0f;
    float n;
    if (fmt == 0) n = log2f(d + 1.0f) * inv_l2m;
    else if (fmt == 1) n = min_d / (d + 1e-8f);
    else n = d / max_d;
    n = fminf(fmaxf(n, 0.0f), 1.0f);
    yb[(long long)ncls * HW + i] = (d > min_d && d < max_d) ? n : 0.0f;
}

}  // namespace

extern "C" int lc_range_postprocess(const float* sample, int64_t s_bs, const float* ray_angles,
                                    float* out, int B, int H, int W, int depth_format,
                                    float min_depth, float max_depth, lc_stream_t s) {
    if (!sample || !ray_angles || !out || B <= 0 || H <= 0 || W <= 0 || depth_format < 0 ||
        depth_format > 2)
        return LC_EINVAL;
    const int HW = H * W;
    hipLaunchKernelGGL(postprocess_kernel, dim3((HW + 255) / 256, B), dim3(256), 0, lc_s(s), sample,
                       (long long)s_bs, ray_angles, out, HW, depth_format, min_depth, max_depth,
                       (float)log2((double)max_depth + 1.0));
    return lc_launch_status();
}

extern "C" int lc_condition_preprocess(const float* condition_mask, int64_t cm_bs, float* out,
                                       int64_t out_bs, int B, int H, int W, int num_classes,
                                       int depth_format, float min_depth, float max_depth,
                                       lc_stream_t s) {
    if (!condition_mask || !out || B <= 0 || H <= 0 || W <= 0 || num_classes <= 0 ||
        depth_format < 0 || depth_format > 2)
        return LC_EINVAL;
    const int HW = H * W;
    hipLaunchKernelGGL(condition_kernel, dim3((HW + 255) / 256, B), dim3(256), 0, lc_s(s),
                       condition_mask, (long long)cm_bs, out, (long long)out_bs, HW, num_classes,
                       depth_format, min_depth, max_depth,
                       (float)(1.0 / log2((double)max_depth + 1.0)));
    return lc_launch_status();
}

LC_TOUCH_TU(lidar, postprocess_kernel)
