// RoI-aware voxel pooling of point features (the "voxel scatter" of lidargen/ops):
//   /root/reference/lidargen/ops/roiaware_pool3d/src/roiaware_pool3d_kernel.cu
//     generate_pts_mask_for_box3d :39-75, collect_inside_pts_for_box3d :78-108,
//     roiaware_maxpool3d :111-157, roiaware_avgpool3d :160-190, backward :236-284.
// Same results as the reference (same voxel encoding, first max_pts-1 points per voxel in point
// order, strict-> max with the lowest index winning, avg summed in point order), different
// parallelisation of the collect step: the reference runs ONE THREAD per box over all points in
// global memory; here one wave per box walks the points 64 at a time and serialises only over the
// (few) lanes whose point is inside the box, with the per-voxel counters of the box in LDS.
#include "common.h"

namespace {

#pragma clang fp contract(off)

__device__ __forceinline__ int pt_in_box_local(const float* pt, const float* bx, float& lx,
                                               float& ly) {
    const float MARGIN = 1e-5f;
    const float x = pt[0], y = pt[1], z = pt[2];
    const float cx = bx[0], cy = bx[1], cz = bx[2], dx = bx[3], dy = bx[4], dz = bx[5], rz = bx[6];
    if ((double)fabsf(z - cz) > (double)dz / 2.0) return 0;
    const float cosa = (float)cos((double)(-rz)), sina = (float)sin((double)(-rz));
    const float sx = x - cx, sy = y - cy;
    lx = sx * cosa + sy * (-sina);
    ly = sx * sina + sy * cosa;
    return ((double)fabsf(lx) < (double)dx / 2.0 + (double)MARGIN) &
           ((double)fabsf(ly) < (double)dy / 2.0 + (double)MARGIN);
}

__global__ __launch_bounds__(256) void roi_mask_kernel(int nb, int np, int ox, int oy, int oz,
                                                      const float* __restrict__ rois,
                                                      const float* __restrict__ pts,
                                                      int* __restrict__ mask) {
    const int p = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (p >= np) return;
    const float* pt = pts + 3ll * p;
    const float* bx = rois + 7ll * b;
    float lx = 0.f, ly = 0.f;
    int enc = -1;
    if (pt_in_box_local(pt, bx, lx, ly)) {
        const float lz = pt[2] - bx[2];
        const float dx = bx[3], dy = bx[4], dz = bx[5];
        const float xr = dx / ox, yr = dy / oy, zr = dz / oz;
        // the reference stores int(...) into UNSIGNED ints and clamps with min(max(u, 0), n-1):
        // a negative int wraps to a huge unsigned and clamps to n-1 (kernel.cu:63-69)
        unsigned xi = (unsigned)(int)((lx + dx / 2) / xr);
        unsigned yi = (unsigned)(int)((ly + dy / 2) / yr);
        unsigned zi = (unsigned)(int)((lz + dz / 2) / zr);
        xi = min(xi, (unsigned)(ox - 1));
        yi = min(yi, (unsigned)(oy - 1));
        zi = min(zi, (unsigned)(oz - 1));
        enc = (int)((xi << 16) + (yi << 8) + zi);
    }
    mask[(long long)b * np + p] = enc;
}

// one wave per box; dynamic LDS = ox*oy*oz counters
__global__ __launch_bounds__(64) void roi_collect_kernel(int np, int cap, int ox, int oy, int oz,
                                                        const int* __restrict__ mask,
                                                        int* __restrict__ vox) {
    extern __shared__ int cnt[];
    const int b = blockIdx.x, lane = threadIdx.x;
    const int nv = ox * oy * oz;
    for (int i = lane; i < nv; i += 64) cnt[i] = 0;
    __syncthreads();
    int* vb = vox + (long long)b * nv * cap;
    const int* mb = mask + (long long)b * np;
    for (int base = 0; base < np; base += 64) {
        const int k = base + lane;
        const int enc = k < np ? mb[k] : -1;
        unsigned long long m = __ballot(enc != -1);
        while (m) {                                     // ascending lane == ascending point index
            const int l = __ffsll((long long)m) - 1;
            m &= m - 1;
            if (lane == l) {
                const unsigned e = (unsigned)enc;
                const int v = ((e >> 16) & 0xFF) * oy * oz + ((e >> 8) & 0xFF) * oz + (e & 0xFF);
                const int c = cnt[v];
                if (c < cap - 1) { vb[(long long)v * cap + c + 1] = k; cnt[v] = c + 1; }
            }
        }
    }
    __syncthreads();
    for (int i = lane; i < nv; i += 64) vb[(long long)i * cap] = cnt[i];
}

__global__ __launch_bounds__(256) void roi_pool_kernel(int nv, int C, int cap,
                                                      const float* __restrict__ feat,
                                                      const int* __restrict__ vox,
                                                      float* __restrict__ pooled,
                                                      int* __restrict__ argmax, int method) {
    const int v = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y, b = blockIdx.z;
    if (v >= nv) return;
    const int* lst = vox + ((long long)b * nv + v) * cap;
    const long long o = ((long long)b * nv + v) * C + c;
    const int total = lst[0];
    if (method == 0) {
        int am = -1;
        float mx = -__builtin_inff();                   // the reference's -1e50 literal is -inf in float
        for (int k = 1; k <= total; ++k) {
            const float f = feat[(long long)lst[k] * C + c];
            if (f > mx) { mx = f; am = lst[k]; }
        }
        if (am != -1) pooled[o] = mx;
        argmax[o] = am;
    } else {
        float s = 0.f;
        for (int k = 1; k <= total; ++k) s += feat[(long long)lst[k] * C + c];
        if (total > 0) pooled[o] = s / total;
    }
}

__global__ __launch_bounds__(256) void roi_pool_bwd_kernel(int nv, int C, int cap,
                                                          const int* __restrict__ vox,
                                                          const int* __restrict__ argmax,
                                                          const float* __restrict__ gout,
                                                          float* __restrict__ gin, int method) {
    const int v = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y, b = blockIdx.z;
    if (v >= nv) return;
    const long long o = ((long long)b * nv + v) * C + c;
    if (method == 0) {
        const int am = argmax[o];
        if (am != -1) atomicAdd(gin + (long long)am * C + c, gout[o]);
    } else {
        const int* lst = vox + ((long long)b * nv + v) * cap;
        const int total = lst[0];
        const float g = gout[o] * (1.0f / fmaxf((float)total, 1.0f));
        for (int k = 1; k <= total; ++k) atomicAdd(gin + (long long)lst[k] * C + c, g);
    }
}

}  // namespace

extern "C" int lc_roiaware_pool3d_fwd(const float* rois, const float* pts, const float* pts_feature,
                                      int n_boxes, int n_pts, int channels, int out_x, int out_y,
                                      int out_z, int max_pts_each_voxel, int pool_method,
                                      int32_t* pts_mask_scratch, int32_t* pts_idx_of_voxels,
                                      int32_t* argmax, float* pooled, lc_stream_t s) {
    if (!rois || !pts || !pts_feature || !pts_mask_scratch || !pts_idx_of_voxels || !pooled ||
        n_boxes <= 0 || n_pts <= 0 || channels <= 0 || max_pts_each_voxel < 2)
        return LC_EINVAL;
    if (out_x <= 0 || out_y <= 0 || out_z <= 0 || out_x >= 256 || out_y >= 256 || out_z >= 256)
        return LC_EINVAL;                              // 8-bit voxel encoding, like the reference
    if (pool_method != 0 && pool_method != 1) return LC_EINVAL;
    if (pool_method == 0 && !argmax) return LC_EINVAL;
    const int nv = out_x * out_y * out_z;
    if ((size_t)nv * sizeof(int) > 150 * 1024) return LC_EUNSUP;
    hipStream_t st = lc_s(s);
    hipLaunchKernelGGL(roi_mask_kernel, dim3((n_pts + 255) / 256, n_boxes), dim3(256), 0, st,
                       n_boxes, n_pts, out_x, out_y, out_z, rois, pts, pts_mask_scratch);
    hipLaunchKernelGGL(roi_collect_kernel, dim3(n_boxes), dim3(64), nv * sizeof(int), st, n_pts,
                       max_pts_each_voxel, out_x, out_y, out_z, pts_mask_scratch,
                       pts_idx_of_voxels);
    hipLaunchKernelGGL(roi_pool_kernel, dim3((nv + 255) / 256, channels, n_boxes), dim3(256), 0, st,
                       nv, channels, max_pts_each_voxel, pts_feature, pts_idx_of_voxels, pooled,
                       argmax, pool_method);
    return lc_launch_status();
}

extern "C" int lc_roiaware_pool3d_bwd(const int32_t* pts_idx_of_voxels, const int32_t* argmax,
                                      const float* grad_out, float* grad_in, int n_boxes,
                                      int channels, int out_x, int out_y, int out_z,
                                      int max_pts_each_voxel, int pool_method, lc_stream_t s) {
    if (!pts_idx_of_voxels || !grad_out || !grad_in || n_boxes <= 0 || channels <= 0)
        return LC_EINVAL;
    if (pool_method == 0 && !argmax) return LC_EINVAL;
    const int nv = out_x * out_y * out_z;
    hipLaunchKernelGGL(roi_pool_bwd_kernel, dim3((nv + 255) / 256, channels, n_boxes), dim3(256), 0,
                       lc_s(s), nv, channels, max_pts_each_voxel, pts_idx_of_voxels, argmax,
                       grad_out, grad_in, pool_method);
    return lc_launch_status();
}

LC_TOUCH_TU(roipool, roi_mask_kernel)
