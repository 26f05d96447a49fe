// Small kernels around the denoiser: dense layers of the time/AdaGN MLPs, sinusoid embedding,
// the fused reverse-diffusion update, strided copy and residual add.  All HBM/latency bound.
#include "common.h"

namespace {

// y[m,n] = act_out( sum_k act_in(x[m,k]) w[n,k] + b[n] ); one wave per output column n, lanes
// split K (coalesced rows of w), all M rows of x handled by the same wave so w is read once.
template <int MT>
__global__ __launch_bounds__(256) void linear_kernel(const float* __restrict__ x,
                                                    const float* __restrict__ w,
                                                    const float* __restrict__ b,
                                                    float* __restrict__ y, int M, int K, int N,
                                                    int act_in, int act_out) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int m0 = blockIdx.y * MT;
    if (n >= N) return;
    float acc[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) acc[i] = 0.f;
    const float* wr = w + (long long)n * K;
    for (int k = lane; k < K; k += 64) {
        const float wv = wr[k];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            if (m0 + i < M) {
                float xv = x[(long long)(m0 + i) * K + k];
                if (act_in) xv = lc_silu(xv);
                acc[i] = fmaf(xv, wv, acc[i]);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        float v = lc_wave_sum(acc[i]);
        if (lane == 0 && m0 + i < M) {
            if (b) v += b[n];
            if (act_out) v = lc_silu(v);
            y[(long long)(m0 + i) * N + n] = v;
        }
    }
}

__global__ void sinusoid_kernel(const float* __restrict__ t, float* __restrict__ y, int M,
                                int channels, float max_period) {
    const int half = channels / 2;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= M * half) return;
    const int m = e / half, k = e - m * half;
    // ops.py:22-25: h = exp(-ln(max_period)/(half-1) * k); arg = t*h; cat[sin, cos]
    const float h = expf((-logf(max_period) / (float)(half - 1)) * (float)k);
    const float a = t[m] * h;
    y[(long long)m * channels + k] = sinf(a);
    y[(long long)m * channels + half + k] = cosf(a);
}

// Point Condition Network epilogue (point_unet.py:22-26): per (sample, channel) row of N points
//   y = act( x * sigmoid(gate_logit[b,c]) + bias[b,c] ) [+ res]
// x = fea_layer(fea) in channel-major layout [B, C, N] (the 1x1-conv kernel's output).
__global__ __launch_bounds__(256) void gate_bias_kernel(const float* __restrict__ x, long long x_bs,
                                                       const float* __restrict__ gate_logit,
                                                       const float* __restrict__ bias,
                                                       long long gb_bs,
                                                       const float* __restrict__ res,
                                                       long long res_bs, float* __restrict__ y,
                                                       long long y_bs, int C, int N, int act) {
    const int c = blockIdx.x, b = blockIdx.y;
    const float g = 1.0f / (1.0f + expf(-gate_logit[b * gb_bs + c]));
    const float s = bias[b * gb_bs + c];
    const float* xr = x + b * x_bs + (long long)c * N;
    const float* rr = res ? res + b * res_bs + (long long)c * N : nullptr;
    float* yr = y + b * y_bs + (long long)c * N;
    for (int n = threadIdx.x; n < N; n += 256) {
        float v = xr[n] * g + s;
        if (act == 1) v = v > 0.f ? v : 0.01f * v;          // F.leaky_relu default slope
        if (rr) v += rr[n];
        yr[n] = v;
    }
}

#pragma clang fp contract(off)
// continuous_time.py:209-231; same op order as the reference so fp32 rounding agrees.
__global__ __launch_bounds__(256) void pstep_kernel(
    const float* __restrict__ x_t, long long xt_bs, const float* __restrict__ pred,
    long long pred_bs, const float* __restrict__ noise, long long noise_bs,
    const float* __restrict__ coef, float* __restrict__ x_s, long long xs_bs, long long n,
    int objective, int mode) {
    const int b = blockIdx.y;
    const float* cf = coef + b * 8;
    const float a_t = cf[0], s_t = cf[1], a_s = cf[2], s_s = cf[3], k0 = cf[4], k1 = cf[5];
    const float clip = cf[6];
    (void)s_s;
    const float* xp = x_t + b * xt_bs;
    const float* pp = pred + b * pred_bs;
    const float* np_ = noise ? noise + b * noise_bs : nullptr;
    float* op = x_s + b * xs_bs;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float xt = xp[i], pr = pp[i];
        float x0;
        if (objective == 0) x0 = (xt - s_t * pr) / a_t;
        else if (objective == 1) x0 = a_t * xt - s_t * pr;
        else x0 = pr;
        if (clip > 0.f) x0 = fminf(fmaxf(x0, -clip), clip);
        const float nz = np_ ? np_[i] : 0.f;
        float out;
        if (mode >= 2) {
            // discrete-time update (discrete_time.py:126-180), coefficient row
            //   {A, Bc, c2, c3, c4, c5, clip, c7}: x0 = A xt - Bc pred (eps / v objectives) or pred;
            //   mode 2 (ddpm): (c2 x0 + c3 xt) + c4 nz;  mode 3 (ddim): c4 x0 + c5 (xt - c2 x0) / c3 [+ c7 nz]
            if (objective != 2) x0 = a_t * xt - s_t * pr;
            if (clip > 0.f) x0 = fminf(fmaxf(x0, -clip), clip);
            if (mode == 2) {
                const float mean = a_s * x0 + s_s * xt;
                out = mean + k0 * nz;
            } else {
                const float eps = (xt - a_s * x0) / s_s;
                out = k0 * x0 + k1 * eps;
                if (np_) out = out + cf[7] * nz;
            }
            op[i] = out;
            continue;
        }
        if (mode == 0) {  // ddpm: k0 = c, k1 = sigma_s*sqrt(c)
            const float mean = a_s * (xt * (1.f - k0) / a_t + k0 * x0);
            out = mean + k1 * nz;
        } else {          // ddim: k0 = c1, k1 = c2
            const float eps = (xt - a_t * x0) / s_t;
            out = a_s * x0 + k0 * nz + k1 * eps;
        }
        op[i] = out;
    }
}

__global__ __launch_bounds__(256) void copy_kernel(const float* __restrict__ x, long long x_bs,
                                                  float* __restrict__ y, long long y_bs,
                                                  long long n) {
    const int b = blockIdx.y;
    const float* xp = x + b * x_bs;
    float* yp = y + b * y_bs;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        yp[i] = xp[i];
}

__global__ __launch_bounds__(256) void add_scale_kernel(const float* __restrict__ a, long long a_bs,
                                                       const float* __restrict__ bb, long long b_bs,
                                                       float* __restrict__ y, long long y_bs,
                                                       long long n, float scale) {
    const int b = blockIdx.y;
    const float* ap = a + b * a_bs;
    const float* bp = bb + b * b_bs;
    float* yp = y + b * y_bs;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        yp[i] = (ap[i] + bp[i]) * scale;
}

// ---- box calibration (bench.py: `box_calibration`) ---------------------------------------------------------------
// A bare v_mfma_f32_32x32x16_f16 loop on operands the CALLER supplies (random fp16: the chip clocks to its power budget,
// and what a matrix pipe sustains depends on the operand bits -- profiles/r01_e_ubench_mfma.txt: 2.24 PFLOP/s on zeros,
// 1.69 on random data), 4 independent accumulators per wave, 8 waves per block; and a streaming float4 copy.
typedef _Float16 cal_half8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(512) void calib_mfma_kernel(const cal_half8* __restrict__ ops, int iters, float* __restrict__ sink) {
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const int t = blockIdx.x * 512 + threadIdx.x;
    cal_half8 a[2], b[2];
    a[0] = ops[4 * t]; a[1] = ops[4 * t + 1]; b[0] = ops[4 * t + 2]; b[1] = ops[4 * t + 3];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i & 1], b[(i >> 1) ^ (rep & 1)], acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    sink[t] = s;
}
__global__ __launch_bounds__(256) void calib_copy_kernel(const f32x4* __restrict__ src, f32x4* __restrict__ dst, long long n4) {
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += (long long)gridDim.x * 256)
        dst[i] = __builtin_nontemporal_load(src + i);
}

inline int grid_for(long long n) {
    long long g = (n + 255) / 256;
    return (int)(g > 4096 ? 4096 : (g < 1 ? 1 : g));
}

}  // namespace

extern "C" int lc_abi_version(void) { return 5; }

extern "C" int64_t lc_calibrate_mfma_f16(const void* operands, int blocks, int iters, float* sink, lc_stream_t s) {
    if (!operands || !sink || blocks <= 0 || iters <= 0) return LC_EINVAL;
    hipLaunchKernelGGL(calib_mfma_kernel, dim3(blocks), dim3(512), 0, lc_s(s), reinterpret_cast<const cal_half8*>(operands),
                       iters, sink);
    const int rc = lc_launch_status();
    if (rc != LC_OK) return rc < 0 ? rc : -rc;
    return 2ll * 32 * 32 * 16 * 16 * (int64_t)iters * 8 * blocks;     // flops of the launch
}

extern "C" int lc_calibrate_stream_copy(const float* src, float* dst, int64_t n, lc_stream_t s) {
    if (!src || !dst || n <= 0 || (n & 3) || ((uintptr_t)src & 15) || ((uintptr_t)dst & 15)) return LC_EINVAL;
    hipLaunchKernelGGL(calib_copy_kernel, dim3(8192), dim3(256), 0, lc_s(s), reinterpret_cast<const f32x4*>(src),
                       reinterpret_cast<f32x4*>(dst), (long long)(n / 4));
    return lc_launch_status();
}

// Loads every translation unit's code object for the current device (HIP defers that to the unit's first launch; a
// first sampling step otherwise pays ~15 loads).  Idempotent; lidarcrafter_amd.ops.prepare_model calls it.
extern "C" {
int lc_touch_attention();
int lc_touch_attention_bwd();
int lc_touch_attention_bwd_h();
int lc_touch_conv();
int lc_touch_conv_bwd();
int lc_touch_conv_f16x2();
int lc_touch_conv_f16x2_tall();
int lc_touch_conv_f16x2_s2();
int lc_touch_geometry();
int lc_touch_layout();
int lc_touch_lidar();
int lc_touch_metrics();
int lc_touch_norm();
int lc_touch_resample();
int lc_touch_roipool();
int lc_touch_temporal();
int lc_touch_upfold();
int lc_touch_voxel();
}
extern "C" int lc_load_code_objects(void) {
    int (*const touch[])() = {lc_touch_attention, lc_touch_attention_bwd, lc_touch_attention_bwd_h, lc_touch_conv, lc_touch_conv_bwd, lc_touch_conv_f16x2, lc_touch_conv_f16x2_tall, lc_touch_conv_f16x2_s2, lc_touch_geometry, lc_touch_layout, lc_touch_lidar, lc_touch_metrics, lc_touch_norm, lc_touch_resample, lc_touch_roipool, lc_touch_temporal, lc_touch_upfold, lc_touch_voxel};
    for (auto f : touch) {
        const int rc = f();
        if (rc != 0) return rc;
    }
    hipFuncAttributes fa;
    return (int)hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&linear_kernel<8>));
}

extern "C" int lc_device_arch(char* buf, int buflen) {
    if (!buf || buflen <= 0) return LC_EINVAL;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    hipDeviceProp_t p;
    e = hipGetDeviceProperties(&p, dev);
    if (e != hipSuccess) return (int)e;
    int i = 0;
    for (; i < buflen - 1 && p.gcnArchName[i]; ++i) buf[i] = p.gcnArchName[i];
    buf[i] = 0;
    return LC_OK;
}

extern "C" int lc_linear_fwd(const float* x, const float* w, const float* b, float* y, int M, int K,
                             int N, int act_in, int act_out, lc_stream_t s) {
    if (!x || !w || !y || M <= 0 || K <= 0 || N <= 0) return LC_EINVAL;
    constexpr int MT = 8;
    dim3 grid((N + 3) / 4, (M + MT - 1) / MT);
    hipLaunchKernelGGL(linear_kernel<MT>, grid, dim3(256), 0, lc_s(s), x, w, b, y, M, K, N, act_in,
                       act_out);
    return lc_launch_status();
}

extern "C" int lc_sinusoid_fwd(const float* t, float* y, int M, int channels, float max_period,
                               lc_stream_t s) {
    if (!t || !y || M <= 0 || channels < 4 || (channels & 1)) return LC_EINVAL;
    const int n = M * (channels / 2);
    hipLaunchKernelGGL(sinusoid_kernel, dim3((n + 255) / 256), dim3(256), 0, lc_s(s), t, y, M,
                       channels, max_period);
    return lc_launch_status();
}

extern "C" int lc_pstep_fwd(const float* x_t, int64_t xt_bs, const float* pred, int64_t pred_bs,
                            const float* noise, int64_t noise_bs, const float* coef, float* x_s,
                            int64_t xs_bs, int B, int64_t n, int objective, int mode,
                            lc_stream_t s) {
    if (!x_t || !pred || !coef || !x_s || B <= 0 || n <= 0) return LC_EINVAL;
    if (objective < 0 || objective > 2 || mode < 0 || mode > 3) return LC_EINVAL;
    hipLaunchKernelGGL(pstep_kernel, dim3(grid_for(n), B), dim3(256), 0, lc_s(s), x_t,
                       (long long)xt_bs, pred, (long long)pred_bs, noise, (long long)noise_bs, coef,
                       x_s, (long long)xs_bs, (long long)n, objective, mode);
    return lc_launch_status();
}

extern "C" int lc_copy_strided(const float* x, int64_t x_bs, float* y, int64_t y_bs, int B,
                               int64_t n, lc_stream_t s) {
    if (!x || !y || B <= 0 || n <= 0) return LC_EINVAL;
    hipLaunchKernelGGL(copy_kernel, dim3(grid_for(n), B), dim3(256), 0, lc_s(s), x, (long long)x_bs,
                       y, (long long)y_bs, (long long)n);
    return lc_launch_status();
}

extern "C" int lc_add_scale(const float* a, int64_t a_bs, const float* b, int64_t b_bs, float* y,
                            int64_t y_bs, int B, int64_t n, float scale, lc_stream_t s) {
    if (!a || !b || !y || B <= 0 || n <= 0) return LC_EINVAL;
    hipLaunchKernelGGL(add_scale_kernel, dim3(grid_for(n), B), dim3(256), 0, lc_s(s), a,
                       (long long)a_bs, b, (long long)b_bs, y, (long long)y_bs, (long long)n, scale);
    return lc_launch_status();
}

extern "C" int lc_gate_bias_act(const float* x, int64_t x_bs, const float* gate_logit,
                                const float* bias, int64_t gb_bs, const float* res, int64_t res_bs,
                                float* y, int64_t y_bs, int B, int C, int N, int act, lc_stream_t s) {
    if (!x || !gate_logit || !bias || !y || B <= 0 || C <= 0 || N <= 0 || act < 0 || act > 1)
        return LC_EINVAL;
    hipLaunchKernelGGL(gate_bias_kernel, dim3(C, B), dim3(256), 0, lc_s(s), x, (long long)x_bs,
                       gate_logit, bias, (long long)gb_bs, res, (long long)res_bs, y,
                       (long long)y_bs, C, N, act);
    return lc_launch_status();
}
