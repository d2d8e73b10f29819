// The steps on either side of the SP / AT hot path, as small HBM-bound kernels:
//   egz_u8_normalize     data/STdatas.py:50-68, data/lateDataset.py:22-33: uint8 planes -> (u8 / 255 - mean[c]) / std[c] fp32
//                        (image: ImageNet mean/std on the BGR-ordered channels; flow: 0.5 / 0.5; gt: 0 / 1).  The batch
//                        crosses PCIe as bytes (a quarter of the fp32 size) and is normalised where it is consumed.
//                        Same three correctly-rounded fp32 operations as the reference's torch expression: bit-exact.
//   egz_crop_mean        AT.crop_feature (AT.py:25-39) + the spatial mean that makes chn_weight (AT.py:229,
//                        extractLSTMw.py:81-90): size x size window of the 14 x 14 map around gaze_point // 16.
//   egz_weighted_minmax  AT.get_weighted (AT.py:58-66): channel-weighted sum of the map, min-max normalised.
#include "egz_common.h"

namespace {

__global__ void u8_normalize_kernel(const unsigned char* __restrict__ src, float* __restrict__ dst, long n, long plane,
                                    int C, const float* __restrict__ mean, const float* __restrict__ stdv) {
    // 4 bytes in, one float4 out per thread; a group of 4 never straddles a plane (plane % 4 == 0 checked by the caller)
    const long n4 = n >> 2;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const unsigned int u = reinterpret_cast<const unsigned int*>(src)[i];
        const int c = (int)(((i << 2) / plane) % C);
        const float m = mean[c], s = stdv[c];
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = ((float)((u >> (8 * e)) & 255u) / 255.f - m) / s;
        reinterpret_cast<f32x4*>(dst)[i] = o;
    }
}

// one thread per (b, c): mean of the size x size window; feat is the channels-last (B, H, W, C) image of the NCHW tensor
__global__ void crop_mean_kernel(const float* __restrict__ feat, const int* __restrict__ gp, float* __restrict__ out,
                                 int B, int H, int W, int C, int size, int cell) {
    const long idx = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (idx >= (long)B * C) return;
    const int b = (int)(idx / C), c = (int)(idx % C);
    const int lo = size / 2, hi = size - lo;                  // hi = ceil(size / 2)
    int fy = gp[2 * b] / cell, fx = gp[2 * b + 1] / cell;
    fy = min(max(fy, lo), H - hi);
    fx = min(max(fx, lo), W - hi);
    float s = 0.f;
    for (int y = fy - lo; y < fy + hi; ++y)
        for (int x = fx - lo; x < fx + hi; ++x) s += feat[(((long)b * H + y) * W + x) * C + c];
    out[idx] = s / (float)(size * size);
}

// one thread per (b, c): mean of the explicit window rows [y0, y1) x columns [x0, x1) of sample b (win: (B, 4) int32)
__global__ void window_mean_kernel(const float* __restrict__ feat, const int* __restrict__ win, float* __restrict__ out,
                                   int B, int H, int W, int C) {
    const long idx = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (idx >= (long)B * C) return;
    const int b = (int)(idx / C), c = (int)(idx % C);
    const int y0 = win[4 * b], y1 = win[4 * b + 1], x0 = win[4 * b + 2], x1 = win[4 * b + 3];
    float s = 0.f;
    for (int y = y0; y < y1; ++y)
        for (int x = x0; x < x1; ++x) s += feat[(((long)b * H + y) * W + x) * C + c];
    out[idx] = s / (float)((y1 - y0) * (x1 - x0));
}

// one thread per (b, c): out[b][c] = sum_p wmap[b][p] * feat[b][p][c]  (a linear functional of the map, e.g. the mean of a
// window of its bilinear x16 upsampling: AT.crop_align_feature + mean, AT.py:41-56,229)
__global__ void pixel_weighted_sum_kernel(const float* __restrict__ feat, const float* __restrict__ wmap,
                                          float* __restrict__ out, int B, int HW, int C) {
    const long idx = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (idx >= (long)B * C) return;
    const int b = (int)(idx / C), c = (int)(idx % C);
    float s = 0.f;
    for (int p = 0; p < HW; ++p) s += wmap[(long)b * HW + p] * feat[((long)b * HW + p) * C + c];
    out[idx] = s;
}

// one block per image: out[p] = sum_c feat[p][c] * w[c]; then (out - min) / max(out - min)
__global__ __launch_bounds__(256) void weighted_minmax_kernel(const float* __restrict__ feat, const float* __restrict__ w,
                                                               float* __restrict__ out, int HW, int C) {
    extern __shared__ float sm[];          // [HW] sums, then [256] x 2 for min / max
    float* vals = sm;
    float* red = sm + HW;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* f = feat + (long)b * HW * C;
    for (int p = wave; p < HW; p += 4) {   // one wave per pixel: lanes stride over the channels (coalesced), shuffle reduce
        float s = 0.f;
        for (int c = lane; c < C; c += 64) s += f[(long)p * C + c] * w[(long)b * C + c];
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
        if (lane == 0) vals[p] = s;
    }
    __syncthreads();
    float mn = INFINITY, mx = -INFINITY;
    for (int p = tid; p < HW; p += 256) { mn = fminf(mn, vals[p]); mx = fmaxf(mx, vals[p]); }
    red[tid] = mn; red[256 + tid] = mx;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (tid < st) { red[tid] = fminf(red[tid], red[tid + st]); red[256 + tid] = fmaxf(red[256 + tid], red[256 + tid + st]); }
        __syncthreads();
    }
    mn = red[0];
    const float den = red[256] - mn;       // max of (x - min)
    for (int p = tid; p < HW; p += 256) out[(long)b * HW + p] = (vals[p] - mn) / den;
}

}  // namespace

// src: n bytes laid out [...][C][plane]; dst: n floats, same order.  mean / std: C floats on the device.
EGZ_API int egz_u8_normalize(const unsigned char* src, float* dst, long n, long plane, int C, const float* mean,
                             const float* stdv, hipStream_t st) {
    EGZ_CHECK_ARG(src && dst && mean && stdv && n > 0 && C > 0, "egz_u8_normalize: bad arguments");
    EGZ_CHECK_ARG(plane % 4 == 0 && n % plane == 0, "egz_u8_normalize: plane size %ld must be a multiple of 4 and divide n", plane);
    const long n4 = n >> 2;
    const int g = egz_cdiv(n4, 256) > 8192 ? 8192 : egz_cdiv(n4, 256);
    hipLaunchKernelGGL(u8_normalize_kernel, dim3(g), dim3(256), 0, st, src, dst, n, plane, C, mean, stdv);
    EGZ_CHECK_LAUNCH("egz_u8_normalize");
    return 0;
}

// feat: (B, H, W, C) channels-last fp32; gp: (B, 2) int32 gaze points in input pixels (row, col); cell = input pixels per
// feature cell (16); out: (B, C) = mean over the clipped size x size window around gp / cell.
EGZ_API int egz_crop_mean(const float* feat, const int* gp, float* out, int B, int H, int W, int C, int size, int cell,
                          hipStream_t st) {
    EGZ_CHECK_ARG(feat && gp && out && B > 0 && C > 0, "egz_crop_mean: bad arguments");
    EGZ_CHECK_ARG(size > 0 && size <= H && size <= W && cell > 0, "egz_crop_mean: window %d does not fit a %d x %d map", size, H, W);
    hipLaunchKernelGGL(crop_mean_kernel, dim3(egz_cdiv((long)B * C, 256)), dim3(256), 0, st, feat, gp, out, B, H, W, C, size, cell);
    EGZ_CHECK_LAUNCH("egz_crop_mean");
    return 0;
}

// feat: (B, H, W, C) channels-last fp32; win: (B, 4) int32 {y0, y1, x0, x1} (host-validated: 0 <= y0 < y1 <= H, same for
// x); out: (B, C) = mean over the window.  Serves extractLSTMw.crop_feature_var (extractLSTMw.py:46-58), whose window
// comes from a float clip and int() truncation and is therefore not always size x size.
EGZ_API int egz_window_mean(const float* feat, const int* win, float* out, int B, int H, int W, int C, hipStream_t st) {
    EGZ_CHECK_ARG(feat && win && out && B > 0 && C > 0 && H > 0 && W > 0, "egz_window_mean: bad arguments");
    hipLaunchKernelGGL(window_mean_kernel, dim3(egz_cdiv((long)B * C, 256)), dim3(256), 0, st, feat, win, out, B, H, W, C);
    EGZ_CHECK_LAUNCH("egz_window_mean");
    return 0;
}

// feat: (B, H*W, C) channels-last, wmap: (B, H*W) per-pixel weights, out: (B, C) = sum_p wmap[b][p] * feat[b][p][:].
EGZ_API int egz_pixel_weighted_sum(const float* feat, const float* wmap, float* out, int B, int HW, int C, hipStream_t st) {
    EGZ_CHECK_ARG(feat && wmap && out && B > 0 && HW > 0 && C > 0, "egz_pixel_weighted_sum: bad arguments");
    hipLaunchKernelGGL(pixel_weighted_sum_kernel, dim3(egz_cdiv((long)B * C, 256)), dim3(256), 0, st, feat, wmap, out, B, HW, C);
    EGZ_CHECK_LAUNCH("egz_pixel_weighted_sum");
    return 0;
}

// feat: (B, H*W, C) channels-last, w: (B, C), out: (B, H*W) min-max normalised channel-weighted sums.
EGZ_API int egz_weighted_minmax(const float* feat, const float* w, float* out, int B, int HW, int C, hipStream_t st) {
    EGZ_CHECK_ARG(feat && w && out && B > 0 && HW > 0 && C > 0, "egz_weighted_minmax: bad arguments");
    EGZ_CHECK_ARG(HW <= 4096, "egz_weighted_minmax: map of %d pixels does not fit the LDS stage", HW);
    hipLaunchKernelGGL(weighted_minmax_kernel, dim3(B), dim3(256), (size_t)(HW + 512) * sizeof(float), st, feat, w, out, HW, C);
    EGZ_CHECK_LAUNCH("egz_weighted_minmax");
    return 0;
}
