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


// one block per image: the reference's host glue  `im = (out * 255).astype(uint8); ndimage.center_of_mass(im)`
// (run_spatialstream.py:99-104,130-131) on the device.  q = (unsigned char)(v * 255.f) is numpy's truncating cast of the fp32
// product; scipy forms sum(q * row) / sum(q) in float64 from integer-valued terms, i.e. the correctly rounded quotient of two
// exact integers -- reproduced bit for bit with 64-bit integer sums and ONE double division.  gp = floor(com) (what
// `np.array(predicted) // 16` then divides); an all-zero image (0 / 0 in the reference) yields com = NaN, gp = 0.
__global__ __launch_bounds__(256) void u8_center_of_mass_kernel(const float* __restrict__ map, int H, int W,
                                                                double* __restrict__ com, int* __restrict__ gp,
                                                                unsigned char* __restrict__ q8) {
    __shared__ unsigned long long red[3][4];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* m = map + (long)b * H * W;
    unsigned long long s0 = 0, sy = 0, sx = 0;
    for (int p = tid; p < H * W; p += 256) {
        const unsigned int q = (unsigned int)(unsigned char)(m[p] * 255.f);
        if (q8) q8[(long)b * H * W + p] = (unsigned char)q;
        const int y = p / W, x = p - y * W;
        s0 += q;
        sy += (unsigned long long)q * (unsigned)y;
        sx += (unsigned long long)q * (unsigned)x;
    }
    for (int off = 32; off > 0; off >>= 1) {
        s0 += __shfl_xor(s0, off);
        sy += __shfl_xor(sy, off);
        sx += __shfl_xor(sx, off);
    }
    if ((tid & 63) == 0) { red[0][tid >> 6] = s0; red[1][tid >> 6] = sy; red[2][tid >> 6] = sx; }
    __syncthreads();
    if (tid == 0) {
        const unsigned long long t0 = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        const unsigned long long ty = red[1][0] + red[1][1] + red[1][2] + red[1][3];
        const unsigned long long tx = red[2][0] + red[2][1] + red[2][2] + red[2][3];
        const double cy = t0 ? (double)ty / (double)t0 : NAN, cx = t0 ? (double)tx / (double)t0 : NAN;
        com[2 * b] = cy;
        com[2 * b + 1] = cx;
        gp[2 * b] = t0 ? (int)floor(cy) : 0;
        gp[2 * b + 1] = t0 ? (int)floor(cx) : 0;
    }
}

// nn.functional.interpolate(x, scale_factor=s, mode='bilinear') of (B, h, w) maps -- torch's upsample_bilinear2d arithmetic in
// fp32: source index r * (dst + 0.5) - 0.5 clamped at 0 with r = 1 / s (align_corners=False, run_spatialstream.py:136) or
// r * dst with r = (h - 1) / (H - 1) (align_corners=True, nn.functional.upsample_bilinear: AT.py:47, extractLSTMw.py:33);
// value = l0y (l0x v00 + l1x v01) + l1y (l0x v10 + l1x v11).  dst rows of sample b start at dst + b * dst_bstride (so the
// result can land in one plane of a wider tensor, e.g. channel 1 of late_fusion's (B, 2, H, W) input).
__global__ __launch_bounds__(256) void bilinear_up_kernel(const float* __restrict__ src, float* __restrict__ dst, int B, int h,
                                                          int w, int H, int W, float rh, float rw, int align, long dst_bstride) {
    const long n = (long)B * H * W;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int x = (int)(i % W);
        const long t = i / W;
        const int y = (int)(t % H), b = (int)(t / H);
        float sy = align ? rh * (float)y : rh * ((float)y + 0.5f) - 0.5f;
        float sx = align ? rw * (float)x : rw * ((float)x + 0.5f) - 0.5f;
        if (!align) { sy = sy < 0.f ? 0.f : sy; sx = sx < 0.f ? 0.f : sx; }
        const int y0 = (int)sy, x0 = (int)sx;
        const int yp = (y0 < h - 1) ? 1 : 0, xp = (x0 < w - 1) ? 1 : 0;
        const float ly1 = sy - (float)y0, ly0 = 1.f - ly1, lx1 = sx - (float)x0, lx0 = 1.f - lx1;
        const float* s = src + ((long)b * h + y0) * w + x0;
        dst[(long)b * dst_bstride + (long)y * W + x] =
            ly0 * (lx0 * s[0] + lx1 * s[xp]) + ly1 * (lx0 * s[(long)yp * w] + lx1 * s[(long)yp * w + xp]);
    }
}

// torch.cat((f, g), dim=1) of two one-channel maps (models/late_fusion.py:19): out[b][0] = f[b], out[b][1] = g[b]
__global__ __launch_bounds__(256) void cat2_planes_kernel(const f32x4* __restrict__ f, const f32x4* __restrict__ g,
                                                          f32x4* __restrict__ out, long hw4, long n4) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const long b = i / hw4, r = i - b * hw4;
        out[2 * b * hw4 + r] = f[i];
        out[(2 * b + 1) * hw4 + r] = g[i];
    }
}
// (planes whose size is not a multiple of 4 floats, or operands that are not 16-byte aligned: one float per thread item)
__global__ __launch_bounds__(256) void cat2_planes_scalar_kernel(const float* __restrict__ f, const float* __restrict__ g,
                                                                 float* __restrict__ out, long hw, long n) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const long b = i / hw, r = i - b * hw;
        out[2 * b * hw + r] = f[i];
        out[(2 * b + 1) * hw + r] = g[i];
    }
}
}  // namespace

// f, g: [B][1][H][W] -> out [B][2][H][W]: the late-fusion stack's input (late_fusion.py:19).  16-byte copies when HW % 4 == 0
// and all three pointers are 16-byte aligned, 4-byte copies otherwise.
EGZ_API int egz_cat2_planes(const float* f, const float* g, float* out, int B, long HW, hipStream_t st) {
    EGZ_CHECK_ARG(f && g && out && B > 0 && HW > 0, "egz_cat2_planes: bad arguments");
    if (HW % 4 != 0 || ((uintptr_t)f | (uintptr_t)g | (uintptr_t)out) % 16 != 0) {
        const long n = (long)B * HW;
        const int grid = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
        hipLaunchKernelGGL(cat2_planes_scalar_kernel, dim3(grid), dim3(256), 0, st, f, g, out, HW, n);
        EGZ_CHECK_LAUNCH("egz_cat2_planes");
        return 0;
    }
    const long n4 = (long)B * HW / 4;
    const int grid = (int)((n4 + 255) / 256 > 4096 ? 4096 : (n4 + 255) / 256);
    hipLaunchKernelGGL(cat2_planes_kernel, dim3(grid), dim3(256), 0, st, reinterpret_cast<const f32x4*>(f),
                       reinterpret_cast<const f32x4*>(g), reinterpret_cast<f32x4*>(out), HW / 4, n4);
    EGZ_CHECK_LAUNCH("egz_cat2_planes");
    return 0;
}

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

// map: (B, H, W) fp32 in [0, 1]; com: (B, 2) doubles (row, col) = scipy.ndimage.center_of_mass of (map * 255).astype(uint8);
// gp: (B, 2) int32 = floor(com) (feeds egz_crop_mean); q8 (optional): the (B, H, W) uint8 image itself.
EGZ_API int egz_u8_center_of_mass(const float* map, int B, int H, int W, double* com, int* gp, unsigned char* q8,
                                  hipStream_t st) {
    EGZ_CHECK_ARG(map && com && gp && B > 0 && H > 0 && W > 0 && (long)H * W < (1L << 24), "egz_u8_center_of_mass: bad arguments");
    hipLaunchKernelGGL(u8_center_of_mass_kernel, dim3(B), dim3(256), 0, st, map, H, W, com, gp, q8);
    EGZ_CHECK_LAUNCH("egz_u8_center_of_mass");
    return 0;
}

// src: (B, h, w) fp32 -> dst: B maps of (h * scale) x (w * scale), sample b at dst + b * dst_bstride floats.
EGZ_API int egz_bilinear_up(const float* src, float* dst, int B, int h, int w, int scale, int align_corners, long dst_bstride,
                            hipStream_t st) {
    EGZ_CHECK_ARG(src && dst && B > 0 && h > 0 && w > 0 && scale >= 1 && dst_bstride >= (long)h * w * scale * scale,
                  "egz_bilinear_up: bad arguments");
    const int H = h * scale, W = w * scale;
    const float rh = align_corners ? (H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f) : 1.f / (float)scale;
    const float rw = align_corners ? (W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f) : 1.f / (float)scale;
    const long n = (long)B * H * W;
    const int g = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    hipLaunchKernelGGL(bilinear_up_kernel, dim3(g), dim3(256), 0, st, src, dst, B, h, w, H, W, rh, rw, align_corners ? 1 : 0, dst_bstride);
    EGZ_CHECK_LAUNCH("egz_bilinear_up");
    return 0;
}
