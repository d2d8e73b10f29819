// HBM-bound streaming passes of the SP/LF path (NHWC fp32, float4 over channels, grid-stride):
//   * train-mode BatchNorm2d statistics finalisation + running-stat update   (utils.py:72, model_SP.py:12,45)
//   * BN-apply + ReLU (+ 2x2/2 max-pool) forward and its backward            (utils.py:68,72)
//   * element-wise max of the two fusion streams (Conv3d k=(1,3,3) + MaxPool3d((2,1,1)), model_SP.py:38-44)
//   * ReLU mask, nearest-x2 upsample backward (2x2 sum), bias gradient (column sums)
// Channel reductions: every block keeps fp64 per-channel partials, writes them to a small workspace and a
// second tiny kernel sums them in a fixed order (deterministic, no atomics).
#include "egz_common.h"
#include "x3_split.h"
#include <cstdlib>

namespace {

constexpr int RED_ROWS = 64;   // row-splits of the generic column reduction (stage A)

// ------------------------------------------------------------------ generic column sums of a [rows][cols] matrix
template <typename T>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const T* __restrict__ in, double* __restrict__ part,
                                                             long rows, int cols) {
    __shared__ double red[8][33];
    const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
    const int col = blockIdx.x * 32 + cx;
    const long per = (rows + gridDim.y - 1) / gridDim.y;
    const long r0 = blockIdx.y * per, r1 = (r0 + per < rows) ? r0 + per : rows;
    double s = 0.0;
    if (col < cols)
        for (long r = r0 + ry; r < r1; r += 8) s += (double)in[r * cols + col];
    red[ry][cx] = s;
    __syncthreads();
    if (ry == 0 && col < cols) {
        double t = 0.0;
#pragma unroll
        for (int i = 0; i < 8; ++i) t += red[i][cx];
        part[(long)blockIdx.y * cols + col] = t;
    }
}

template <typename T>
int colsum_partial(const T* in, double* part, long rows, int cols, hipStream_t st) {
    dim3 grid(egz_cdiv(cols, 32), RED_ROWS);
    hipLaunchKernelGGL(colsum_partial_kernel<T>, grid, dim3(256), 0, st, in, part, rows, cols);
    EGZ_CHECK_LAUNCH("colsum_partial");
    return 0;
}

// ------------------------------------------------------------------ BN forward finalise
// part2: [RED_ROWS][2][K] fp64 (sum, sumsq).  Batch mean / biased var -> invstd; running stats use the
// unbiased variance (torch BatchNorm2d semantics).  scale = gamma*invstd, shift = beta - mean*scale.
__global__ void bn_finalize_kernel(const double* __restrict__ part2, int nparts, int K, double count,
                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                   float* __restrict__ running_mean, float* __restrict__ running_var,
                                   float momentum, float eps, float* __restrict__ mean_out,
                                   float* __restrict__ invstd_out, float* __restrict__ scale,
                                   float* __restrict__ shift, long long* __restrict__ num_batches_tracked,
                                   const unsigned int* __restrict__ minmax, unsigned int* __restrict__ absmax_out) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;                                 // (K % 64 == 0 whenever minmax is given: whole waves stay)
    if (k == 0 && num_batches_tracked) *num_batches_tracked += 1;      // BatchNorm2d bookkeeping, same launch
    double s1 = 0.0, s2 = 0.0;
#pragma unroll 16                                   // independent loads in flight; the sum keeps its fixed order
    for (int p = 0; p < nparts; ++p) {
        s1 += part2[((long)p * 2 + 0) * K + k];
        s2 += part2[((long)p * 2 + 1) * K + k];
    }
    const double mean = s1 / count;
    double var = s2 / count - mean * mean;
    if (var < 0.0) var = 0.0;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    const float g = gamma ? gamma[k] : 1.f, bt = beta ? beta[k] : 0.f;
    const float sc = g * invstd;
    mean_out[k] = (float)mean;
    invstd_out[k] = invstd;
    scale[k] = sc;
    shift[k] = bt - (float)mean * sc;
    if (running_mean) {
        const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
        running_mean[k] = (1.f - momentum) * running_mean[k] + momentum * (float)mean;
        running_var[k] = (1.f - momentum) * running_var[k] + momentum * (float)unbiased;
    }
    if (minmax) {
        // the exact maximum of relu(fma(y, scale, shift)) over this channel from the channel's max / min of y (the map is
        // monotonic in y): the f16 x3 scale of the block output is known BEFORE the pass that writes it runs, so that pass
        // can store pre-split pairs (egz_bn_relu_pool_fwd_presplit).  minmax: order-preserving integer images of max y
        // (slot k) and max -y (slot K + k), conv3x3_igemm_x3s.hip.
        auto from_ordered = [](unsigned int u) -> float {
            return __uint_as_float((u & 0x80000000u) ? (u ^ 0x80000000u) : ~u);
        };
        unsigned int umx = 0, umn = 0;                       // 1024 / (2 K) slot sets of 2 K uints (unused sets hold zeros = lowest)
        for (int sl = 0; sl < (512 / K > 0 ? 512 / K : 1); ++sl) {
            const unsigned int a = minmax[sl * 2 * K + k], b = minmax[sl * 2 * K + K + k];
            umx = a > umx ? a : umx;
            umn = b > umn ? b : umn;
        }
        const float ymax = from_ordered(umx), ymin = -from_ordered(umn);
        float bound = fmaxf(fmaxf(__builtin_fmaf(ymax, sc, shift[k]), __builtin_fmaf(ymin, sc, shift[k])), 0.f);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) bound = fmaxf(bound, __shfl_xor(bound, o));
        if ((threadIdx.x & 63) == 0) absmax_commit(absmax_out, blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), bound);
    }
}

// Finalize of a DEFERRED BatchNorm (K <= 64; late_fusion.py:11-12): the same coefficients, computed by one 256-thread block that
// also sums the partial rows itself (no column-sum launch) and bounds the block output it never materialises:
//   max over channels of relu(y * scale + shift) = max_k max(0, fma(ymax_k, scale_k, shift_k), fma(ymin_k, scale_k, shift_k))
// exactly (the map is monotonic in y per channel), from the per-channel max / min rows the conv epilogue wrote.  That value
// is the f16 split scale source (egz_absmax layout, slot 0) of the convolution that applies the BatchNorm while staging y.
__global__ __launch_bounds__(1024) void bn_finalize_deferred_kernel(
    const double* __restrict__ part, int nparts, int K, double count, const float* __restrict__ gamma,
    const float* __restrict__ beta, float* __restrict__ running_mean, float* __restrict__ running_var, float momentum, float eps,
    float* __restrict__ mean_out, float* __restrict__ invstd_out, float* __restrict__ scale, float* __restrict__ shift,
    long long* __restrict__ num_batches_tracked, const float* __restrict__ mm, int mm_rows, unsigned int* __restrict__ absmax_out) {
    __shared__ double fsum[32][128];
    __shared__ float fmm[32][128];
    __shared__ float bound[64];
    const int cols = 2 * K, ngrp = 1024 / cols;                // K = 32: 16 row groups of 64 columns
    const int col = threadIdx.x % cols, grp = threadIdx.x / cols;
    {
        // four independent chains per thread (rows g, g + ngrp, ...: a one-chain loop ran at one L2 latency per row, 35 us for
        // 256 rows), combined in a fixed order
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        int r = grp;
        for (; r + 3 * ngrp < nparts; r += 4 * ngrp) {
            s0 += part[(long)r * cols + col];
            s1 += part[(long)(r + ngrp) * cols + col];
            s2 += part[(long)(r + 2 * ngrp) * cols + col];
            s3 += part[(long)(r + 3 * ngrp) * cols + col];
        }
        for (; r < nparts; r += ngrp) s0 += part[(long)r * cols + col];
        fsum[grp][col] = (s0 + s1) + (s2 + s3);
        const bool ismax = col < K;
        float m = ismax ? -INFINITY : INFINITY;
#pragma unroll 4
        for (int q = grp; q < mm_rows; q += ngrp) {
            const float v = mm[(long)q * cols + col];
            m = ismax ? fmaxf(m, v) : fminf(m, v);
        }
        fmm[grp][col] = m;
    }
    __syncthreads();
    if (threadIdx.x < K) {
        const int k = threadIdx.x;
        if (k == 0 && num_batches_tracked) *num_batches_tracked += 1;
        double s1 = 0.0, s2 = 0.0;
        float mx = -INFINITY, mn = INFINITY;
        for (int g = 0; g < ngrp; ++g) {
            s1 += fsum[g][k];
            s2 += fsum[g][K + k];
            mx = fmaxf(mx, fmm[g][k]);
            mn = fminf(mn, fmm[g][K + k]);
        }
        const double mean = s1 / count;
        double var = s2 / count - mean * mean;
        if (var < 0.0) var = 0.0;
        const float invstd = (float)(1.0 / sqrt(var + (double)eps));
        const float g = gamma ? gamma[k] : 1.f, bt = beta ? beta[k] : 0.f;
        const float sc = g * invstd, sh = bt - (float)mean * sc;
        mean_out[k] = (float)mean;
        invstd_out[k] = invstd;
        scale[k] = sc;
        shift[k] = sh;
        if (running_mean) {
            const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
            running_mean[k] = (1.f - momentum) * running_mean[k] + momentum * (float)mean;
            running_var[k] = (1.f - momentum) * running_var[k] + momentum * (float)unbiased;
        }
        bound[k] = fmaxf(fmaxf(__builtin_fmaf(mx, sc, sh), __builtin_fmaf(mn, sc, sh)), 0.f);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float m = 0.f;
        for (int k = 0; k < K; ++k) m = fmaxf(m, bound[k]);
        absmax_out[0] = __float_as_uint(m);
    }
}

__global__ void bn_eval_coeffs_kernel(int K, const float* __restrict__ gamma, const float* __restrict__ beta,
                                      const float* __restrict__ rm, const float* __restrict__ rv, float eps,
                                      float* __restrict__ scale, float* __restrict__ shift) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    const float invstd = 1.f / sqrtf(rv[k] + eps);
    const float sc = (gamma ? gamma[k] : 1.f) * invstd;
    scale[k] = sc;
    shift[k] = (beta ? beta[k] : 0.f) - rm[k] * sc;
}

// max |v| of everything the launch wrote: the block's maximum goes into the abs-max buffer (layout and rationale: egz_common.h)
__device__ __forceinline__ void block_absmax_commit(float m, unsigned int* __restrict__ absmax) {
    if (!absmax) return;                                  // uniform
    __shared__ float s_am[16];
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    const int wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    if ((threadIdx.x & 63) == 0) s_am[wave] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < nw; ++w) m = fmaxf(m, s_am[w]);
        absmax_commit(absmax, blockIdx.x, m);
    }
}
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, long n4, unsigned int* __restrict__ absmax) {
    float m = 0.f;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
        m = fmaxf(m, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
    }
    block_absmax_commit(m, absmax);
}

// ------------------------------------------------------------------ BN-apply + ReLU (+pool) forward
// SPLIT (round 5): `absmax` is an INPUT -- the exact max of what this pass writes (egz_bn_finalize_bound) -- and the output is
// stored PRE-SPLIT: per 4-channel quad the 16 bytes [4 hi halves | 4 lo halves] of the f16 pair of value * absmax_scale(absmax),
// exactly the pair the consuming convolution / weight-gradient kernels would form while staging the fp32 value (same
// footprint; those kernels then stage without a split: conv3x3_igemm_x3s_kernel<..., PRE>, conv3x3_wgrad9_x3_kernel<..., XPRE>).
template <bool POOL, bool SPLIT = false>
__global__ __launch_bounds__(256) void bn_relu_pool_fwd_kernel(const float* __restrict__ y,
                                                               const float* __restrict__ scale,
                                                               const float* __restrict__ shift,
                                                               float* __restrict__ out, int B, int H, int W, int K,
                                                               unsigned int* __restrict__ absmax) {
    const float a_scale = SPLIT ? absmax_scale(absmax) : 1.f;
    const int K4 = K >> 2;
    const int Ho = POOL ? (H >> 1) : H, Wo = POOL ? (W >> 1) : W;
    const long n = (long)B * Ho * Wo * K4;
    float amx = 0.f;                                      // max of what this thread wrote (outputs are >= 0)
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % K4);
        const long pix = i / K4;
        const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + c4 * 4);
        const f32x4 sh = *reinterpret_cast<const f32x4*>(shift + c4 * 4);
        f32x4 r;
        if (!POOL) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(y + pix * K + c4 * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e] = fmaxf(v[e] * sc[e] + sh[e], 0.f);
        } else {
            const int xo = (int)(pix % Wo);
            const long t = pix / Wo;
            const int yo = (int)(t % Ho);
            const long b = t / Ho;
            const float* p = y + (((b * H + 2 * yo) * (long)W + 2 * xo) * K + c4 * 4);
            const f32x4 v00 = *reinterpret_cast<const f32x4*>(p);
            const f32x4 v01 = *reinterpret_cast<const f32x4*>(p + K);
            const f32x4 v10 = *reinterpret_cast<const f32x4*>(p + (long)W * K);
            const f32x4 v11 = *reinterpret_cast<const f32x4*>(p + (long)W * K + K);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float a = fmaxf(v00[e] * sc[e] + sh[e], v01[e] * sc[e] + sh[e]);
                const float c = fmaxf(v10[e] * sc[e] + sh[e], v11[e] * sc[e] + sh[e]);
                r[e] = fmaxf(fmaxf(a, c), 0.f);
            }
        }
        if constexpr (SPLIT) {
            x3::u32x2 hi, lo;
            x3::Half<_Float16>::split4(r * a_scale, hi, lo);
            *reinterpret_cast<x3::u32x4*>(out + pix * K + c4 * 4) = x3::u32x4{hi[0], hi[1], lo[0], lo[1]};
        } else {
            *reinterpret_cast<f32x4*>(out + pix * K + c4 * 4) = r;
            amx = fmaxf(amx, fmaxf(fmaxf(r[0], r[1]), fmaxf(r[2], r[3])));
        }
    }
    if constexpr (!SPLIT) block_absmax_commit(amx, absmax);
}

// dz at the four (or one) input positions of an output pixel: ReLU mask and first-max-wins pool routing
// (torch's max_pool2d scan order (0,0),(0,1),(1,0),(1,1); strictly-greater replaces).
__device__ __forceinline__ void pool_route(const float z[4], float dout, float dz[4]) {
    int best = 0;
    float bv = z[0];
#pragma unroll
    for (int q = 1; q < 4; ++q)
        if (z[q] > bv) {
            bv = z[q];
            best = q;
        }
#pragma unroll
    for (int q = 0; q < 4; ++q) dz[q] = (q == best && bv > 0.f) ? dout : 0.f;
}

// ------------------------------------------------------------------ BN backward, pass 1: per-channel sum(dz), sum(dz*xhat)
template <bool POOL>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ y, const float* __restrict__ dout,
                                                            const float* __restrict__ scale, const float* __restrict__ shift,
                                                            const float* __restrict__ mean, const float* __restrict__ invstd,
                                                            double* __restrict__ part, int B, int H, int W, int K) {
    extern __shared__ double sred[];   // [rows_per_block][2][K]
    const int K4 = K >> 2;
    const int rpb = blockDim.x / K4;               // output-pixel rows handled concurrently by a block
    const int c4 = threadIdx.x % K4, rr = threadIdx.x / K4;
    const int Ho = POOL ? (H >> 1) : H, Wo = POOL ? (W >> 1) : W;
    const long npix = (long)B * Ho * Wo;
    // per-thread partials in fp32 (a thread sees <= a few hundred pixels), fp64 across threads / blocks
    float s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
    if (rr < rpb) {
        const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + c4 * 4);
        const f32x4 sh = *reinterpret_cast<const f32x4*>(shift + c4 * 4);
        const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c4 * 4);
        const f32x4 is = *reinterpret_cast<const f32x4*>(invstd + c4 * 4);
        const long stride = (long)gridDim.x * rpb;
        if (!POOL) {
            // two pixels per iteration: four 16-byte loads in flight per thread instead of two (a narrow tensor -- the
            // late-fusion widths, K = 32 -- ran at 3.9 TB/s, latency-bound); same accumulation order as one pixel at a time
            for (long pix = (long)blockIdx.x * rpb + rr; pix < npix; pix += 2 * stride) {
                const bool two = pix + stride < npix;
                const f32x4 g0 = *reinterpret_cast<const f32x4*>(dout + pix * K + c4 * 4);
                const f32x4 v0 = *reinterpret_cast<const f32x4*>(y + pix * K + c4 * 4);
                f32x4 g1 = {0.f, 0.f, 0.f, 0.f}, v1 = {0.f, 0.f, 0.f, 0.f};
                if (two) {
                    g1 = *reinterpret_cast<const f32x4*>(dout + (pix + stride) * K + c4 * 4);
                    v1 = *reinterpret_cast<const f32x4*>(y + (pix + stride) * K + c4 * 4);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float z = v0[e] * sc[e] + sh[e];
                    const float dz = z > 0.f ? g0[e] : 0.f;
                    s1[e] += dz;
                    s2[e] += dz * ((v0[e] - mu[e]) * is[e]);
                }
                if (two) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float z = v1[e] * sc[e] + sh[e];
                        const float dz = z > 0.f ? g1[e] : 0.f;
                        s1[e] += dz;
                        s2[e] += dz * ((v1[e] - mu[e]) * is[e]);
                    }
                }
            }
        }
        for (long pix = (long)blockIdx.x * rpb + rr; POOL && pix < npix; pix += stride) {
            const f32x4 g = *reinterpret_cast<const f32x4*>(dout + pix * K + c4 * 4);
            if (!POOL) {
            } else {
                const int xo = (int)(pix % Wo);
                const long t = pix / Wo;
                const int yo = (int)(t % Ho);
                const long b = t / Ho;
                const float* p = y + (((b * H + 2 * yo) * (long)W + 2 * xo) * K + c4 * 4);
                f32x4 v[4];
                v[0] = *reinterpret_cast<const f32x4*>(p);
                v[1] = *reinterpret_cast<const f32x4*>(p + K);
                v[2] = *reinterpret_cast<const f32x4*>(p + (long)W * K);
                v[3] = *reinterpret_cast<const f32x4*>(p + (long)W * K + K);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float z[4], dz[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) z[q] = v[q][e] * sc[e] + sh[e];
                    pool_route(z, g[e], dz);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        s1[e] += dz[q];
                        s2[e] += dz[q] * ((v[q][e] - mu[e]) * is[e]);
                    }
                }
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            sred[((long)rr * 2 + 0) * K + c4 * 4 + e] = (double)s1[e];
            sred[((long)rr * 2 + 1) * K + c4 * 4 + e] = (double)s2[e];
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * K; i += blockDim.x) {
        double t = 0.0;
        for (int r = 0; r < rpb; ++r) t += sred[(long)r * 2 * K + i];
        part[(long)blockIdx.x * 2 * K + i] = t;
    }
}

// part: [nparts][2][K] -> dgamma, dbeta and the two per-channel means the apply pass needs
// bound (absmax_out != null; K % 64 == 0): an upper bound of max |dy| of the apply pass that follows, before it runs --
//   |dy| = |sc| |dz - m1 - xhat m2| <= |sc| (D + |m1| + X |m2|),  D = max |dout| (dout_absmax, from the kernel that produced dout),
//   X = max |xhat| over the channel = max(|ymax - mean|, |ymin - mean|) invstd (y_minmax: the conv epilogue's per-channel max / min)
// -- so that the apply pass can store dy as pre-split f16 pairs (bn_bwd_apply_kernel<..., SPLIT>).  Rigorous (every term is a
// bound; 0.1 % slack covers the fp32 rounding of the apply pass), and tight: the dz term dominates and |dz| <= |dout|.
__global__ void bn_bwd_finalize_kernel(const double* __restrict__ part, int nparts, int K, double count,
                                       float* __restrict__ dgamma, float* __restrict__ dbeta,
                                       float* __restrict__ mdz, float* __restrict__ mdzx,
                                       const float* __restrict__ scale = nullptr, const float* __restrict__ mean = nullptr,
                                       const float* __restrict__ invstd = nullptr, const unsigned int* __restrict__ y_minmax = nullptr,
                                       const unsigned int* __restrict__ dout_absmax = nullptr,
                                       unsigned int* __restrict__ absmax_out = nullptr) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    double s1 = 0.0, s2 = 0.0;
#pragma unroll 16
    for (int p = 0; p < nparts; ++p) {
        s1 += part[((long)p * 2 + 0) * K + k];
        s2 += part[((long)p * 2 + 1) * K + k];
    }
    if (dbeta) dbeta[k] = (float)s1;
    if (dgamma) dgamma[k] = (float)s2;
    const float m1 = (float)(s1 / count), m2 = (float)(s2 / count);
    mdz[k] = m1;
    mdzx[k] = m2;
    if (absmax_out) {                                     // (block-uniform; whole waves: K % 64 == 0, 64-thread blocks)
        const float D = __uint_as_float(absmax_bits(dout_absmax));
        unsigned int umx = 0, umn = 0;
        for (int sl = 0; sl < (512 / K > 0 ? 512 / K : 1); ++sl) {
            const unsigned int a = y_minmax[sl * 2 * K + k], b = y_minmax[sl * 2 * K + K + k];
            umx = a > umx ? a : umx;
            umn = b > umn ? b : umn;
        }
        auto from_ordered = [](unsigned int u) -> float { return __uint_as_float((u & 0x80000000u) ? (u ^ 0x80000000u) : ~u); };
        const float ymax = from_ordered(umx), ymin = -from_ordered(umn);
        const float X = fmaxf(fabsf(ymax - mean[k]), fabsf(ymin - mean[k])) * invstd[k];
        float bound = fabsf(scale[k]) * (D + fabsf(m1) + X * fabsf(m2)) * 1.001f;
        if (!(bound < INFINITY)) bound = 0.f;             // (a non-finite gradient: scale 1, like absmax_scale on a NaN maximum)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) bound = fmaxf(bound, __shfl_xor(bound, o));
        if ((threadIdx.x & 63) == 0) absmax_commit(absmax_out, blockIdx.x, bound);
    }
}

// The finalize step inside a consumer kernel (K <= 64): every block sums the few partial rows ([nrows][2][K] fp64, written by
// the 256 blocks of the persistent narrow data-gradient kernel) itself, in a fixed order -- identical values in every block --
// and block 0 also writes dgamma / dbeta.  The two tiny launches this replaces (column sums + finalize) would each wait for a
// CU slot behind the weight-gradient kernel running on the helper stream (100 us apiece in the late-fusion backward pass,
// profiles/r03_lf_timeline.txt).  fm[0][k] = mean(dz), fm[1][k] = mean(dz * xhat).
__device__ inline void bn_bwd_finalize_in_block(const double* __restrict__ rows, int nrows, int K, double count,
                                                float* __restrict__ dgamma, float* __restrict__ dbeta, double (*fsum)[128],
                                                float (*fm)[64]) {
    const int cols = 2 * K, ngrp = 256 / cols;                 // K = 32: 4 row groups of 64 columns
    const int col = threadIdx.x % cols, grp = threadIdx.x / cols;
    if (grp < ngrp) {
        double s = 0.0;
        for (int r = grp; r < nrows; r += ngrp) s += rows[(long)r * cols + col];
        fsum[grp][col] = s;
    }
    __syncthreads();
    if (threadIdx.x < cols) {
        double t = 0.0;
        for (int g = 0; g < ngrp; ++g) t += fsum[g][threadIdx.x];
        const int k = threadIdx.x % K, plane = threadIdx.x / K;
        fm[plane][k] = (float)(t / count);
        if (blockIdx.x == 0) {
            float* dst = plane ? dgamma : dbeta;
            if (dst) dst[k] = (float)t;
        }
    }
    __syncthreads();
}

// ------------------------------------------------------------------ BN backward, pass 2: dy = gamma*invstd*(dz - mean(dz) - xhat*mean(dz*xhat))
// SPLIT: absmax is an INPUT (the bound of bn_bwd_finalize_kernel) and dy is stored as pre-split f16 pairs scaled by
// absmax_scale(absmax) (layout: bn_relu_pool_fwd_kernel); consumers: egz_conv3x3_fwd_streamed(mode | 0x100) as the data gradient's
// operand, egz_conv3x3_wgrad(flags | 0x10000).
template <bool POOL, bool FIN = false, bool SPLIT = false>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ y, const float* __restrict__ dout,
                                                           const float* __restrict__ scale, const float* __restrict__ shift,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           const float* mdz, const float* mdzx,
                                                           float* __restrict__ dy, int B, int H, int W, int K,
                                                           unsigned int* __restrict__ absmax, const double* __restrict__ rows = nullptr,
                                                           int nrows = 0, double count = 1.0, float* __restrict__ dgamma = nullptr,
                                                           float* __restrict__ dbeta = nullptr) {
    __shared__ double fsum[FIN ? 16 : 1][128];
    __shared__ float fm[2][64];
    if (FIN) {          // FIN: mdz / mdzx are not inputs -- this block derives them from the partial rows (K <= 64)
        bn_bwd_finalize_in_block(rows, nrows, K, count, dgamma, dbeta, fsum, fm);
        mdz = fm[0];
        mdzx = fm[1];
    }
    const float a_scale = SPLIT ? absmax_scale(absmax) : 1.f;
    auto put = [&](float* dst, const f32x4 r) {
        if constexpr (SPLIT) {
            x3::u32x2 hi, lo;
            x3::Half<_Float16>::split4(r * a_scale, hi, lo);
            *reinterpret_cast<x3::u32x4*>(dst) = x3::u32x4{hi[0], hi[1], lo[0], lo[1]};
        } else {
            *reinterpret_cast<f32x4*>(dst) = r;
        }
    };
    const int K4 = K >> 2;
    const int Ho = POOL ? (H >> 1) : H, Wo = POOL ? (W >> 1) : W;
    const long n = (long)B * Ho * Wo * K4;
    float amx = 0.f;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % K4);
        const long pix = i / K4;
        const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + c4 * 4);
        const f32x4 sh = *reinterpret_cast<const f32x4*>(shift + c4 * 4);
        const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c4 * 4);
        const f32x4 is = *reinterpret_cast<const f32x4*>(invstd + c4 * 4);
        const f32x4 m1 = *reinterpret_cast<const f32x4*>(mdz + c4 * 4);
        const f32x4 m2 = *reinterpret_cast<const f32x4*>(mdzx + c4 * 4);
        const f32x4 g = *reinterpret_cast<const f32x4*>(dout + pix * K + c4 * 4);
        if (!POOL) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(y + pix * K + c4 * 4);
            f32x4 r;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float z = v[e] * sc[e] + sh[e];
                const float dz = z > 0.f ? g[e] : 0.f;
                const float xh = (v[e] - mu[e]) * is[e];
                r[e] = sc[e] * (dz - m1[e] - xh * m2[e]);
                amx = fmaxf(amx, fabsf(r[e]));
            }
            put(dy + pix * K + c4 * 4, r);
        } else {
            const int xo = (int)(pix % Wo);
            const long t = pix / Wo;
            const int yo = (int)(t % Ho);
            const long b = t / Ho;
            const long base = ((b * H + 2 * yo) * (long)W + 2 * xo) * K + c4 * 4;
            const long off[4] = {0, (long)K, (long)W * K, (long)W * K + K};
            f32x4 v[4], r[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = *reinterpret_cast<const f32x4*>(y + base + off[q]);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float z[4], dz[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) z[q] = v[q][e] * sc[e] + sh[e];
                pool_route(z, g[e], dz);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float xh = (v[q][e] - mu[e]) * is[e];
                    r[q][e] = sc[e] * (dz[q] - m1[e] - xh * m2[e]);
                    amx = fmaxf(amx, fabsf(r[q][e]));
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) put(dy + base + off[q], r[q]);
        }
    }
    if constexpr (!SPLIT) block_absmax_commit(amx, absmax);
}

// ------------------------------------------------------------------ fusion: z = max(ys, yt) (ys wins ties)
__global__ __launch_bounds__(256) void pairmax_fwd_kernel(const float* __restrict__ y2, float* __restrict__ z, long n4) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const f32x4 a = reinterpret_cast<const f32x4*>(y2)[i];
        const f32x4 b = reinterpret_cast<const f32x4*>(y2)[n4 + i];
        f32x4 r;
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = a[e] >= b[e] ? a[e] : b[e];
        reinterpret_cast<f32x4*>(z)[i] = r;
    }
}
__global__ __launch_bounds__(256) void pairmax_bwd_kernel(const float* __restrict__ y2, const float* __restrict__ dz,
                                                          float* __restrict__ dy2, long n4, unsigned int* __restrict__ absmax) {
    float amx = 0.f;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const f32x4 a = reinterpret_cast<const f32x4*>(y2)[i];
        const f32x4 b = reinterpret_cast<const f32x4*>(y2)[n4 + i];
        const f32x4 g = reinterpret_cast<const f32x4*>(dz)[i];
        f32x4 ra, rb;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool first = a[e] >= b[e];
            ra[e] = first ? g[e] : 0.f;
            rb[e] = first ? 0.f : g[e];
            amx = fmaxf(amx, fabsf(g[e]));
        }
        reinterpret_cast<f32x4*>(dy2)[i] = ra;
        reinterpret_cast<f32x4*>(dy2)[n4 + i] = rb;
    }
    block_absmax_commit(amx, absmax);
}

// dy = dout * (out > 0)   (nn.ReLU backward; in place allowed)
__global__ __launch_bounds__(256) void relu_bwd_kernel(const float* __restrict__ out, const float* __restrict__ dout,
                                                       float* __restrict__ dy, long n4) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const f32x4 o = reinterpret_cast<const f32x4*>(out)[i];
        const f32x4 g = reinterpret_cast<const f32x4*>(dout)[i];
        f32x4 r;
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = o[e] > 0.f ? g[e] : 0.f;
        reinterpret_cast<f32x4*>(dy)[i] = r;
    }
}

// dy = dout * (out > 0) plus per-channel partial sums of dy (the conv bias gradient), one pass
__global__ __launch_bounds__(256) void relu_bwd_bias_kernel(const float* __restrict__ out, const float* __restrict__ dout,
                                                            float* __restrict__ dy, double* __restrict__ part,
                                                            long rows, int K, unsigned int* __restrict__ absmax) {
    extern __shared__ double sred[];   // [rpb][K]
    const int K4 = K >> 2;
    const int rpb = blockDim.x / K4;
    const int c4 = threadIdx.x % K4, rr = threadIdx.x / K4;
    float s[4] = {0, 0, 0, 0};
    float amx = 0.f;
    if (rr < rpb) {
        for (long r = (long)blockIdx.x * rpb + rr; r < rows; r += (long)gridDim.x * rpb) {
            const f32x4 o = *reinterpret_cast<const f32x4*>(out + r * K + c4 * 4);
            const f32x4 g = *reinterpret_cast<const f32x4*>(dout + r * K + c4 * 4);
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = o[e] > 0.f ? g[e] : 0.f;
                s[e] += v[e];
                amx = fmaxf(amx, fabsf(v[e]));
            }
            *reinterpret_cast<f32x4*>(dy + r * K + c4 * 4) = v;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) sred[(long)rr * K + c4 * 4 + e] = (double)s[e];
    }
    block_absmax_commit(amx, absmax);
    __syncthreads();
    for (int i = threadIdx.x; i < K; i += blockDim.x) {
        double t = 0.0;
        for (int r = 0; r < rpb; ++r) t += sred[(long)r * K + i];
        part[(long)blockIdx.x * K + i] = t;
    }
}

// nn.Upsample(scale_factor=2, nearest) backward: dx[b][y][x][c] = sum of the 2x2 block of dxu
__global__ __launch_bounds__(256) void upsample2x_bwd_kernel(const float* __restrict__ dxu, float* __restrict__ dx,
                                                             int B, int H, int W, int C) {
    const int C4 = C >> 2;
    const long n = (long)B * H * W * C4;
    const int W2 = 2 * W;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        const long pix = i / C4;
        const int xo = (int)(pix % W);
        const long t = pix / W;
        const int yo = (int)(t % H);
        const long b = t / H;
        const float* p = dxu + (((b * 2 * H + 2 * yo) * (long)W2 + 2 * xo) * C + c4 * 4);
        const f32x4 a = *reinterpret_cast<const f32x4*>(p);
        const f32x4 bb = *reinterpret_cast<const f32x4*>(p + C);
        const f32x4 c = *reinterpret_cast<const f32x4*>(p + (long)W2 * C);
        const f32x4 d = *reinterpret_cast<const f32x4*>(p + (long)W2 * C + C);
        f32x4 r;
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = (a[e] + bb[e]) + (c[e] + d[e]);
        *reinterpret_cast<f32x4*>(dx + pix * C + c4 * 4) = r;
    }
}

__global__ void colsum_final_kernel(const double* __restrict__ part, int nparts, int cols, float* __restrict__ out) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= cols) return;
    double s = 0.0;
#pragma unroll 16
    for (int p = 0; p < nparts; ++p) s += part[(long)p * cols + k];
    out[k] = (float)s;
}

// layout transposes between the reference's NCHW tensors and the library's NHWC activations
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                           int B, int C, long HW, int Cp) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const long p0 = (long)blockIdx.x * 32;
    const int c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int j = ty; j < 32; j += 8) {
        const int c = c0 + j;
        const long p = p0 + tx;
        tile[j][tx] = (c < C && p < HW) ? in[((long)b * C + c) * HW + p] : 0.f;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const long p = p0 + j;
        const int c = c0 + tx;
        if (c < Cp && p < HW) out[((long)b * HW + p) * Cp + c] = tile[tx][j];      // channels C .. Cp-1 are zero padding
    }
}
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                           int B, int C, long HW) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const long p0 = (long)blockIdx.x * 32;
    const int c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int j = ty; j < 32; j += 8) {
        const long p = p0 + j;
        const int c = c0 + tx;
        tile[j][tx] = (c < C && p < HW) ? in[((long)b * HW + p) * C + c] : 0.f;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int c = c0 + j;
        const long p = p0 + tx;
        if (c < C && p < HW) out[((long)b * C + c) * HW + p] = tile[tx][j];
    }
}

// Grid of the element-wise passes: n float4 items over 256-thread blocks, grid-stride, capped at 8192 blocks.  (Narrower grids --
// fewer of the CU slots that conv blocks free, held for longer -- change nothing down to 1024 blocks and lose below:
// profiles/r03_ew_cap_ab.txt.)
inline int ew_cap() { return 8192; }
inline int ew_grid(long n) {
    long g = (n + 255) / 256;
    const int cap = ew_cap();
    return (int)(g > cap ? cap : (g < 1 ? 1 : g));
}

constexpr int BWD_BLOCKS = 1024;
constexpr int FIN_MAX_ROWS = 512, FIN_GRID = 2048;     // in-kernel finalize: at most this many partial rows / blocks re-summing them
inline bool fin_in_kernel() { return true; }          // few partial rows: the apply pass sums them itself

}  // namespace

// ------------------------------------------------------------------------------------------------ C ABI
EGZ_API size_t egz_bn_ws_bytes(int K) {
    // stage-A partials of the forward stats ([RED_ROWS][2][K]) or the backward reduction ([BWD_BLOCKS][2][K])
    const size_t a = (size_t)RED_ROWS * 2 * K * sizeof(double);
    const size_t b = (size_t)BWD_BLOCKS * 2 * K * sizeof(double);
    return a > b ? a : b;
}

// stat_partial: [rows][2][K] fp64 as written by the conv epilogues.  Produces batch mean / invstd and the fused
// affine (scale, shift); updates running_mean / running_var in place when they are non-null and increments the int64
// num_batches_tracked counter when that is non-null.
EGZ_API int egz_bn_finalize(const double* stat_partial, int rows, int K, double count, const float* gamma,
                            const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                            float* mean_out, float* invstd_out, float* scale, float* shift,
                            long long* num_batches_tracked, void* workspace, size_t ws_bytes, hipStream_t st) {
    EGZ_CHECK_ARG(stat_partial && mean_out && invstd_out && scale && shift && workspace, "egz_bn_finalize: null pointer");
    EGZ_CHECK_ARG(ws_bytes >= (size_t)RED_ROWS * 2 * K * sizeof(double), "egz_bn_finalize: workspace too small");
    EGZ_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), "egz_bn_finalize: running stats must come in pairs");
    double* part2 = static_cast<double*>(workspace);
    const double* src = stat_partial;
    int nparts = rows;
    if (rows > RED_ROWS) {
        int rc = colsum_partial<double>(stat_partial, part2, rows, 2 * K, st);
        if (rc) return rc;
        src = part2;
        nparts = RED_ROWS;
    }
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(egz_cdiv(K, 128)), dim3(128), 0, st, src, nparts, K, count, gamma, beta,
                       running_mean, running_var, momentum, eps, mean_out, invstd_out, scale, shift, num_batches_tracked,
                       nullptr, nullptr);
    EGZ_CHECK_LAUNCH("egz_bn_finalize");
    return 0;
}

// egz_bn_finalize that also bounds the block output: minmax = the 1024 uints egz_conv3x3_fwd_streamed's minmax_out received on the
// 64- / 128-column tiles (1024 / (2 K) sets of order-preserving images of the per-channel max of y and of -y), absmax_out (egz_absmax layout,
// zero-filled by the caller) receives the EXACT max of relu(y * scale + shift) over the whole tensor -- what
// egz_bn_relu_pool_fwd_presplit needs before it writes the first element.  K % 64 == 0.
EGZ_API int egz_bn_finalize_bound(const double* stat_partial, int rows, int K, double count, const float* gamma,
                                  const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                                  float* mean_out, float* invstd_out, float* scale, float* shift,
                                  long long* num_batches_tracked, void* workspace, size_t ws_bytes,
                                  const unsigned int* minmax, unsigned int* absmax_out, hipStream_t st) {
    EGZ_CHECK_ARG(stat_partial && mean_out && invstd_out && scale && shift && workspace && minmax && absmax_out,
                  "egz_bn_finalize_bound: null pointer");
    EGZ_CHECK_ARG(K % 64 == 0, "egz_bn_finalize_bound: K=%d must be a multiple of 64", K);
    EGZ_CHECK_ARG(ws_bytes >= (size_t)RED_ROWS * 2 * K * sizeof(double), "egz_bn_finalize_bound: workspace too small");
    EGZ_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), "egz_bn_finalize_bound: running stats must come in pairs");
    double* part2 = static_cast<double*>(workspace);
    const double* src = stat_partial;
    int nparts = rows;
    if (rows > RED_ROWS) {
        int rc = colsum_partial<double>(stat_partial, part2, rows, 2 * K, st);
        if (rc) return rc;
        src = part2;
        nparts = RED_ROWS;
    }
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(egz_cdiv(K, 128)), dim3(128), 0, st, src, nparts, K, count, gamma, beta,
                       running_mean, running_var, momentum, eps, mean_out, invstd_out, scale, shift, num_batches_tracked,
                       minmax, absmax_out);
    EGZ_CHECK_LAUNCH("egz_bn_finalize_bound");
    return 0;
}

// egz_bn_finalize for a BatchNorm whose output is never materialised (the consuming convolution applies it while staging y):
// minmax = [mm_rows][2][K] per-channel max / min rows of y (egz_conv_first_fwd / egz_conv3x3_fwd_streamed minmax_out),
// absmax_out (egz_absmax layout) receives the exact max of relu(y * scale + shift).  K in {16, 32, 64}; one launch.
EGZ_API int egz_bn_finalize_deferred(const double* stat_partial, int rows, int K, double count, const float* gamma,
                                     const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                                     float* mean_out, float* invstd_out, float* scale, float* shift,
                                     long long* num_batches_tracked, const float* minmax, int mm_rows,
                                     unsigned int* absmax_out, hipStream_t st) {
    EGZ_CHECK_ARG(stat_partial && mean_out && invstd_out && scale && shift && minmax && absmax_out,
                  "egz_bn_finalize_deferred: null pointer");
    EGZ_CHECK_ARG((K == 16 || K == 32 || K == 64) && rows > 0 && mm_rows > 0,
                  "egz_bn_finalize_deferred: K=%d must be 16, 32 or 64", K);
    EGZ_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), "egz_bn_finalize_deferred: running stats must come in pairs");
    hipLaunchKernelGGL(bn_finalize_deferred_kernel, dim3(1), dim3(1024), 0, st, stat_partial, rows, K, count, gamma, beta,
                       running_mean, running_var, momentum, eps, mean_out, invstd_out, scale, shift, num_batches_tracked,
                       minmax, mm_rows, absmax_out);
    EGZ_CHECK_LAUNCH("egz_bn_finalize_deferred");
    return 0;
}

EGZ_API int egz_bn_eval_coeffs(int K, const float* gamma, const float* beta, const float* running_mean,
                               const float* running_var, float eps, float* scale, float* shift, hipStream_t st) {
    EGZ_CHECK_ARG(running_mean && running_var && scale && shift, "egz_bn_eval_coeffs: null pointer");
    hipLaunchKernelGGL(bn_eval_coeffs_kernel, dim3(egz_cdiv(K, 128)), dim3(128), 0, st, K, gamma, beta, running_mean,
                       running_var, eps, scale, shift);
    EGZ_CHECK_LAUNCH("egz_bn_eval_coeffs");
    return 0;
}

// out = relu(y*scale + shift), optionally followed by MaxPool2d(2,2) (out is then [B][H/2][W/2][K]).
// absmax (optional): receives max |out| (egz_absmax layout) -- the f16 x3 scaling of the convolution that consumes `out`.
EGZ_API int egz_bn_relu_pool_fwd(const float* y, const float* scale, const float* shift, float* out, int B, int H,
                                 int W, int K, int pool, unsigned int* absmax, hipStream_t st) {
    EGZ_CHECK_ARG(y && scale && shift && out, "egz_bn_relu_pool_fwd: null pointer");
    EGZ_CHECK_ARG(K % 4 == 0, "egz_bn_relu_pool_fwd: K=%d must be a multiple of 4", K);
    EGZ_CHECK_ARG(!pool || (H % 2 == 0 && W % 2 == 0), "egz_bn_relu_pool_fwd: pooled map must be even");
    const long n = (long)B * (pool ? H / 2 : H) * (pool ? W / 2 : W) * (K / 4);
    if (pool) hipLaunchKernelGGL(bn_relu_pool_fwd_kernel<true>, dim3(ew_grid(n)), dim3(256), 0, st, y, scale, shift, out, B, H, W, K, absmax);
    else      hipLaunchKernelGGL(bn_relu_pool_fwd_kernel<false>, dim3(ew_grid(n)), dim3(256), 0, st, y, scale, shift, out, B, H, W, K, absmax);
    EGZ_CHECK_LAUNCH("egz_bn_relu_pool_fwd");
    return 0;
}

// egz_bn_relu_pool_fwd with the output stored PRE-SPLIT (see bn_relu_pool_fwd_kernel): absmax is an INPUT here, the exact max of
// the output (egz_bn_finalize_bound).  The consumers are egz_conv3x3_fwd_streamed(mode | 0x100) and egz_conv3x3_wgrad(flags |
// 0x8000); nothing else can read the tensor.
EGZ_API int egz_bn_relu_pool_fwd_presplit(const float* y, const float* scale, const float* shift, float* out, int B, int H,
                                          int W, int K, int pool, const unsigned int* absmax, hipStream_t st) {
    EGZ_CHECK_ARG(y && scale && shift && out && absmax, "egz_bn_relu_pool_fwd_presplit: null pointer");
    EGZ_CHECK_ARG(K % 4 == 0, "egz_bn_relu_pool_fwd_presplit: K=%d must be a multiple of 4", K);
    EGZ_CHECK_ARG(!pool || (H % 2 == 0 && W % 2 == 0), "egz_bn_relu_pool_fwd_presplit: pooled map must be even");
    const long n = (long)B * (pool ? H / 2 : H) * (pool ? W / 2 : W) * (K / 4);
    unsigned int* am = const_cast<unsigned int*>(absmax);
    if (pool) hipLaunchKernelGGL((bn_relu_pool_fwd_kernel<true, true>), dim3(ew_grid(n)), dim3(256), 0, st, y, scale, shift, out, B, H, W, K, am);
    else      hipLaunchKernelGGL((bn_relu_pool_fwd_kernel<false, true>), dim3(ew_grid(n)), dim3(256), 0, st, y, scale, shift, out, B, H, W, K, am);
    EGZ_CHECK_LAUNCH("egz_bn_relu_pool_fwd_presplit");
    return 0;
}

EGZ_API size_t egz_bn_relu_pool_bwd_ws_bytes(int K) {
    return ((size_t)BWD_BLOCKS + RED_ROWS) * 2 * K * sizeof(double) + 2 * (size_t)K * sizeof(float);
}

// Backward of [BN(train) -> ReLU -> (pool)] given the saved pre-BN tensor y and the batch statistics.
// dgamma/dbeta may be null (frozen BN).  dy gets the gradient w.r.t. y ([B][H][W][K]).
static int bn_relu_pool_bwd_impl(const float* y, const float* dout, const float* scale, const float* shift,
                                 const float* mean, const float* invstd, float* dy, float* dgamma, float* dbeta,
                                 int B, int H, int W, int K, int pool, void* workspace, size_t ws_bytes,
                                 unsigned int* absmax, const double* sums, int sums_rows, const unsigned int* y_minmax,
                                 const unsigned int* dout_absmax, hipStream_t st) {
    const bool presplit = y_minmax != nullptr;
    // sums (optional): [sums_rows][2][K] partial rows of (sum dz, sum dz * xhat) already accumulated by the producer of dout
    // (egz_conv3x3_fwd_streamed epi 5): the reduce pass over y and dout is skipped.
    EGZ_CHECK_ARG(y && dout && scale && shift && mean && invstd && dy && workspace, "egz_bn_relu_pool_bwd: null pointer");
    EGZ_CHECK_ARG(!sums || (sums_rows > 0 && !pool), "egz_bn_relu_pool_bwd: precomputed sums need sums_rows > 0 and no pooling");
    EGZ_CHECK_ARG(K % 4 == 0 && K <= 1024, "egz_bn_relu_pool_bwd: K=%d must be a multiple of 4, <= 1024", K);
    EGZ_CHECK_ARG(!pool || (H % 2 == 0 && W % 2 == 0), "egz_bn_relu_pool_bwd: pooled map must be even");
    const int K4 = K / 4;
    const int threads = 256 > K4 ? 256 : K4;
    const int rpb = threads / K4;
    const long npix = (long)B * (pool ? H / 2 : H) * (pool ? W / 2 : W);
    int blocks = (int)((npix + rpb - 1) / rpb);
    if (blocks > BWD_BLOCKS) blocks = BWD_BLOCKS;
    if (blocks > ew_cap()) blocks = ew_cap();
    const size_t need = egz_bn_relu_pool_bwd_ws_bytes(K);
    EGZ_CHECK_ARG(ws_bytes >= need, "egz_bn_relu_pool_bwd: workspace too small (%zu < %zu)", ws_bytes, need);
    double* part = static_cast<double*>(workspace);
    double* part2 = part + (size_t)BWD_BLOCKS * 2 * K;
    float* mdz = reinterpret_cast<float*>(part2 + (size_t)RED_ROWS * 2 * K);
    float* mdzx = mdz + K;
    const size_t shm = (size_t)rpb * 2 * K * sizeof(double);
    if (!presplit && sums && sums_rows <= FIN_MAX_ROWS && K <= 64 && 256 % (2 * K) == 0 && fin_in_kernel()) {
        // few partial rows (persistent narrow kernel): the apply pass sums them itself -- no column-sum / finalize launches
        const long n = npix * K4;
        const int grid = ew_grid(n) < FIN_GRID ? ew_grid(n) : FIN_GRID;
        hipLaunchKernelGGL((bn_bwd_apply_kernel<false, true>), dim3(grid), dim3(256), 0, st, y, dout, scale, shift, mean, invstd,
                           (const float*)nullptr, (const float*)nullptr, dy, B, H, W, K, absmax, sums, sums_rows, (double)B * H * W, dgamma, dbeta);
        EGZ_CHECK_LAUNCH("egz_bn_relu_pool_bwd(finalize + apply)");
        return 0;
    }
    if (sums) {
        part = const_cast<double*>(sums);
        blocks = sums_rows;
    } else {
        if (pool) hipLaunchKernelGGL(bn_bwd_reduce_kernel<true>, dim3(blocks), dim3(threads), shm, st, y, dout, scale, shift, mean, invstd, part, B, H, W, K);
        else      hipLaunchKernelGGL(bn_bwd_reduce_kernel<false>, dim3(blocks), dim3(threads), shm, st, y, dout, scale, shift, mean, invstd, part, B, H, W, K);
        EGZ_CHECK_LAUNCH("egz_bn_relu_pool_bwd(reduce)");
    }
    const double* fin = part;
    int nfin = blocks;
    if (blocks > RED_ROWS) {            // two-stage: keep the serial per-channel tail at <= RED_ROWS terms
        int rc = colsum_partial<double>(part, part2, blocks, 2 * K, st);
        if (rc) return rc;
        fin = part2;
        nfin = RED_ROWS;
    }
    const long n = npix * K4;
    if (presplit) {
        hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(egz_cdiv(K, 64)), dim3(64), 0, st, fin, nfin, K,
                           (double)B * H * W, dgamma, dbeta, mdz, mdzx, scale, mean, invstd, y_minmax, dout_absmax, absmax);
        EGZ_CHECK_LAUNCH("egz_bn_relu_pool_bwd_presplit(finalize)");
        if (pool) hipLaunchKernelGGL((bn_bwd_apply_kernel<true, false, true>), dim3(ew_grid(n)), dim3(256), 0, st, y, dout, scale, shift, mean, invstd, mdz, mdzx, dy, B, H, W, K, absmax);
        else      hipLaunchKernelGGL((bn_bwd_apply_kernel<false, false, true>), dim3(ew_grid(n)), dim3(256), 0, st, y, dout, scale, shift, mean, invstd, mdz, mdzx, dy, B, H, W, K, absmax);
        EGZ_CHECK_LAUNCH("egz_bn_relu_pool_bwd_presplit(apply)");
        return 0;
    }
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(egz_cdiv(K, 64)), dim3(64), 0, st, fin, nfin, K,
                       (double)B * H * W, dgamma, dbeta, mdz, mdzx);
    EGZ_CHECK_LAUNCH("egz_bn_relu_pool_bwd(finalize)");
    if (pool) hipLaunchKernelGGL(bn_bwd_apply_kernel<true>, dim3(ew_grid(n)), dim3(256), 0, st, y, dout, scale, shift, mean, invstd, mdz, mdzx, dy, B, H, W, K, absmax);
    else      hipLaunchKernelGGL(bn_bwd_apply_kernel<false>, dim3(ew_grid(n)), dim3(256), 0, st, y, dout, scale, shift, mean, invstd, mdz, mdzx, dy, B, H, W, K, absmax);
    EGZ_CHECK_LAUNCH("egz_bn_relu_pool_bwd(apply)");
    return 0;
}

EGZ_API int egz_bn_relu_pool_bwd(const float* y, const float* dout, const float* scale, const float* shift,
                                 const float* mean, const float* invstd, float* dy, float* dgamma, float* dbeta,
                                 int B, int H, int W, int K, int pool, void* workspace, size_t ws_bytes,
                                 unsigned int* absmax, const double* sums, int sums_rows, hipStream_t st) {
    return bn_relu_pool_bwd_impl(y, dout, scale, shift, mean, invstd, dy, dgamma, dbeta, B, H, W, K, pool, workspace, ws_bytes,
                                 absmax, sums, sums_rows, nullptr, nullptr, st);
}

// egz_bn_relu_pool_bwd with dy stored PRE-SPLIT (f16 hi / lo pairs, see egz_conv3x3_wgrad_presplit_ok): the pairs are scaled by a
// BOUND of max |dy| that the finalize step derives before the apply pass runs -- from y_minmax (the 1024 uints the conv's
// statistics epilogue wrote in the forward pass: per-channel max / min of y), dout_absmax (egz_absmax layout: max |dout|, from
// the data-gradient kernel that produced dout) and the two per-channel sums -- and leaves in `absmax` (egz_absmax layout,
// zero-filled by the caller), which the consumers take as dy's abs-max.  K % 64 == 0, K <= 512.
EGZ_API int egz_bn_relu_pool_bwd_presplit(const float* y, const float* dout, const float* scale, const float* shift,
                                          const float* mean, const float* invstd, float* dy, float* dgamma, float* dbeta,
                                          int B, int H, int W, int K, int pool, void* workspace, size_t ws_bytes,
                                          unsigned int* absmax, const double* sums, int sums_rows,
                                          const unsigned int* y_minmax, const unsigned int* dout_absmax, hipStream_t st) {
    EGZ_CHECK_ARG(absmax && y_minmax && dout_absmax && K % 64 == 0 && K <= 512,
                  "egz_bn_relu_pool_bwd_presplit: needs absmax, y_minmax, dout_absmax and K %% 64 == 0, K <= 512");
    return bn_relu_pool_bwd_impl(y, dout, scale, shift, mean, invstd, dy, dgamma, dbeta, B, H, W, K, pool, workspace, ws_bytes,
                                 absmax, sums, sums_rows, y_minmax, dout_absmax, st);
}


namespace {
// ------------------------------------------------------------------ first block of a narrow stack: BN backward + weight gradient in one pass
// late_fusion.py:10-12 starts with Conv2d(2 -> 32) -> BatchNorm2d -> ReLU on the (B, 2, H, W) network input, which needs no data
// gradient: the gradient w.r.t. the conv output, dy = scale * (dz - mean(dz) - xhat * mean(dz xhat)), is consumed by the weight
// gradient only.  This kernel computes dy in registers (same arithmetic as bn_bwd_apply_kernel) and accumulates
//   dw[k][c][tap] = sum_pixels dy[pixel][k] * x[b][c][py + tap / 3 - 1][px + tap % 3 - 1]
// straight away: the 205 MB dy tensor (B = 32, 224 x 224) is neither written nor re-read, and the generic first-layer weight
// gradient (built for 3 / 20 -> 64 channels on fp32 MFMA) leaves the end of the LF backward pass.
// Thread = (pixel lane pl = tid >> 3, channel quad k4 = tid & 7): one 16-byte load of dout and y per pixel, CIN * 9 taps of the
// NCHW input (the eight channel quads of a pixel read the same address), CIN * 9 * 4 fp32 accumulators.  A block walks a
// contiguous pixel range 32 pixels at a time; its partial sums are reduced over the pixel lanes with wave shuffles and over the
// four waves through LDS, in a fixed order, and written as one partial row; first_wgrad_rows_kernel sums the rows in fp64.
constexpr int FWG_BLOCKS = 1024;
// KQ = filter quads = K / 4: 8 (late fusion, 32 filters) or 16 (Conv2d(3, 64), the RGB encoder's first layer).
template <int CIN, int PX, bool FIN, int KQ>      // PX = horizontally adjacent pixels per thread (4 when W % 4 == 0, see conv_first_direct_kernel)
__global__ __launch_bounds__(256) void bn_bwd_first_wgrad_kernel(
    const float* __restrict__ y, const float* __restrict__ dout, const float* __restrict__ scale, const float* __restrict__ shift,
    const float* __restrict__ mean, const float* __restrict__ invstd, const float* mdz, const float* mdzx,
    const float* __restrict__ x, float* __restrict__ part, int B, int H, int W, int ppb, const double* __restrict__ rows,
    int nrows, double count, float* __restrict__ dgamma, float* __restrict__ dbeta) {
    constexpr int K = 4 * KQ, NT = CIN * 9, PL = 256 / KQ;
    __shared__ float red[4][KQ][NT * 4];
    __shared__ double fsum[FIN ? 4 : 1][128];
    __shared__ float fm[2][64];
    if (FIN) {                                                  // the BatchNorm finalize step, redone by every block (see above)
        bn_bwd_finalize_in_block(rows, nrows, K, count, dgamma, dbeta, fsum, fm);
        mdz = fm[0];
        mdzx = fm[1];
    }
    const int tid = threadIdx.x, k4 = tid % KQ, pl = tid / KQ, lane = tid & 63, wave = tid >> 6;
    const int HW = H * W, M = B * HW;
    const int m0 = blockIdx.x * ppb, m1 = (m0 + ppb < M) ? m0 + ppb : M;
    const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + k4 * 4), sh = *reinterpret_cast<const f32x4*>(shift + k4 * 4);
    const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + k4 * 4), is = *reinterpret_cast<const f32x4*>(invstd + k4 * 4);
    const f32x4 a1 = *reinterpret_cast<const f32x4*>(mdz + k4 * 4), a2 = *reinterpret_cast<const f32x4*>(mdzx + k4 * 4);
    float acc[NT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[t][e] = 0.f;
    int m = m0 + pl * PX;                                      // first of the thread's PX pixels (same image row: W % PX == 0)
    int b = m / HW, py = (m - b * HW) / W, px = m - b * HW - py * W;          // advanced by 32 PX pixels per iteration below
    for (; m < m1; m += PL * PX) {
        f32x4 g[PX], v[PX];
#pragma unroll
        for (int p = 0; p < PX; ++p) {
            g[p] = *reinterpret_cast<const f32x4*>(dout + (long)(m + p) * K + k4 * 4);
            v[p] = *reinterpret_cast<const f32x4*>(y + (long)(m + p) * K + k4 * 4);
        }
        float xw[CIN][3][PX + 2];
        const float* xb = x + (long)b * CIN * HW + (long)py * W + px;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const bool rok = (unsigned)(py + r - 1) < (unsigned)H;
#pragma unroll
            for (int j = 0; j < PX + 2; ++j) {
                const bool ok = rok && (unsigned)(px + j - 1) < (unsigned)W;
#pragma unroll
                for (int c = 0; c < CIN; ++c) xw[c][r][j] = ok ? xb[(long)c * HW + (r - 1) * W + (j - 1)] : 0.f;
            }
        }
#pragma unroll
        for (int p = 0; p < PX; ++p) {
            f32x4 r;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float z = v[p][e] * sc[e] + sh[e];
                const float dz = z > 0.f ? g[p][e] : 0.f;
                const float xh = (v[p][e] - mu[e]) * is[e];
                r[e] = sc[e] * (dz - a1[e] - xh * a2[e]);
            }
#pragma unroll
            for (int c = 0; c < CIN; ++c)
#pragma unroll
                for (int t = 0; t < 9; ++t)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[c * 9 + t][e] += r[e] * xw[c][t / 3][p + t % 3];
        }
        px += PL * PX;
        while (px >= W) {
            px -= W;
            if (++py == H) {
                py = 0;
                ++b;
            }
        }
    }
    // pixel lanes of a wave: lane = (pl % (64 / KQ)) * KQ + k4
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float a = acc[t][e];
#pragma unroll
            for (int o = KQ; o < 64; o <<= 1) a += __shfl_xor(a, o);
            if (lane < KQ) red[wave][lane][t * 4 + e] = a;
        }
    __syncthreads();
    // part[block][k][c * 9 + tap]
    for (int i = tid; i < K * NT; i += 256) {
        const int k = i / NT, t = i - k * NT;
        const int q = k >> 2, e = k & 3;
        part[(long)blockIdx.x * K * NT + i] = ((red[0][q][t * 4 + e] + red[1][q][t * 4 + e]) + red[2][q][t * 4 + e]) + red[3][q][t * 4 + e];
    }
}

// dw[i] = sum over the partial rows in fp64: a block = 16 outputs x 16 row lanes (row lane q sums rows q, q + 16, ... with
// four independent chains), the sixteen lane sums added in lane order
__global__ __launch_bounds__(256) void first_wgrad_rows_kernel(const float* __restrict__ part, float* __restrict__ dw, int rows, int n) {
    __shared__ double red[16][17];
    const int o = threadIdx.x & 15, q = threadIdx.x >> 4;
    const int i = blockIdx.x * 16 + o;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    if (i < n) {
        int r = q;
        for (; r + 48 < rows; r += 64) {
            s0 += (double)part[(long)r * n + i];
            s1 += (double)part[(long)(r + 16) * n + i];
            s2 += (double)part[(long)(r + 32) * n + i];
            s3 += (double)part[(long)(r + 48) * n + i];
        }
        for (; r < rows; r += 16) s0 += (double)part[(long)r * n + i];
    }
    red[q][o] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (q == 0 && i < n) {
        double t = 0.0;
#pragma unroll
        for (int j = 0; j < 16; ++j) t += red[j][o];
        dw[i] = (float)t;
    }
}

}  // namespace

EGZ_API size_t egz_bn_bwd_first_wgrad_ws_bytes(int C, int K) {
    return egz_bn_relu_pool_bwd_ws_bytes(K) + (size_t)FWG_BLOCKS * K * C * 9 * sizeof(float);
}

// Backward of the FIRST block [Conv2d(C -> 32, 3x3) -> BN(train) -> ReLU] of a narrow stack (late_fusion.py:10-12; C <= 3):
// dgamma / dbeta of the BatchNorm and dw (K, C, 3, 3) of the conv, without materialising the gradient w.r.t. the conv output.
// y: pre-BN conv output [B][H][W][32], dout: gradient w.r.t. the block output (same layout), x: the block input [B][C][H][W]
// (NCHW, as the reference loader yields it).  sums (optional): as for egz_bn_relu_pool_bwd.
EGZ_API int egz_bn_bwd_first_wgrad(const float* y, const float* dout, const float* scale, const float* shift, const float* mean,
                                   const float* invstd, const float* x, float* dw, float* dgamma, float* dbeta, int B, int H,
                                   int W, int C, int K, void* workspace, size_t ws_bytes, const double* sums, int sums_rows,
                                   hipStream_t st) {
    EGZ_CHECK_ARG(y && dout && scale && shift && mean && invstd && x && dw && workspace, "egz_bn_bwd_first_wgrad: null pointer");
    EGZ_CHECK_ARG((K == 32 || K == 64) && C >= 1 && C <= 3, "egz_bn_bwd_first_wgrad: covers C <= 3 input channels and 32 / 64 filters (got %d -> %d)", C, K);
    EGZ_CHECK_ARG(B > 0 && H > 0 && W > 0 && (long)B * H * W * K < (1l << 31), "egz_bn_bwd_first_wgrad: bad shape");
    EGZ_CHECK_ARG(!sums || sums_rows > 0, "egz_bn_bwd_first_wgrad: sums need sums_rows > 0");
    const size_t need = egz_bn_bwd_first_wgrad_ws_bytes(C, K);
    EGZ_CHECK_ARG(ws_bytes >= need, "egz_bn_bwd_first_wgrad: workspace too small (%zu < %zu)", ws_bytes, need);
    const int K4 = K / 4, threads = 256, rpb = threads / K4;
    const long npix = (long)B * H * W;
    int blocks = (int)((npix + rpb - 1) / rpb);
    if (blocks > BWD_BLOCKS) blocks = BWD_BLOCKS;
    if (blocks > ew_cap()) blocks = ew_cap();
    double* part = static_cast<double*>(workspace);
    double* part2 = part + (size_t)BWD_BLOCKS * 2 * K;
    float* mdz = reinterpret_cast<float*>(part2 + (size_t)RED_ROWS * 2 * K);
    float* mdzx = mdz + K;
    float* wpart = reinterpret_cast<float*>(static_cast<char*>(workspace) + egz_bn_relu_pool_bwd_ws_bytes(K));
    const bool fin_fused = sums && sums_rows <= FIN_MAX_ROWS && K <= 64 && fin_in_kernel();
    if (sums) {
        part = const_cast<double*>(sums);
        blocks = sums_rows;
    } else {
        hipLaunchKernelGGL(bn_bwd_reduce_kernel<false>, dim3(blocks), dim3(threads), (size_t)rpb * 2 * K * sizeof(double), st, y, dout,
                           scale, shift, mean, invstd, part, B, H, W, K);
        EGZ_CHECK_LAUNCH("egz_bn_bwd_first_wgrad(reduce)");
    }
    if (!fin_fused) {
        const double* fin = part;
        int nfin = blocks;
        if (blocks > RED_ROWS) {
            int rc = colsum_partial<double>(part, part2, blocks, 2 * K, st);
            if (rc) return rc;
            fin = part2;
            nfin = RED_ROWS;
        }
        hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(egz_cdiv(K, 64)), dim3(64), 0, st, fin, nfin, K, (double)npix, dgamma, dbeta, mdz, mdzx);
        EGZ_CHECK_LAUNCH("egz_bn_bwd_first_wgrad(finalize)");
    }
    // pixels per thread: 4 (W % 4 == 0; C = 3: 2 -- with 108 accumulators the 4-pixel form is left with one wave per SIMD) or 1
    const int pxt = (C == 3) ? ((W % 2 == 0) ? 2 : 1) : ((W % 4 == 0) ? 4 : 1);
    const int step = (256 / (K / 4)) * pxt;                    // pixels per block iteration
    int ppb = (int)((npix + FWG_BLOCKS - 1) / FWG_BLOCKS);
    ppb = (ppb + step - 1) / step * step;
    const int nb = (int)((npix + ppb - 1) / ppb);
#define EGZ_FWG3(CC, PP, QQ)                                                                                                     \
    do {                                                                                                                         \
        if (fin_fused)                                                                                                           \
            hipLaunchKernelGGL((bn_bwd_first_wgrad_kernel<CC, PP, true, QQ>), dim3(nb), dim3(256), 0, st, y, dout, scale, shift, mean, invstd, \
                               (const float*)nullptr, (const float*)nullptr, x, wpart, B, H, W, ppb, sums, sums_rows, (double)npix, dgamma, dbeta); \
        else                                                                                                                     \
            hipLaunchKernelGGL((bn_bwd_first_wgrad_kernel<CC, PP, false, QQ>), dim3(nb), dim3(256), 0, st, y, dout, scale, shift, mean, invstd, \
                               mdz, mdzx, x, wpart, B, H, W, ppb, (const double*)nullptr, 0, 1.0, (float*)nullptr, (float*)nullptr); \
    } while (0)
#define EGZ_FWG2(CC, PP)                                                                                                         \
    do {                                                                                                                         \
        if (K == 32) EGZ_FWG3(CC, PP, 8);                                                                                        \
        else EGZ_FWG3(CC, PP, 16);                                                                                               \
    } while (0)
#define EGZ_FWG(CC)                                                                                                              \
    do {                                                                                                                         \
        if (pxt == 4) EGZ_FWG2(CC, 4);                                                                                           \
        else EGZ_FWG2(CC, 1);                                                                                                    \
    } while (0)
    if (C == 1) EGZ_FWG(1);
    else if (C == 2) EGZ_FWG(2);
    else if (pxt == 2) EGZ_FWG2(3, 2);
    else EGZ_FWG2(3, 1);
#undef EGZ_FWG
#undef EGZ_FWG2
#undef EGZ_FWG3
    EGZ_CHECK_LAUNCH("egz_bn_bwd_first_wgrad(apply + wgrad)");
    const int n = K * C * 9;
    hipLaunchKernelGGL(first_wgrad_rows_kernel, dim3(egz_cdiv(n, 16)), dim3(256), 0, st, wpart, dw, nb, n);
    EGZ_CHECK_LAUNCH("egz_bn_bwd_first_wgrad(rows)");
    return 0;
}

// y2: [2][n] (stream s then stream t), z: [n];  n must be a multiple of 4.
EGZ_API int egz_pairmax_fwd(const float* y2, float* z, long n, hipStream_t st) {
    EGZ_CHECK_ARG(y2 && z && n % 4 == 0, "egz_pairmax_fwd: bad arguments");
    hipLaunchKernelGGL(pairmax_fwd_kernel, dim3(ew_grid(n / 4)), dim3(256), 0, st, y2, z, n / 4);
    EGZ_CHECK_LAUNCH("egz_pairmax_fwd");
    return 0;
}
EGZ_API int egz_pairmax_bwd(const float* y2, const float* dz, float* dy2, long n, unsigned int* absmax, hipStream_t st) {
    EGZ_CHECK_ARG(y2 && dz && dy2 && n % 4 == 0, "egz_pairmax_bwd: bad arguments");
    hipLaunchKernelGGL(pairmax_bwd_kernel, dim3(ew_grid(n / 4)), dim3(256), 0, st, y2, dz, dy2, n / 4, absmax);
    EGZ_CHECK_LAUNCH("egz_pairmax_bwd");
    return 0;
}

// max |x| into an abs-max buffer of egz_absmax_elems() uints (layout: egz_common.h).  Every OTHER entry point that takes an
// `absmax` / `absmax_out` buffer folds this reduction into its own pass and expects the buffer ZERO-FILLED by the caller; this one
// serves callers that hold a bare tensor and zero-fills the buffer itself (stream-ordered).  n must be a multiple of 4.
EGZ_API int egz_absmax_elems(void) { return EGZ_AM_SLOTS * EGZ_AM_STRIDE; }
EGZ_API int egz_absmax(const float* x, long n, unsigned int* absmax, hipStream_t st) {
    EGZ_CHECK_ARG(x && absmax && n > 0 && n % 4 == 0, "egz_absmax: bad arguments");
    hipError_t e = hipMemsetAsync(absmax, 0, sizeof(unsigned int) * EGZ_AM_SLOTS * EGZ_AM_STRIDE, st);
    if (e != hipSuccess) { egz_set_error("egz_absmax: memset failed: %s", hipGetErrorString(e)); return (int)e; }
    hipLaunchKernelGGL(absmax_kernel, dim3(ew_grid(n / 4)), dim3(256), 0, st, x, n / 4, absmax);
    EGZ_CHECK_LAUNCH("egz_absmax");
    return 0;
}

// per-channel sum / sum-of-squares partials of an NHWC tensor [rows][K] in the conv-epilogue format, so that
// egz_bn_finalize can consume them: stat_partial must hold RED_ROWS*2*K doubles.
EGZ_API int egz_channel_stats(const float* x, long rows, int K, double* stat_partial, hipStream_t st);
EGZ_API int egz_channel_stats_rows(void) { return RED_ROWS; }

namespace {
__global__ __launch_bounds__(256) void channel_stats_kernel(const float* __restrict__ x, double* __restrict__ part,
                                                            long rows, int K) {
    __shared__ double red[2][8][33];
    const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
    const int col = blockIdx.x * 32 + cx;
    const long per = (rows + gridDim.y - 1) / gridDim.y;
    const long r0 = blockIdx.y * per, r1 = (r0 + per < rows) ? r0 + per : rows;
    double s1 = 0.0, s2 = 0.0;
    if (col < K)
        for (long r = r0 + ry; r < r1; r += 8) {
            const double v = (double)x[r * K + col];
            s1 += v;
            s2 += v * v;
        }
    red[0][ry][cx] = s1;
    red[1][ry][cx] = s2;
    __syncthreads();
    if (ry < 2 && col < K) {
        double t = 0.0;
#pragma unroll
        for (int i = 0; i < 8; ++i) t += red[ry][i][cx];
        part[((long)blockIdx.y * 2 + ry) * K + col] = t;
    }
}
}  // namespace

EGZ_API int egz_channel_stats(const float* x, long rows, int K, double* stat_partial, hipStream_t st) {
    EGZ_CHECK_ARG(x && stat_partial && rows > 0 && K > 0, "egz_channel_stats: bad arguments");
    hipLaunchKernelGGL(channel_stats_kernel, dim3(egz_cdiv(K, 32), RED_ROWS), dim3(256), 0, st, x, stat_partial, rows, K);
    EGZ_CHECK_LAUNCH("egz_channel_stats");
    return 0;
}

EGZ_API int egz_relu_bwd(const float* out, const float* dout, float* dy, long n, hipStream_t st) {
    EGZ_CHECK_ARG(out && dout && dy && n % 4 == 0, "egz_relu_bwd: bad arguments");
    hipLaunchKernelGGL(relu_bwd_kernel, dim3(ew_grid(n / 4)), dim3(256), 0, st, out, dout, dy, n / 4);
    EGZ_CHECK_LAUNCH("egz_relu_bwd");
    return 0;
}

EGZ_API size_t egz_relu_bwd_bias_ws_bytes(int K) { return ((size_t)BWD_BLOCKS + RED_ROWS) * K * sizeof(double); }

// ReLU backward fused with the bias gradient of the conv that produced `out`: dy = dout*(out>0), db[k] = sum_rows dy.
EGZ_API int egz_relu_bwd_bias(const float* out, const float* dout, float* dy, float* db, long rows, int K,
                              void* workspace, size_t ws_bytes, unsigned int* absmax, hipStream_t st) {
    EGZ_CHECK_ARG(out && dout && dy && db && workspace, "egz_relu_bwd_bias: null pointer");
    EGZ_CHECK_ARG(K % 4 == 0 && K <= 1024, "egz_relu_bwd_bias: K=%d must be a multiple of 4, <= 1024", K);
    EGZ_CHECK_ARG(ws_bytes >= egz_relu_bwd_bias_ws_bytes(K), "egz_relu_bwd_bias: workspace too small");
    const int K4 = K / 4;
    const int threads = 256 > K4 ? 256 : K4;
    const int rpb = threads / K4;
    int blocks = (int)((rows + rpb - 1) / rpb);
    if (blocks > BWD_BLOCKS) blocks = BWD_BLOCKS;
    if (blocks > ew_cap()) blocks = ew_cap();
    double* part = static_cast<double*>(workspace);
    double* part2 = part + (size_t)BWD_BLOCKS * K;
    hipLaunchKernelGGL(relu_bwd_bias_kernel, dim3(blocks), dim3(threads), (size_t)rpb * K * sizeof(double), st, out,
                       dout, dy, part, rows, K, absmax);
    EGZ_CHECK_LAUNCH("egz_relu_bwd_bias");
    const double* fin = part;
    int nfin = blocks;
    if (blocks > RED_ROWS) {
        int rc = colsum_partial<double>(part, part2, blocks, K, st);
        if (rc) return rc;
        fin = part2;
        nfin = RED_ROWS;
    }
    hipLaunchKernelGGL(colsum_final_kernel, dim3(egz_cdiv(K, 64)), dim3(64), 0, st, fin, nfin, K, db);
    EGZ_CHECK_LAUNCH("egz_relu_bwd_bias(final)");
    return 0;
}

// dxu: [B][2H][2W][C] -> dx: [B][H][W][C]
EGZ_API int egz_upsample2x_bwd(const float* dxu, float* dx, int B, int H, int W, int C, hipStream_t st) {
    EGZ_CHECK_ARG(dxu && dx && C % 4 == 0, "egz_upsample2x_bwd: bad arguments");
    hipLaunchKernelGGL(upsample2x_bwd_kernel, dim3(ew_grid((long)B * H * W * (C / 4))), dim3(256), 0, st, dxu, dx, B, H, W, C);
    EGZ_CHECK_LAUNCH("egz_upsample2x_bwd");
    return 0;
}

// out[k] = sum_r x[r][k]  (conv bias gradient).  workspace: RED_ROWS*K doubles.
EGZ_API int egz_colsum(const float* x, long rows, int K, float* out, void* workspace, size_t ws_bytes, hipStream_t st) {
    EGZ_CHECK_ARG(x && out && workspace, "egz_colsum: null pointer");
    EGZ_CHECK_ARG(ws_bytes >= (size_t)RED_ROWS * K * sizeof(double), "egz_colsum: workspace too small");
    double* part = static_cast<double*>(workspace);
    int rc = colsum_partial<float>(x, part, rows, K, st);
    if (rc) return rc;
    hipLaunchKernelGGL(colsum_final_kernel, dim3(egz_cdiv(K, 128)), dim3(128), 0, st, part, RED_ROWS, K, out);
    EGZ_CHECK_LAUNCH("egz_colsum");
    return 0;
}

EGZ_API int egz_nchw_to_nhwc(const float* in, float* out, int B, int C, int H, int W, hipStream_t st) {
    EGZ_CHECK_ARG(in && out, "egz_nchw_to_nhwc: null pointer");
    const long HW = (long)H * W;
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(egz_cdiv(HW, 32), egz_cdiv(C, 32), B), dim3(256), 0, st, in, out, B, C, HW, C);
    EGZ_CHECK_LAUNCH("egz_nchw_to_nhwc");
    return 0;
}
// Same transpose with the channel dimension zero-padded to Cp: (B, C, H, W) -> (B, H, W, Cp).  Lets the 20-channel flow
// stack of the temporal encoder's first conv (SP.py:53) run on the split-half kernels, which want Cin % 32 == 0.
EGZ_API int egz_nchw_to_nhwc_pad(const float* in, float* out, int B, int C, int H, int W, int Cp, hipStream_t st) {
    EGZ_CHECK_ARG(in && out && Cp >= C && C > 0, "egz_nchw_to_nhwc_pad: bad arguments");
    const long HW = (long)H * W;
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(egz_cdiv(HW, 32), egz_cdiv(Cp, 32), B), dim3(256), 0, st, in, out, B, C, HW, Cp);
    EGZ_CHECK_LAUNCH("egz_nchw_to_nhwc_pad");
    return 0;
}

EGZ_API int egz_nhwc_to_nchw(const float* in, float* out, int B, int C, int H, int W, hipStream_t st) {
    EGZ_CHECK_ARG(in && out, "egz_nhwc_to_nchw: null pointer");
    const long HW = (long)H * W;
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(egz_cdiv(HW, 32), egz_cdiv(C, 32), B), dim3(256), 0, st, in, out, B, C, HW);
    EGZ_CHECK_LAUNCH("egz_nhwc_to_nchw");
    return 0;
}

// Stream-ordered device copy / zero fill (hipMemcpyAsync / hipMemsetAsync on the caller's stream): plumbing for the
// host side where the reference writes torch.cat((x_s, x_t), 2) (models/model_SP.py:39) and optimizer.zero_grad()
// (SP.py:138) -- no kernels of the tensor library in the step.
// Both are plain kernels when the buffers are 16-byte aligned (float4 grid-stride; 5-6 TB/s on large buffers, ~2 us on the
// small state / gradient buffers of the AT per-sample step, where the runtime's blit kernels behind hipMemcpyAsync /
// hipMemsetAsync were a quarter of a replayed graph's device time -- profiles/r02_at_sample_loop.txt); the runtime calls
// remain for unaligned buffers.
namespace {
__global__ __launch_bounds__(256) void copy16_kernel(const f32x4* __restrict__ src, f32x4* __restrict__ dst, long n4) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) dst[i] = src[i];
}
__global__ __launch_bounds__(256) void zero16_kernel(f32x4* __restrict__ dst, long n4) {
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) dst[i] = z;
}
inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
}  // namespace
EGZ_API int egz_copy(const float* src, float* dst, long n, hipStream_t st) {
    EGZ_CHECK_ARG(src && dst && n >= 0, "egz_copy: bad arguments");
    if (n == 0) return 0;
    if (al16(src) && al16(dst) && n % 4 == 0) {
        hipLaunchKernelGGL(copy16_kernel, dim3(ew_grid(n / 4)), dim3(256), 0, st, reinterpret_cast<const f32x4*>(src),
                           reinterpret_cast<f32x4*>(dst), n / 4);
        EGZ_CHECK_LAUNCH("egz_copy");
        return 0;
    }
    hipError_t e = hipMemcpyAsync(dst, src, (size_t)n * sizeof(float), hipMemcpyDeviceToDevice, st);
    EGZ_CHECK_ARG(e == hipSuccess, "egz_copy: %s", hipGetErrorString(e));
    return 0;
}
EGZ_API int egz_fill_zero(void* dst, size_t bytes, hipStream_t st) {
    EGZ_CHECK_ARG(dst || bytes == 0, "egz_fill_zero: null pointer");
    if (bytes == 0) return 0;
    if (al16(dst) && bytes % 16 == 0) {
        hipLaunchKernelGGL(zero16_kernel, dim3(ew_grid((long)(bytes / 16))), dim3(256), 0, st, static_cast<f32x4*>(dst), (long)(bytes / 16));
        EGZ_CHECK_LAUNCH("egz_fill_zero");
        return 0;
    }
    hipError_t e = hipMemsetAsync(dst, 0, bytes, st);
    EGZ_CHECK_ARG(e == hipSuccess, "egz_fill_zero: %s", hipGetErrorString(e));
    return 0;
}

namespace {
// out[k] = sum_p part[p][k] for k < nout (row stride `cols`)
__global__ void colsum_final_n_kernel(const double* __restrict__ part, int nparts, int cols, int nout, float* __restrict__ out) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nout) return;
    double s = 0.0;
#pragma unroll 16
    for (int p = 0; p < nparts; ++p) s += part[(long)p * cols + k];
    out[k] = (float)s;
}
}  // namespace

// out[c] = sum over rows of part[row][c] for c < ncols_out; part: [rows][cols] fp64 (e.g. the stat rows of a conv epilogue,
// cols = 2 K, ncols_out = K = the sum plane).  workspace: RED_ROWS * cols doubles.  Fixed summation order.
EGZ_API int egz_colsum_f64(const double* part, int rows, int cols, int ncols_out, float* out, void* workspace,
                           size_t ws_bytes, hipStream_t st) {
    EGZ_CHECK_ARG(part && out && workspace && rows > 0 && cols > 0 && ncols_out > 0 && ncols_out <= cols, "egz_colsum_f64: bad arguments");
    EGZ_CHECK_ARG(ws_bytes >= (size_t)RED_ROWS * cols * sizeof(double), "egz_colsum_f64: workspace too small");
    const double* src = part;
    int n = rows;
    if (rows > RED_ROWS) {
        double* part2 = static_cast<double*>(workspace);
        int rc = colsum_partial<double>(part, part2, rows, cols, st);
        if (rc) return rc;
        src = part2;
        n = RED_ROWS;
    }
    hipLaunchKernelGGL(colsum_final_n_kernel, dim3(egz_cdiv(ncols_out, 128)), dim3(128), 0, st, src, n, cols, ncols_out, out);
    EGZ_CHECK_LAUNCH("egz_colsum_f64");
    return 0;
}
