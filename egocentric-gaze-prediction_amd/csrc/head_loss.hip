// Gaze-map head and the saliency loss (HBM-bound, wave-shuffle reductions):
//   * Conv2d(C -> 1, 1x1) + Sigmoid, forward and backward        (models/model_SP.py:30,32,49; late_fusion.py:13,15,22)
//   * floss: gaze-distance-weighted binary cross-entropy          (floss.py:9-41) -- the per-sample centroid of
//     ALL arg-max pixels and the weight map are built on the device, removing the reference's per-step
//     device->host->device round trip (floss.py:16,11);  plain BCELoss when no weights are requested
//     (SP.py:103-106 `--loss_function` switch).
#include "egz_common.h"


namespace {

__device__ __forceinline__ float sigmoidf_(float z) { return 1.f / (1.f + expf(-z)); }

// LPP = C/4 lanes cooperate on one pixel (float4 each); pixels are [M][C] rows (NHWC with any H, W).
template <int LPP>
__global__ __launch_bounds__(256) void conv1x1_sigmoid_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                  const float* __restrict__ bias, float* __restrict__ out,
                                                                  float* __restrict__ logits, long M) {
    constexpr int C = LPP * 4;
    const int sub = threadIdx.x % LPP;
    const f32x4 wv = *reinterpret_cast<const f32x4*>(w + sub * 4);
    const float bz = bias ? bias[0] : 0.f;
    const long ppb = blockDim.x / LPP;
    for (long m = blockIdx.x * ppb + threadIdx.x / LPP; m < M; m += (long)gridDim.x * ppb) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + m * C + sub * 4);
        float s = v[0] * wv[0] + v[1] * wv[1] + v[2] * wv[2] + v[3] * wv[3];
#pragma unroll
        for (int o = LPP / 2; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (sub == 0) {
            const float z = s + bz;
            if (logits) logits[m] = z;
            out[m] = sigmoidf_(z);
        }
    }
}

// dlogit = dout*out*(1-out); dx[m][c] = dlogit*w[c]; partial dw[c] = sum_m dlogit*x[m][c]; partial db = sum dlogit
// MASK: x is the post-ReLU output of the conv block below (models/model_SP.py:28-30: Conv2d 3x3 -> ReLU -> Conv2d 1x1).  The
// ReLU backward of that block, the column sums of the masked gradient (= its bias gradient) and max |dx| for the f16 scaling
// of its conv backward are taken here, where x is in registers anyway -- the standalone pass (egz_relu_bwd_bias: 1.2 GB per
// step at batch 32, nothing to overlap with at the head of the backward pass) goes away.  mstat: [gridDim.x][C] fp64 partial
// rows, absmax: the abs-max buffer (zero-filled by the caller; egz_common.h).
template <int LPP, bool MASK>
__global__ __launch_bounds__(256) void conv1x1_sigmoid_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                  const float* __restrict__ out, const float* __restrict__ dout,
                                                                  float* __restrict__ dx, double* __restrict__ part, long M,
                                                                  double* __restrict__ mstat, unsigned int* __restrict__ absmax) {
    constexpr int C = LPP * 4;
    __shared__ double red[256 / LPP][C + 1];
    const int sub = threadIdx.x % LPP, grp = threadIdx.x / LPP;
    const f32x4 wv = *reinterpret_cast<const f32x4*>(w + sub * 4);
    const long ppb = blockDim.x / LPP;
    double aw[4] = {0, 0, 0, 0}, ab = 0.0;
    float ms[4] = {0.f, 0.f, 0.f, 0.f}, amx = 0.f;
    for (long m = blockIdx.x * ppb + grp; m < M; m += (long)gridDim.x * ppb) {
        const float o = out[m];
        const float dl = dout[m] * o * (1.f - o);
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + m * C + sub * 4);
        if (dx) {
            f32x4 r;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                r[e] = dl * wv[e];
                if (MASK) {
                    r[e] = v[e] > 0.f ? r[e] : 0.f;
                    ms[e] += r[e];
                    amx = fmaxf(amx, fabsf(r[e]));
                }
            }
            *reinterpret_cast<f32x4*>(dx + m * C + sub * 4) = r;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) aw[e] += (double)(dl * v[e]);
        if (sub == 0) ab += (double)dl;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) red[grp][sub * 4 + e] = aw[e];
    if (sub == 0) red[grp][C] = ab;
    __syncthreads();
    for (int i = threadIdx.x; i < C + 1; i += blockDim.x) {
        double t = 0.0;
        for (int g = 0; g < 256 / LPP; ++g) t += red[g][i];
        part[(long)blockIdx.x * (C + 1) + i] = t;
    }
    if (MASK) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 4; ++e) red[grp][sub * 4 + e] = (double)ms[e];
        __shared__ float s_am[4];
        for (int off = 32; off > 0; off >>= 1) amx = fmaxf(amx, __shfl_xor(amx, off));
        if ((threadIdx.x & 63) == 0) s_am[threadIdx.x >> 6] = amx;
        __syncthreads();
        for (int i = threadIdx.x; i < C; i += blockDim.x) {
            double t = 0.0;
            for (int g = 0; g < 256 / LPP; ++g) t += red[g][i];
            mstat[(long)blockIdx.x * C + i] = t;
        }
        if (threadIdx.x == 0)
            absmax_commit(absmax, blockIdx.x, fmaxf(fmaxf(s_am[0], s_am[1]), fmaxf(s_am[2], s_am[3])));
    }
}

// dw[i] = sum_p part[p][i] (i < C), db = column C.  8 columns x 32 row groups per block: group g sums the partials
// p = g, g + 32, ... (independent loads in flight), the groups are combined in a fixed order -- deterministic and
// ~20 x faster than one thread walking all partials of a column.
__global__ __launch_bounds__(256) void head_bwd_final_kernel(const double* __restrict__ part, int nparts, int C,
                                                             float* __restrict__ dw, float* __restrict__ db) {
    __shared__ double red[32][9];
    const int o = threadIdx.x & 7, g = threadIdx.x >> 3;
    const int i = blockIdx.x * 8 + o;
    double s = 0.0;
    if (i <= C) {
#pragma unroll 8
        for (int p = g; p < nparts; p += 32) s += part[(long)p * (C + 1) + i];
    }
    red[g][o] = s;
    __syncthreads();
    if (g == 0 && i <= C) {
        double t = 0.0;
#pragma unroll
        for (int q = 0; q < 32; ++q) t += red[q][o];
        if (i < C) dw[i] = (float)t;
        else if (db) db[0] = (float)t;
    }
}

// ---------------------------------------------------------------------------------------------- floss
// One 1024-thread block per sample: max over the map, then count / row-sum / col-sum of every pixel equal to it (two
// sweeps of 16-byte loads; the map is L2-resident for the second).  Integer sums are exact, the centroid is their fp64
// quotient -- the same value numpy's rows.mean() / cols.mean() yields (floss.py:24-26).
template <bool VEC>
__global__ __launch_bounds__(1024) void floss_centroid_kernel(const float* __restrict__ target, double* __restrict__ cen,
                                                              int H, int W) {
    __shared__ float smax[16];
    __shared__ unsigned long long scnt[16], srow[16], scol[16];
    const float* t = target + (long)blockIdx.x * H * W;
    const int n = H * W, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float mx = -INFINITY;
    if (VEC) {
        const float4* t4 = reinterpret_cast<const float4*>(t);
        for (int i = threadIdx.x; i < n / 4; i += blockDim.x) {
            const float4 v = t4[i];
            mx = fmaxf(fmaxf(mx, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
        }
    } else {
        for (int i = threadIdx.x; i < n; i += blockDim.x) mx = fmaxf(mx, t[i]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if (lane == 0) smax[wave] = mx;
    __syncthreads();
    mx = smax[0];
#pragma unroll
    for (int w = 1; w < 16; ++w) mx = fmaxf(mx, smax[w]);
    unsigned long long cnt = 0, rs = 0, cs = 0;
    auto hit = [&](int i) {
        cnt += 1;
        rs += (unsigned)(i / W);
        cs += (unsigned)(i % W);
    };
    if (VEC) {
        const float4* t4 = reinterpret_cast<const float4*>(t);
        for (int i = threadIdx.x; i < n / 4; i += blockDim.x) {
            const float4 v = t4[i];
            if (v.x == mx) hit(4 * i);
            if (v.y == mx) hit(4 * i + 1);
            if (v.z == mx) hit(4 * i + 2);
            if (v.w == mx) hit(4 * i + 3);
        }
    } else {
        for (int i = threadIdx.x; i < n; i += blockDim.x)
            if (t[i] == mx) hit(i);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        cnt += __shfl_xor(cnt, o);
        rs += __shfl_xor(rs, o);
        cs += __shfl_xor(cs, o);
    }
    if (lane == 0) {
        scnt[wave] = cnt;
        srow[wave] = rs;
        scol[wave] = cs;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long c = 0, r = 0, q = 0;
        for (int w = 0; w < 16; ++w) {
            c += scnt[w];
            r += srow[w];
            q += scol[w];
        }
        cen[2 * blockIdx.x + 0] = (double)r / (double)c;
        cen[2 * blockIdx.x + 1] = (double)q / (double)c;
    }
}

// weight (fp64 arithmetic in the reference's operation order, rounded to fp32 on store -- floss.py:27-39)
// and the clamped BCE terms; per-block fp64 partial sums of the weighted loss.
template <bool WEIGHTED>
__global__ __launch_bounds__(256) void bce_fwd_kernel(const float* __restrict__ inp, const float* __restrict__ target,
                                                      const double* __restrict__ cen, float* __restrict__ weights,
                                                      double* __restrict__ part, int H, int W, long n) {
    __shared__ double red[4];
    const int HW = H * W;
    double acc = 0.0;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float wgt = 1.f;
        if (WEIGHTED) {
            const int b = (int)(i / HW), rem = (int)(i - (long)b * HW);
            const int r = rem / W, c = rem - r * W;
            const double a = (double)r - cen[2 * b], bb = (double)c - cen[2 * b + 1];
            const double dist = (sqrt(a * a + bb * bb) + 1.0) / (double)W;
            wgt = (float)(1.0 / dist);
            weights[i] = wgt;
        }
        const float x = inp[i], t = target[i];
        const float lx = fmaxf(logf(x), -100.f);
        const float l1x = fmaxf(log1pf(-x), -100.f);
        const float l = ((t - 1.f) * l1x - t * lx) * wgt;
        acc += (double)l;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ void loss_final_kernel(const double* __restrict__ part, int nparts, double denom, float* __restrict__ loss) {
    __shared__ double red[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < nparts; i += blockDim.x) s += part[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = (float)(red[0] / denom);
}

// d loss / d input = gout * w * (x - t) / max((1-x)*x, 1e-12) / N     (aten binary_cross_entropy_backward)
__global__ __launch_bounds__(256) void bce_bwd_kernel(const float* __restrict__ inp, const float* __restrict__ target,
                                                      const float* __restrict__ weights, const float* __restrict__ gout,
                                                      float* __restrict__ dinp, long n) {
    const float go = gout ? gout[0] : 1.f;
    const float invn = 1.f / (float)n;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float x = inp[i], t = target[i];
        float g = (x - t) / fmaxf((1.f - x) * x, 1e-12f);
        if (weights) g *= weights[i];
        dinp[i] = go * g * invn;
    }
}

// mean squared error (nn.MSELoss, AT.py:83,138) forward partials and backward
// tanh_b: the target is tanh(b) (AT.py:138: `criterion(pred, tanh(target))` -- the tanh pass of the reference folded in)
__global__ __launch_bounds__(256) void mse_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                      double* __restrict__ part, long n, int tanh_b) {
    __shared__ double red[4];
    double acc = 0.0;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float d = a[i] - (tanh_b ? tanhf(b[i]) : b[i]);
        acc += (double)(d * d);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(256) void mse_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                      const float* __restrict__ gout, float* __restrict__ da, long n,
                                                      int tanh_b) {
    const float s = (gout ? gout[0] : 1.f) * 2.f / (float)n;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        da[i] = s * (a[i] - (tanh_b ? tanhf(b[i]) : b[i]));
}

// small n (the AT per-sample step: 512 values): partial sums and the final division in ONE block, one launch instead of two.
// Same summation tree as the two-kernel form would use with a single partial (fp64 throughout).
__global__ __launch_bounds__(256) void mse_small_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                        float* __restrict__ loss, long n, int tanh_b) {
    __shared__ double red[4];
    double acc = 0.0;
    for (long i = threadIdx.x; i < n; i += 256) {
        const float d = a[i] - (tanh_b ? tanhf(b[i]) : b[i]);
        acc += (double)(d * d);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) loss[0] = (float)(((red[0] + red[1]) + (red[2] + red[3])) / (double)n);
}

// mse_small_kernel + the gradient w.r.t. a for a unit seed, da = (2 / n) (a - b) (mse_bwd_kernel's arithmetic with gout = 1), in the
// same launch; optionally parks the loss in ring[counter[0] % ring_n] (the AT per-sample loop reads its losses back once per ring:
// the slot index follows the optimizer's device-side step counter, so the captured step needs no copy launch per replay).
__global__ __launch_bounds__(256) void mse_small_grad_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                             float* __restrict__ loss, float* __restrict__ da, long n, int tanh_b,
                                                             float* __restrict__ ring, int ring_n, const int* __restrict__ counter) {
    __shared__ double red[4];
    double acc = 0.0;
    const float s = 1.f * 2.f / (float)n;
    for (long i = threadIdx.x; i < n; i += 256) {
        const float d = a[i] - (tanh_b ? tanhf(b[i]) : b[i]);
        da[i] = s * d;
        acc += (double)(d * d);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float v = (float)(((red[0] + red[1]) + (red[2] + red[3])) / (double)n);
        loss[0] = v;
        if (ring) ring[*static_cast<const volatile int*>(counter) % ring_n] = v;
    }
}

constexpr int LOSS_BLOCKS = 1024;
constexpr int HEAD_BLOCKS = 1024;

}  // namespace

// x: [M][C] NHWC rows, w: (1, C, 1, 1), bias: (1,), out / logits: [M].  C in {8, 16, 32, 64, 128, 256}.
EGZ_API int egz_conv1x1_sigmoid_fwd(const float* x, const float* w, const float* bias, float* out, float* logits,
                                    long M, int C, hipStream_t st) {
    EGZ_CHECK_ARG(x && w && out && M > 0, "egz_conv1x1_sigmoid_fwd: bad arguments");
    long g = (M * (C / 4) + 255) / 256;
    const int grid = (int)(g > 4096 ? 4096 : g);
#define EGZ_HEAD_FWD(L) hipLaunchKernelGGL(conv1x1_sigmoid_fwd_kernel<L>, dim3(grid), dim3(256), 0, st, x, w, bias, out, logits, M)
    switch (C) {
        case 8: EGZ_HEAD_FWD(2); break;
        case 16: EGZ_HEAD_FWD(4); break;
        case 32: EGZ_HEAD_FWD(8); break;
        case 64: EGZ_HEAD_FWD(16); break;
        case 128: EGZ_HEAD_FWD(32); break;
        case 256: EGZ_HEAD_FWD(64); break;
        default: egz_set_error("egz_conv1x1_sigmoid_fwd: unsupported C=%d", C); return (int)hipErrorInvalidValue;
    }
#undef EGZ_HEAD_FWD
    EGZ_CHECK_LAUNCH("egz_conv1x1_sigmoid_fwd");
    return 0;
}

// blocks of the backward launch = rows of its partial-sum workspaces
EGZ_API int egz_conv1x1_sigmoid_bwd_rows(long M, int C) {
    const long g = (M * (C / 4) + 255) / 256;
    return (int)(g > HEAD_BLOCKS ? HEAD_BLOCKS : (g < 1 ? 1 : g));
}
EGZ_API size_t egz_conv1x1_sigmoid_bwd_ws_bytes(int C) { return (size_t)HEAD_BLOCKS * (C + 1) * sizeof(double); }

static int head_bwd_launch(const float* x, const float* w, const float* out, const float* dout, float* dx, float* dw, float* db,
                           long M, int C, void* workspace, size_t ws_bytes, double* mstat, unsigned int* absmax,
                           hipStream_t st, const char* what) {
    if (!(x && w && out && dout && dw && workspace)) {
        egz_set_error("%s: null pointer", what);
        return (int)hipErrorInvalidValue;
    }
    if (ws_bytes < egz_conv1x1_sigmoid_bwd_ws_bytes(C)) {
        egz_set_error("%s: workspace too small", what);
        return (int)hipErrorInvalidValue;
    }
    double* part = static_cast<double*>(workspace);
    const int grid = egz_conv1x1_sigmoid_bwd_rows(M, C);
    const bool mask = mstat != nullptr;
#define EGZ_HEAD_BWD(L)                                                                                                   \
    do {                                                                                                                  \
        if (mask)                                                                                                         \
            hipLaunchKernelGGL((conv1x1_sigmoid_bwd_kernel<L, true>), dim3(grid), dim3(256), 0, st, x, w, out, dout, dx,  \
                               part, M, mstat, absmax);                                                                   \
        else                                                                                                              \
            hipLaunchKernelGGL((conv1x1_sigmoid_bwd_kernel<L, false>), dim3(grid), dim3(256), 0, st, x, w, out, dout, dx, \
                               part, M, mstat, absmax);                                                                   \
    } while (0)
    switch (C) {
        case 8: EGZ_HEAD_BWD(2); break;
        case 16: EGZ_HEAD_BWD(4); break;
        case 32: EGZ_HEAD_BWD(8); break;
        case 64: EGZ_HEAD_BWD(16); break;
        case 128: EGZ_HEAD_BWD(32); break;
        case 256: EGZ_HEAD_BWD(64); break;
        default: egz_set_error("%s: unsupported C=%d", what, C); return (int)hipErrorInvalidValue;
    }
#undef EGZ_HEAD_BWD
    EGZ_CHECK_LAUNCH(what);
    hipLaunchKernelGGL(head_bwd_final_kernel, dim3(egz_cdiv(C + 1, 8)), dim3(256), 0, st, part, grid, C, dw, db);
    EGZ_CHECK_LAUNCH(what);
    return 0;
}

// dout: gradient w.r.t. the sigmoid output [M]; dx may be null.  dw: (1,C,1,1), db: (1,).
EGZ_API int egz_conv1x1_sigmoid_bwd(const float* x, const float* w, const float* out, const float* dout, float* dx,
                                    float* dw, float* db, long M, int C, void* workspace, size_t ws_bytes,
                                    hipStream_t st) {
    return head_bwd_launch(x, w, out, dout, dx, dw, db, M, C, workspace, ws_bytes, nullptr, nullptr, st,
                           "egz_conv1x1_sigmoid_bwd");
}

// The same with the ReLU backward of the block that produced x (x = its post-ReLU output) applied to dx:
// dx = (x > 0) ? dlogit * w : 0;  mstat: [egz_conv1x1_sigmoid_bwd_rows(M, C)][C] fp64 partial rows whose column sums are that
// block's bias gradient (egz_colsum_f64);  absmax: an egz_absmax_elems() buffer, slot 0 = max |dx| on return.
EGZ_API int egz_conv1x1_sigmoid_bwd_masked(const float* x, const float* w, const float* out, const float* dout, float* dx,
                                           float* dw, float* db, double* mstat, unsigned int* absmax, long M, int C,
                                           void* workspace, size_t ws_bytes, hipStream_t st) {
    EGZ_CHECK_ARG(dx && mstat && absmax, "egz_conv1x1_sigmoid_bwd_masked: null pointer");
    int rc = head_bwd_launch(x, w, out, dout, dx, dw, db, M, C, workspace, ws_bytes, mstat, absmax, st,
                             "egz_conv1x1_sigmoid_bwd_masked");
    if (rc) return rc;
    return 0;
}

EGZ_API size_t egz_loss_ws_bytes(int B) { return (size_t)LOSS_BLOCKS * sizeof(double) + (size_t)2 * B * sizeof(double); }

// floss.forward (floss.py:9-13).  inp/target: [B][1][H][W]; weights_out: [B*H*W] (kept for backward);
// loss_out: device scalar.  weighted = 0 gives torch.nn.BCELoss() (weights_out may then be null).
EGZ_API int egz_floss_fwd(const float* inp, const float* target, float* weights_out, float* loss_out, int B, int H,
                          int W, int weighted, void* workspace, size_t ws_bytes, hipStream_t st) {
    EGZ_CHECK_ARG(inp && target && loss_out && workspace, "egz_floss_fwd: null pointer");
    EGZ_CHECK_ARG(!weighted || weights_out, "egz_floss_fwd: weighted loss needs weights_out");
    EGZ_CHECK_ARG(ws_bytes >= egz_loss_ws_bytes(B), "egz_floss_fwd: workspace too small");
    EGZ_CHECK_ARG(!weighted || H == W, "egz_floss_fwd: the reference's weight map assumes square maps (floss.py:18)");
    double* part = static_cast<double*>(workspace);
    double* cen = part + LOSS_BLOCKS;
    const long n = (long)B * H * W;
    long g = (n + 255) / 256;
    const int grid = (int)(g > LOSS_BLOCKS ? LOSS_BLOCKS : g);
    if (weighted) {
        if ((H * W) % 4 == 0 && (reinterpret_cast<uintptr_t>(target) & 15) == 0)
            hipLaunchKernelGGL(floss_centroid_kernel<true>, dim3(B), dim3(1024), 0, st, target, cen, H, W);
        else
            hipLaunchKernelGGL(floss_centroid_kernel<false>, dim3(B), dim3(1024), 0, st, target, cen, H, W);
        EGZ_CHECK_LAUNCH("egz_floss_fwd(centroid)");
        hipLaunchKernelGGL(bce_fwd_kernel<true>, dim3(grid), dim3(256), 0, st, inp, target, cen, weights_out, part, H, W, n);
    } else {
        hipLaunchKernelGGL(bce_fwd_kernel<false>, dim3(grid), dim3(256), 0, st, inp, target, cen, weights_out, part, H, W, n);
    }
    EGZ_CHECK_LAUNCH("egz_floss_fwd(bce)");
    hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(256), 0, st, part, grid, (double)n, loss_out);
    EGZ_CHECK_LAUNCH("egz_floss_fwd(final)");
    return 0;
}

// grad_out: device scalar (d L / d loss) or null for 1.  weights may be null (plain BCE).
EGZ_API int egz_floss_bwd(const float* inp, const float* target, const float* weights, const float* grad_out,
                          float* dinp, long n, hipStream_t st) {
    EGZ_CHECK_ARG(inp && target && dinp && n > 0, "egz_floss_bwd: bad arguments");
    long g = (n + 255) / 256;
    hipLaunchKernelGGL(bce_bwd_kernel, dim3((int)(g > 4096 ? 4096 : g)), dim3(256), 0, st, inp, target, weights, grad_out, dinp, n);
    EGZ_CHECK_LAUNCH("egz_floss_bwd");
    return 0;
}

EGZ_API int egz_mse_fwd(const float* a, const float* b, float* loss_out, long n, void* workspace, size_t ws_bytes,
                        int tanh_b, hipStream_t st) {
    EGZ_CHECK_ARG(a && b && loss_out && workspace && n > 0, "egz_mse_fwd: bad arguments");
    EGZ_CHECK_ARG(ws_bytes >= LOSS_BLOCKS * sizeof(double), "egz_mse_fwd: workspace too small");
    if (n <= 4096) {
        hipLaunchKernelGGL(mse_small_kernel, dim3(1), dim3(256), 0, st, a, b, loss_out, n, tanh_b);
        EGZ_CHECK_LAUNCH("egz_mse_fwd(small)");
        return 0;
    }
    double* part = static_cast<double*>(workspace);
    long g = (n + 255) / 256;
    const int grid = (int)(g > LOSS_BLOCKS ? LOSS_BLOCKS : g);
    hipLaunchKernelGGL(mse_fwd_kernel, dim3(grid), dim3(256), 0, st, a, b, part, n, tanh_b);
    EGZ_CHECK_LAUNCH("egz_mse_fwd");
    hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(256), 0, st, part, grid, (double)n, loss_out);
    EGZ_CHECK_LAUNCH("egz_mse_fwd(final)");
    return 0;
}
// nn.MSELoss forward AND its gradient for a unit seed in one single-block launch (n <= 4096: AT.py:138-141 on one (1, 1, 512)
// sample: loss = criterion(pred, tanh(target)); loss.backward()).  loss_out = mean((a - b')^2), da = (2 / n) (a - b') with b' = b or
// tanh(b): bit-identical to egz_mse_fwd + egz_mse_bwd(grad_out = 1).  ring (optional): the loss is also stored to
// ring[counter[0] % ring_n] (counter = a device int, e.g. the step counter of egz_adam_step_dev).
EGZ_API int egz_mse_fwd_grad(const float* a, const float* b, float* loss_out, float* da, long n, int tanh_b, float* ring,
                             int ring_n, const int* counter, hipStream_t st) {
    EGZ_CHECK_ARG(a && b && loss_out && da && n > 0 && n <= 4096, "egz_mse_fwd_grad: bad arguments (n <= 4096)");
    EGZ_CHECK_ARG(!ring || (ring_n > 0 && counter), "egz_mse_fwd_grad: a ring needs its size and a counter");
    hipLaunchKernelGGL(mse_small_grad_kernel, dim3(1), dim3(256), 0, st, a, b, loss_out, da, n, tanh_b, ring, ring_n, counter);
    EGZ_CHECK_LAUNCH("egz_mse_fwd_grad");
    return 0;
}
EGZ_API int egz_mse_bwd(const float* a, const float* b, const float* grad_out, float* da, long n, int tanh_b,
                        hipStream_t st) {
    EGZ_CHECK_ARG(a && b && da && n > 0, "egz_mse_bwd: bad arguments");
    long g = (n + 255) / 256;
    hipLaunchKernelGGL(mse_bwd_kernel, dim3((int)(g > 4096 ? 4096 : g)), dim3(256), 0, st, a, b, grad_out, da, n, tanh_b);
    EGZ_CHECK_LAUNCH("egz_mse_bwd");
    return 0;
}
