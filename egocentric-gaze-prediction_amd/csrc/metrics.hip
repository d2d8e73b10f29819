// utils.computeAAEAUC (utils.py:96-140) on the device: one 256-thread block per sample, no map ever leaves HBM.
//   com   = scipy.ndimage.center_of_mass(output)            -> fp64 sums of o, i*o, j*o in a fixed reduction order
//   gp    = np.unravel_index(target.argmax(), ...)          -> FIRST index of the maximum (integer work: bit-exact)
//   AAE   = atan2(|r1 x r2|, r1 . r2) in degrees, r = (row - 112, col - 112, 112 / tan(pi/6))
//   AUC   = 1 - #{z > z[gp]} / (H*W) with z = gaussian_filter(delta(int(com)), sigma 14), min-max normalised.
// The filtered delta is never materialised: scipy's separable, symmetric 1-D correlation (truncate 4 sigma = 56 taps a
// side, 'reflect' boundary) of a single non-zero has at most three non-zero terms per output sample -- the direct tap and
// the two mirror images -- and they are accumulated here in scipy's own order (centre tap first, then |k| = 56 .. 1) in
// fp64 with the kernel weights computed by the caller exactly as scipy computes them, so z reproduces scipy's array bit
// for bit (tests: against scipy on border / interior centres) and the count is an integer comparison of equal doubles.
// The one place the reference is not reproducible to the bit is its own float32 numpy sum of the map (SIMD-dependent
// summation tree); com therefore agrees to ~1e-7 relative and int(com) can differ only for a centroid within ~1e-5 px
// of an integer.
#include "egz_common.h"

namespace {

constexpr int MT = 256;

__device__ __forceinline__ int reflect(int q, int n) { return q < 0 ? -q - 1 : (q >= n ? 2 * n - 1 - q : q); }

// response at position p of the 1-D filter applied to a line that is `v` at index c and zero elsewhere
__device__ double line_resp(int p, int c, int n, double v, const double* __restrict__ gw, int R) {
    double tmp = (p == c ? v : 0.0) * gw[R];
    for (int ll = R; ll >= 1; --ll) {
        const double a = (reflect(p + ll, n) == c) ? v : 0.0;
        const double b = (reflect(p - ll, n) == c) ? v : 0.0;
        const double ab = a + b;
        if (ab != 0.0) tmp += ab * gw[R + ll];
    }
    return tmp;
}

__global__ __launch_bounds__(MT) void aae_auc_kernel(const float* __restrict__ out, const float* __restrict__ gt, int H, int W,
                                                     const double* __restrict__ gw, int R, double dist,
                                                     double* __restrict__ res) {
    extern __shared__ double sm[];
    double* red = sm;                    // [3][MT]
    double* vrow = sm + 3 * MT;          // [H]
    double* ccoef = vrow + H;            // [W][3]   weights of the (<= 3) non-zero column terms, scipy order
    int* cmul = reinterpret_cast<int*>(ccoef + 3 * W);    // [W][3] multiplicities (0 = unused)
    __shared__ float s_gmax[MT];
    __shared__ int s_gidx[MT];
    __shared__ double s_bc[8];
    __shared__ int s_cnt[MT];

    const int b = blockIdx.x, tid = threadIdx.x, n = H * W;
    const float* o = out + (long)b * n;
    const float* g = gt + (long)b * n;

    // 1. fp64 sums and the first arg-max of the target
    double s = 0.0, si = 0.0, sj = 0.0;
    float gm = -INFINITY;
    int gi = n;
    for (int idx = tid; idx < n; idx += MT) {
        const double v = (double)o[idx];
        const int i = idx / W, j = idx - i * W;
        s += v;
        si += v * (double)i;
        sj += v * (double)j;
        const float t = g[idx];
        if (t > gm) { gm = t; gi = idx; }                 // strided scan: idx increases, strict > keeps the first
    }
    red[tid] = s; red[MT + tid] = si; red[2 * MT + tid] = sj;
    s_gmax[tid] = gm; s_gidx[tid] = gi;
    __syncthreads();
    for (int st = MT / 2; st > 0; st >>= 1) {
        if (tid < st) {
            red[tid] += red[tid + st];
            red[MT + tid] += red[MT + tid + st];
            red[2 * MT + tid] += red[2 * MT + tid + st];
            const float m2 = s_gmax[tid + st];
            const int i2 = s_gidx[tid + st];
            if (m2 > s_gmax[tid] || (m2 == s_gmax[tid] && i2 < s_gidx[tid])) { s_gmax[tid] = m2; s_gidx[tid] = i2; }
        }
        __syncthreads();
    }
    // 2. centroid, gaze point, angular error
    if (tid == 0) {
        const double c0 = red[MT] / red[0], c1 = red[2 * MT] / red[0];
        const int pi = s_gidx[0] / W, pj = s_gidx[0] - pi * W;
        const double r1x = c0 - 112.0, r1y = c1 - 112.0, r2x = (double)pi - 112.0, r2y = (double)pj - 112.0;
        const double cx = r1y * dist - dist * r2y, cy = dist * r2x - r1x * dist, cz = r1x * r2y - r1y * r2x;
        const double cn = sqrt(cx * cx + cy * cy + cz * cz), dt = r1x * r2x + r1y * r2y + dist * dist;
        s_bc[0] = c0; s_bc[1] = c1; s_bc[2] = (double)pi; s_bc[3] = (double)pj;
        s_bc[4] = atan2(cn, dt) * (180.0 / 3.14159265358979323846);
    }
    __syncthreads();
    const int ci = (int)s_bc[0], cj = (int)s_bc[1], pi = (int)s_bc[2], pj = (int)s_bc[3];

    // 3. the separable response: rows (first pass along axis 0), then per column the <= 3 terms of the second pass
    for (int i = tid; i < H; i += MT) vrow[i] = line_resp(i, ci, H, 1.0, gw, R);
    for (int j = tid; j < W; j += MT) {
        int k = 0;
        for (int q = 0; q < 3; ++q) { cmul[3 * j + q] = 0; ccoef[3 * j + q] = 0.0; }
        for (int ll = R; ll >= 1; --ll) {
            const int m = (reflect(j + ll, W) == cj ? 1 : 0) + (reflect(j - ll, W) == cj ? 1 : 0);
            if (m && k < 3) { cmul[3 * j + k] = m; ccoef[3 * j + k] = gw[R + ll]; ++k; }
        }
    }
    __syncthreads();
    auto zval = [&](int i, int j) -> double {
        const double v = vrow[i];
        double tmp = (j == cj ? v : 0.0) * gw[R];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int m = cmul[3 * j + q];
            if (m) tmp += ((double)m * v) * ccoef[3 * j + q];      // (a + b) * w with a, b in {v, 0}
        }
        return tmp;
    };
    // 4. min / max of z, then the count above the (normalised) value at the gaze point
    double zmin = INFINITY, zmax = -INFINITY;
    for (int idx = tid; idx < n; idx += MT) {
        const int i = idx / W, j = idx - i * W;
        const double z = zval(i, j);
        zmin = fmin(zmin, z);
        zmax = fmax(zmax, z);
    }
    red[tid] = zmin; red[MT + tid] = zmax;
    __syncthreads();
    for (int st = MT / 2; st > 0; st >>= 1) {
        if (tid < st) {
            red[tid] = fmin(red[tid], red[tid + st]);
            red[MT + tid] = fmax(red[MT + tid], red[MT + tid + st]);
        }
        __syncthreads();
    }
    zmin = red[0];
    const double zden = red[MT] - zmin;                   // (z - z.min()).max()
    const double zg = (zval(pi, pj) - zmin) / zden;
    int cnt = 0;
    for (int idx = tid; idx < n; idx += MT) {
        const int i = idx / W, j = idx - i * W;
        cnt += ((zval(i, j) - zmin) / zden > zg) ? 1 : 0;
    }
    s_cnt[tid] = cnt;
    __syncthreads();
    for (int st = MT / 2; st > 0; st >>= 1) {
        if (tid < st) s_cnt[tid] += s_cnt[tid + st];
        __syncthreads();
    }
    if (tid == 0) {
        double* r = res + (long)b * 6;
        r[0] = s_bc[4];                 // angular error, degrees
        r[1] = (double)s_cnt[0];        // fp = #{z > z[gp]}
        r[2] = s_bc[2];                 // gaze point row
        r[3] = s_bc[3];                 // gaze point col
        r[4] = s_bc[0];                 // centroid row
        r[5] = s_bc[1];                 // centroid col
    }
}

}  // namespace

// out, gt: (B, H, W) fp32 maps, H = W = 224 (the reference hard-codes 224 / 112, utils.py:104-113).  gw: the 2R+1 weights
// of scipy's gaussian kernel (sigma 14, radius R = 56), dist = 112 / tan(pi/6).  res: (B, 6) doubles =
// (AAE deg, fp count, gaze row, gaze col, centroid row, centroid col); the batch means are the caller's (np.mean).
EGZ_API int egz_aae_auc(const float* out, const float* gt, int B, int H, int W, const double* gw, int R, double dist,
                        double* res, hipStream_t st) {
    EGZ_CHECK_ARG(out && gt && gw && res && B > 0, "egz_aae_auc: null pointer or empty batch");
    EGZ_CHECK_ARG(H == 224 && W == 224, "egz_aae_auc: the reference metric is defined for 224 x 224 maps (got %d x %d)", H, W);
    EGZ_CHECK_ARG(R > 0 && R < H, "egz_aae_auc: bad kernel radius %d", R);
    const size_t shm = (size_t)(3 * MT + H + 3 * W) * sizeof(double) + (size_t)3 * W * sizeof(int);
    hipLaunchKernelGGL(aae_auc_kernel, dim3(B), dim3(MT), shm, st, out, gt, H, W, gw, R, dist, res);
    EGZ_CHECK_LAUNCH("egz_aae_auc");
    return 0;
}
