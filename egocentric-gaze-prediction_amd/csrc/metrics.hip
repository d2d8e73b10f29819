// utils.computeAAEAUC (utils.py:96-140) on the device: one 1024-thread block per sample, no map ever leaves HBM.
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

constexpr int MT = 1024;             // one block per sample, 16 waves
constexpr int NW = MT / 64;

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

__device__ __forceinline__ double shfl_xor_f64(double v, int m) {
    const long long b = __double_as_longlong(v);
    const int lo = __shfl_xor((int)(b & 0xffffffffll), m), hi = __shfl_xor((int)(b >> 32), m);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// Round 5: the pixel loops of the z field are gone.  For a fixed column j the filtered value is z(i, j) = F_j(vrow[i]) with
//   F_j(v) = fl(fl([j == cj] v) gw[R]) (+) fl(fl(m_1 v) c_1) (+) fl(fl(m_2 v) c_2) (+) fl(fl(m_3 v) c_3)      (scipy's order)
// -- every term is a product of non-negative factors and rounding is monotone, so F_j is a non-decreasing function of v >= 0,
// and so is G_j(v) = fl(fl(F_j(v) - zmin) / zden).  Hence, with the SAME doubles scipy's array holds,
//   min z = min_j F_j(min_i vrow[i]),  max z = max_j F_j(max_i vrow[i]),
//   #{(i, j): G_j(vrow[i]) > zg} = sum_j #{i: G_j(vrow[i]) > zg} = sum_j (H - first k with G_j(sorted vrow[k]) > zg):
// 224 binary searches over the sorted row responses instead of two fp64 sweeps over 50,176 pixels (the kernel took 195 us per
// batch of 32 inside LF.trainLate's iteration, profiles/r04_lf_kernel_stats.txt; the count is the same integer).
__global__ __launch_bounds__(MT) void aae_auc_kernel(const float* __restrict__ out, const float* __restrict__ gt, int H, int W,
                                                     const double* __restrict__ gw, int R, double dist,
                                                     double* __restrict__ res) {
    __shared__ double red[3][NW];
    __shared__ float s_gmax[NW];
    __shared__ int s_gidx[NW];
    __shared__ double vrow[256];         // [H] row responses
    __shared__ double vsort[256];        // the same, ascending (padded with +inf)
    __shared__ double ccoef[3 * 256];    // [W][3] weights of the (<= 3) non-zero column terms, scipy order
    __shared__ int cmul[3 * 256];        // [W][3] multiplicities (0 = unused)
    __shared__ double s_bc[8];
    __shared__ double zred[2][256];
    __shared__ int s_cnt[256];

    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = H * W;
    const f32x4* o4 = reinterpret_cast<const f32x4*>(out + (long)b * n);
    const f32x4* g4 = reinterpret_cast<const f32x4*>(gt + (long)b * n);

    // 1. fp64 sums and the first arg-max of the target (W % 4 == 0: the four pixels of a quad share their row).  Fixed order:
    // per thread in index order, then a shuffle tree per wave, then the 16 wave sums in wave order.
    double s = 0.0, si = 0.0, sj = 0.0;
    float gm = -INFINITY;
    int gi = n;
    for (int q = tid; q < n / 4; q += MT) {
        const f32x4 ov = o4[q], gv = g4[q];
        const int idx = 4 * q, i = idx / W, j = idx - i * W;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const double v = (double)ov[e];
            s += v;
            si += v * (double)i;
            sj += v * (double)(j + e);
            if (gv[e] > gm) { gm = gv[e]; gi = idx + e; }      // idx increases within a thread: strict > keeps the first
        }
    }
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) {
        s += shfl_xor_f64(s, m);
        si += shfl_xor_f64(si, m);
        sj += shfl_xor_f64(sj, m);
        const float m2 = __shfl_xor(gm, m);
        const int i2 = __shfl_xor(gi, m);
        if (m2 > gm || (m2 == gm && i2 < gi)) { gm = m2; gi = i2; }
    }
    if (lane == 0) {
        red[0][wave] = s; red[1][wave] = si; red[2][wave] = sj;
        s_gmax[wave] = gm; s_gidx[wave] = gi;
    }
    __syncthreads();
    // 2. centroid, gaze point, angular error
    if (tid == 0) {
        double t0 = 0.0, t1 = 0.0, t2 = 0.0;
        float bm = s_gmax[0];
        int bi = s_gidx[0];
        for (int w = 0; w < NW; ++w) {
            t0 += red[0][w]; t1 += red[1][w]; t2 += red[2][w];
            if (s_gmax[w] > bm || (s_gmax[w] == bm && s_gidx[w] < bi)) { bm = s_gmax[w]; bi = s_gidx[w]; }
        }
        const double c0 = t1 / t0, c1 = t2 / t0;
        const int pi = bi / W, pj = bi - pi * W;
        const double r1x = c0 - 112.0, r1y = c1 - 112.0, r2x = (double)pi - 112.0, r2y = (double)pj - 112.0;
        const double cx = r1y * dist - dist * r2y, cy = dist * r2x - r1x * dist, cz = r1x * r2y - r1y * r2x;
        const double cn = sqrt(cx * cx + cy * cy + cz * cz), dt = r1x * r2x + r1y * r2y + dist * dist;
        s_bc[0] = c0; s_bc[1] = c1; s_bc[2] = (double)pi; s_bc[3] = (double)pj;
        s_bc[4] = atan2(cn, dt) * (180.0 / 3.14159265358979323846);
    }
    __syncthreads();
    const int ci = (int)s_bc[0], cj = (int)s_bc[1], pi = (int)s_bc[2], pj = (int)s_bc[3];

    // 3. the separable response: rows (first pass along axis 0), then per column the <= 3 terms of the second pass
    if (tid < 256) {
        const double v = tid < H ? line_resp(tid, ci, H, 1.0, gw, R) : INFINITY;
        vrow[tid] = v;
        vsort[tid] = v;
    } else if (tid < 256 + W) {
        const int j = tid - 256;
        int k = 0;
        for (int q = 0; q < 3; ++q) { cmul[3 * j + q] = 0; ccoef[3 * j + q] = 0.0; }
        for (int ll = R; ll >= 1; --ll) {
            const int m = (reflect(j + ll, W) == cj ? 1 : 0) + (reflect(j - ll, W) == cj ? 1 : 0);
            if (m && k < 3) { cmul[3 * j + k] = m; ccoef[3 * j + k] = gw[R + ll]; ++k; }
        }
    }
    __syncthreads();
    auto fval = [&](double v, int j) -> double {
        double tmp = (j == cj ? v : 0.0) * gw[R];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int m = cmul[3 * j + q];
            if (m) tmp += ((double)m * v) * ccoef[3 * j + q];      // (a + b) * w with a, b in {v, 0}
        }
        return tmp;
    };
    // 4. sort the row responses (bitonic, 256 slots), then min / max of z from the extreme rows
    for (int k = 2; k <= 256; k <<= 1)
        for (int jj = k >> 1; jj > 0; jj >>= 1) {
            if (tid < 256) {
                const int ixj = tid ^ jj;
                if (ixj > tid) {
                    const double a = vsort[tid], c = vsort[ixj];
                    const bool up = (tid & k) == 0;
                    if ((a > c) == up) { vsort[tid] = c; vsort[ixj] = a; }
                }
            }
            __syncthreads();
        }
    if (tid < 256) {
        const double vmin = vsort[0], vmax = vsort[H - 1];
        zred[0][tid] = tid < W ? fval(vmin, tid) : INFINITY;
        zred[1][tid] = tid < W ? fval(vmax, tid) : -INFINITY;
    }
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (tid < st) {
            zred[0][tid] = fmin(zred[0][tid], zred[0][tid + st]);
            zred[1][tid] = fmax(zred[1][tid], zred[1][tid + st]);
        }
        __syncthreads();
    }
    const double zmin = zred[0][0];
    const double zden = zred[1][0] - zmin;                // (z - z.min()).max()
    const double zg = (fval(vrow[pi], pj) - zmin) / zden;
    // 5. per column: the first sorted row whose normalised value exceeds the one at the gaze point
    if (tid < 256) {
        int cnt = 0;
        if (tid < W) {
            int lo = 0, hi = H;                            // invariant: G(vsort[k]) <= zg for k < lo, > zg for k >= hi
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if ((fval(vsort[mid], tid) - zmin) / zden > zg) hi = mid; else lo = mid + 1;
            }
            cnt = H - lo;
        }
        s_cnt[tid] = cnt;
    }
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
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
    EGZ_CHECK_ARG(((uintptr_t)out | (uintptr_t)gt) % 16 == 0, "egz_aae_auc: maps must be 16-byte aligned");
    hipLaunchKernelGGL(aae_auc_kernel, dim3(B), dim3(MT), 0, st, out, gt, H, W, gw, R, dist, res);
    EGZ_CHECK_LAUNCH("egz_aae_auc");
    return 0;
}
