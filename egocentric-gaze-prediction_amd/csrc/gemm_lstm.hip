// AT module (models/LSTMnet.py:15-37): tanh -> nn.LSTM(512, 512, num_layers=2) -> Linear(512,512) -> ReLU.
// Building blocks, all fp32:
//   * egz_gemm: generic strided GEMM on exact-f32 MFMA (v_mfma_f32_32x32x2_f32), C = alpha*op(A)*op(B) [+ C] [+ bias]
//     [ReLU]; used for the input projections over all T at once, the per-step recurrent product h*W_hh^T,
//     the Linear layer and every backward product (dX = dY*W, dW = dY^T*X).
//   * LSTM cell point-wise forward / backward (gate order i,f,g,o, torch semantics), tanh forward/backward.
// The recurrence is latency-bound (M = batch rows per step): one GEMM + one point-wise launch per step.
#include "egz_common.h"

namespace {

constexpr int GM = 64, GN = 64, GK = 32, GLD = 65;   // odd row stride: conflict-free transposing writes

// C[m][n] (ldc) = sum_k A(m,k) * B(k,n); A(m,k) = A[m*sam + k*sak], B(k,n) = B[k*sbk + n*sbn]
// flags: bit0 accumulate into C, bit1 ReLU after bias.  bias: per-n or null.
__global__ __launch_bounds__(256) void gemm_kernel(const float* __restrict__ A, const float* __restrict__ Bm,
                                                   float* __restrict__ C, const float* __restrict__ bias, int M, int N,
                                                   int K, long sam, long sak, long sbk, long sbn, long ldc, int flags) {
    __shared__ float As[2][GK * GLD];   // [k][m]
    __shared__ float Bs[2][GK * GLD];   // [k][n]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hl = lane >> 5, l31 = lane & 31;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * GM, n0 = blockIdx.x * GN;
    const bool a_kfast = (sak == 1), b_nfast = (sbn == 1);

    float ra[8], rb[8];
    auto gload = [&](int k0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int e = tid + 256 * j;
            int m, k;
            if (a_kfast) { m = e >> 5; k = e & 31; } else { k = e >> 6; m = e & 63; }
            ra[j] = (m0 + m < M && k0 + k < K) ? A[(long)(m0 + m) * sam + (long)(k0 + k) * sak] : 0.f;
            int n, kb;
            if (b_nfast) { kb = e >> 6; n = e & 63; } else { n = e >> 5; kb = e & 31; }
            rb[j] = (n0 + n < N && k0 + kb < K) ? Bm[(long)(k0 + kb) * sbk + (long)(n0 + n) * sbn] : 0.f;
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int e = tid + 256 * j;
            int m, k;
            if (a_kfast) { m = e >> 5; k = e & 31; } else { k = e >> 6; m = e & 63; }
            As[buf][k * GLD + m] = ra[j];
            int n, kb;
            if (b_nfast) { kb = e >> 6; n = e & 63; } else { n = e >> 5; kb = e & 31; }
            Bs[buf][kb * GLD + n] = rb[j];
        }
    };

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int nk = (K + GK - 1) / GK;
    gload(0);
    lstore(0);
    __syncthreads();
    for (int s = 0; s < nk; ++s) {
        const int buf = s & 1;
        if (s + 1 < nk) gload((s + 1) * GK);
        const float* Ab = As[buf] + hl * GLD + wm * 32 + l31;
        const float* Bb = Bs[buf] + hl * GLD + wn * 32 + l31;
#pragma unroll
        for (int t = 0; t < GK / 2; ++t)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Ab[(2 * t) * GLD], Bb[(2 * t) * GLD], acc, 0, 0, 0);
        if (s + 1 < nk) lstore(buf ^ 1);
        __syncthreads();
    }
    const int n = n0 + wn * 32 + l31;
    if (n < N) {
        const float bz = bias ? bias[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * 32 + egz_acc_row(r, lane);
            if (m < M) {
                float v = acc[r] + bz;
                if (flags & 1) v += C[(long)m * ldc + n];
                if (flags & 2) v = fmaxf(v, 0.f);
                C[(long)m * ldc + n] = v;
            }
        }
    }
}

// Fast path of egz_gemm (round 4): M, N multiples of 64, K a multiple of 64, unit stride along ONE index of each operand, 16-byte
// aligned rows.  The generic kernel above fetches every operand element with a scalar, bounds-checked load (16 loads and ~200
// address / predicate instructions per thread per 32-deep slab, one wave per SIMD): 30 us for the 512 x 2048 x 512 products of
// the AT step against an 8 us exact-f32 MFMA floor, and the 17 GEMMs are half of the step's device time.  Here a slab is 64
// deep, every thread moves four float4 per operand per slab, and nothing in the loop is predicated.  Same exact-f32 MFMA,
// same k order within an accumulator: results are bit-identical to the generic kernel.
constexpr int FK = 64;                             // slab depth
// WMW x WNW waves of 32 x 32: block tile (32 WMW) x (32 WNW), 64 WMW WNW threads.  2 x 2 = the 64 x 64 tile; 1 x 1 (one wave per
// block) for products that would otherwise leave CUs empty (egz_gemm picks the tile).
template <int WMW, int WNW, bool A_KFAST, bool B_NFAST>
__device__ __forceinline__ void gemm_fast_body(const float* __restrict__ A, const float* __restrict__ Bm,
                                               float* __restrict__ C, const float* __restrict__ bias, int M, int N,
                                               int K, long lda, long ldb, long ldc, int flags) {
    constexpr int TM = 32 * WMW, TN = 32 * WNW, NTHR = 64 * WMW * WNW;
    constexpr int LDA_ = TM + 4, LDB_ = TN + 4;                // LDS row pitch (16-byte aligned rows, 4-row skew of the banks)
    constexpr int NA = (TM * FK / 4) / NTHR, NB = (TN * FK / 4) / NTHR;       // float4 per thread per slab: 8 / WNW, 8 / WMW
    __shared__ __attribute__((aligned(16))) float As[2][FK * LDA_];   // [k][m]
    __shared__ __attribute__((aligned(16))) float Bs[2][FK * LDB_];   // [k][n]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hl = lane >> 5, l31 = lane & 31;
    const int wm = wave / WNW, wn = wave % WNW;
    const int m0 = blockIdx.y * TM, n0 = blockIdx.x * TN;
    // A_KFAST: A[m * lda + k] (row-major M x K): thread -> (m = e >> 4, k4 = e & 15): one float4 along k (16 per row of 64 k)
    // else     A[k * lda + m] (m contiguous):     thread -> (k = e / (TM / 4), m4 = e % (TM / 4)): one float4 along m
    // TWO slabs in flight in registers (sets 0 / 1): a slab's MFMAs take 0.85 us, a fetch from L2 ~1.5 us -- with one slab in
    // flight every one of the K / 64 iterations waited for its successor's fetch (15 - 21 us for products whose MFMA work is 7 us)
    f32x4 ra[2][NA], rb[2][NB];
    auto gload = [&](int k0, const int set) {
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const int e = tid + NTHR * j;
            if (A_KFAST) ra[set][j] = *reinterpret_cast<const f32x4*>(A + (long)(m0 + (e >> 4)) * lda + k0 + (e & 15) * 4);
            else         ra[set][j] = *reinterpret_cast<const f32x4*>(A + (long)(k0 + e / (TM / 4)) * lda + m0 + (e % (TM / 4)) * 4);
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int e = tid + NTHR * j;
            if (B_NFAST) rb[set][j] = *reinterpret_cast<const f32x4*>(Bm + (long)(k0 + e / (TN / 4)) * ldb + n0 + (e % (TN / 4)) * 4);
            else         rb[set][j] = *reinterpret_cast<const f32x4*>(Bm + (long)(n0 + (e >> 4)) * ldb + k0 + (e & 15) * 4);
        }
    };
    auto lstore = [&](int buf, const int set) {
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const int e = tid + NTHR * j;
            if (A_KFAST) {
#pragma unroll
                for (int q = 0; q < 4; ++q) As[buf][((e & 15) * 4 + q) * LDA_ + (e >> 4)] = ra[set][j][q];
            } else {
                *reinterpret_cast<f32x4*>(&As[buf][(e / (TM / 4)) * LDA_ + (e % (TM / 4)) * 4]) = ra[set][j];
            }
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int e = tid + NTHR * j;
            if (B_NFAST) {
                *reinterpret_cast<f32x4*>(&Bs[buf][(e / (TN / 4)) * LDB_ + (e % (TN / 4)) * 4]) = rb[set][j];
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) Bs[buf][((e & 15) * 4 + q) * LDB_ + (e >> 4)] = rb[set][j][q];
            }
        }
    };
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int nk = K / FK;
    gload(0, 0);
    if (nk > 1) gload(FK, 1);
    lstore(0, 0);                                              // slab 0 -> LDS image 0; set 0 is free again
    if (nk > 2) gload(2 * FK, 0);
    __syncthreads();
    auto slab = [&](const int s, const int set_next) {          // image s & 1 holds slab s; register set `set_next` holds slab s + 1
        const int buf = s & 1;
        const float* Ab = As[buf] + hl * LDA_ + wm * 32 + l31;
        const float* Bb = Bs[buf] + hl * LDB_ + wn * 32 + l31;
#pragma unroll
        for (int t = 0; t < FK / 2; ++t)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Ab[(2 * t) * LDA_], Bb[(2 * t) * LDB_], acc, 0, 0, 0);
        if (s + 1 < nk) lstore(buf ^ 1, set_next);
        if (s + 3 < nk) gload((s + 3) * FK, set_next);          // the set just stored is free: slab s + 3 (s + 2 is in the other set)
        __syncthreads();
    };
    for (int s = 0; s < nk; s += 2) {                           // (two slabs per trip: the register sets keep compile-time names)
        slab(s, 1);
        if (s + 1 < nk) slab(s + 1, 0);
    }
    const int n = n0 + wn * 32 + l31;
    const float bz = bias ? bias[n] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 32 + egz_acc_row(r, lane);
        float v = acc[r] + bz;
        if (flags & 1) v += C[(long)m * ldc + n];
        if (flags & 2) v = fmaxf(v, 0.f);
        C[(long)m * ldc + n] = v;
    }
}

template <int WMW, int WNW, bool A_KFAST, bool B_NFAST>
__global__ __launch_bounds__(64 * WMW * WNW) void gemm_fast_kernel(const float* __restrict__ A, const float* __restrict__ Bm,
                                                                  float* __restrict__ C, const float* __restrict__ bias, int M, int N,
                                                                  int K, long lda, long ldb, long ldc, int flags) {
    gemm_fast_body<WMW, WNW, A_KFAST, B_NFAST>(A, Bm, C, bias, M, N, K, lda, ldb, ldc, flags);
}
// Up to 8 products of ONE shape / stride set in one launch (blockIdx.z picks the operands): the four weight-gradient products
// dgates^T [x | h_prev] of the AT step's backward (models/LSTMnet.py:18 under autograd) are 256 tiles each -- one launch of 1024
// tiles instead of four dependent-looking launches of one tile per CU.
constexpr int GEMM_MAX_BATCH = 8;
struct GemmPtrs {
    const float* A[GEMM_MAX_BATCH];
    const float* B[GEMM_MAX_BATCH];
    float* C[GEMM_MAX_BATCH];
};
template <int WMW, int WNW, bool A_KFAST, bool B_NFAST>
__global__ __launch_bounds__(64 * WMW * WNW) void gemm_fast_batched_kernel(const GemmPtrs p, int M, int N, int K, long lda, long ldb,
                                                                          long ldc, int flags) {
    gemm_fast_body<WMW, WNW, A_KFAST, B_NFAST>(p.A[blockIdx.z], p.B[blockIdx.z], p.C[blockIdx.z], nullptr, M, N, K, lda, ldb, ldc, flags);
}

__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + expf(-x)); }

// gates: [B][4H] pre-activations (i,f,g,o); c_prev/h_out/c_out: [B][H]; act (saved for backward): [B][4H] activated gates
__global__ __launch_bounds__(256) void lstm_cell_fwd_kernel(const float* __restrict__ gates, const float* __restrict__ c_prev,
                                                            float* __restrict__ h_out, float* __restrict__ c_out,
                                                            float* __restrict__ act, int B, int Hd) {
    const int n = B * Hd;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int b = i / Hd, j = i - b * Hd;
        const float* g = gates + (long)b * 4 * Hd;
        const float gi = sigm(g[j]), gf = sigm(g[Hd + j]), gg = tanhf(g[2 * Hd + j]), go = sigm(g[3 * Hd + j]);
        const float c = gf * c_prev[i] + gi * gg;
        c_out[i] = c;
        h_out[i] = go * tanhf(c);
        if (act) {
            float* a = act + (long)b * 4 * Hd;
            a[j] = gi; a[Hd + j] = gf; a[2 * Hd + j] = gg; a[3 * Hd + j] = go;
        }
    }
}

// dh, dc_in: gradients w.r.t. h_t and c_t; writes dgates [B][4H] (pre-activation) and dc_prev.
__global__ __launch_bounds__(256) void lstm_cell_bwd_kernel(const float* __restrict__ act, const float* __restrict__ c,
                                                            const float* __restrict__ c_prev, const float* __restrict__ dh,
                                                            const float* __restrict__ dc_in, float* __restrict__ dgates,
                                                            float* __restrict__ dc_prev, int B, int Hd) {
    const int n = B * Hd;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int b = i / Hd, j = i - b * Hd;
        const float* a = act + (long)b * 4 * Hd;
        const float gi = a[j], gf = a[Hd + j], gg = a[2 * Hd + j], go = a[3 * Hd + j];
        const float tc = tanhf(c[i]);
        const float dhv = dh[i];
        const float dct = (dc_in ? dc_in[i] : 0.f) + dhv * go * (1.f - tc * tc);
        float* d = dgates + (long)b * 4 * Hd;
        d[j] = dct * gg * gi * (1.f - gi);
        d[Hd + j] = dct * c_prev[i] * gf * (1.f - gf);
        d[2 * Hd + j] = dct * gi * (1.f - gg * gg);
        d[3 * Hd + j] = dhv * tc * go * (1.f - go);
        dc_prev[i] = dct * gf;
    }
}

__global__ __launch_bounds__(256) void tanh_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long n) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) y[i] = tanhf(x[i]);
}
// dx = dy * (1 - y^2) with y = tanh(x)
__global__ __launch_bounds__(256) void tanh_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy,
                                                       float* __restrict__ dx, long n) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float t = y[i];
        dx[i] = dy[i] * (1.f - t * t);
    }
}
// out = a + b (element-wise; used to merge the two LSTM bias vectors and to sum gradient paths)
__global__ __launch_bounds__(256) void add_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                  float* __restrict__ out, long n) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) out[i] = a[i] + b[i];
}

inline int grid1d(long n) {
    long g = (n + 255) / 256;
    return (int)(g > 2048 ? 2048 : (g < 1 ? 1 : g));
}

}  // namespace

// C[M][N] (row stride ldc) = op(A)[M][K] * op(B)[K][N]  (+ C if flags&1) (+ bias[n]) (ReLU if flags&2).
// Element (m,k) of op(A) is A[m*sam + k*sak]; element (k,n) of op(B) is B[k*sbk + n*sbn].
EGZ_API int egz_gemm(const float* A, const float* B, float* C, const float* bias, int M, int N, int K, long sam, long sak,
                     long sbk, long sbn, long ldc, int flags, hipStream_t st) {
    EGZ_CHECK_ARG(A && B && C && M > 0 && N > 0 && K > 0, "egz_gemm: bad arguments");
    dim3 grid(egz_cdiv(N, GN), egz_cdiv(M, GM));
    // fast path: whole 64-deep slabs of whole tiles, one unit stride per operand, float4-aligned rows (the AT step's products all qualify)
    const bool a_k = (sak == 1), a_m = (sam == 1), b_n = (sbn == 1), b_k = (sbk == 1);
    const long lda = a_k ? sam : sak, ldb = b_n ? sbk : sbn;
    if (M % GM == 0 && N % GN == 0 && K % FK == 0 && (a_k || a_m) && (b_n || b_k) && lda % 4 == 0 && ldb % 4 == 0 &&
        (reinterpret_cast<uintptr_t>(A) & 15) == 0 && (reinterpret_cast<uintptr_t>(B) & 15) == 0) {
        // tile: 64 x 64 when that fills the chip, else 32 x 32 (a 512 x 512 output is 64 tiles of 64 x 64 on 256 CUs: 18.1 -> 15.1 us;
        // smaller tiles on products that already give every CU a block are SLOWER -- 19.7 -> 24.9 us at 512 x 2048 -- their weight
        // of fetches per MFMA doubles; profiles/r04_ab_notes.txt)
        const long t64 = (long)(M / 64) * (N / 64);
        const int tile = t64 >= 256 ? 0 : 2;
#define EGZ_GF(WMW, WNW)                                                                                                        \
        do {                                                                                                                   \
            const dim3 g(N / (32 * WNW), M / (32 * WMW)), b(64 * WMW * WNW);                                                   \
            if (a_k && b_n)  hipLaunchKernelGGL((gemm_fast_kernel<WMW, WNW, true, true>), g, b, 0, st, A, B, C, bias, M, N, K, lda, ldb, ldc, flags);   \
            else if (a_k)    hipLaunchKernelGGL((gemm_fast_kernel<WMW, WNW, true, false>), g, b, 0, st, A, B, C, bias, M, N, K, lda, ldb, ldc, flags);  \
            else if (b_n)    hipLaunchKernelGGL((gemm_fast_kernel<WMW, WNW, false, true>), g, b, 0, st, A, B, C, bias, M, N, K, lda, ldb, ldc, flags);  \
            else             hipLaunchKernelGGL((gemm_fast_kernel<WMW, WNW, false, false>), g, b, 0, st, A, B, C, bias, M, N, K, lda, ldb, ldc, flags); \
        } while (0)
        if (tile == 0) EGZ_GF(2, 2); else EGZ_GF(1, 1);
#undef EGZ_GF
        EGZ_CHECK_LAUNCH("egz_gemm(fast)");
        return 0;
    }
    hipLaunchKernelGGL(gemm_kernel, grid, dim3(256), 0, st, A, B, C, bias, M, N, K, sam, sak, sbk, sbn, ldc, flags);
    EGZ_CHECK_LAUNCH("egz_gemm");
    return 0;
}

// count <= 8 products C_i = op(A_i) op(B_i) (+ C_i if flags & 1) of one shape and one stride set in ONE launch; A, B, C: HOST arrays
// of `count` device pointers.  Geometries outside the fast path of egz_gemm (whole 64 x 64 x 64 tiles, one unit stride per
// operand, 16-byte aligned rows) run as `count` egz_gemm calls.  No bias, no ReLU.
EGZ_API int egz_gemm_batched(const float* const* A, const float* const* B, float* const* C, int count, int M, int N, int K,
                             long sam, long sak, long sbk, long sbn, long ldc, int flags, hipStream_t st) {
    EGZ_CHECK_ARG(A && B && C && count > 0 && count <= GEMM_MAX_BATCH && M > 0 && N > 0 && K > 0 && !(flags & 2),
                  "egz_gemm_batched: bad arguments (1 <= count <= 8, no ReLU)");
    const bool a_k = (sak == 1), a_m = (sam == 1), b_n = (sbn == 1), b_k = (sbk == 1);
    const long lda = a_k ? sam : sak, ldb = b_n ? sbk : sbn;
    bool fast = M % GM == 0 && N % GN == 0 && K % FK == 0 && (a_k || a_m) && (b_n || b_k) && lda % 4 == 0 && ldb % 4 == 0;
    GemmPtrs p;
    for (int i = 0; i < count; ++i) {
        EGZ_CHECK_ARG(A[i] && B[i] && C[i], "egz_gemm_batched: null operand %d", i);
        fast = fast && (reinterpret_cast<uintptr_t>(A[i]) & 15) == 0 && (reinterpret_cast<uintptr_t>(B[i]) & 15) == 0;
        p.A[i] = A[i]; p.B[i] = B[i]; p.C[i] = C[i];
    }
    if (!fast) {
        for (int i = 0; i < count; ++i) {
            const int rc = egz_gemm(A[i], B[i], C[i], nullptr, M, N, K, sam, sak, sbk, sbn, ldc, flags, st);
            if (rc) return rc;
        }
        return 0;
    }
    const dim3 g(N / 64, M / 64, count), b(256);
    if (a_k && b_n)  hipLaunchKernelGGL((gemm_fast_batched_kernel<2, 2, true, true>), g, b, 0, st, p, M, N, K, lda, ldb, ldc, flags);
    else if (a_k)    hipLaunchKernelGGL((gemm_fast_batched_kernel<2, 2, true, false>), g, b, 0, st, p, M, N, K, lda, ldb, ldc, flags);
    else if (b_n)    hipLaunchKernelGGL((gemm_fast_batched_kernel<2, 2, false, true>), g, b, 0, st, p, M, N, K, lda, ldb, ldc, flags);
    else             hipLaunchKernelGGL((gemm_fast_batched_kernel<2, 2, false, false>), g, b, 0, st, p, M, N, K, lda, ldb, ldc, flags);
    EGZ_CHECK_LAUNCH("egz_gemm_batched");
    return 0;
}

EGZ_API int egz_lstm_cell_fwd(const float* gates, const float* c_prev, float* h_out, float* c_out, float* act, int B,
                              int Hd, hipStream_t st) {
    EGZ_CHECK_ARG(gates && c_prev && h_out && c_out && B > 0 && Hd > 0, "egz_lstm_cell_fwd: bad arguments");
    hipLaunchKernelGGL(lstm_cell_fwd_kernel, dim3(grid1d((long)B * Hd)), dim3(256), 0, st, gates, c_prev, h_out, c_out, act, B, Hd);
    EGZ_CHECK_LAUNCH("egz_lstm_cell_fwd");
    return 0;
}

EGZ_API int egz_lstm_cell_bwd(const float* act, const float* c, const float* c_prev, const float* dh, const float* dc_in,
                              float* dgates, float* dc_prev, int B, int Hd, hipStream_t st) {
    EGZ_CHECK_ARG(act && c && c_prev && dh && dgates && dc_prev && B > 0 && Hd > 0, "egz_lstm_cell_bwd: bad arguments");
    hipLaunchKernelGGL(lstm_cell_bwd_kernel, dim3(grid1d((long)B * Hd)), dim3(256), 0, st, act, c, c_prev, dh, dc_in, dgates,
                       dc_prev, B, Hd);
    EGZ_CHECK_LAUNCH("egz_lstm_cell_bwd");
    return 0;
}

EGZ_API int egz_tanh_fwd(const float* x, float* y, long n, hipStream_t st) {
    EGZ_CHECK_ARG(x && y && n > 0, "egz_tanh_fwd: bad arguments");
    hipLaunchKernelGGL(tanh_fwd_kernel, dim3(grid1d(n)), dim3(256), 0, st, x, y, n);
    EGZ_CHECK_LAUNCH("egz_tanh_fwd");
    return 0;
}
EGZ_API int egz_tanh_bwd(const float* y, const float* dy, float* dx, long n, hipStream_t st) {
    EGZ_CHECK_ARG(y && dy && dx && n > 0, "egz_tanh_bwd: bad arguments");
    hipLaunchKernelGGL(tanh_bwd_kernel, dim3(grid1d(n)), dim3(256), 0, st, y, dy, dx, n);
    EGZ_CHECK_LAUNCH("egz_tanh_bwd");
    return 0;
}
EGZ_API int egz_add(const float* a, const float* b, float* out, long n, hipStream_t st) {
    EGZ_CHECK_ARG(a && b && out && n > 0, "egz_add: bad arguments");
    hipLaunchKernelGGL(add_kernel, dim3(grid1d(n)), dim3(256), 0, st, a, b, out, n);
    EGZ_CHECK_LAUNCH("egz_add");
    return 0;
}
