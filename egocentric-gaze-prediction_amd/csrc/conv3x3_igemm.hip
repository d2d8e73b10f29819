// 3x3 / pad 1 / stride 1 convolution as an implicit GEMM on the exact-f32 matrix cores of gfx950
// (v_mfma_f32_32x32x2_f32).  Replaces every nn.Conv2d(Cin>=32, Cout>=64, 3, padding=1) the SP path
// executes: the 12 wide encoder convs per stream (reference utils.py:70), the shared `fusion` conv
// (models/model_SP.py:10,41) and the 12 decoder convs (models/model_SP.py:13-29).  The same kernel
// computes the data gradient when it is handed dgrad-packed (tap-flipped, transposed) weights.
//
// GEMM view:  Y[m][n] = sum_{tap,c} X[pix(m) + tap][c] * Wp[tap][c][n],   m = (b,y,x) flat, NHWC.
//   block tile 128(m) x BN(n) (BN = 128 / 64, or 32 for the late-fusion widths), K-slice = one tap x 32 channels,
//   4 waves as 2(m) x 2(n), each wave 64 x BN/2 = 2 x (BN/64) MFMA 32x32 tiles (BN = 32: 4(m) x 1(n)),
//   channel counts that are not multiples of 32 are zero-padded in the packed weights and masked in x / y,
//   LDS double-buffered, register-staged global loads (issue next slice -> MFMAs -> write LDS).
// Optional fusions: nearest x2 upsample folded into the input gather (decoder.4/.11/.18/.23),
// bias, ReLU, and per-channel sum / sum-of-squares partials (fp64) for train-mode BatchNorm.
#include "egz_common.h"

namespace {

constexpr int BK = 32;
constexpr int LDA = BK + 4;   // 36 floats: 16-B aligned rows, conflict-free ds_read_b128

enum { EPI_BIAS = 0, EPI_BIAS_RELU = 1, EPI_BIAS_STATS = 2 };

template <int BM, int BN, bool UPS, int EPI>
__global__ __launch_bounds__(256, (BM == 64) ? 4 : 2) void conv3x3_igemm_kernel(
    const float* __restrict__ x, const float* __restrict__ wp, const float* __restrict__ bias,
    float* __restrict__ y, double* __restrict__ stat, int B, int H, int W, int C, int K, int Cp, int Kp) {
    // C / K: real channel counts (strides of x / y); Cp / Kp: the padded extents of the packed weights.
    constexpr int WAVES_N = (BN >= 64) ? 2 : 1, WAVES_M = 4 / WAVES_N;
    constexpr int MR = BM / (32 * WAVES_M);    // 32-row m-tiles per wave (2, or 1 for BN = 32)
    constexpr int NR = BN / (32 * WAVES_N);    // 32-wide n-tiles per wave
    constexpr int WM = BM / WAVES_M;           // m-extent per wave
    constexpr int WN = BN / WAVES_N;           // n-extent per wave
    constexpr int BLD = BN / 32;               // float4 B loads per thread per slice (4, 2, 1)
    constexpr int ALD = BM / 32;               // float4 A loads per thread per slice (4, 2)

    // A: [pixel row][k], B: [output channel row][k]; both k-contiguous with the same padded row stride so both
    // MFMA operands are fetched with conflict-free ds_read_b128 (4 consecutive k of the lane's half per read)
    __shared__ __attribute__((aligned(16))) float As[2 * BM * LDA];
    __shared__ __attribute__((aligned(16))) float Bs[2 * BN * LDA];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (WAVES_N == 2) ? (wave >> 1) : wave, wn = (WAVES_N == 2) ? (wave & 1) : 0;
    const int hl = lane >> 5, l31 = lane & 31;
    const int ntn = Kp / BN;
    const int tile_n = blockIdx.x % ntn, tile_m = blockIdx.x / ntn;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const long HW = (long)H * W;
    const long M = (long)B * HW;
    const int Hs = UPS ? (H >> 1) : H, Ws = UPS ? (W >> 1) : W;

    // ---- per-thread rows (fixed for the whole K loop): 4 A rows (pixels) and BLD B rows (output channels)
    const int a_c4 = tid & 7, r0 = tid >> 3;
    int a_y[ALD], a_x[ALD];
    long a_img[ALD];
#pragma unroll
    for (int j = 0; j < ALD; ++j) {
        const long m = m0 + r0 + 32 * j;
        if (m < M) {
            const long b = m / HW;
            const int rem = (int)(m - b * HW);
            a_y[j] = rem / W;
            a_x[j] = rem - a_y[j] * W;
            a_img[j] = b * (long)Hs * Ws;
        } else {
            a_y[j] = -(1 << 20);
            a_x[j] = 0;
            a_img[j] = 0;
        }
    }

    f32x4 ra[ALD], rb[BLD];
    auto gload = [&](int s) {
        const int cblk = s / 9, tap = s - cblk * 9;
        const int c0 = cblk * BK;
        const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
#pragma unroll
        for (int j = 0; j < ALD; ++j) {
            const int iy = a_y[j] + dy, ix = a_x[j] + dx;
            const bool ok = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W && (c0 + a_c4 * 4 < C);
            const int sy = UPS ? (iy >> 1) : iy, sx = UPS ? (ix >> 1) : ix;
            const float* p = x + ((a_img[j] + (long)sy * Ws + sx) * C + c0 + a_c4 * 4);
            ra[j] = ok ? *reinterpret_cast<const f32x4*>(p) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int j = 0; j < BLD; ++j) {
            const float* p = wp + ((long)(tap * Kp + n0 + r0 + 32 * j) * Cp + c0 + a_c4 * 4);
            rb[j] = *reinterpret_cast<const f32x4*>(p);
        }
    };
    auto lstore = [&](int buf) {
        float* a = As + buf * BM * LDA;
        float* b = Bs + buf * BN * LDA;
#pragma unroll
        for (int j = 0; j < ALD; ++j)
            *reinterpret_cast<f32x4*>(a + (r0 + 32 * j) * LDA + a_c4 * 4) = ra[j];
#pragma unroll
        for (int j = 0; j < BLD; ++j)
            *reinterpret_cast<f32x4*>(b + (r0 + 32 * j) * LDA + a_c4 * 4) = rb[j];
    };

    f32x16 acc[MR][NR];
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NR; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int S = (Cp / BK) * 9;
    gload(0);
    lstore(0);
    __syncthreads();
    for (int s = 0; s < S; ++s) {
        const int buf = s & 1;
        if (s + 1 < S) gload(s + 1);
        const float* Ab = As + buf * BM * LDA + (wm * WM + l31) * LDA + 4 * hl;
        const float* Bb = Bs + buf * BN * LDA + (wn * WN + l31) * LDA + 4 * hl;
        // fragments of k-group q+1 are fetched while the 4*MR*NR MFMAs of group q issue (register double buffer)
        f32x4 af[2][MR], bf[2][NR];
#pragma unroll
        for (int mr = 0; mr < MR; ++mr) af[0][mr] = *reinterpret_cast<const f32x4*>(Ab + mr * 32 * LDA);
#pragma unroll
        for (int nr = 0; nr < NR; ++nr) bf[0][nr] = *reinterpret_cast<const f32x4*>(Bb + nr * 32 * LDA);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (q < 3) {
#pragma unroll
                for (int mr = 0; mr < MR; ++mr)
                    af[(q + 1) & 1][mr] = *reinterpret_cast<const f32x4*>(Ab + mr * 32 * LDA + 8 * (q + 1));
#pragma unroll
                for (int nr = 0; nr < NR; ++nr)
                    bf[(q + 1) & 1][nr] = *reinterpret_cast<const f32x4*>(Bb + nr * 32 * LDA + 8 * (q + 1));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int nr = 0; nr < NR; ++nr)
#pragma unroll
                    for (int mr = 0; mr < MR; ++mr)
                        acc[mr][nr] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q & 1][mr][j], bf[q & 1][nr][j],
                                                                          acc[mr][nr], 0, 0, 0);
            // the other LDS buffer was released by the barrier that ended slice s-1: stage slice s+1 into it
            // half-way through this slice's MFMAs (its global loads were issued ~2k cycles ago)
            if (q == 1 && s + 1 < S) lstore(buf ^ 1);
        }
        __syncthreads();
    }

    // ---- epilogue: bias (+ReLU) (+BN statistic partials), NHWC store
    double* red = reinterpret_cast<double*>(As);   // [WAVES_M][2 (sum, sumsq)][BN]
#pragma unroll
    for (int nr = 0; nr < NR; ++nr) {
        const int col = wn * WN + nr * 32 + l31;
        const bool nok = n0 + col < K;
        const float bz = (bias && nok) ? bias[n0 + col] : 0.f;
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int mr = 0; mr < MR; ++mr) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long m = m0 + wm * WM + mr * 32 + egz_acc_row(r, lane);
                if (m < M && nok) {
                    float v = acc[mr][nr][r] + bz;
                    if (EPI == EPI_BIAS_RELU) v = fmaxf(v, 0.f);
                    y[m * K + n0 + col] = v;
                    if (EPI == EPI_BIAS_STATS) {
                        s1 += (double)v;
                        s2 += (double)v * (double)v;
                    }
                }
            }
        }
        if (EPI == EPI_BIAS_STATS) {
            s1 += __shfl_xor(s1, 32);
            s2 += __shfl_xor(s2, 32);
            if (hl == 0) {
                red[(wm * 2 + 0) * BN + col] = s1;
                red[(wm * 2 + 1) * BN + col] = s2;
            }
        }
    }
    if (EPI == EPI_BIAS_STATS) {
        __syncthreads();
        if (tid < BN && n0 + tid < K) {
            double t1 = 0.0, t2 = 0.0;
#pragma unroll
            for (int w = 0; w < WAVES_M; ++w) {
                t1 += red[(w * 2 + 0) * BN + tid];
                t2 += red[(w * 2 + 1) * BN + tid];
            }
            stat[((long)tile_m * 2 + 0) * K + n0 + tid] = t1;
            stat[((long)tile_m * 2 + 1) * K + n0 + tid] = t2;
        }
    }
}

// wp[(tap*Kp + k)*Cp + c] = w[(k*C + c)*9 + tap], zero in the padding (Cp, Kp = C, K rounded up to 32):
// one k-contiguous (here: input-channel-contiguous) row per output channel, like the A tile's pixel rows
__global__ void pack_fwd_kernel(const float* __restrict__ w, float* __restrict__ wp, int C, int K, int Cp, int Kp) {
    const long n = (long)9 * Cp * Kp;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % Cp);
        const long t = i / Cp;
        const int k = (int)(t % Kp), tap = (int)(t / Kp);
        wp[i] = (c < C && k < K) ? w[((long)k * C + c) * 9 + tap] : 0.f;
    }
}
// dgrad view: dX = conv3x3(dY, Wd); the GEMM's output channel is c and its reduction index is k:
// wp[((8-tap)*Cp + c)*Kp + k] = w[(k*C + c)*9 + tap]  (tap flip + in/out transpose), padded
__global__ void pack_dgrad_kernel(const float* __restrict__ w, float* __restrict__ wp, int C, int K, int Cp, int Kp) {
    const long n = (long)9 * Cp * Kp;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int k = (int)(i % Kp);
        const long t = i / Kp;
        const int c = (int)(t % Cp), tapf = (int)(t / Cp);
        wp[i] = (c < C && k < K) ? w[((long)k * C + c) * 9 + (8 - tapf)] : 0.f;
    }
}

template <int BM, int BN, bool UPS, int EPI>
int launch_igemm(const float* x, const float* wp, const float* bias, float* y, double* stat, int B, int H,
                 int W, int C, int K, hipStream_t st) {
    const long M = (long)B * H * W;
    const int Cp = (C + 31) / 32 * 32, Kp = (K + 31) / 32 * 32;
    const int grid = egz_cdiv(M, BM) * (Kp / BN);
    hipLaunchKernelGGL((conv3x3_igemm_kernel<BM, BN, UPS, EPI>), dim3(grid), dim3(256), 0, st, x, wp, bias, y,
                       stat, B, H, W, C, K, Cp, Kp);
    EGZ_CHECK_LAUNCH("egz_conv3x3_fwd");
    return 0;
}

template <int BM, int BN, bool UPS>
int dispatch_epi(int epi, const float* x, const float* wp, const float* bias, float* y, double* stat, int B,
                 int H, int W, int C, int K, hipStream_t st) {
    switch (epi) {
        case EPI_BIAS: return launch_igemm<BM, BN, UPS, EPI_BIAS>(x, wp, bias, y, stat, B, H, W, C, K, st);
        case EPI_BIAS_RELU: return launch_igemm<BM, BN, UPS, EPI_BIAS_RELU>(x, wp, bias, y, stat, B, H, W, C, K, st);
        default: return launch_igemm<BM, BN, UPS, EPI_BIAS_STATS>(x, wp, bias, y, stat, B, H, W, C, K, st);
    }
}

// Tile choice, from measurements on MI355X at B=32 (tools/bench_conv.py; whole-step A/B in bench.py):
//   * Cout = 64 layers (224^2): the 64x64 tile at 4 blocks/CU wins by 6-15 % (a 128x64 block has half the MFMA
//     work per staged byte and only 18 K-slices to amortise its prologue / epilogue over);
//   * Cout >= 128: in isolation 64x64 is 1-5 % faster (finer tail: 784 / 1568 coarse tiles leave up to 23 % of the
//     last wave of 512 slots idle), but inside the training step -- where the flow / RGB encoders and wgrad || dgrad
//     overlap on separate HIP streams and fill those tails -- the 128x128 tile is 1.5 % faster end to end.
struct Tile { int bm, bn; };
Tile pick_tile(long M, int K, int flags) {
    (void)M;
    if (K % 64 != 0) return {128, 32};   // late-fusion widths (32, 8): one 32-wide n-tile, 4 waves along m
    if (flags & 0x400) return {64, 64};
    if (flags & 0x100) return {128, 64};
    if (flags & 0x200) return {128, K % 128 == 0 ? 128 : 64};
    return (K % 128 == 0) ? Tile{128, 128} : Tile{64, 64};
}

}  // namespace

EGZ_API int egz_conv3x3_stat_rows(int B, int H, int W, int K, int flags) {
    return egz_cdiv((long)B * H * W, pick_tile((long)B * H * W, K, flags).bm);
}

EGZ_API size_t egz_pack_w3x3_elems(int C, int K) {
    return (size_t)9 * ((C + 31) / 32 * 32) * ((K + 31) / 32 * 32);
}

EGZ_API int egz_pack_w3x3_fwd(const float* w, float* wp, int C, int K, hipStream_t st) {
    EGZ_CHECK_ARG(w && wp && C > 0 && K > 0, "egz_pack_w3x3_fwd: bad arguments");
    const int Cp = (C + 31) / 32 * 32, Kp = (K + 31) / 32 * 32;
    const long n = (long)9 * Cp * Kp;
    hipLaunchKernelGGL(pack_fwd_kernel, dim3(egz_cdiv(n, 256) > 4096 ? 4096 : egz_cdiv(n, 256)), dim3(256), 0, st, w,
                       wp, C, K, Cp, Kp);
    EGZ_CHECK_LAUNCH("egz_pack_w3x3_fwd");
    return 0;
}

EGZ_API int egz_pack_w3x3_dgrad(const float* w, float* wp, int C, int K, hipStream_t st) {
    EGZ_CHECK_ARG(w && wp && C > 0 && K > 0, "egz_pack_w3x3_dgrad: bad arguments");
    const int Cp = (C + 31) / 32 * 32, Kp = (K + 31) / 32 * 32;
    const long n = (long)9 * Cp * Kp;
    hipLaunchKernelGGL(pack_dgrad_kernel, dim3(egz_cdiv(n, 256) > 4096 ? 4096 : egz_cdiv(n, 256)), dim3(256), 0, st,
                       w, wp, C, K, Cp, Kp);
    EGZ_CHECK_LAUNCH("egz_pack_w3x3_dgrad");
    return 0;
}

// flags: bit0 = input is nearest-x2 upsampled on the fly (x is [B][H/2][W/2][C]);
//        bits 4-5 = epilogue (0 bias, 1 bias+relu, 2 bias + BN stat partials);
//        0x100 / 0x200 / 0x400 force the 128x64 / 128x128 / 64x64 tile (benchmarking).
EGZ_API int egz_conv3x3_fwd(const float* x, const float* wp, const float* bias, float* y, double* stat_partial,
                            int B, int H, int W, int C, int K, int flags, hipStream_t st) {
    EGZ_CHECK_ARG(x && wp && y, "egz_conv3x3_fwd: null pointer");
    EGZ_CHECK_ARG(B > 0 && H > 0 && W > 0, "egz_conv3x3_fwd: bad shape %dx%dx%d", B, H, W);
    EGZ_CHECK_ARG(C % 4 == 0 && C > 0, "egz_conv3x3_fwd: Cin=%d must be a positive multiple of 4", C);
    EGZ_CHECK_ARG(K > 0 && (K % 64 == 0 || K <= 32), "egz_conv3x3_fwd: Cout=%d must be a multiple of 64 or <= 32", K);
    EGZ_CHECK_ARG(((uintptr_t)x % 16 == 0), "egz_conv3x3_fwd: x must be 16-byte aligned");
    const bool ups = flags & 1;
    EGZ_CHECK_ARG(!ups || (H % 2 == 0 && W % 2 == 0), "egz_conv3x3_fwd: upsampled output must be even");
    const int epi = (flags >> 4) & 3;
    EGZ_CHECK_ARG(epi <= 2, "egz_conv3x3_fwd: bad epilogue %d", epi);
    EGZ_CHECK_ARG(epi != EPI_BIAS_STATS || stat_partial, "egz_conv3x3_fwd: stats epilogue needs stat_partial");
    const Tile t = pick_tile((long)B * H * W, K, flags);
#define EGZ_DISPATCH(BM_, BN_)                                                                                  \
    return ups ? dispatch_epi<BM_, BN_, true>(epi, x, wp, bias, y, stat_partial, B, H, W, C, K, st)            \
               : dispatch_epi<BM_, BN_, false>(epi, x, wp, bias, y, stat_partial, B, H, W, C, K, st)
    if (t.bn == 32) { EGZ_DISPATCH(128, 32); }
    if (t.bm == 64) { EGZ_DISPATCH(64, 64); }
    if (t.bn == 64) { EGZ_DISPATCH(128, 64); }
    EGZ_DISPATCH(128, 128);
#undef EGZ_DISPATCH
}
