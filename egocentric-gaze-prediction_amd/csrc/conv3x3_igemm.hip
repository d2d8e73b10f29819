// 3x3 / pad 1 / stride 1 convolution as an implicit GEMM on the exact-f32 matrix cores of gfx950
// (v_mfma_f32_32x32x2_f32).  Replaces every nn.Conv2d(Cin>=32, Cout, 3, padding=1) the SP / LF paths execute:
// the 12 wide encoder convs per stream (reference utils.py:70), the shared `fusion` conv
// (models/model_SP.py:10,41), the 12 decoder convs (models/model_SP.py:13-29) and the late-fusion convs
// (models/late_fusion.py:10-12).  The same kernel computes the data gradient when it is handed dgrad-packed
// (tap-flipped, transposed) weights.
//
// GEMM view:  Y[m][n] = sum_{tap,c} X[pix(m) + tap][c] * Wp[tap][n][c],   m = flat pixel index, NHWC.
//   block tile BM(m) x BN(n) in {128x128, 128x64, 64x64, 128x32}, K-slice = one tap x 32 channels,
//   4 waves (64 lanes), each wave MR x NR MFMA 32x32 tiles; A = [pixel][k] and B = [out-channel][k] tiles in LDS,
//   both k-contiguous with a 36-float row stride so both operands are fetched with conflict-free ds_read_b128;
//   LDS double-buffered, register-staged global loads running two slices ahead (two register sets),
//   fragment reads register-double-buffered against the MFMAs.
//   Channel counts that are not multiples of 32 are zero-padded in the packed weights and masked in x / y.
//
// Gather modes (template MODE) -- the nearest x2 upsample of the decoder (nn.Upsample, model_SP.py:16,20,24,27)
// never materialises:
//   0 PLAIN     3x3 taps on x[B][H][W][C]
//   1 UPS_FOLD  3x3 taps on the virtual upsampled image, source pixel = (iy>>1, ix>>1)   (9/9 of the MACs)
//   2 UPS_PHASE forward of [upsample -> conv] as four 2x2 convolutions on the LOW-res input, one per output
//               phase (py,px), with pre-summed weights: 4/9 of the MACs, writes y[2y+py][2x+px]
//   3 UPS_DGRAD data gradient of [upsample -> conv] w.r.t. the LOW-res input as one 4x4 / stride-2 gather over the
//               hi-res dY (16 taps on a quarter of the pixels = 4/9 of the MACs); replaces dgrad + 2x2-sum
// Epilogues: bias, bias+ReLU, bias + per-channel sum / sum-of-squares partials (fp64) for train-mode BatchNorm.
#include "egz_common.h"
#include <type_traits>

namespace {

constexpr int BK = 32;
constexpr int LDA = BK + 4;   // 36 floats: 16-B aligned rows, conflict-free ds_read_b128

enum { EPI_BIAS = 0, EPI_BIAS_RELU = 1, EPI_BIAS_STATS = 2 };
enum { PLAIN = 0, UPS_FOLD = 1, UPS_PHASE = 2, UPS_DGRAD = 3 };

// H, W are always the HI-res (conv output) dims of the layer for the UPS_* modes.
template <int BM, int BN, int MODE, int EPI>
__global__ __launch_bounds__(256, (BM == 64) ? 4 : 2) void conv3x3_igemm_kernel(
    const float* __restrict__ x, const float* __restrict__ wp, const float* __restrict__ bias,
    float* __restrict__ y, double* __restrict__ stat, int B, int H, int W, int C, int K, int Cp, int Kp) {
    // C / K: real channel counts (strides of x / y); Cp / Kp: the padded extents of the packed weights.
    constexpr int WAVES_N = (BN >= 64) ? 2 : 1, WAVES_M = 4 / WAVES_N;
    constexpr int MR = BM / (32 * WAVES_M);    // 32-row m-tiles per wave
    constexpr int NR = BN / (32 * WAVES_N);    // 32-wide n-tiles per wave
    constexpr int WM = BM / WAVES_M;           // m-extent per wave
    constexpr int WN = BN / WAVES_N;           // n-extent per wave
    constexpr int BLD = BN / 32;               // float4 B loads per thread per slice (4, 2, 1)
    constexpr int ALD = BM / 32;               // float4 A loads per thread per slice (4, 2)
    constexpr int NTAP = (MODE == UPS_PHASE) ? 4 : (MODE == UPS_DGRAD) ? 16 : 9;

    __shared__ __attribute__((aligned(16))) float As[2 * BM * LDA];
    __shared__ __attribute__((aligned(16))) float Bs[2 * BN * LDA];
    __shared__ long Ro[BM];                    // output element offset of each tile row (-1: row out of range)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (WAVES_N == 2) ? (wave >> 1) : wave, wn = (WAVES_N == 2) ? (wave & 1) : 0;
    const int hl = lane >> 5, l31 = lane & 31;
    const int ntn = Kp / BN;
    const int tile_n = blockIdx.x % ntn, tile_m = blockIdx.x / ntn;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int phase = (MODE == UPS_PHASE) ? blockIdx.y : 0, py = phase >> 1, px = phase & 1;
    // row domain (what m enumerates) and gather-source dims
    const int Hr = (MODE >= UPS_PHASE) ? (H >> 1) : H, Wr = (MODE >= UPS_PHASE) ? (W >> 1) : W;
    const int Hg = (MODE == UPS_FOLD || MODE == UPS_PHASE) ? (H >> 1) : H;
    const int Wg = (MODE == UPS_FOLD || MODE == UPS_PHASE) ? (W >> 1) : W;
    const long HWr = (long)Hr * Wr;
    const long M = (long)B * HWr;

    // ---- per-thread rows (fixed for the whole K loop): ALD A rows (pixels) and BLD B rows (output channels)
    const int a_c4 = tid & 7, r0 = tid >> 3;
    int a_y[ALD], a_x[ALD];
    long a_img[ALD];
#pragma unroll
    for (int j = 0; j < ALD; ++j) {
        const long m = m0 + r0 + 32 * j;
        long off = -1;
        if (m < M) {
            const long b = m / HWr;
            const int rem = (int)(m - b * HWr);
            a_y[j] = rem / Wr;
            a_x[j] = rem - a_y[j] * Wr;
            a_img[j] = b * (long)Hg * Wg;
            off = (MODE == UPS_PHASE) ? ((b * H + 2 * a_y[j] + py) * (long)W + 2 * a_x[j] + px) * K : m * K;
        } else {
            a_y[j] = -(1 << 20);
            a_x[j] = 0;
            a_img[j] = 0;
        }
        if (a_c4 == 0) Ro[r0 + 32 * j] = off;
    }

    // two register sets: the global loads run TWO slices ahead of the MFMAs (slice s+2 is in flight while slice
    // s+1 is being staged into LDS and slice s is multiplied), so a wave that is alone on its SIMD -- the tail of
    // a launch -- still hides the full HBM latency
    f32x4 ra[2][ALD], rb[2][BLD];
    auto gload_a = [&](int s, auto SETC) {
        constexpr int SET = decltype(SETC)::value;
        const int cblk = s / NTAP, tap = s - cblk * NTAP;
        const int c0 = cblk * BK;
        int dy, dx;
        if (MODE == UPS_PHASE) {
            dy = (tap >> 1) + py - 1;
            dx = (tap & 1) + px - 1;
        } else if (MODE == UPS_DGRAD) {
            dy = (tap >> 2) - 1;
            dx = (tap & 3) - 1;
        } else {
            dy = tap / 3 - 1;
            dx = tap - (tap / 3) * 3 - 1;
        }
#pragma unroll
        for (int j = 0; j < ALD; ++j) {
            int iy, ix;
            if (MODE == UPS_DGRAD) {
                iy = 2 * a_y[j] + dy;
                ix = 2 * a_x[j] + dx;
            } else {
                iy = a_y[j] + dy;
                ix = a_x[j] + dx;
            }
            // bounds are those of the image the taps walk on: the virtual hi-res image for UPS_FOLD
            const int Hb = (MODE == UPS_FOLD) ? H : Hg, Wb = (MODE == UPS_FOLD) ? W : Wg;
            const bool ok = (unsigned)iy < (unsigned)Hb && (unsigned)ix < (unsigned)Wb && (c0 + a_c4 * 4 < C);
            const int sy = (MODE == UPS_FOLD) ? (iy >> 1) : iy, sx = (MODE == UPS_FOLD) ? (ix >> 1) : ix;
            const float* p = x + ((a_img[j] + (long)sy * Wg + sx) * C + c0 + a_c4 * 4);
            ra[SET][j] = ok ? *reinterpret_cast<const f32x4*>(p) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto gload_b = [&](int s, auto SETC) {
        constexpr int SET = decltype(SETC)::value;
        const int cblk = s / NTAP, tap = s - cblk * NTAP;
        const int c0 = cblk * BK;
#pragma unroll
        for (int j = 0; j < BLD; ++j) {
            const float* p = wp + ((long)((phase * NTAP + tap) * Kp + n0 + r0 + 32 * j) * Cp + c0 + a_c4 * 4);
            rb[SET][j] = *reinterpret_cast<const f32x4*>(p);
        }
    };
    auto gload = [&](int s, auto SETC) {
        gload_a(s, SETC);
        gload_b(s, SETC);
    };
    auto lstore = [&](int buf, auto SETC) {
        constexpr int SET = decltype(SETC)::value;
        float* a = As + buf * BM * LDA;
        float* b = Bs + buf * BN * LDA;
#pragma unroll
        for (int j = 0; j < ALD; ++j)
            *reinterpret_cast<f32x4*>(a + (r0 + 32 * j) * LDA + a_c4 * 4) = ra[SET][j];
#pragma unroll
        for (int j = 0; j < BLD; ++j)
            *reinterpret_cast<f32x4*>(b + (r0 + 32 * j) * LDA + a_c4 * 4) = rb[SET][j];
    };

    f32x16 acc[MR][NR];
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NR; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int S = (Cp / BK) * NTAP;
    using Set0 = std::integral_constant<int, 0>;
    using Set1 = std::integral_constant<int, 1>;
    gload(0, Set0{});
    lstore(0, Set0{});
    if (S > 1) gload(1, Set1{});
    __syncthreads();
    // Fragment registers persist across slices: group q lives in set q&1, and the q=0 fragments of slice s+1 are
    // fetched during the q=3 MFMAs of slice s, so the matrix pipe never drains at a slice boundary.
    f32x4 af[2][MR], bf[2][NR];
    {
        const float* Ab = As + (wm * WM + l31) * LDA + 4 * hl;
        const float* Bb = Bs + (wn * WN + l31) * LDA + 4 * hl;
#pragma unroll
        for (int mr = 0; mr < MR; ++mr) af[0][mr] = *reinterpret_cast<const f32x4*>(Ab + mr * 32 * LDA);
#pragma unroll
        for (int nr = 0; nr < NR; ++nr) bf[0][nr] = *reinterpret_cast<const f32x4*>(Bb + nr * 32 * LDA);
    }
    // One slice = 4 k-groups of 4*MR*NR MFMAs on LDS buffer s&1.  Interleaved in their shadow (branch-free, one basic
    // block): the address arithmetic + issue of the global loads of slice s+2 (into the register set slice s
    // vacated), the staging of slice s+1 into the other LDS buffer, ONE barrier right after that staging (RAW for
    // slice s+1; WAR is safe because every read of buffer s&1 was issued before this barrier of the same slice),
    // and the first fragments of slice s+1.  Past the end the prefetches re-read the last slice (clamped index).
    auto slice = [&](int s, auto CURC) {
        constexpr int CUR = decltype(CURC)::value;
        const int buf = s & 1;
        const int sp = (s + 2 < S) ? s + 2 : S - 1;
        const float* Ab = As + buf * BM * LDA + (wm * WM + l31) * LDA + 4 * hl;
        const float* Bb = Bs + buf * BN * LDA + (wn * WN + l31) * LDA + 4 * hl;
        const float* An = As + (buf ^ 1) * BM * LDA + (wm * WM + l31) * LDA + 4 * hl;
        const float* Bn = Bs + (buf ^ 1) * BN * LDA + (wn * WN + l31) * LDA + 4 * hl;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (q < 3) {
#pragma unroll
                for (int mr = 0; mr < MR; ++mr)
                    af[(q + 1) & 1][mr] = *reinterpret_cast<const f32x4*>(Ab + mr * 32 * LDA + 8 * (q + 1));
#pragma unroll
                for (int nr = 0; nr < NR; ++nr)
                    bf[(q + 1) & 1][nr] = *reinterpret_cast<const f32x4*>(Bb + nr * 32 * LDA + 8 * (q + 1));
            } else {
#pragma unroll
                for (int mr = 0; mr < MR; ++mr) af[0][mr] = *reinterpret_cast<const f32x4*>(An + mr * 32 * LDA);
#pragma unroll
                for (int nr = 0; nr < NR; ++nr) bf[0][nr] = *reinterpret_cast<const f32x4*>(Bn + nr * 32 * LDA);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int nr = 0; nr < NR; ++nr)
#pragma unroll
                    for (int mr = 0; mr < MR; ++mr)
                        acc[mr][nr] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q & 1][mr][j], bf[q & 1][nr][j],
                                                                          acc[mr][nr], 0, 0, 0);
            if (q == 0) gload_a(sp, std::integral_constant<int, CUR>{});
            if (q == 1) gload_b(sp, std::integral_constant<int, CUR>{});
            if (q == 2) {
                lstore(buf ^ 1, std::integral_constant<int, CUR ^ 1>{});
                __syncthreads();
            }
        }
    };
    for (int s = 0; s < S; s += 2) {
        slice(s, Set0{});
        if (s + 1 < S) slice(s + 1, Set1{});
    }
    __syncthreads();      // all fragment reads done before the epilogue reuses As

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
                const long off = Ro[wm * WM + mr * 32 + egz_acc_row(r, lane)];
                if (off >= 0 && nok) {
                    float v = acc[mr][nr][r] + bz;
                    if (EPI == EPI_BIAS_RELU) v = fmaxf(v, 0.f);
                    y[off + n0 + col] = v;
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
            const long srow = (long)phase * (gridDim.x / ntn) + tile_m;
            stat[(srow * 2 + 0) * K + n0 + tid] = t1;
            stat[(srow * 2 + 1) * K + n0 + tid] = t2;
        }
    }
}

// ---------------------------------------------------------------------------------------------- weight packing
// All packed layouts hold one reduction-index-contiguous row per GEMM output channel (like the A tile's pixel
// rows), zero in the padding (Cp, Kp = C, K rounded up to 32).
// forward : wp[(tap*Kp + k)*Cp + c] = w[k][c][tap]
__global__ void pack_fwd_kernel(const float* __restrict__ w, float* __restrict__ wp, int C, int K, int Cp, int Kp) {
    const long n = (long)9 * Cp * Kp;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % Cp);
        const long t = i / Cp;
        const int k = (int)(t % Kp), tap = (int)(t / Kp);
        wp[i] = (c < C && k < K) ? w[((long)k * C + c) * 9 + tap] : 0.f;
    }
}
// dgrad   : dX = conv3x3(dY, Wd); the GEMM's output channel is c and its reduction index is k:
//           wp[((8-tap)*Cp + c)*Kp + k] = w[k][c][tap]  (tap flip + in/out transpose)
__global__ void pack_dgrad_kernel(const float* __restrict__ w, float* __restrict__ wp, int C, int K, int Cp, int Kp) {
    const long n = (long)9 * Cp * Kp;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int k = (int)(i % Kp);
        const long t = i / Kp;
        const int c = (int)(t % Cp), tapf = (int)(t / Cp);
        wp[i] = (c < C && k < K) ? w[((long)k * C + c) * 9 + (8 - tapf)] : 0.f;
    }
}
// [upsample x2 -> conv3x3] collapses, per output phase p (p = Y&1 or X&1) and 2-tap index a, to the sum of the
// 3x3 taps r in R(p,a):  R(0,0)={0}, R(0,1)={1,2}, R(1,0)={0,1}, R(1,1)={2}   (the two hi-res taps that land on the
// same low-res pixel are merged).  Low-res source offset of tap a in phase p: a + p - 1.
__device__ __forceinline__ float weff(const float* __restrict__ w9, int py, int a, int px, int b) {
    const int rlo = (py == 0) ? (a == 0 ? 0 : 1) : (a == 0 ? 0 : 2), rhi = (py == 0) ? (a == 0 ? 0 : 2) : (a == 0 ? 1 : 2);
    const int slo = (px == 0) ? (b == 0 ? 0 : 1) : (b == 0 ? 0 : 2), shi = (px == 0) ? (b == 0 ? 0 : 2) : (b == 0 ? 1 : 2);
    float s = 0.f;
    for (int r = rlo; r <= rhi; ++r)
        for (int q = slo; q <= shi; ++q) s += w9[r * 3 + q];
    return s;
}
// phase fwd: wp[((phase*4 + a*2+b)*Kp + k)*Cp + c] = Weff[py][px][a][b][k][c]
__global__ void pack_ups_fwd_kernel(const float* __restrict__ w, float* __restrict__ wp, int C, int K, int Cp, int Kp) {
    const long n = (long)16 * Cp * Kp;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % Cp);
        const long t = i / Cp;
        const int k = (int)(t % Kp), pt = (int)(t / Kp);
        const int phase = pt >> 2, tap = pt & 3;
        wp[i] = (c < C && k < K) ? weff(w + ((long)k * C + c) * 9, phase >> 1, tap >> 1, phase & 1, tap & 1) : 0.f;
    }
}
// ups dgrad: dx_low[y][x] = sum_{oy,ox in -1..2} Wd[oy][ox] dY[2y+oy][2x+ox];  hi-res row offset oy belongs to
//            (py,a) = (1,1), (0,1), (1,0), (0,0) for oy = -1, 0, 1, 2.  wp[(((oy+1)*4 + ox+1)*Cp + c)*Kp + k] = Weff[..][k][c]
__global__ void pack_ups_dgrad_kernel(const float* __restrict__ w, float* __restrict__ wp, int C, int K, int Cp, int Kp) {
    const long n = (long)16 * Cp * Kp;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int k = (int)(i % Kp);
        const long t = i / Kp;
        const int c = (int)(t % Cp), tap = (int)(t / Cp);
        const int oy = (tap >> 2) - 1, ox = (tap & 3) - 1;
        const int py = (oy == -1 || oy == 1) ? 1 : 0, a = (oy <= 0) ? 1 : 0;
        const int px = (ox == -1 || ox == 1) ? 1 : 0, b = (ox <= 0) ? 1 : 0;
        wp[i] = (c < C && k < K) ? weff(w + ((long)k * C + c) * 9, py, a, px, b) : 0.f;
    }
}

// ---------------------------------------------------------------------------------------------- launch plumbing
struct Tile { int bm, bn; };
// Tile choice, from measurements on MI355X at B=32 (tools/bench_conv.py; whole-step A/B in bench.py):
//   * Cout = 64 layers (224^2): the 64x64 tile at 4 blocks/CU wins by 6-15 % (a 128x64 block has half the MFMA
//     work per staged byte and only 18 K-slices to amortise its prologue / epilogue over);
//   * Cout >= 128: in isolation 64x64 is 1-5 % faster (finer tail: 784 / 1568 coarse tiles leave up to 23 % of the
//     last wave of 512 slots idle), but inside the training step -- where the flow / RGB encoders and wgrad || dgrad
//     overlap on separate HIP streams and fill those tails -- the 128x128 tile is 1.5 % faster end to end.
Tile pick_tile(int K, int flags) {
    if (K % 64 != 0) return {128, 32};   // late-fusion widths (32, 8): one 32-wide n-tile, 4 waves along m
    if (flags & 0x400) return {64, 64};
    if (flags & 0x100) return {128, 64};
    if (flags & 0x200) return {128, K % 128 == 0 ? 128 : 64};
    return (K % 128 == 0) ? Tile{128, 128} : Tile{64, 64};
}

template <int BM, int BN, int MODE, int EPI>
int launch_igemm(const float* x, const float* wp, const float* bias, float* y, double* stat, int B, int H,
                 int W, int C, int K, hipStream_t st) {
    const long M = (MODE >= UPS_PHASE) ? (long)B * (H / 2) * (W / 2) : (long)B * H * W;
    const int Cp = (C + 31) / 32 * 32, Kp = (K + 31) / 32 * 32;
    dim3 grid(egz_cdiv(M, BM) * (Kp / BN), MODE == UPS_PHASE ? 4 : 1);
    hipLaunchKernelGGL((conv3x3_igemm_kernel<BM, BN, MODE, EPI>), grid, dim3(256), 0, st, x, wp, bias, y,
                       stat, B, H, W, C, K, Cp, Kp);
    EGZ_CHECK_LAUNCH("egz_conv3x3_fwd");
    return 0;
}

template <int BM, int BN, int MODE>
int dispatch_epi(int epi, const float* x, const float* wp, const float* bias, float* y, double* stat, int B,
                 int H, int W, int C, int K, hipStream_t st) {
    if (MODE == UPS_DGRAD) return launch_igemm<BM, BN, MODE, EPI_BIAS>(x, wp, bias, y, stat, B, H, W, C, K, st);
    switch (epi) {
        case EPI_BIAS: return launch_igemm<BM, BN, MODE, EPI_BIAS>(x, wp, bias, y, stat, B, H, W, C, K, st);
        case EPI_BIAS_RELU: return launch_igemm<BM, BN, MODE, EPI_BIAS_RELU>(x, wp, bias, y, stat, B, H, W, C, K, st);
        default: return launch_igemm<BM, BN, MODE, EPI_BIAS_STATS>(x, wp, bias, y, stat, B, H, W, C, K, st);
    }
}

template <int MODE>
int dispatch_tile(Tile t, int epi, const float* x, const float* wp, const float* bias, float* y, double* stat, int B,
                  int H, int W, int C, int K, hipStream_t st) {
    if (t.bn == 32) return dispatch_epi<128, 32, MODE>(epi, x, wp, bias, y, stat, B, H, W, C, K, st);
    if (t.bm == 64) return dispatch_epi<64, 64, MODE>(epi, x, wp, bias, y, stat, B, H, W, C, K, st);
    if (t.bn == 64) return dispatch_epi<128, 64, MODE>(epi, x, wp, bias, y, stat, B, H, W, C, K, st);
    return dispatch_epi<128, 128, MODE>(epi, x, wp, bias, y, stat, B, H, W, C, K, st);
}

int pack_grid(long n) { return egz_cdiv(n, 256) > 4096 ? 4096 : egz_cdiv(n, 256); }

}  // namespace

// rows of the stats partial buffer for a given launch configuration (flags as for egz_conv3x3_fwd)
EGZ_API int egz_conv3x3_stat_rows(int B, int H, int W, int K, int flags) {
    const int bm = pick_tile(K, flags).bm;
    if ((flags & 3) == 3) return 4 * egz_cdiv((long)B * (H / 2) * (W / 2), bm);   // phase-decomposed upsample
    return egz_cdiv((long)B * H * W, bm);
}

// elements of a packed weight buffer; kind 0 = plain fwd / dgrad (9 taps), 1 = upsample-fused fwd / dgrad (16)
EGZ_API size_t egz_pack_w3x3_elems(int C, int K, int kind) {
    return (size_t)(kind ? 16 : 9) * ((C + 31) / 32 * 32) * ((K + 31) / 32 * 32);
}

#define EGZ_PACK(NAME, KERNEL, TAPS)                                                                          \
    EGZ_API int NAME(const float* w, float* wp, int C, int K, hipStream_t st) {                               \
        EGZ_CHECK_ARG(w && wp && C > 0 && K > 0, #NAME ": bad arguments");                                    \
        const int Cp = (C + 31) / 32 * 32, Kp = (K + 31) / 32 * 32;                                           \
        hipLaunchKernelGGL(KERNEL, dim3(pack_grid((long)TAPS * Cp * Kp)), dim3(256), 0, st, w, wp, C, K, Cp, Kp); \
        EGZ_CHECK_LAUNCH(#NAME);                                                                              \
        return 0;                                                                                             \
    }
EGZ_PACK(egz_pack_w3x3_fwd, pack_fwd_kernel, 9)
EGZ_PACK(egz_pack_w3x3_dgrad, pack_dgrad_kernel, 9)
EGZ_PACK(egz_pack_w3x3_ups_fwd, pack_ups_fwd_kernel, 16)
EGZ_PACK(egz_pack_w3x3_ups_dgrad, pack_ups_dgrad_kernel, 16)
#undef EGZ_PACK

// flags: bit0 = the conv input is the nearest-x2 upsampling of x ([B][H/2][W/2][C]), folded into the gather;
//        bit1 (with bit0) = phase-decomposed form (needs egz_pack_w3x3_ups_fwd weights, 4/9 of the MACs);
//        bits 4-5 = epilogue (0 bias, 1 bias+relu, 2 bias + BN stat partials);
//        0x100 / 0x200 / 0x400 force the 128x64 / 128x128 / 64x64 tile (benchmarking).
EGZ_API int egz_conv3x3_fwd(const float* x, const float* wp, const float* bias, float* y, double* stat_partial,
                            int B, int H, int W, int C, int K, int flags, hipStream_t st) {
    EGZ_CHECK_ARG(x && wp && y, "egz_conv3x3_fwd: null pointer");
    EGZ_CHECK_ARG(B > 0 && H > 0 && W > 0, "egz_conv3x3_fwd: bad shape %dx%dx%d", B, H, W);
    EGZ_CHECK_ARG(C % 4 == 0 && C > 0, "egz_conv3x3_fwd: Cin=%d must be a positive multiple of 4", C);
    EGZ_CHECK_ARG(K > 0 && (K % 64 == 0 || K <= 32), "egz_conv3x3_fwd: Cout=%d must be a multiple of 64 or <= 32", K);
    EGZ_CHECK_ARG(((uintptr_t)x % 16 == 0), "egz_conv3x3_fwd: x must be 16-byte aligned");
    const int ups = flags & 3;
    EGZ_CHECK_ARG(ups != 2, "egz_conv3x3_fwd: flag bit1 needs bit0");
    EGZ_CHECK_ARG(!ups || (H % 2 == 0 && W % 2 == 0), "egz_conv3x3_fwd: upsampled output must be even");
    const int epi = (flags >> 4) & 3;
    EGZ_CHECK_ARG(epi <= 2, "egz_conv3x3_fwd: bad epilogue %d", epi);
    EGZ_CHECK_ARG(epi != EPI_BIAS_STATS || stat_partial, "egz_conv3x3_fwd: stats epilogue needs stat_partial");
    const Tile t = pick_tile(K, flags);
    if (ups == 3) return dispatch_tile<UPS_PHASE>(t, epi, x, wp, bias, y, stat_partial, B, H, W, C, K, st);
    if (ups == 1) return dispatch_tile<UPS_FOLD>(t, epi, x, wp, bias, y, stat_partial, B, H, W, C, K, st);
    return dispatch_tile<PLAIN>(t, epi, x, wp, bias, y, stat_partial, B, H, W, C, K, st);
}

// Data gradient of [nn.Upsample(x2) -> nn.Conv2d(C, K, 3, padding=1)] w.r.t. the low-res input:
// dy: [B][H][W][K] (hi-res), wp: egz_pack_w3x3_ups_dgrad weights, dx: [B][H/2][W/2][C].
EGZ_API int egz_conv3x3_ups_dgrad(const float* dy, const float* wp, float* dx, int B, int H, int W, int C, int K,
                                  int flags, hipStream_t st) {
    EGZ_CHECK_ARG(dy && wp && dx, "egz_conv3x3_ups_dgrad: null pointer");
    EGZ_CHECK_ARG(H % 2 == 0 && W % 2 == 0 && B > 0, "egz_conv3x3_ups_dgrad: hi-res dims must be even");
    EGZ_CHECK_ARG(K % 4 == 0 && (C % 64 == 0 || C <= 32), "egz_conv3x3_ups_dgrad: unsupported C=%d K=%d", C, K);
    // GEMM roles: reduction over the conv's K (input channels of this GEMM), output channels = the conv's C
    return dispatch_tile<UPS_DGRAD>(pick_tile(C, flags), EPI_BIAS, dy, wp, nullptr, dx, nullptr, B, H, W, K, C, st);
}
