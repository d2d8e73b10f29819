// First encoder convolution of each stream: nn.Conv2d(3 -> 64) on the RGB frame and nn.Conv2d(20 -> 64)
// on the 10-pair optical-flow stack (reference utils.py:70 instantiated at SP.py:53; inputs follow
// data/STdatas.py:50-73 and arrive NCHW exactly as the reference's DataLoader yields them).
// K = 9*Cin is tiny (27 / 180), so the im2col A tile is built in LDS straight from the NCHW planes --
// consecutive lanes read consecutive x of one channel plane (coalesced 256-B rows of the 20-channel
// flow stack) -- and multiplied on v_mfma_f32_32x32x2_f32.  Output is NHWC (the library's internal
// activation layout) with the same bias / BN-statistics epilogue as the wide kernel.
// The weight gradient (no data gradient: the network input needs none) reduces over all pixels with
// split-K partials + a deterministic second pass.
#include "egz_common.h"

namespace {

constexpr int FM = 128;        // pixels per block
constexpr int CCH = 4;         // channels per K chunk
constexpr int KCH = CCH * 9;   // 36 k per chunk
constexpr int FLDA = KCH + 1;  // 37: odd stride -> conflict-free column reads

// y[m][k] (NHWC, 64 channels) = bias[k] + sum_{c,tap} x[b][c][y+dy][x+dx] * w[k][c][tap]
template <bool STATS, int KT>   // KT = Cout / 32 (2: SP encoders, 1: late-fusion first conv)
__global__ __launch_bounds__(256, 2) void conv_first_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
    float* __restrict__ y, double* __restrict__ stat, int B, int H, int W, int C) {
    constexpr int K = 32 * KT;
    __shared__ float As[FM * FLDA];
    __shared__ float Bs[KCH * K];
    __shared__ double red[4 * 2 * K];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hl = lane >> 5, l31 = lane & 31;
    const long HW = (long)H * W, M = (long)B * HW;
    const long m0 = (long)blockIdx.x * FM;

    // the pixel this thread gathers for (p = tid & 127)
    const int p = tid & 127;
    const long mp = m0 + p;
    int py = -(1 << 20), px = 0;
    long pimg = 0;
    if (mp < M) {
        const long b = mp / HW;
        const int rem = (int)(mp - b * HW);
        py = rem / W;
        px = rem - py * W;
        pimg = b * C * HW;
    }

    f32x16 acc[KT];
#pragma unroll
    for (int j = 0; j < KT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    for (int cc0 = 0; cc0 < C; cc0 += CCH) {
        // ---- im2col gather: As[p][kk], kk = (c - cc0)*9 + tap
#pragma unroll
        for (int j = 0; j < KCH / 2; ++j) {
            const int kk = (tid >> 7) + 2 * j;
            const int c = cc0 + kk / 9, tap = kk % 9;
            const int iy = py + tap / 3 - 1, ix = px + tap % 3 - 1;
            float v = 0.f;
            if (c < C && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
                v = x[pimg + (long)c * HW + (long)iy * W + ix];
            As[p * FLDA + kk] = v;
        }
        // ---- weights: Bs[kk][k] = w[k][cc0 + kk/9][kk%9]
        for (int i = tid; i < KCH * K; i += 256) {
            const int k = i / KCH, kk = i - k * KCH;     // consecutive threads walk (c,tap) of one filter
            const int c = cc0 + kk / 9;
            Bs[kk * K + k] = (c < C) ? w[((long)k * C + c) * 9 + (kk % 9)] : 0.f;
        }
        __syncthreads();
        const float* Ab = As + (wave * 32 + l31) * FLDA + hl;
        const float* Bb = Bs + hl * K + l31;
#pragma unroll
        for (int t = 0; t < KCH / 2; ++t) {
            const float a = Ab[2 * t];
#pragma unroll
            for (int j = 0; j < KT; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, Bb[(2 * t) * K + 32 * j], acc[j], 0, 0, 0);
        }
        __syncthreads();
    }

#pragma unroll
    for (int nr = 0; nr < KT; ++nr) {
        const int col = nr * 32 + l31;
        const float bz = bias ? bias[col] : 0.f;
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const long m = m0 + wave * 32 + egz_acc_row(r, lane);
            if (m < M) {
                const float v = acc[nr][r] + bz;
                y[m * K + col] = v;
                if (STATS) {
                    s1 += (double)v;
                    s2 += (double)v * (double)v;
                }
            }
        }
        if (STATS) {
            s1 += __shfl_xor(s1, 32);
            s2 += __shfl_xor(s2, 32);
            if (hl == 0) {
                red[(wave * 2 + 0) * K + col] = s1;
                red[(wave * 2 + 1) * K + col] = s2;
            }
        }
    }
    if (STATS) {
        __syncthreads();
        if (tid < 2 * K) {
            const int which = tid / K, col = tid % K;
            double s = 0.0;
#pragma unroll
            for (int wv = 0; wv < 4; ++wv) s += red[(wv * 2 + which) * K + col];
            stat[((long)blockIdx.x * 2 + which) * K + col] = s;
        }
    }
}

// Direct form for the first late-fusion conv (late_fusion.py:10: Conv2d(2 -> 32) on the (B, 2, 224, 224) pair of maps; CIN <= 3,
// 32 filters).  With 18-27 MACs per output the layer is a 205 MB write (B = 32) and nothing else; the im2col + fp32-MFMA kernel
// above (built for 3 / 20 -> 64 channels) spent 110-120 us on it (1.8 TB/s).  Here thread = (pixel lane pl = tid >> 3, filter
// quad k4 = tid & 7): its CIN * 9 * 4 weights live in registers, the CIN * 9 taps of the NCHW input are read once per pixel
// (the eight quads of a pixel share the address), and the eight 16-byte stores of a pixel form one 128-byte line of the NHWC
// output.  A block walks a contiguous pixel range 32 pixels at a time and writes ONE row of fp64 BN partial sums at the end
// (pixel lanes reduced with wave shuffles, the four waves through LDS, fixed order).
constexpr int FD_BLOCKS = 512;          // one round at two resident blocks per CU; also the rows the BatchNorm finalize sums
// PX = horizontally adjacent pixels per thread (4 when W % 4 == 0: the 3 x (PX + 2) input window of a channel is read once for
// the four of them -- 9 instead of 18 tap loads, bounds checks and address computations per pixel; the one-pixel form spent
// more issue slots on those than on the 72 FMAs and ran at 1.9 TB/s).
// KQ = filter quads = K / 4: 8 (the late-fusion conv, 32 filters) or 16 (Conv2d(3, 64), the RGB encoder's first layer,
// utils.py:70 at SP.py:53): thread = (pixel lane tid / KQ, quad tid % KQ), 256 / KQ pixel lanes per block.
template <int CIN, int PX, bool STATS, int KQ>
__global__ __launch_bounds__(256) void conv_first_direct_kernel(
    const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias, float* __restrict__ y,
    double* __restrict__ stat, int B, int H, int W, int ppb, float* __restrict__ mm_out, unsigned int* __restrict__ mm_ord) {
    constexpr int K = 4 * KQ, NT = CIN * 9, PL = 256 / KQ;
    __shared__ double red[4][KQ][8];
    __shared__ float rmm[4][KQ][8];
    float vmx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY}, vmn[4] = {INFINITY, INFINITY, INFINITY, INFINITY};
    const int tid = threadIdx.x, k4 = tid % KQ, pl = tid / KQ, lane = tid & 63, wave = tid >> 6;
    const int HW = H * W, M = B * HW;
    const int m0 = blockIdx.x * ppb, m1 = (m0 + ppb < M) ? m0 + ppb : M;
    float wr[NT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) wr[t][e] = w[(long)(k4 * 4 + e) * NT + t];          // w: (K, CIN, 3, 3)
    f32x4 bz = {0.f, 0.f, 0.f, 0.f};
    if (bias) bz = *reinterpret_cast<const f32x4*>(bias + k4 * 4);
    double s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0};
    int m = m0 + pl * PX;                                      // first of the thread's PX pixels (same image row: W % PX == 0)
    int b = m / HW, py = (m - b * HW) / W, px = m - b * HW - py * W;
    for (; m < m1; m += PL * PX) {
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
        float q1[4] = {0.f, 0.f, 0.f, 0.f}, q2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int p = 0; p < PX; ++p) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < CIN; ++c)
#pragma unroll
                for (int t = 0; t < 9; ++t)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[e] += xw[c][t / 3][p + t % 3] * wr[c * 9 + t][e];
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] += bz[e];
            *reinterpret_cast<f32x4*>(y + (long)(m + p) * K + k4 * 4) = acc;
            if (STATS) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    vmx[e] = fmaxf(vmx[e], acc[e]);            // per-channel max / min of y (mm_out: a deferred BatchNorm
                    vmn[e] = fminf(vmn[e], acc[e]);            // bounds its output with them, egz_bn_finalize_deferred)
                    if (PX == 1) {
                        s1[e] += (double)acc[e];
                        s2[e] += (double)acc[e] * (double)acc[e];
                    } else {                                     // the PX pixels in fp32, then one fp64 add per iteration
                        q1[e] += acc[e];
                        q2[e] += acc[e] * acc[e];
                    }
                }
            }
        }
        if (STATS && PX > 1) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                s1[e] += (double)q1[e];
                s2[e] += (double)q2[e];
            }
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
    if (STATS) {                                                // lane = (pixel lane % (64 / KQ)) * KQ + quad
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            double a = s1[e], q = s2[e];
            float hi = vmx[e], lo = vmn[e];
#pragma unroll
            for (int o = KQ; o < 64; o <<= 1) {
                a += __shfl_xor(a, o);
                q += __shfl_xor(q, o);
                hi = fmaxf(hi, __shfl_xor(hi, o));
                lo = fminf(lo, __shfl_xor(lo, o));
            }
            if (lane < KQ) {
                red[wave][lane][e] = a;
                red[wave][lane][4 + e] = q;
                rmm[wave][lane][e] = hi;
                rmm[wave][lane][4 + e] = lo;
            }
        }
        __syncthreads();
        if (tid < 2 * K) {                                      // (K <= 64: at most 128 of the 256 threads)
            const int which = tid / K, col = tid % K;
            double t = 0.0;
#pragma unroll
            for (int wv = 0; wv < 4; ++wv) t += red[wv][col >> 2][which * 4 + (col & 3)];
            stat[((long)blockIdx.x * 2 + which) * K + col] = t;
            if (mm_out || mm_ord) {
                float r = rmm[0][col >> 2][which * 4 + (col & 3)];
#pragma unroll
                for (int wv = 1; wv < 4; ++wv) {
                    const float o = rmm[wv][col >> 2][which * 4 + (col & 3)];
                    r = which ? fminf(r, o) : fmaxf(r, o);
                }
                if (mm_out) mm_out[((long)blockIdx.x * 2 + which) * K + col] = r;
                if (mm_ord) {
                    // the 2 K-uint form egz_bn_finalize_bound reads (conv3x3_igemm_x3s.hip): order-preserving images of max y
                    // (slot col) and max -y (slot K + col), folded in with atomic max -- exact and order independent
                    const unsigned int b = __float_as_uint(which ? -r : r);
                    const unsigned int u = b ^ (((int)b < 0) ? 0xffffffffu : 0x80000000u);
                    unsigned int* pm = mm_ord + which * K + col;
                    if (u > __hip_atomic_load(pm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                        __hip_atomic_fetch_max(pm, u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
    }
}

inline bool first_direct_ok(int C, int K) { return (K == 32 || K == 64) && C >= 1 && C <= 3; }
inline int first_direct_ppb(long M, int W, int K) {
    const int step = (256 / (K / 4)) * ((W % 4 == 0) ? 4 : 1);  // pixels per block iteration (4 / 1 per thread)
    long ppb = (M + FD_BLOCKS - 1) / FD_BLOCKS;
    return (int)((ppb + step - 1) / step * step);
}

// partial[split * NPG + pixel group][k][kkp] with kkp = c*9 + tap padded to KP (multiple of 32):
//   sum over the group's pixels of dy[m][k] * x[b][c][y+dy][x+dx]
// Cin = 20: the four waves tile (k half) x (kkp half).  Cin <= 3 has ONE kkp tile (and one or two k tiles), so the spare
// waves take further 32-pixel groups of a 32 * NPG pixel stage (NPG = 4 / KT) and write their own partial rows.
template <int NT, int KT>   // NT = KP/32 : 1 (Cin <= 3) or 6 (Cin = 20);  KT = Cout/32
__global__ __launch_bounds__(256, 2) void conv_first_wgrad_kernel(
    const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ part, int B, int H, int W,
    int C, long pix_per_split) {
    constexpr int NPG = (NT == 1) ? 4 / KT : 1;     // pixel groups (one wave each along the reduction)
    constexpr int K = 32 * KT, KP = NT * 32, PKS = 32 * NPG;
    constexpr int XLD = KP + 1;                     // odd row stride: conflict-free transposing writes
    constexpr int TW = (NT == 1) ? 1 : NT / 2;      // n-tiles per wave
    __shared__ __attribute__((aligned(16))) float Ds[PKS * K];   // [pixel][k]
    __shared__ float Xs[PKS * XLD];                 // [pixel][kkp]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hl = lane >> 5, l31 = lane & 31;
    const int wk = (NT == 1) ? wave / NPG : wave >> 1;     // k tile (32 filters)
    const int wn = (NT == 1) ? 0 : wave & 1;               // kkp half
    const int pg = (NT == 1) ? wave % NPG : 0;             // pixel group
    const long HW = (long)H * W, M = (long)B * HW;
    const long mbeg = (long)blockIdx.x * pix_per_split;
    const long mend = (mbeg + pix_per_split < M) ? (mbeg + pix_per_split) : M;
    const int KK = C * 9;

    f32x16 acc[TW];
#pragma unroll
    for (int j = 0; j < TW; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    for (long mb = mbeg; mb < mend; mb += PKS) {
        // dy tile: PKS pixels x K filters
#pragma unroll
        for (int j = 0; j < KT * NPG; ++j) {
            const int i = tid + 256 * j;
            const int pp = i / (K / 4), k4 = i % (K / 4);
            const long m = mb + pp;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (m < mend) v = *reinterpret_cast<const f32x4*>(dy + m * K + k4 * 4);
            *reinterpret_cast<f32x4*>(Ds + pp * K + k4 * 4) = v;
        }
        // im2col tile: lanes walk the pixels (consecutive x -> coalesced plane reads)
        {
            const int pp = tid % PKS;
            const long m = mb + pp;
            int yy = -(1 << 20), xx = 0;
            long img = 0;
            if (m < mend) {
                const long b = m / HW;
                const int rem = (int)(m - b * HW);
                yy = rem / W;
                xx = rem - yy * W;
                img = b * C * HW;
            }
#pragma unroll 4
            for (int kk = tid / PKS; kk < KP; kk += 256 / PKS) {
                float v = 0.f;
                if (kk < KK) {
                    const int c = kk / 9, tap = kk - c * 9;
                    const int iy = yy + tap / 3 - 1, ix = xx + tap % 3 - 1;
                    if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
                        v = x[img + (long)c * HW + (long)iy * W + ix];
                }
                Xs[pp * XLD + kk] = v;
            }
        }
        __syncthreads();
        {
            const float* Ab = Ds + (pg * 32 + hl) * K + wk * 32 + l31;
            const float* Bb = Xs + (pg * 32 + hl) * XLD + wn * (TW * 32) + l31;
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const float a = Ab[(2 * t) * K];
#pragma unroll
                for (int j = 0; j < TW; ++j)
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, Bb[(2 * t) * XLD + j * 32], acc[j], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    float* out = part + ((long)blockIdx.x * NPG + pg) * K * KP;
#pragma unroll
    for (int j = 0; j < TW; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int k = wk * 32 + egz_acc_row(r, lane);
            const int kk = wn * (TW * 32) + j * 32 + l31;
            out[k * KP + kk] = acc[j][r];
        }
}

// dw[k][c][tap] (= [k][kk], kk < 9C) = sum_s part[s][k][kk].  32 outputs x 8 split groups per block: group g sums the
// splits g, g + 8, ... with independent loads in flight, the groups are combined in a fixed order (deterministic).
__global__ __launch_bounds__(256) void first_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw,
                                                                 int K, int KK, int KP, int S) {
    __shared__ float red[8][33];
    const int o = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int n = K * KK, i = blockIdx.x * 32 + o;
    float s = 0.f;
    if (i < n) {
        const int k = i / KK, kk = i - k * KK;
#pragma unroll 8
        for (int sp = g; sp < S; sp += 8) s += part[((long)sp * K + k) * KP + kk];
    }
    red[g][o] = s;
    __syncthreads();
    if (g == 0 && i < n) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) t += red[q][o];
        dw[i] = t;
    }
}

int first_splits(long M) {
    long s = (M + 2047) / 2048;
    if (s > 1024) s = 1024;
    if (s < 1) s = 1;
    return (int)s;
}
int first_kp(int C) { return (9 * C + 31) / 32 * 32; }

}  // namespace

EGZ_API int egz_conv_first_stat_rows(int B, int H, int W) { return egz_cdiv((long)B * H * W, FM); }
// rows of stat_partial for a given channel configuration (the direct kernel for C <= 3 -> 32 writes one row per block)
EGZ_API int egz_conv_first_stat_rows_for(int B, int H, int W, int C, int K) {
    const long M = (long)B * H * W;
    if (first_direct_ok(C, K) && M * K < (1l << 31)) return egz_cdiv(M, first_direct_ppb(M, W, K));
    return egz_cdiv(M, FM);
}

// x: [B][C][H][W] (NCHW, as the reference DataLoader yields it), w: (64, C, 3, 3), y: [B][H][W][64].
// minmax_out (optional; C <= 3 -> 32 with stat_partial only): [rows][2][32] per-channel max / min of y, rows as stat_partial.
// minmax_ordered (optional, same conditions): 2 K uints, zero-filled by the caller -- the atomic form egz_bn_finalize_bound reads.
EGZ_API int egz_conv_first_fwd(const float* x, const float* w, const float* bias, float* y, double* stat_partial,
                               int B, int H, int W, int C, int K, float* minmax_out, unsigned int* minmax_ordered,
                               hipStream_t st) {
    EGZ_CHECK_ARG(x && w && y, "egz_conv_first_fwd: null pointer");
    EGZ_CHECK_ARG(!(minmax_out || minmax_ordered) || (stat_partial && first_direct_ok(C, K) && (long)B * H * W * K < (1l << 31)),
                  "egz_conv_first_fwd: minmax_out / minmax_ordered exist on the direct kernel only (C <= 3 -> 32 / 64 filters, with stat_partial)");
    EGZ_CHECK_ARG(K == 64 || K == 32, "egz_conv_first_fwd: Cout must be 64 or 32 (got %d)", K);
    EGZ_CHECK_ARG(C > 0 && C <= 64 && B > 0 && H > 0 && W > 0, "egz_conv_first_fwd: bad shape");
    const long M = (long)B * H * W;
    if (first_direct_ok(C, K) && M * K < (1l << 31)) {
        const int ppb = first_direct_ppb(M, W, K), nb = egz_cdiv(M, ppb);
#define EGZ_FD3(CC, PP, QQ)                                                                                                    \
    do {                                                                                                                       \
        if (stat_partial) hipLaunchKernelGGL((conv_first_direct_kernel<CC, PP, true, QQ>), dim3(nb), dim3(256), 0, st, x, w, bias, y, stat_partial, B, H, W, ppb, minmax_out, minmax_ordered); \
        else              hipLaunchKernelGGL((conv_first_direct_kernel<CC, PP, false, QQ>), dim3(nb), dim3(256), 0, st, x, w, bias, y, stat_partial, B, H, W, ppb, (float*)nullptr, (unsigned int*)nullptr); \
    } while (0)
#define EGZ_FD2(CC, PP)                                                                                                        \
    do {                                                                                                                       \
        if (K == 32) EGZ_FD3(CC, PP, 8);                                                                                       \
        else EGZ_FD3(CC, PP, 16);                                                                                              \
    } while (0)
#define EGZ_FD(CC)                                                                                                             \
    do {                                                                                                                       \
        if (W % 4 == 0) EGZ_FD2(CC, 4);                                                                                        \
        else EGZ_FD2(CC, 1);                                                                                                   \
    } while (0)
        if (C == 1) EGZ_FD(1);
        else if (C == 2) EGZ_FD(2);
        else EGZ_FD(3);
#undef EGZ_FD
#undef EGZ_FD2
#undef EGZ_FD3
        EGZ_CHECK_LAUNCH("egz_conv_first_fwd(direct)");
        return 0;
    }
    const int grid = egz_cdiv((long)B * H * W, FM);
    if (K == 64) {
        if (stat_partial) hipLaunchKernelGGL((conv_first_fwd_kernel<true, 2>), dim3(grid), dim3(256), 0, st, x, w, bias, y, stat_partial, B, H, W, C);
        else              hipLaunchKernelGGL((conv_first_fwd_kernel<false, 2>), dim3(grid), dim3(256), 0, st, x, w, bias, y, stat_partial, B, H, W, C);
    } else {
        if (stat_partial) hipLaunchKernelGGL((conv_first_fwd_kernel<true, 1>), dim3(grid), dim3(256), 0, st, x, w, bias, y, stat_partial, B, H, W, C);
        else              hipLaunchKernelGGL((conv_first_fwd_kernel<false, 1>), dim3(grid), dim3(256), 0, st, x, w, bias, y, stat_partial, B, H, W, C);
    }
    EGZ_CHECK_LAUNCH("egz_conv_first_fwd");
    return 0;
}

EGZ_API size_t egz_conv_first_wgrad_ws_bytes(int B, int H, int W, int C) {
    // sized for Cout = 64; the one-kkp-tile form (Cin <= 3) writes 4 / (Cout / 32) partial rows per split: 128 filters' worth
    const int KP = first_kp(C);
    return (size_t)first_splits((long)B * H * W) * (KP == 32 ? 128 : 64) * KP * sizeof(float);
}

EGZ_API int egz_conv_first_wgrad(const float* x, const float* dy, float* dw, int B, int H, int W, int C, int K,
                                 void* workspace, size_t ws_bytes, hipStream_t st) {
    EGZ_CHECK_ARG(x && dy && dw && workspace, "egz_conv_first_wgrad: null pointer");
    EGZ_CHECK_ARG(K == 64 || K == 32, "egz_conv_first_wgrad: Cout must be 64 or 32 (got %d)", K);
    const int KP = first_kp(C);
    EGZ_CHECK_ARG(KP == 32 || KP == 192, "egz_conv_first_wgrad: Cin=%d unsupported (1..3 or 18..21)", C);
    const long M = (long)B * H * W;
    const int S = first_splits(M);
    const int NPG = (KP == 32) ? 128 / K : 1;          // pixel groups (partial rows) per split
    EGZ_CHECK_ARG(ws_bytes >= (size_t)S * NPG * K * KP * sizeof(float), "egz_conv_first_wgrad: workspace too small");
    long pps = (M + S - 1) / S;
    pps = (pps + 127) / 128 * 128;
    float* part = static_cast<float*>(workspace);
    if (K == 64) {
        if (KP == 32) hipLaunchKernelGGL((conv_first_wgrad_kernel<1, 2>), dim3(S), dim3(256), 0, st, x, dy, part, B, H, W, C, pps);
        else          hipLaunchKernelGGL((conv_first_wgrad_kernel<6, 2>), dim3(S), dim3(256), 0, st, x, dy, part, B, H, W, C, pps);
    } else {
        if (KP == 32) hipLaunchKernelGGL((conv_first_wgrad_kernel<1, 1>), dim3(S), dim3(256), 0, st, x, dy, part, B, H, W, C, pps);
        else          hipLaunchKernelGGL((conv_first_wgrad_kernel<6, 1>), dim3(S), dim3(256), 0, st, x, dy, part, B, H, W, C, pps);
    }
    EGZ_CHECK_LAUNCH("egz_conv_first_wgrad");
    hipLaunchKernelGGL(first_wgrad_reduce_kernel, dim3(egz_cdiv(K * 9 * C, 32)), dim3(256), 0, st, part, dw, K, 9 * C, KP, S * NPG);
    EGZ_CHECK_LAUNCH("egz_conv_first_wgrad(reduce)");
    return 0;
}
