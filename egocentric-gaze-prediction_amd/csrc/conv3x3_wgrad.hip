// Weight gradient of the 3x3 / pad 1 convolution on exact-f32 MFMA (v_mfma_f32_32x32x2_f32).
//   dW[k][c][tap] = sum_{m=(b,y,x)} dY[m][k] * X[pix(m)+tap][c]          (autograd of nn.Conv2d;
//   reference call sites: loss.backward() in SP.py:136, LF.py:99)
// GEMM view per tap: OUT[c][k] (C x K) with the reduction over the B*H*W pixels.  The pixel range is
// split S ways (split-K) so the grid fills the chip; each block writes its fp32 partial tile to the
// workspace and a second kernel sums the S partials in a fixed order (deterministic, no atomics) and
// writes dW in the reference (Cout, Cin, 3, 3) layout.
//   block tile BT(c) x BT(k), BT = 128 or 64, 4 waves as 2 x 2; reduction chunk = 32 pixels;
//   both operands are pixel-major in LDS ([pixel][channel]) which is exactly the MFMA A/B fragment
//   order (lane -> consecutive channel) so every ds_read_b32 is conflict-free without padding.
#include "egz_common.h"
#include <cstdlib>

namespace {

constexpr int PK = 32;   // pixels per LDS stage

template <int BT, bool UPS>
__global__ __launch_bounds__(256, 2) void conv3x3_wgrad_kernel(
    const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ part, int B, int H, int W,
    int C, int K, long pix_per_split) {
    constexpr int TR = BT / 64;             // 32-wide MFMA tiles per wave per dim
    constexpr int LD4 = BT / 4;             // float4 per LDS row
    constexpr int NLD = (PK * LD4) / 256;   // float4 loads per thread per operand per stage (4 or 2)
    constexpr int PSTEP = 256 / LD4;        // pixel stride between a thread's loads (8 or 16)

    __shared__ __attribute__((aligned(16))) float Xs[2 * PK * BT];
    __shared__ __attribute__((aligned(16))) float Ds[2 * PK * BT];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wc = wave >> 1, wk = wave & 1, hl = lane >> 5, l31 = lane & 31;
    const int tk = K / BT;
    const int tap = blockIdx.x % 9;
    const int tile = blockIdx.x / 9;
    const int c0 = (tile / tk) * BT, k0 = (tile % tk) * BT;
    const int dyy = tap / 3 - 1, dxx = tap % 3 - 1;
    const long HW = (long)H * W, M = (long)B * HW;
    const long mbeg = (long)blockIdx.y * pix_per_split;
    const long mend = (mbeg + pix_per_split < M) ? (mbeg + pix_per_split) : M;
    const int Hs = UPS ? (H >> 1) : H, Ws = UPS ? (W >> 1) : W;

    const int ch4 = tid % LD4, p0 = tid / LD4;

    f32x4 rx[NLD], rd[NLD];
    auto gload = [&](long mb) {
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const long m = mb + p0 + PSTEP * j;
            f32x4 vx = {0.f, 0.f, 0.f, 0.f}, vd = {0.f, 0.f, 0.f, 0.f};
            if (m < mend) {
                vd = *reinterpret_cast<const f32x4*>(dy + m * K + k0 + ch4 * 4);
                const long b = m / HW;
                const int rem = (int)(m - b * HW);
                const int yy = rem / W, xx = rem - yy * W;
                const int iy = yy + dyy, ix = xx + dxx;
                if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) {
                    const int sy = UPS ? (iy >> 1) : iy, sx = UPS ? (ix >> 1) : ix;
                    vx = *reinterpret_cast<const f32x4*>(x + ((b * Hs + sy) * (long)Ws + sx) * C + c0 + ch4 * 4);
                }
            }
            rx[j] = vx;
            rd[j] = vd;
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            *reinterpret_cast<f32x4*>(Xs + buf * PK * BT + (p0 + PSTEP * j) * BT + ch4 * 4) = rx[j];
            *reinterpret_cast<f32x4*>(Ds + buf * PK * BT + (p0 + PSTEP * j) * BT + ch4 * 4) = rd[j];
        }
    };

    f32x16 acc[TR][TR];
#pragma unroll
    for (int i = 0; i < TR; ++i)
#pragma unroll
        for (int j = 0; j < TR; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nst = (int)((mend - mbeg + PK - 1) / PK);
    if (nst > 0) {
        gload(mbeg);
        lstore(0);
    }
    __syncthreads();
    for (int s = 0; s < nst; ++s) {
        const int buf = s & 1;
        if (s + 1 < nst) gload(mbeg + (long)(s + 1) * PK);
        const float* Ab = Xs + buf * PK * BT + hl * BT + wc * (BT / 2) + l31;
        const float* Bb = Ds + buf * PK * BT + hl * BT + wk * (BT / 2) + l31;
#pragma unroll
        for (int t = 0; t < PK / 2; ++t) {
            float av[TR], bv[TR];
#pragma unroll
            for (int i = 0; i < TR; ++i) av[i] = Ab[(2 * t) * BT + i * 32];
#pragma unroll
            for (int j = 0; j < TR; ++j) bv[j] = Bb[(2 * t) * BT + j * 32];
#pragma unroll
            for (int i = 0; i < TR; ++i)
#pragma unroll
                for (int j = 0; j < TR; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
        if (s + 1 < nst) lstore(buf ^ 1);
        __syncthreads();
    }

    // partial tile -> workspace [split][tap][C][K]
    float* out = part + ((long)blockIdx.y * 9 + tap) * C * K;
#pragma unroll
    for (int i = 0; i < TR; ++i)
#pragma unroll
        for (int j = 0; j < TR; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = c0 + wc * (BT / 2) + i * 32 + egz_acc_row(r, lane);
                const int k = k0 + wk * (BT / 2) + j * 32 + l31;
                out[(long)c * K + k] = acc[i][j][r];
            }
}

// ---------------------------------------------------------------------------------------------------------
// 9-tap fused variant (the default for the SP shapes): one block owns a 64(c) x 64(k) tile for ALL nine taps.
// A stage is a row segment of L pixels; its 3 x (L+2) input halo and its L dY rows are staged in LDS once and
// feed 9 MFMAs per k-step (one per tap, reading the halo at the tap's shift), so x and dY are fetched from
// L2/HBM once per tile instead of nine times and the arithmetic intensity per staged byte is 4.5x higher.
// Each wave keeps 9 accumulators (144 VGPRs) for its 32 x 32 corner of the tile.
template <bool UPS, int L>
__global__ __launch_bounds__(256, 2) void conv3x3_wgrad9_kernel(
    const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ part, int B, int H, int W,
    int C, int K, int segs_per_split) {
    constexpr int HP = L + 2;                         // halo pixels per row
    constexpr int NX = (3 * HP * 16 + 255) / 256;     // float4 halo loads per thread
    constexpr int ND = (L * 16 + 255) / 256;          // float4 dY loads per thread
    __shared__ __attribute__((aligned(16))) float Xh[2 * 3 * HP * 64];
    __shared__ __attribute__((aligned(16))) float Ds[2 * L * 64];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wc = wave >> 1, wk = wave & 1, hl = lane >> 5, l31 = lane & 31;
    const int tk = K / 64;
    const int c0 = (blockIdx.x / tk) * 64, k0 = (blockIdx.x % tk) * 64;
    const int spr = W / L;                            // segments per image row
    const long nseg = (long)B * H * spr;
    const long g0 = (long)blockIdx.y * segs_per_split;
    const long g1 = (g0 + segs_per_split < nseg) ? (g0 + segs_per_split) : nseg;
    const int Hs = UPS ? (H >> 1) : H, Ws = UPS ? (W >> 1) : W;

    f32x4 rx[NX], rd[ND];
    auto gload = [&](long g) {
        const int xs = (int)(g % spr) * L;
        const long t = g / spr;
        const int yy = (int)(t % H);
        const long b = t / H;
#pragma unroll
        for (int j = 0; j < NX; ++j) {
            const int i = tid + 256 * j;
            const int pos = i >> 4, c4 = i & 15;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (pos < 3 * HP) {
                const int r = pos / HP, px = pos - r * HP;
                const int iy = yy + r - 1, ix = xs + px - 1;
                if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) {
                    const int sy = UPS ? (iy >> 1) : iy, sx = UPS ? (ix >> 1) : ix;
                    v = *reinterpret_cast<const f32x4*>(x + ((b * Hs + sy) * (long)Ws + sx) * C + c0 + c4 * 4);
                }
            }
            rx[j] = v;
        }
        const long m0 = (b * H + yy) * (long)W + xs;
#pragma unroll
        for (int j = 0; j < ND; ++j) {
            const int i = tid + 256 * j;
            const int pp = i >> 4, k4 = i & 15;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (pp < L) v = *reinterpret_cast<const f32x4*>(dy + (m0 + pp) * K + k0 + k4 * 4);
            rd[j] = v;
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int j = 0; j < NX; ++j) {
            const int i = tid + 256 * j;
            if (i < 3 * HP * 16) *reinterpret_cast<f32x4*>(Xh + buf * 3 * HP * 64 + i * 4) = rx[j];
        }
#pragma unroll
        for (int j = 0; j < ND; ++j) {
            const int i = tid + 256 * j;
            if (i < L * 16) *reinterpret_cast<f32x4*>(Ds + buf * L * 64 + i * 4) = rd[j];
        }
    };

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    if (g0 < g1) {
        gload(g0);
        lstore(0);
    }
    __syncthreads();
    for (long g = g0; g < g1; ++g) {
        const int buf = (int)((g - g0) & 1);
        if (g + 1 < g1) gload(g + 1);
        const float* Xb = Xh + buf * 3 * HP * 64 + hl * 64 + wc * 32 + l31;
        const float* Db = Ds + buf * L * 64 + hl * 64 + wk * 32 + l31;
#pragma unroll
        for (int t = 0; t < L / 2; ++t) {
            const float bv = Db[(2 * t) * 64];
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const float av = Xb[((tap / 3) * HP + 2 * t + (tap % 3)) * 64];
                acc[tap] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[tap], 0, 0, 0);
            }
        }
        if (g + 1 < g1) lstore(buf ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        float* out = part + ((long)blockIdx.y * 9 + tap) * C * K;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int c = c0 + wc * 32 + egz_acc_row(r, lane);
            const int k = k0 + wk * 32 + l31;
            out[(long)c * K + k] = acc[tap][r];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Split-half form of the 9-tap fused kernel (operand type T = bf16, the default, or f16 with abs-max scaled dY -- see
// W16 below): x and dY stay fp32 in HBM and are split into 16-bit hi + lo
// halves while they are staged into LDS; every product is  x*dy ~= xh*dh + xh*dl + xl*dh  on
// v_mfma_f32_32x32x16_{bf16,f16} (16x the rate of the exact-f32 MFMA, 3 per 16 pixels instead of 8), fp32 accumulate.
// The reduction dimension of this GEMM is the PIXEL, but both operands are channel-contiguous in HBM and in LDS
// ([pixel][channel]); the 16-bit MFMA wants 8 consecutive reduction elements per lane, i.e. the transposed image.
// gfx950's ds_read_b64_tr_b16 does that transpose inside the LDS read: each 16-lane group hands in sixteen 8-byte
// addresses (4 pixels x 4 chunks of 4 channels) and lane t gets the 4 pixels of channel t.  Per-lane addresses make
// the pixel -> halo mapping free, so the stage is a 2-D patch of R x WD pixels (R * WD = 32) with an (R+2) x (WD+2)
// halo, the nine taps are immediate offsets off ONE per-lane base address, and row widths that are not a
// multiple of 16 (28, 14) run as masked WD = 32 / 16 patches (zero dY beyond the row end).
//   LDS image: [plane hi/lo][channel half][halo pixel][32 ch] 16-bit -- a wave only touches the 64-byte rows of its own
//   channel half, and the four pixel rows a 16-lane group reads tile one 256-byte bank row exactly.
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
#define EGZ_LDS __attribute__((address_space(3)))

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
// 4 floats -> packed bf16 hi / lo halves: v_cvt_pk_bf16_f32 (RNE), two bit ops to widen the halves back, one packed
// subtract, v_cvt_pk_bf16_f32 of the residuals -- 2.5 VALU per float
__device__ __forceinline__ void bf16_split4(const f32x4 v, u32x2_t& hi, u32x2_t& lo) {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const f32x2_t x = {v[2 * e], v[2 * e + 1]};
        const unsigned hu = __builtin_bit_cast(unsigned, __builtin_convertvector(x, bf16x2_t));
        const f32x2_t hf = {__builtin_bit_cast(float, hu << 16), __builtin_bit_cast(float, hu & 0xffff0000u)};
        hi[e] = hu;
        lo[e] = __builtin_bit_cast(unsigned, __builtin_convertvector(x - hf, bf16x2_t));
    }
}

typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef __fp16 fp16x4_t __attribute__((ext_vector_type(4)));
// 4 floats -> packed f16 hi / lo halves (22 significant bits): hi = v_cvt_pkrtz (exact residual, saturating), lo = RNE of
// the residual -- the same split as the forward implicit GEMM.  The gradient operand is pre-multiplied by absmax_scale().
__device__ __forceinline__ void f16_split4(const f32x4 v, u32x2_t& hi, u32x2_t& lo) {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const f16x2_t h = __builtin_bit_cast(f16x2_t, __builtin_amdgcn_cvt_pkrtz(v[2 * e], v[2 * e + 1]));
        const f16x2_t l = __builtin_convertvector(f32x2_t{v[2 * e] - (float)h[0], v[2 * e + 1] - (float)h[1]}, f16x2_t);
        hi[e] = __builtin_bit_cast(unsigned, h);
        lo[e] = __builtin_bit_cast(unsigned, l);
    }
}
// The hi-only (x) operand of the two-product arithmetic, rounded to NEAREST: of 4 floats, and of a stored pair quad (its hi halves
// are round-toward-zero images; hi + lo in packed f16 arithmetic is the nearest f16 of the value the pair holds): zero-mean
// rounding errors instead of the -2^-11 relative bias of the truncated hi half.
__device__ __forceinline__ void f16_rne4(const f32x4 v, u32x2_t& hi) {
#pragma unroll
    for (int e = 0; e < 2; ++e)
        hi[e] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{v[2 * e], v[2 * e + 1]}, f16x2_t));
}
__device__ __forceinline__ u32x2_t f16_pair_rne(const u32x2_t h, const u32x2_t l) {
    return __builtin_bit_cast(u32x2_t, __builtin_bit_cast(f16x4_t, h) + __builtin_bit_cast(f16x4_t, l));      // two v_pk_add_f16
}
__device__ __forceinline__ f16x8_t tr_frag(const EGZ_LDS unsigned short* p0, const EGZ_LDS unsigned short* p1) {
    const f16x4_t a = __builtin_bit_cast(f16x4_t, __builtin_amdgcn_ds_read_tr16_b64_v4f16((EGZ_LDS fp16x4_t*)p0));
    const f16x4_t b = __builtin_bit_cast(f16x4_t, __builtin_amdgcn_ds_read_tr16_b64_v4f16((EGZ_LDS fp16x4_t*)p1));
    return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}

// 16-bit operand type of the split-half weight gradient: bf16 x3 (16 significant bits, fp32 exponent range, no scaling --
// the default) or f16 x3 (22 bits; dY is multiplied by absmax_scale(*dy_absmax) before the split and the accumulators divided
// by it at the end -- selected by passing dy_absmax).
template <typename T> struct W16;
template <> struct W16<__bf16> {
    typedef bf16x8_t vec8;
    static __device__ __forceinline__ void split4(const f32x4 v, u32x2_t& hi, u32x2_t& lo) { bf16_split4(v, hi, lo); }
    static __device__ __forceinline__ void split4s(const f32x4 v, const float s, u32x2_t& hi, u32x2_t& lo) { bf16_split4(v * s, hi, lo); }
    static __device__ __forceinline__ vec8 frag(const EGZ_LDS unsigned short* p0, const EGZ_LDS unsigned short* p1) {
        const bf16x4_t a = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((EGZ_LDS bf16x4_t*)p0);
        const bf16x4_t b = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((EGZ_LDS bf16x4_t*)p1);
        return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
    }
    static __device__ __forceinline__ f32x16 mfma(vec8 a, vec8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ float scale(const unsigned int*) { return 1.f; }
};
template <> struct W16<_Float16> {
    typedef f16x8_t vec8;
    static __device__ __forceinline__ void split4(const f32x4 v, u32x2_t& hi, u32x2_t& lo) { f16_split4(v, hi, lo); }
    static __device__ __forceinline__ void split4s(const f32x4 v, const float s, u32x2_t& hi, u32x2_t& lo) {
#ifdef EGZ_TIMING_NOSPLIT
        const u32x4_t b = __builtin_bit_cast(u32x4_t, v);      // timing-only variant, see x3_split.h
        hi = u32x2_t{b[0] & 0x7bff7bffu, b[1] & 0x7bff7bffu};
        lo = u32x2_t{b[2] & 0x7bff7bffu, b[3] & 0x7bff7bffu};
#else
        f16_split4(v * s, hi, lo);
#endif
    }
    static __device__ __forceinline__ vec8 frag(const EGZ_LDS unsigned short* p0, const EGZ_LDS unsigned short* p1) { return tr_frag(p0, p1); }
    static __device__ __forceinline__ f32x16 mfma(vec8 a, vec8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ float scale(const unsigned int* am) { return absmax_scale(am); }
};
template <> struct W16<egz_f16p2> : W16<_Float16> {};      // two products per MAC: x enters hi-only (egz_common.h)

// XCD-aware (tile, split) of a block of a (tiles, splits[, z]) grid.  Hardware places consecutive flat block ids on
// consecutive XCDs (id % 8) and every XCD has its own L2.  The natural map puts the 64 (c, k) tiles of ONE pixel range on all
// eight XCDs, so each XCD's L2 fetches that range of x and dy from HBM separately (measured 2.5 x the algorithmic bytes);
// here XCD j runs whole splits j, j + 8, ... -- all tiles of a pixel range share one L2 and stream through it together.
// Needs gridDim.y % 8 == 0 (otherwise the natural map is kept); the result is independent of the map.
__device__ __forceinline__ void wgrad_block(int& tile, int& split) {
    tile = blockIdx.x;
    split = blockIdx.y;
    const int nt = gridDim.x, ns = gridDim.y;
    if ((ns & 7) == 0) {
        const int flat = blockIdx.y * nt + blockIdx.x;
        const int r = flat >> 3;
        tile = r % nt;
        split = (flat & 7) + 8 * (r / nt);
    }
}

// XPRE (round 5): x holds PRE-SPLIT activations ([4 hi halves | 4 lo halves] per 4-channel quad, scaled by
// absmax_scale(x_absmax); egz_bn_relu_pool_fwd's presplit form) -- the halo staging copies the quad into the two planes of the
// LDS image without a split on the vector ALU (two thirds of this kernel's staged floats are x: its halo is 1.9x the patch).
// DPRE: the same for the gradient operand dy (egz_bn_relu_pool_bwd_presplit: pairs scaled by a BOUND of max |dy|, dy_absmax).
template <typename T, bool UPS, int R, int WD, bool XPRE = false, bool DPRE = false>
__global__ __launch_bounds__(256, 2) void conv3x3_wgrad9_x3_kernel(
    const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ part, int B, int H, int W,
    int C, int K, int patches_per_split, const unsigned int* __restrict__ dy_absmax,
    const unsigned int* __restrict__ x_absmax) {
    static_assert(R * WD == 32 && (WD == 32 || WD == 16 || WD == 8), "patch = 32 pixels");
    const float d_scale = W16<T>::scale(dy_absmax), d_inv = 1.f / d_scale;
    const float x_scale = W16<T>::scale(x_absmax), x_inv = 1.f / x_scale;      // the activation operand, scaled the same way
    constexpr int NP = R * WD, KS = NP / 16;
    constexpr int HPW = WD + 2, NH = (R + 2) * HPW;            // halo row width / halo pixels
    constexpr int XH = NH * 32 + ((NH & 1) ? 0 : 32);          // channel-half stride (elements): bytes % 128 == 64
    constexpr int DH = NP * 32 + 32;
    constexpr int XB = 4 * XH, DB = 4 * DH;                    // per-buffer strides ([plane][half])
    constexpr int NX = (NH * 16 + 255) / 256;                  // float4 halo loads per thread
    constexpr int ND = (NP * 16) / 256;                        // float4 dY loads per thread (2)
    __shared__ __attribute__((aligned(16))) unsigned short Xs[2 * XB];
    __shared__ __attribute__((aligned(16))) unsigned short Ds[2 * DB];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wc = wave >> 1, wk = wave & 1, l31 = lane & 31;
    const int tk = (K + 63) / 64;
    int tile, split;
    wgrad_block(tile, split);
    const int c0 = (tile / tk) * 64, k0 = (tile % tk) * 64;
    const int cpr = (W + WD - 1) / WD, rpi = (H + R - 1) / R;  // patches per row / patch rows per image
    const long npatch = (long)B * rpi * cpr;
    const long g0 = (long)split * patches_per_split;
    const long g1 = (g0 + patches_per_split < npatch) ? (g0 + patches_per_split) : npatch;
    const int Hs = UPS ? (H >> 1) : H, Ws = UPS ? (W >> 1) : W;

    // Operand fetch (direct conv): buffer loads.  A thread's halo slot (hr, hx) and dY pixel (py, px) never change, so its
    // byte offsets relative to the patch origin are fixed; the origin is one scalar offset per stage, and a slot outside
    // the image gets offset 0xFFFFFFFF, which the buffer bounds check turns into zeros (~5 VALU per load instead of ~20).
    // The x resource starts (W + 1) * C floats early so the (-1, -1) halo corner keeps lane offsets non-negative.
    const unsigned x_bias = (unsigned)(W + 1) * (unsigned)C * 4u;
    const __amdgpu_buffer_rsrc_t x_rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(x)) - x_bias, 0, (int)((unsigned)B * H * W * C * 4u + x_bias), 0x00020000);
    const __amdgpu_buffer_rsrc_t d_rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(dy), 0, (int)((unsigned)B * H * W * K * 4u), 0x00020000);
    unsigned x_vo[NX], d_vo[ND];                          // byte offsets of the slot inside the patch
#pragma unroll
    for (int j = 0; j < NX; ++j) {
        const int i = tid + 256 * j, pos = i >> 4, c4 = i & 15;
        const int hr = pos / HPW, hx = pos - hr * HPW;
        x_vo[j] = (pos < NH && c0 + c4 * 4 < C) ? (unsigned)((hr * W + hx) * C * 4 + c4 * 16) : 0xFFFFFFFFu;   // C = 32: half tile
    }
#pragma unroll
    for (int j = 0; j < ND; ++j) {
        const int i = tid + 256 * j, pp = i >> 4, k4 = i & 15;
        d_vo[j] = (k0 + k4 * 4 < K) ? (unsigned)(((pp / WD) * W + pp % WD) * K * 4 + k4 * 16) : 0xFFFFFFFFu;   // K < 64: masked k-tile
    }
    // (row << 8 | col) of a slot inside the patch, for the border tests: recomputed per stage from an opaque copy of the thread
    // index (a few integer ops per load) instead of held in registers across the MFMA loop -- the kernel sits at 256 VGPRs and
    // every register kept out of the loop is a spill reload less in it
    // (only where it pays: the 2 x 16 patch variant spilled 60 bytes per lane and its 112 x 112 layers ran 16 % slower for
    // it; the other variants fit and keep the codes in registers -- recomputing costs them 2-5 %)
    constexpr bool RC_RECOMP = (R == 2 && WD == 16);
    unsigned x_rc[RC_RECOMP ? 1 : NX], d_rc[RC_RECOMP ? 1 : ND];
    if constexpr (!RC_RECOMP) {
#pragma unroll
        for (int j = 0; j < NX; ++j) {
            const int pos = (tid + 256 * j) >> 4, hr = pos / HPW;
            x_rc[j] = (unsigned)(hr << 8 | (pos - hr * HPW));
        }
#pragma unroll
        for (int j = 0; j < ND; ++j) {
            const int pp = (tid + 256 * j) >> 4;
            d_rc[j] = (unsigned)((pp / WD) << 8 | (pp % WD));
        }
    }
    auto x_rc_of = [&](const int j) -> unsigned {
        if constexpr (!RC_RECOMP) {
            return x_rc[j];
        } else {
            int t_ = tid;
            asm volatile("" : "+v"(t_));
            const int pos = (t_ + 256 * j) >> 4, hr = pos / HPW;
            return (unsigned)(hr << 8 | (pos - hr * HPW));
        }
    };
    auto d_rc_of = [&](const int j) -> unsigned {
        if constexpr (!RC_RECOMP) {
            return d_rc[j];
        } else {
            int t_ = tid;
            asm volatile("" : "+v"(t_));
            const int pp = (t_ + 256 * j) >> 4;
            return (unsigned)((pp / WD) << 8 | (pp % WD));
        }
    };

    f32x4 rx[NX], rd[ND];
    // patch cursor of the NEXT fetch, advanced by one patch per call (stages are consecutive patches): three scalar adds and
    // compares per stage instead of four 64-bit divisions on the vector ALU (which also made the scalar offsets of the buffer
    // loads look divergent, wrapping every load in a waterfall loop)
    int nx0 = (int)(g0 % cpr) * WD, ny0 = (int)((g0 / cpr) % rpi) * R, nb = (int)((g0 / cpr) / rpi);
    auto gload = [&]() {
        const int x0 = nx0, y0 = ny0;
        const unsigned b = (unsigned)nb;
        nx0 += WD;
        if (nx0 >= cpr * WD) {
            nx0 = 0;
            ny0 += R;
            if (ny0 >= rpi * R) {
                ny0 = 0;
                ++nb;
            }
        }
        if constexpr (!UPS) {
            // valid halo rows hr in [rlo, rhi], cols hx in [clo, chi]  (source pixel = (y0 + hr - 1, x0 + hx - 1))
            const unsigned rlo = (y0 == 0) ? 1u : 0u, rn = (unsigned)((H - y0 < R + 1) ? (H - y0) : (R + 1)) - rlo;
            const unsigned clo = (x0 == 0) ? 1u : 0u, cn = (unsigned)((W - x0 < WD + 1) ? (W - x0) : (WD + 1)) - clo;
            const unsigned so_x = (unsigned)((((b * H + y0) * W + x0) * C + c0) * 4);     // + x_bias - x_bias
            const unsigned so_d = (unsigned)((((b * H + y0) * W + x0) * K + k0) * 4);
#pragma unroll
            for (int j = 0; j < NX; ++j) {
                const unsigned rc = x_rc_of(j);
                const bool ok = ((rc >> 8) - rlo <= rn) && ((rc & 255u) - clo <= cn);
                rx[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(x_rs, ok ? x_vo[j] : 0xFFFFFFFFu, so_x, 0));
            }
            const unsigned rmax = (unsigned)(H - y0), cmax = (unsigned)(W - x0);              // py < rmax, px < cmax
#pragma unroll
            for (int j = 0; j < ND; ++j) {
                const unsigned rc = d_rc_of(j);
                const bool ok = ((rc >> 8) < rmax) && ((rc & 255u) < cmax);
                rd[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(d_rs, ok ? d_vo[j] : 0xFFFFFFFFu, so_d, 0));
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < NX; ++j) {
            const int i = tid + 256 * j;
            const int pos = i >> 4, c4 = i & 15;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (pos < NH) {
                const int hr = pos / HPW, hx = pos - hr * HPW;
                const int iy = y0 + hr - 1, ix = x0 + hx - 1;
                if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) {
                    const int sy = UPS ? (iy >> 1) : iy, sx = UPS ? (ix >> 1) : ix;
                    v = *reinterpret_cast<const f32x4*>(x + ((b * Hs + sy) * (long)Ws + sx) * C + c0 + c4 * 4);
                }
            }
            rx[j] = v;
        }
#pragma unroll
        for (int j = 0; j < ND; ++j) {
            const int i = tid + 256 * j;
            const int pp = i >> 4, k4 = i & 15;
            const int yy = y0 + pp / WD, xx = x0 + pp % WD;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (yy < H && xx < W) v = *reinterpret_cast<const f32x4*>(dy + ((b * H + yy) * (long)W + xx) * K + k0 + k4 * 4);
            rd[j] = v;
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int j = 0; j < NX; ++j) {
            const int i = tid + 256 * j;
            const int pos = i >> 4, c4 = i & 15;
            if (pos < NH) {
                u32x2_t hi, lo;
                if constexpr (XPRE) {
                    const u32x4_t bq = __builtin_bit_cast(u32x4_t, rx[j]);
                    hi = u32x2_t{bq[0], bq[1]};
                    lo = u32x2_t{bq[2], bq[3]};
                    if constexpr (egz_drop_alo<T>::value) hi = f16_pair_rne(hi, lo);
                } else if constexpr (egz_drop_alo<T>::value) {
                    f16_rne4(rx[j] * x_scale, hi);
                } else {
                    W16<T>::split4s(rx[j], x_scale, hi, lo);
                }
                unsigned short* d = Xs + buf * XB + (c4 >> 3) * XH + pos * 32 + (c4 & 7) * 4;
                *reinterpret_cast<u32x2_t*>(d) = hi;
                if constexpr (!egz_drop_alo<T>::value) *reinterpret_cast<u32x2_t*>(d + 2 * XH) = lo;      // (x hi-only: its lo plane is never read)
            }
        }
#pragma unroll
        for (int j = 0; j < ND; ++j) {
            const int i = tid + 256 * j;
            const int pp = i >> 4, k4 = i & 15;
            u32x2_t hi, lo;
            if constexpr (DPRE) {
                const u32x4_t bq = __builtin_bit_cast(u32x4_t, rd[j]);
                hi = u32x2_t{bq[0], bq[1]};
                lo = u32x2_t{bq[2], bq[3]};
            } else {
                W16<T>::split4s(rd[j], d_scale, hi, lo);
            }
            unsigned short* d = Ds + buf * DB + (k4 >> 3) * DH + pp * 32 + (k4 & 7) * 4;
            *reinterpret_cast<u32x2_t*>(d) = hi;
            *reinterpret_cast<u32x2_t*>(d + 2 * DH) = lo;
        }
    };


    // transpose-read addressing: 16-lane group g = lane >> 4 covers channels 16 * (g & 1) .. +15 of the wave's half and
    // reduction elements 8 * (g >> 1) .. +7 (two reads of 4 pixels); lane u of the group hands in pixel (u >> 2),
    // channel chunk (u & 3)
    const int u = lane & 15, hh = lane >> 5;
    const int chan = 16 * ((lane >> 4) & 1) + 4 * (u & 3);
    const int lp = 8 * hh + (u >> 2);                                  // pixel of the 16-pixel chunk (first read)
    const int xlane = (WD >= 16) ? lp : (hh * HPW + (u >> 2));          // halo position of that pixel (tap 0,0)
    const EGZ_LDS unsigned short* Xl = (const EGZ_LDS unsigned short*)(Xs) + wc * XH + xlane * 32 + chan;
    const EGZ_LDS unsigned short* Dl = (const EGZ_LDS unsigned short*)(Ds) + wk * DH + lp * 32 + chan;
    auto xoff = [&](int ks, int q, int tap) -> int {                   // compile-time after unrolling
        const int base = (WD == 32) ? (16 * ks + 4 * q) : (WD == 16) ? (ks * HPW + 4 * q) : (2 * ks * HPW + 4 * q);
        return (base + (tap / 3) * HPW + (tap % 3)) * 32;
    };
    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    if (g0 < g1) {
        gload();
        lstore(0);
    }
    __syncthreads();
    for (long g = g0; g < g1; ++g) {
        const int buf = (int)((g - g0) & 1);
        if (g + 1 < g1) gload();
        const EGZ_LDS unsigned short* Xb = Xl + buf * XB;
        const EGZ_LDS unsigned short* Db = Dl + buf * DB;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const typename W16<T>::vec8 dh = W16<T>::frag(Db + (16 * ks) * 32, Db + (16 * ks + 4) * 32);
            const typename W16<T>::vec8 dl = W16<T>::frag(Db + 2 * DH + (16 * ks) * 32, Db + 2 * DH + (16 * ks + 4) * 32);
            // taps in groups of three: the three products of one accumulator are issued three MFMAs apart
#pragma unroll
            for (int tr = 0; tr < 3; ++tr) {
                typename W16<T>::vec8 xh[3], xl[3];
#pragma unroll
                for (int ts = 0; ts < 3; ++ts) {
                    const int tap = tr * 3 + ts;
                    xh[ts] = W16<T>::frag(Xb + xoff(ks, 0, tap), Xb + xoff(ks, 1, tap));
                    xl[ts] = W16<T>::frag(Xb + 2 * XH + xoff(ks, 0, tap), Xb + 2 * XH + xoff(ks, 1, tap));
                }
                // (issuing the next stage's split + LDS stores piecewise behind the last MFMAs, as the forward kernel does, measured
                // slower here -- 4.14 vs 3.82 ms over the 12 layer shapes, profiles/r03_wgrad_ab.txt -- and was removed)
#pragma unroll
                for (int term = 0; term < 3; ++term)
#pragma unroll
                    for (int ts = 0; ts < 3; ++ts)
                        if (!(egz_drop_alo<T>::value && term == 0)) acc[tr * 3 + ts] = W16<T>::mfma(term == 0 ? xl[ts] : xh[ts], term == 1 ? dl : dh, acc[tr * 3 + ts]);
            }
        }
        if (g + 1 < g1) lstore(buf ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        float* out = part + ((long)split * 9 + tap) * C * K;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int c = c0 + wc * 32 + egz_acc_row(r, lane);
            const int k = k0 + wk * 32 + l31;
            if (c < C && k < K) out[(long)c * K + k] = acc[tap][r] * d_inv * x_inv;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Narrow form of the split-half 9-tap kernel for C <= 32 and K <= 32 (the late-fusion widths, late_fusion.py:10-13): the
// whole (c, k) range is ONE 32 x 32 MFMA tile, so the 2 x 2 wave grid of the kernel above would leave three waves multiplying
// zeros.  Here the four waves split the REDUCTION instead: a stage is a patch of R x WD = 64 pixels, wave w owns its w-th
// 16-pixel run (one MFMA k-step, 27 MFMAs for the nine taps) and keeps its own nine accumulators; the four partial sums are
// added through LDS in a fixed order at the end.  Only 32 channels are staged (8 float4 per pixel), half of the wide
// kernel's fetch / split work per pixel.  Same LDS image per plane and the same ds_read_b64_tr_b16 addressing.
// BNIN: x is the PRE-BatchNorm output of the block below (its normalised form is never materialised, see
// conv3x3_x3p_narrow_kernel) and x_bn its BatchNorm's 4 x C coefficient rows: relu(x * scale + shift) is applied per channel
// while the halo is staged (out-of-image positions stay zero); x_absmax = the max of the normalised values.
template <typename T, int R, int WD, bool BNIN>
__global__ __launch_bounds__(256, 2) void conv3x3_wgrad9_x3n_kernel(
    const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ part, int B, int H, int W,
    int C, int K, int patches_per_split, const unsigned int* __restrict__ dy_absmax,
    const unsigned int* __restrict__ x_absmax, const float* __restrict__ x_bn) {
    static_assert(R * WD == 64 && (WD == 32 || WD == 16), "patch = 64 pixels in rows of 16-pixel runs");
    const float d_scale = W16<T>::scale(dy_absmax), d_inv = 1.f / d_scale;
    const float x_scale = W16<T>::scale(x_absmax), x_inv = 1.f / x_scale;      // the activation operand, scaled the same way
    constexpr int NP = R * WD;
    constexpr int HPW = WD + 2, NH = (R + 2) * HPW;            // halo row width / halo pixels
    constexpr int XH = NH * 32 + 32, DH = NP * 32 + 32;        // plane strides (elements)
    constexpr int XB = 2 * XH, DB = 2 * DH;                    // per-buffer strides ([plane hi / lo])
    constexpr int NX = (NH * 8 + 255) / 256;                   // float4 halo loads per thread
    constexpr int ND = (NP * 8) / 256;                         // float4 dY loads per thread (2)
    __shared__ __attribute__((aligned(16))) unsigned short Xs[2 * XB];
    __shared__ __attribute__((aligned(16))) unsigned short Ds[2 * DB];
    static_assert(sizeof(unsigned short) * 2 * XB >= 4 * 1024 * sizeof(float), "final reduction reuses the halo buffers");

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31;
    const int split = blockIdx.y;
    const int cpr = (W + WD - 1) / WD, rpi = (H + R - 1) / R;  // patches per row / patch rows per image
    const long npatch = (long)B * rpi * cpr;
    const long g0 = (long)split * patches_per_split;
    const long g1 = (g0 + patches_per_split < npatch) ? (g0 + patches_per_split) : npatch;

    // operand fetch as in conv3x3_wgrad9_x3_kernel: buffer loads, fixed per-thread slot offsets, out-of-image -> zeros
    const unsigned x_bias = (unsigned)(W + 1) * (unsigned)C * 4u;
    const __amdgpu_buffer_rsrc_t x_rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(x)) - x_bias, 0, (int)((unsigned)B * H * W * C * 4u + x_bias), 0x00020000);
    const __amdgpu_buffer_rsrc_t d_rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(dy), 0, (int)((unsigned)B * H * W * K * 4u), 0x00020000);
    unsigned x_vo[NX], x_rc[NX], d_vo[ND], d_rc[ND];
#pragma unroll
    for (int j = 0; j < NX; ++j) {
        const int i = tid + 256 * j, pos = i >> 3, c4 = i & 7;
        const int hr = pos / HPW, hx = pos - hr * HPW;
        x_vo[j] = (pos < NH && c4 * 4 < C) ? (unsigned)((hr * W + hx) * C * 4 + c4 * 16) : 0xFFFFFFFFu;
        x_rc[j] = (unsigned)(hr << 8 | hx);
    }
#pragma unroll
    for (int j = 0; j < ND; ++j) {
        const int i = tid + 256 * j, pp = i >> 3, k4 = i & 7;
        d_vo[j] = (k4 * 4 < K) ? (unsigned)(((pp / WD) * W + pp % WD) * K * 4 + k4 * 16) : 0xFFFFFFFFu;
        d_rc[j] = (unsigned)((pp / WD) << 8 | (pp % WD));
    }
    f32x4 rx[NX], rd[ND];
    unsigned rx_ok = 0;                                         // BNIN: bit j = rx[j] came from inside the image
    f32x4 in_sc = {0.f, 0.f, 0.f, 0.f}, in_sh = {0.f, 0.f, 0.f, 0.f};
    if (BNIN && (tid & 7) * 4 < C) {                           // the thread's channel quad is fixed (c4 = tid & 7)
        in_sc = *reinterpret_cast<const f32x4*>(x_bn + 2 * C + (tid & 7) * 4);
        in_sh = *reinterpret_cast<const f32x4*>(x_bn + 3 * C + (tid & 7) * 4);
    }
    // patch cursor of the NEXT fetch, advanced by one patch per call (stages are consecutive patches): three scalar adds and
    // compares per stage instead of four 64-bit divisions on the vector ALU (which also made the scalar offsets of the buffer
    // loads look divergent, wrapping every load in a waterfall loop)
    int nx0 = (int)(g0 % cpr) * WD, ny0 = (int)((g0 / cpr) % rpi) * R, nb = (int)((g0 / cpr) / rpi);
    auto gload = [&]() {
        const int x0 = nx0, y0 = ny0;
        const unsigned b = (unsigned)nb;
        nx0 += WD;
        if (nx0 >= cpr * WD) {
            nx0 = 0;
            ny0 += R;
            if (ny0 >= rpi * R) {
                ny0 = 0;
                ++nb;
            }
        }
        const unsigned rlo = (y0 == 0) ? 1u : 0u, rn = (unsigned)((H - y0 < R + 1) ? (H - y0) : (R + 1)) - rlo;
        const unsigned clo = (x0 == 0) ? 1u : 0u, cn = (unsigned)((W - x0 < WD + 1) ? (W - x0) : (WD + 1)) - clo;
        const unsigned so_x = (unsigned)((((b * H + y0) * W + x0) * C) * 4);
        const unsigned so_d = (unsigned)((((b * H + y0) * W + x0) * K) * 4);
        unsigned okm = 0;
#pragma unroll
        for (int j = 0; j < NX; ++j) {
            const bool ok = ((x_rc[j] >> 8) - rlo <= rn) && ((x_rc[j] & 255u) - clo <= cn);
            if (BNIN && ok && x_vo[j] != 0xFFFFFFFFu) okm |= 1u << j;
            rx[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(x_rs, ok ? x_vo[j] : 0xFFFFFFFFu, so_x, 0));
        }
        rx_ok = okm;
        const unsigned rmax = (unsigned)(H - y0), cmax = (unsigned)(W - x0);
#pragma unroll
        for (int j = 0; j < ND; ++j) {
            const bool ok = ((d_rc[j] >> 8) < rmax) && ((d_rc[j] & 255u) < cmax);
            rd[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(d_rs, ok ? d_vo[j] : 0xFFFFFFFFu, so_d, 0));
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int j = 0; j < NX; ++j) {
            const int i = tid + 256 * j, pos = i >> 3, c4 = i & 7;
            if (pos < NH) {
                u32x2_t hi, lo;
                if (BNIN) {                                     // BatchNorm + ReLU of the block below, on the way into LDS;
                    const float ms = ((rx_ok >> j) & 1u) ? x_scale : 0.f;  // the zero-padding mask rides on the scale factor
#pragma unroll
                    for (int e = 0; e < 4; ++e) rx[j][e] = fmaxf(__builtin_fmaf(rx[j][e], in_sc[e], in_sh[e]), 0.f) * ms;
                    if constexpr (egz_drop_alo<T>::value) f16_rne4(rx[j], hi);
                    else W16<T>::split4(rx[j], hi, lo);
                } else if constexpr (egz_drop_alo<T>::value) {
                    f16_rne4(rx[j] * x_scale, hi);
                } else {
                    W16<T>::split4s(rx[j], x_scale, hi, lo);
                }
                unsigned short* d = Xs + buf * XB + pos * 32 + c4 * 4;
                *reinterpret_cast<u32x2_t*>(d) = hi;
                if constexpr (!egz_drop_alo<T>::value) *reinterpret_cast<u32x2_t*>(d + XH) = lo;
            }
        }
#pragma unroll
        for (int j = 0; j < ND; ++j) {
            const int i = tid + 256 * j, pp = i >> 3, k4 = i & 7;
            u32x2_t hi, lo;
            W16<T>::split4s(rd[j], d_scale, hi, lo);
            unsigned short* d = Ds + buf * DB + pp * 32 + k4 * 4;
            *reinterpret_cast<u32x2_t*>(d) = hi;
            *reinterpret_cast<u32x2_t*>(d + DH) = lo;
        }
    };

    // transpose-read addressing of the wide kernel; the wave's 16-pixel run starts at halo position wrun (tap 0, 0)
    const int u = lane & 15, hh = lane >> 5;
    const int chan = 16 * ((lane >> 4) & 1) + 4 * (u & 3);
    const int lp = 8 * hh + (u >> 2);
    const int wrun = (WD == 32) ? ((wave >> 1) * HPW + 16 * (wave & 1)) : (wave * HPW);
    const EGZ_LDS unsigned short* Xl = (const EGZ_LDS unsigned short*)(Xs) + (wrun + lp) * 32 + chan;
    const EGZ_LDS unsigned short* Dl = (const EGZ_LDS unsigned short*)(Ds) + (16 * wave + lp) * 32 + chan;
    auto xoff = [&](int q, int tap) -> int { return (4 * q + (tap / 3) * HPW + (tap % 3)) * 32; };
    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    if (g0 < g1) {
        gload();
        lstore(0);
    }
    __syncthreads();
    for (long g = g0; g < g1; ++g) {
        const int buf = (int)((g - g0) & 1);
        if (g + 1 < g1) gload();
        const EGZ_LDS unsigned short* Xb = Xl + buf * XB;
        const EGZ_LDS unsigned short* Db = Dl + buf * DB;
        const typename W16<T>::vec8 dh = W16<T>::frag(Db, Db + 4 * 32);
        const typename W16<T>::vec8 dl = W16<T>::frag(Db + DH, Db + DH + 4 * 32);
#pragma unroll
        for (int tr = 0; tr < 3; ++tr) {
            typename W16<T>::vec8 xh[3], xl[3];
#pragma unroll
            for (int ts = 0; ts < 3; ++ts) {
                const int tap = tr * 3 + ts;
                xh[ts] = W16<T>::frag(Xb + xoff(0, tap), Xb + xoff(1, tap));
                xl[ts] = W16<T>::frag(Xb + XH + xoff(0, tap), Xb + XH + xoff(1, tap));
            }
#pragma unroll
            for (int term = 0; term < 3; ++term)
#pragma unroll
                for (int ts = 0; ts < 3; ++ts)
                    if (!(egz_drop_alo<T>::value && term == 0)) acc[tr * 3 + ts] = W16<T>::mfma(term == 0 ? xl[ts] : xh[ts], term == 1 ? dl : dh, acc[tr * 3 + ts]);
        }
        if (g + 1 < g1) lstore(buf ^ 1);
        __syncthreads();
    }
    // the four waves' partial tiles, summed in wave order through LDS (one tap at a time)
    float* red = reinterpret_cast<float*>(Xs);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[tap][r];
        __syncthreads();
        float* out = part + ((long)split * 9 + tap) * C * K;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int e = tid + 256 * m, r = e >> 6, ln = e & 63;
            const float v = ((red[e] + red[1024 + e]) + red[2048 + e]) + red[3072 + e];
            const int c = egz_acc_row(r, ln), k = ln & 31;
            if (c < C && k < K) out[(long)c * K + k] = v * d_inv * x_inv;
        }
        __syncthreads();
    }
    (void)l31;
}

// ---------------------------------------------------------------------------------------------------------
// Tap-packed form of the narrow kernel for K <= 8 filters (late_fusion.py:12: Conv2d(32, 8)).  With the nine taps as nine 32-column
// tiles, three quarters of every MFMA of conv3x3_wgrad9_x3n_kernel multiply the zero padding of the 8 -> 32 column tile (the
// launch was matrix-core bound: 107 us for 59 algorithmic GFLOP).  Here the SHIFT moves to the gradient operand,
//     dw[c][k][tap] = sum_q x[q][c] * dy[q - off(tap)][k]          (q over the image, dy zero outside),
// so the activation fragment of a 16-pixel run is read once, unshifted, and the GEMM columns are (tap, k) pairs: 72 columns in
// three 32-column tiles (four taps each; the last tile carries tap 8 and three copies whose results are dropped) -- 9 MFMAs per 16
// pixels instead of 27.  The per-lane addresses of ds_read_b64_tr_b16 make the gather free: a 16-lane group's four channel
// chunks are (tap a, k 0-3), (tap a, k 4-7), (tap a + 1, k 0-3), (tap a + 1, k 4-7), each reading its own shifted pixel of the
// staged dy halo ([halo pixel][8 k], 16 bytes per pixel).  Stage = the same R x WD = 64-pixel patch, wave w owns its w-th
// 16-pixel run, partial tiles summed through LDS in wave order, [split][tap][C][K] partials for wgrad_reduce: as the kernel above.
template <typename T, int R, int WD, bool BNIN>
__global__ __launch_bounds__(256, 2) void conv3x3_wgrad9_x3t_kernel(
    const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ part, int B, int H, int W,
    int C, int K, int patches_per_split, const unsigned int* __restrict__ dy_absmax,
    const unsigned int* __restrict__ x_absmax, const float* __restrict__ x_bn) {
    static_assert(R * WD == 64 && (WD == 32 || WD == 16), "patch = 64 pixels in rows of 16-pixel runs");
    const float d_scale = W16<T>::scale(dy_absmax), d_inv = 1.f / d_scale;
    const float x_scale = W16<T>::scale(x_absmax), x_inv = 1.f / x_scale;
    constexpr int NP = R * WD;
    constexpr int HPW = WD + 2, NH = (R + 2) * HPW;            // dy halo row width / halo pixels
    constexpr int XH = NP * 32 + 32;                           // x plane stride (elements): [pixel][32 channels]
    constexpr int DH = NH * 8 + 8;                             // dy plane stride: [halo pixel][8 k]
    constexpr int XB = 2 * XH, DB = 2 * DH;
    constexpr int NX = (NP * 8) / 256;                         // float4 x loads per thread (2)
    constexpr int ND = (NH * 2 + 255) / 256;                   // float4 dy halo loads per thread (1 or 2)
    __shared__ __attribute__((aligned(16))) unsigned short Xs[2 * XB];
    __shared__ __attribute__((aligned(16))) unsigned short Ds[2 * DB];
    static_assert(sizeof(unsigned short) * 2 * XB >= 4 * 1024 * sizeof(float), "final reduction reuses the activation buffers");

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int split = blockIdx.y;
    const int cpr = (W + WD - 1) / WD, rpi = (H + R - 1) / R;
    const long npatch = (long)B * rpi * cpr;
    const long g0 = (long)split * patches_per_split;
    const long g1 = (g0 + patches_per_split < npatch) ? (g0 + patches_per_split) : npatch;

    // the dy resource starts (W + 1) * K floats early so that the (-1, -1) halo corner keeps lane offsets non-negative
    const unsigned d_bias = (unsigned)(W + 1) * (unsigned)K * 4u;
    const __amdgpu_buffer_rsrc_t x_rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(x), 0, (int)((unsigned)B * H * W * C * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t d_rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(dy)) - d_bias, 0, (int)((unsigned)B * H * W * K * 4u + d_bias), 0x00020000);
    unsigned x_vo[NX], x_rc[NX], d_vo[ND], d_rc[ND];
#pragma unroll
    for (int j = 0; j < NX; ++j) {
        const int i = tid + 256 * j, pp = i >> 3, c4 = i & 7;
        x_vo[j] = (c4 * 4 < C) ? (unsigned)(((pp / WD) * W + pp % WD) * C * 4 + c4 * 16) : 0xFFFFFFFFu;
        x_rc[j] = (unsigned)((pp / WD) << 8 | (pp % WD));
    }
#pragma unroll
    for (int j = 0; j < ND; ++j) {
        const int i = tid + 256 * j, pos = i >> 1, k4 = i & 1;
        const int hr = pos / HPW, hx = pos - hr * HPW;
        d_vo[j] = (pos < NH && k4 * 4 < K) ? (unsigned)((hr * W + hx) * K * 4 + k4 * 16) : 0xFFFFFFFFu;
        d_rc[j] = (unsigned)(hr << 8 | hx);
    }
    f32x4 rx[NX], rd[ND];
    unsigned rx_ok = 0;
    f32x4 in_sc = {0.f, 0.f, 0.f, 0.f}, in_sh = {0.f, 0.f, 0.f, 0.f};
    if (BNIN && (tid & 7) * 4 < C) {
        in_sc = *reinterpret_cast<const f32x4*>(x_bn + 2 * C + (tid & 7) * 4);
        in_sh = *reinterpret_cast<const f32x4*>(x_bn + 3 * C + (tid & 7) * 4);
    }
    int nx0 = (int)(g0 % cpr) * WD, ny0 = (int)((g0 / cpr) % rpi) * R, nb = (int)((g0 / cpr) / rpi);
    auto gload = [&]() {
        const int x0 = nx0, y0 = ny0;
        const unsigned b = (unsigned)nb;
        nx0 += WD;
        if (nx0 >= cpr * WD) {
            nx0 = 0;
            ny0 += R;
            if (ny0 >= rpi * R) {
                ny0 = 0;
                ++nb;
            }
        }
        // valid dy halo rows hr in [rlo, rlo + rn], cols hx in [clo, clo + cn]  (source pixel = (y0 + hr - 1, x0 + hx - 1))
        const unsigned rlo = (y0 == 0) ? 1u : 0u, rn = (unsigned)((H - y0 < R + 1) ? (H - y0) : (R + 1)) - rlo;
        const unsigned clo = (x0 == 0) ? 1u : 0u, cn = (unsigned)((W - x0 < WD + 1) ? (W - x0) : (WD + 1)) - clo;
        const unsigned so_x = (unsigned)((((b * H + y0) * W + x0) * C) * 4);
        const unsigned so_d = (unsigned)((((b * H + y0) * W + x0) * K) * 4);         // + d_bias - d_bias
        const unsigned rmax = (unsigned)(H - y0), cmax = (unsigned)(W - x0);
        unsigned okm = 0;
#pragma unroll
        for (int j = 0; j < NX; ++j) {
            const bool ok = ((x_rc[j] >> 8) < rmax) && ((x_rc[j] & 255u) < cmax);
            if (BNIN && ok && x_vo[j] != 0xFFFFFFFFu) okm |= 1u << j;
            rx[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(x_rs, ok ? x_vo[j] : 0xFFFFFFFFu, so_x, 0));
        }
        rx_ok = okm;
#pragma unroll
        for (int j = 0; j < ND; ++j) {
            const bool ok = ((d_rc[j] >> 8) - rlo <= rn) && ((d_rc[j] & 255u) - clo <= cn);
            rd[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(d_rs, ok ? d_vo[j] : 0xFFFFFFFFu, so_d, 0));
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int j = 0; j < NX; ++j) {
            const int i = tid + 256 * j, pp = i >> 3, c4 = i & 7;
            u32x2_t hi, lo;
            if (BNIN) {
                const float ms = ((rx_ok >> j) & 1u) ? x_scale : 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) rx[j][e] = fmaxf(__builtin_fmaf(rx[j][e], in_sc[e], in_sh[e]), 0.f) * ms;
                if constexpr (egz_drop_alo<T>::value) f16_rne4(rx[j], hi);
                else W16<T>::split4(rx[j], hi, lo);
            } else if constexpr (egz_drop_alo<T>::value) {
                f16_rne4(rx[j] * x_scale, hi);
            } else {
                W16<T>::split4s(rx[j], x_scale, hi, lo);
            }
            unsigned short* d = Xs + buf * XB + pp * 32 + c4 * 4;
            *reinterpret_cast<u32x2_t*>(d) = hi;
            if constexpr (!egz_drop_alo<T>::value) *reinterpret_cast<u32x2_t*>(d + XH) = lo;
        }
#pragma unroll
        for (int j = 0; j < ND; ++j) {
            const int i = tid + 256 * j, pos = i >> 1, k4 = i & 1;
            if (pos < NH) {
                u32x2_t hi, lo;
                W16<T>::split4s(rd[j], d_scale, hi, lo);
                unsigned short* d = Ds + buf * DB + pos * 8 + k4 * 4;
                *reinterpret_cast<u32x2_t*>(d) = hi;
                *reinterpret_cast<u32x2_t*>(d + DH) = lo;
            }
        }
    };

    // transpose-read addressing (see conv3x3_wgrad9_x3_kernel): lane u of a 16-lane group hands in pixel (u >> 2) of the group's
    // four, chunk (u & 3); group g covers reduction elements 8 (g >> 1) .. +7 and the columns / channels 16 (g & 1) .. +15
    const int u = lane & 15, hh = lane >> 5, gh = (lane >> 4) & 1;
    const int lp = 8 * hh + (u >> 2);                          // pixel of the wave's 16-pixel run (first read; the second: + 4)
    const EGZ_LDS unsigned short* Xl = (const EGZ_LDS unsigned short*)(Xs) + (16 * wave + lp) * 32 + 16 * gh + 4 * (u & 3);
    // dy: column chunk (u & 3) of group half gh of tile j = (tap 4 j + 2 gh + ((u & 3) >> 1), k quad (u & 3) & 1)
    const int prow = (WD == 32) ? (wave >> 1) : wave, pcol = ((WD == 32) ? 16 * (wave & 1) : 0) + lp;
    int doff[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        int tap = 4 * j + 2 * gh + ((u & 3) >> 1);
        tap = tap > 8 ? 8 : tap;
        const int oy = tap / 3 - 1, ox = tap % 3 - 1;
        doff[j] = ((prow + 1 - oy) * HPW + (pcol + 1 - ox)) * 8 + ((u & 3) & 1) * 4;
    }
    const EGZ_LDS unsigned short* Dl = (const EGZ_LDS unsigned short*)(Ds);
    f32x16 acc[3];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    if (g0 < g1) {
        gload();
        lstore(0);
    }
    __syncthreads();
    for (long g = g0; g < g1; ++g) {
        const int buf = (int)((g - g0) & 1);
        if (g + 1 < g1) gload();
        const EGZ_LDS unsigned short* Xb = Xl + buf * XB;
        const EGZ_LDS unsigned short* Db = Dl + buf * DB;
        const typename W16<T>::vec8 xh = W16<T>::frag(Xb, Xb + 4 * 32);
        const typename W16<T>::vec8 xl = W16<T>::frag(Xb + XH, Xb + XH + 4 * 32);
        typename W16<T>::vec8 dh[3], dl[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {                           // (the second read: 4 pixels further along the run = 4 halo columns)
            dh[j] = W16<T>::frag(Db + doff[j], Db + doff[j] + 4 * 8);
            dl[j] = W16<T>::frag(Db + DH + doff[j], Db + DH + doff[j] + 4 * 8);
        }
#pragma unroll
        for (int term = 0; term < 3; ++term)
#pragma unroll
            for (int j = 0; j < 3; ++j)
                if (!(egz_drop_alo<T>::value && term == 0)) acc[j] = W16<T>::mfma(term == 0 ? xl : xh, term == 1 ? dl[j] : dh[j], acc[j]);
        if (g + 1 < g1) lstore(buf ^ 1);
        __syncthreads();
    }
    // the four waves' partial tiles, summed in wave order through LDS (one column tile = four taps at a time)
    float* red = reinterpret_cast<float*>(Xs);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[j][r];
        __syncthreads();
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int e = tid + 256 * m, r = e >> 6, ln = e & 63;
            const float v = ((red[e] + red[1024 + e]) + red[2048 + e]) + red[3072 + e];
            const int c = egz_acc_row(r, ln), n = ln & 31, tap = 4 * j + (n >> 3), k = n & 7;
            if (c < C && k < K && tap < 9) part[((long)split * 9 + tap) * C * K + (long)c * K + k] = v * d_inv * x_inv;
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------
// Split-half weight gradient of [nearest x2 upsample -> conv3x3] in phase form (4/9 of the MACs of the folded 9-tap form):
//   dWeff[py][px][a][b][c][k] = sum_{b,y,x} X[y+a+py-1][x+b+px-1][c] * dY[2y+py][2x+px][k]      (LOW-res y, x)
// One block = one 64(c) x 64(k) tile of BOTH column phases of one row phase py (blockIdx.z): 8 accumulators per wave
// (px, a, b).  A stage is an R x WD patch of low-res pixels: its (R+1) x (WD+2) input halo and the two stride-2 dY
// sub-grids (px = 0, 1) are split into 16-bit hi / lo halves and staged once; the A fragment at halo column offset px + b is
// shared by the two (px, b) pairs that reach it.  Same LDS image / ds_read_b64_tr_b16 addressing as
// conv3x3_wgrad9_x3_kernel; partial tiles in the [split][16 phase-taps][C][K] layout of wgrad_reduce_ups_kernel.
template <typename T, int R, int WD>
__global__ __launch_bounds__(256, 2) void conv3x3_wgrad_ups_x3_kernel(
    const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ part, int B, int H, int W,
    int C, int K, int patches_per_split, const unsigned int* __restrict__ dy_absmax,
    const unsigned int* __restrict__ x_absmax) {
    static_assert(R * WD == 32 && (WD == 32 || WD == 16 || WD == 8), "patch = 32 low-res pixels");
    const float d_scale = W16<T>::scale(dy_absmax), d_inv = 1.f / d_scale;
    const float x_scale = W16<T>::scale(x_absmax), x_inv = 1.f / x_scale;      // the activation operand, scaled the same way
    constexpr int NP = R * WD, KS = NP / 16;
    constexpr int HPW = WD + 2, NH = (R + 1) * HPW;
    constexpr int XH = NH * 32 + ((NH & 1) ? 0 : 32);          // channel-half stride (elements): bytes % 128 == 64
    constexpr int DH = 2 * NP * 32 + 32;                       // both column phases: slot = px * NP + pixel
    constexpr int XB = 4 * XH, DB = 4 * DH;
    constexpr int NX = (NH * 16 + 255) / 256;
    constexpr int ND = (2 * NP * 16) / 256;                    // 4
    __shared__ __attribute__((aligned(16))) unsigned short Xs[2 * XB];
    __shared__ __attribute__((aligned(16))) unsigned short Ds[2 * DB];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wc = wave >> 1, wk = wave & 1, l31 = lane & 31;
    const int tk = K / 64;
    int tile, split;
    wgrad_block(tile, split);
    const int c0 = (tile / tk) * 64, k0 = (tile % tk) * 64;
    const int py = blockIdx.z;
    const int Hl = H >> 1, Wl = W >> 1;                        // low-res dims (H, W: the conv's hi-res output dims)
    const int cpr = (Wl + WD - 1) / WD, rpi = (Hl + R - 1) / R;
    const long npatch = (long)B * rpi * cpr;
    const long g0 = (long)split * patches_per_split;
    const long g1 = (g0 + patches_per_split < npatch) ? (g0 + patches_per_split) : npatch;

    const unsigned x_bias = (unsigned)(Wl + 1) * (unsigned)C * 4u;
    const __amdgpu_buffer_rsrc_t x_rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(x)) - x_bias, 0, (int)((unsigned)B * Hl * Wl * C * 4u + x_bias), 0x00020000);
    const __amdgpu_buffer_rsrc_t d_rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(dy), 0, (int)((unsigned)B * H * W * K * 4u), 0x00020000);
    unsigned x_vo[NX], x_rc[NX], d_vo[ND], d_rc[ND];
#pragma unroll
    for (int j = 0; j < NX; ++j) {
        const int i = tid + 256 * j, pos = i >> 4, c4 = i & 15;
        const int hr = pos / HPW, hx = pos - hr * HPW;
        x_vo[j] = (pos < NH) ? (unsigned)((hr * Wl + hx) * C * 4 + c4 * 16) : 0xFFFFFFFFu;
        x_rc[j] = (unsigned)(hr << 8 | hx);
    }
#pragma unroll
    for (int j = 0; j < ND; ++j) {
        const int i = tid + 256 * j, slot = i >> 4, k4 = i & 15;
        const int px = slot / NP, pp = slot - px * NP, ry = pp / WD, rx = pp % WD;
        d_vo[j] = (unsigned)(((2 * ry) * W + 2 * rx + px) * K * 4 + k4 * 16);
        d_rc[j] = (unsigned)(ry << 8 | rx);
    }

    f32x4 rx_[NX], rd[ND];
    // patch cursor of the NEXT fetch, advanced by one patch per call (stages are consecutive patches): three scalar adds and
    // compares per stage instead of four 64-bit divisions on the vector ALU (which also made the scalar offsets of the buffer
    // loads look divergent, wrapping every load in a waterfall loop)
    int nx0 = (int)(g0 % cpr) * WD, ny0 = (int)((g0 / cpr) % rpi) * R, nb = (int)((g0 / cpr) / rpi);
    auto gload = [&]() {
        const int x0 = nx0, y0 = ny0;
        const unsigned b = (unsigned)nb;
        nx0 += WD;
        if (nx0 >= cpr * WD) {
            nx0 = 0;
            ny0 += R;
            if (ny0 >= rpi * R) {
                ny0 = 0;
                ++nb;
            }
        }
        // halo row hr <-> low-res row y0 + py - 1 + hr, halo col hx <-> x0 - 1 + hx
        const int ytop = y0 + py;                                            // low-res row of hr = 1
        const unsigned rlo = (ytop == 0) ? 1u : 0u, rn = (unsigned)((Hl - ytop < R) ? (Hl - ytop) : R) - rlo;
        const unsigned clo = (x0 == 0) ? 1u : 0u, cn = (unsigned)((Wl - x0 < WD + 1) ? (Wl - x0) : (WD + 1)) - clo;
        const unsigned so_x = (unsigned)((((b * Hl + ytop) * Wl + x0) * C + c0) * 4);
        const unsigned so_d = (unsigned)((((b * H + 2 * y0 + py) * W + 2 * x0) * K + k0) * 4);
#pragma unroll
        for (int j = 0; j < NX; ++j) {
            const bool ok = ((x_rc[j] >> 8) - rlo <= rn) && ((x_rc[j] & 255u) - clo <= cn);
            rx_[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(x_rs, ok ? x_vo[j] : 0xFFFFFFFFu, so_x, 0));
        }
        const unsigned rmax = (unsigned)(Hl - y0), cmax = (unsigned)(Wl - x0);
#pragma unroll
        for (int j = 0; j < ND; ++j) {
            const bool ok = ((d_rc[j] >> 8) < rmax) && ((d_rc[j] & 255u) < cmax);
            rd[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(d_rs, ok ? d_vo[j] : 0xFFFFFFFFu, so_d, 0));
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int j = 0; j < NX; ++j) {
            const int i = tid + 256 * j;
            const int pos = i >> 4, c4 = i & 15;
            if (pos < NH) {
                u32x2_t hi, lo;
                if constexpr (egz_drop_alo<T>::value) f16_rne4(rx_[j] * x_scale, hi);
                else W16<T>::split4s(rx_[j], x_scale, hi, lo);
                unsigned short* d = Xs + buf * XB + (c4 >> 3) * XH + pos * 32 + (c4 & 7) * 4;
                *reinterpret_cast<u32x2_t*>(d) = hi;
                if constexpr (!egz_drop_alo<T>::value) *reinterpret_cast<u32x2_t*>(d + 2 * XH) = lo;
            }
        }
#pragma unroll
        for (int j = 0; j < ND; ++j) {
            const int i = tid + 256 * j;
            const int slot = i >> 4, k4 = i & 15;
            u32x2_t hi, lo;
            W16<T>::split4s(rd[j], d_scale, hi, lo);
            unsigned short* d = Ds + buf * DB + (k4 >> 3) * DH + slot * 32 + (k4 & 7) * 4;
            *reinterpret_cast<u32x2_t*>(d) = hi;
            *reinterpret_cast<u32x2_t*>(d + 2 * DH) = lo;
        }
    };

    const int u = lane & 15, hh = lane >> 5;
    const int chan = 16 * ((lane >> 4) & 1) + 4 * (u & 3);
    const int lp = 8 * hh + (u >> 2);
    const int xlane = (WD >= 16) ? lp : (hh * HPW + (u >> 2));
    const EGZ_LDS unsigned short* Xl = (const EGZ_LDS unsigned short*)(Xs) + wc * XH + xlane * 32 + chan;
    const EGZ_LDS unsigned short* Dl = (const EGZ_LDS unsigned short*)(Ds) + wk * DH + lp * 32 + chan;
    auto xoff = [&](int ks, int q, int a, int o) -> int {        // o = px + b: halo column offset
        const int base = (WD == 32) ? (16 * ks + 4 * q) : (WD == 16) ? (ks * HPW + 4 * q) : (2 * ks * HPW + 4 * q);
        return (base + a * HPW + o) * 32;
    };
    f32x16 acc[8];                                               // [px][a][b]
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    if (g0 < g1) {
        gload();
        lstore(0);
    }
    __syncthreads();
    for (long g = g0; g < g1; ++g) {
        const int buf = (int)((g - g0) & 1);
        if (g + 1 < g1) gload();
        const EGZ_LDS unsigned short* Xb = Xl + buf * XB;
        const EGZ_LDS unsigned short* Db = Dl + buf * DB;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            typename W16<T>::vec8 dh[2], dl[2];
#pragma unroll
            for (int px = 0; px < 2; ++px) {
                dh[px] = W16<T>::frag(Db + (px * NP + 16 * ks) * 32, Db + (px * NP + 16 * ks + 4) * 32);
                dl[px] = W16<T>::frag(Db + 2 * DH + (px * NP + 16 * ks) * 32, Db + 2 * DH + (px * NP + 16 * ks + 4) * 32);
            }
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                typename W16<T>::vec8 xh[3], xl[3];
#pragma unroll
                for (int o = 0; o < 3; ++o) {
                    xh[o] = W16<T>::frag(Xb + xoff(ks, 0, a, o), Xb + xoff(ks, 1, a, o));
                    xl[o] = W16<T>::frag(Xb + 2 * XH + xoff(ks, 0, a, o), Xb + 2 * XH + xoff(ks, 1, a, o));
                }
#pragma unroll
                for (int term = 0; term < 3; ++term)
#pragma unroll
                    for (int px = 0; px < 2; ++px)
#pragma unroll
                        for (int b = 0; b < 2; ++b)
                            if (!(egz_drop_alo<T>::value && term == 0)) acc[px * 4 + a * 2 + b] = W16<T>::mfma(term == 0 ? xl[px + b] : xh[px + b], term == 1 ? dl[px] : dh[px], acc[px * 4 + a * 2 + b]);
            }
        }
        if (g + 1 < g1) lstore(buf ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const int px = t >> 2, ab = t & 3;
        float* out = part + ((long)split * 16 + (py * 2 + px) * 4 + ab) * C * K;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int c = c0 + wc * 32 + egz_acc_row(r, lane);
            const int k = k0 + wk * 32 + l31;
            out[(long)c * K + k] = acc[t][r] * d_inv * x_inv;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Weight gradient of [nearest x2 upsample -> conv3x3] in phase form (the transpose of the UPS_PHASE forward of
// conv3x3_igemm.hip): for each output phase (py,px) the 3x3 taps collapse to 2x2 taps on the LOW-res input, so
//   dWeff[py][px][a][b][c][k] = sum_{b,y,x} X[y+a+py-1][x+b+px-1][c] * dY[2y+py][2x+px][k]
// (16 accumulations over a quarter of the pixels = 4/9 of the MACs of the folded form) and
//   dW[r][s] = sum over the (py,a) with r in R(py,a) and the (px,b) with s in R(px,b) of dWeff[py][px][a][b].
// One block = one 64(c) x 64(k) tile of ONE phase (blockIdx.z), 4 accumulators per wave; a stage is a segment of L
// low-res pixels: 2 x (L+1) input halo + L strided dY rows in LDS.
template <int L>
__global__ __launch_bounds__(256, 2) void conv3x3_wgrad_ups_kernel(
    const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ part, int B, int H, int W,
    int C, int K, int segs_per_split) {
    constexpr int HP = L + 1;
    constexpr int NX = (2 * HP * 16 + 255) / 256;
    constexpr int ND = (L * 16 + 255) / 256;
    __shared__ __attribute__((aligned(16))) float Xh[2 * 2 * HP * 64];
    __shared__ __attribute__((aligned(16))) float Ds[2 * L * 64];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wc = wave >> 1, wk = wave & 1, hl = lane >> 5, l31 = lane & 31;
    const int tk = K / 64;
    const int c0 = (blockIdx.x / tk) * 64, k0 = (blockIdx.x % tk) * 64;
    const int phase = blockIdx.z, py = phase >> 1, px = phase & 1;
    const int Hl = H >> 1, Wl = W >> 1;               // low-res dims (H, W are the hi-res conv output dims)
    const int spr = Wl / L;
    const long nseg = (long)B * Hl * spr;
    const long g0 = (long)blockIdx.y * segs_per_split;
    const long g1 = (g0 + segs_per_split < nseg) ? (g0 + segs_per_split) : nseg;

    f32x4 rx[NX], rd[ND];
    auto gload = [&](long g) {
        const int xs = (int)(g % spr) * L;
        const long t = g / spr;
        const int yy = (int)(t % Hl);
        const long b = t / Hl;
#pragma unroll
        for (int j = 0; j < NX; ++j) {
            const int i = tid + 256 * j;
            const int pos = i >> 4, c4 = i & 15;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (pos < 2 * HP) {
                const int a = pos / HP, pxi = pos - a * HP;
                const int iy = yy + a + py - 1, ix = xs + pxi + px - 1;
                if ((unsigned)iy < (unsigned)Hl && (unsigned)ix < (unsigned)Wl)
                    v = *reinterpret_cast<const f32x4*>(x + ((b * Hl + iy) * (long)Wl + ix) * C + c0 + c4 * 4);
            }
            rx[j] = v;
        }
        const long m0 = (b * H + 2 * yy + py) * (long)W + 2 * xs + px;      // hi-res pixel of segment pixel 0
#pragma unroll
        for (int j = 0; j < ND; ++j) {
            const int i = tid + 256 * j;
            const int pp = i >> 4, k4 = i & 15;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (pp < L) v = *reinterpret_cast<const f32x4*>(dy + (m0 + 2 * pp) * K + k0 + k4 * 4);
            rd[j] = v;
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int j = 0; j < NX; ++j) {
            const int i = tid + 256 * j;
            if (i < 2 * HP * 16) *reinterpret_cast<f32x4*>(Xh + buf * 2 * HP * 64 + i * 4) = rx[j];
        }
#pragma unroll
        for (int j = 0; j < ND; ++j) {
            const int i = tid + 256 * j;
            if (i < L * 16) *reinterpret_cast<f32x4*>(Ds + buf * L * 64 + i * 4) = rd[j];
        }
    };

    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    if (g0 < g1) {
        gload(g0);
        lstore(0);
    }
    __syncthreads();
    for (long g = g0; g < g1; ++g) {
        const int buf = (int)((g - g0) & 1);
        if (g + 1 < g1) gload(g + 1);
        const float* Xb = Xh + buf * 2 * HP * 64 + hl * 64 + wc * 32 + l31;
        const float* Db = Ds + buf * L * 64 + hl * 64 + wk * 32 + l31;
#pragma unroll
        for (int t = 0; t < L / 2; ++t) {
            const float bv = Db[(2 * t) * 64];
#pragma unroll
            for (int tap = 0; tap < 4; ++tap) {
                const float av = Xb[((tap >> 1) * HP + 2 * t + (tap & 1)) * 64];
                acc[tap] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[tap], 0, 0, 0);
            }
        }
        if (g + 1 < g1) lstore(buf ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int tap = 0; tap < 4; ++tap) {
        float* out = part + ((long)blockIdx.y * 16 + phase * 4 + tap) * C * K;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int c = c0 + wc * 32 + egz_acc_row(r, lane);
            const int k = k0 + wk * 32 + l31;
            out[(long)c * K + k] = acc[tap][r];
        }
    }
}

// dw[k][c][r][s] = sum_rows sum_{(py,a) : r in R(py,a)} sum_{(px,b) : s in R(px,b)} part[row][py*2+px][a*2+b][c][k]
//   R(0,0)={0}, R(0,1)={1,2}, R(1,0)={0,1}, R(1,1)={2}  =>  r=0: (0,0),(1,0);  r=1: (0,1),(1,0);  r=2: (0,1),(1,1)
__global__ __launch_bounds__(256) void wgrad_reduce_ups_kernel(const float* __restrict__ part, float* __restrict__ dw,
                                                               int C, int K, int rows) {
    const long ck = (long)C * K, n = 16 * ck;
    const int PA[3][2][2] = {{{0, 0}, {1, 0}}, {{0, 1}, {1, 0}}, {{0, 1}, {1, 1}}};   // [r][which] -> (p, a)
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < 9 * ck; i += (long)gridDim.x * blockDim.x) {
        const int k = (int)(i % K);
        const long t = i / K;
        const int c = (int)(t % C), tap = (int)(t / C);
        const int r = tap / 3, q = tap % 3;
        float s = 0.f;
        for (int row = 0; row < rows; ++row)
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int v = 0; v < 2; ++v) {
                    const int py = PA[r][u][0], a = PA[r][u][1], px = PA[q][v][0], b = PA[q][v][1];
                    s += part[(long)row * n + ((py * 2 + px) * 4 + a * 2 + b) * ck + (long)c * K + k];
                }
        dw[((long)k * C + c) * 9 + tap] = s;
    }
}

// Narrow layers (late_fusion: C = 32, K = 32 / 8): one 32(c) x 32(k) MFMA tile per block; the four waves split
// each 32-pixel stage four ways (intra-block split-K) and are summed through LDS at the end.  Channel counts
// that are not multiples of 32 are masked on load / store.
template <bool UPS>
__global__ __launch_bounds__(256, 2) void conv3x3_wgrad32_kernel(
    const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ part, int B, int H, int W,
    int C, int K, long pix_per_split) {
    __shared__ __attribute__((aligned(16))) float Xs[2 * PK * 32];
    __shared__ __attribute__((aligned(16))) float Ds[2 * PK * 32];
    __shared__ float red[4 * 32 * 33];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hl = lane >> 5, l31 = lane & 31;
    const int tk = (K + 31) / 32;
    const int tap = blockIdx.x % 9;
    const int tile = blockIdx.x / 9;
    const int c0 = (tile / tk) * 32, k0 = (tile % tk) * 32;
    const int dyy = tap / 3 - 1, dxx = tap % 3 - 1;
    const long HW = (long)H * W, M = (long)B * HW;
    const long mbeg = (long)blockIdx.y * pix_per_split;
    const long mend = (mbeg + pix_per_split < M) ? (mbeg + pix_per_split) : M;
    const int Hs = UPS ? (H >> 1) : H, Ws = UPS ? (W >> 1) : W;
    const int ch4 = tid & 7, p0 = tid >> 3;
    const bool cok = c0 + ch4 * 4 < C, kok = k0 + ch4 * 4 < K;

    f32x4 rx, rd;
    auto gload = [&](long mb) {
        const long m = mb + p0;
        rx = f32x4{0.f, 0.f, 0.f, 0.f};
        rd = f32x4{0.f, 0.f, 0.f, 0.f};
        if (m < mend) {
            if (kok) rd = *reinterpret_cast<const f32x4*>(dy + m * K + k0 + ch4 * 4);
            const long b = m / HW;
            const int rem = (int)(m - b * HW);
            const int yy = rem / W, xx = rem - yy * W;
            const int iy = yy + dyy, ix = xx + dxx;
            if (cok && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) {
                const int sy = UPS ? (iy >> 1) : iy, sx = UPS ? (ix >> 1) : ix;
                rx = *reinterpret_cast<const f32x4*>(x + ((b * Hs + sy) * (long)Ws + sx) * C + c0 + ch4 * 4);
            }
        }
    };
    auto lstore = [&](int buf) {
        *reinterpret_cast<f32x4*>(Xs + buf * PK * 32 + p0 * 32 + ch4 * 4) = rx;
        *reinterpret_cast<f32x4*>(Ds + buf * PK * 32 + p0 * 32 + ch4 * 4) = rd;
    };

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int nst = (int)((mend - mbeg + PK - 1) / PK);
    if (nst > 0) {
        gload(mbeg);
        lstore(0);
    }
    __syncthreads();
    for (int s = 0; s < nst; ++s) {
        const int buf = s & 1;
        if (s + 1 < nst) gload(mbeg + (long)(s + 1) * PK);
        const float* Ab = Xs + buf * PK * 32 + (8 * wave + hl) * 32 + l31;
        const float* Bb = Ds + buf * PK * 32 + (8 * wave + hl) * 32 + l31;
#pragma unroll
        for (int t = 0; t < 4; ++t)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Ab[(2 * t) * 32], Bb[(2 * t) * 32], acc, 0, 0, 0);
        if (s + 1 < nst) lstore(buf ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) red[(wave * 32 + egz_acc_row(r, lane)) * 33 + l31] = acc[r];
    __syncthreads();
    float* out = part + ((long)blockIdx.y * 9 + tap) * C * K;
    for (int i = tid; i < 32 * 32; i += 256) {
        const int c = i >> 5, k = i & 31;
        if (c0 + c < C && k0 + k < K)
            out[(long)(c0 + c) * K + k0 + k] = (red[(0 * 32 + c) * 33 + k] + red[(1 * 32 + c) * 33 + k]) +
                                               (red[(2 * 32 + c) * 33 + k] + red[(3 * 32 + c) * 33 + k]);
    }
}

// Split-K reduction, fixed order (deterministic, no atomics).  Many splits (up to 1024 for the one-tile 64x64
// layers) are first folded 32 at a time by a 2-D grid (every thread sums 32 rows with 4 independent accumulators);
// the final pass sums the <= 32 remaining rows and transposes into the reference layout:
//   dw[(k*C + c)*9 + tap] = sum_s part[s][tap][c][k]
constexpr int RG = 32;
__global__ __launch_bounds__(256) void wgrad_fold_kernel(const float* __restrict__ part, float* __restrict__ part2,
                                                         long n, int S) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int s0 = blockIdx.y * RG, s1 = (s0 + RG < S) ? s0 + RG : S;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int sp = s0;
    for (; sp + 3 < s1; sp += 4) {
        a0 += part[(long)sp * n + i];
        a1 += part[(long)(sp + 1) * n + i];
        a2 += part[(long)(sp + 2) * n + i];
        a3 += part[(long)(sp + 3) * n + i];
    }
    for (; sp < s1; ++sp) a0 += part[(long)sp * n + i];
    part2[(long)blockIdx.y * n + i] = (a0 + a1) + (a2 + a3);
}
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw,
                                                           int C, int K, int S) {
    const long n = (long)9 * C * K;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        // i enumerates (tap, c, k) with k fastest -> coalesced reads of the partials
        const int k = (int)(i % K);
        const long t = i / K;
        const int c = (int)(t % C), tap = (int)(t / C);
        float s = 0.f;
        for (int sp = 0; sp < S; ++sp) s += part[(long)sp * n + i];
        dw[((long)k * C + c) * 9 + tap] = s;
    }
}
// Tiled form of the final pass for C % 8 == 0, K % 32 == 0: a block owns an 8(c) x 32(k) x 9-tap tile, reads the partials
// coalesced along k (nine independent accumulators per thread, same row order as above -> identical sums), transposes
// through LDS and writes dw[k][c0..c0+7][0..8] as 288-byte contiguous runs instead of 4-byte stores 36 bytes apart.
__global__ __launch_bounds__(256) void wgrad_reduce_tile_kernel(const float* __restrict__ part, float* __restrict__ dw,
                                                                int C, int K, int S) {
    __shared__ float tile[32 * 73];                    // [k][c * 9 + tap], row padded 72 -> 73
    const long ck = (long)C * K, n = 9 * ck;
    const int c0 = blockIdx.x * 8, k0 = blockIdx.y * 32;
    const int kl = threadIdx.x & 31, cs = threadIdx.x >> 5;
    float acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = 0.f;
    const float* p = part + (long)(c0 + cs) * K + k0 + kl;
    // four rows' loads in flight per step, added in row order (the sums are those of the plain loop): the 32 x 32 filters of the
    // late-fusion head have 4 blocks here and a row-by-row loop cost them one L2 round trip per row (90 us for 16 rows)
    int sp = 0;
    for (; sp + 3 < S; sp += 4, p += 4 * n) {
        float v[4][9];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int t = 0; t < 9; ++t) v[r][t] = p[r * n + t * ck];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int t = 0; t < 9; ++t) acc[t] += v[r][t];
    }
    for (; sp < S; ++sp, p += n)
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[t] += p[t * ck];
#pragma unroll
    for (int t = 0; t < 9; ++t) tile[kl * 73 + cs * 9 + t] = acc[t];
    __syncthreads();
    for (int e = threadIdx.x; e < 32 * 72; e += 256) {
        const int k = e / 72, r = e - k * 72;
        dw[((long)(k0 + k) * C + c0) * 9 + r] = tile[k * 73 + r];
    }
}
// workspace layout: [S][n] partial tiles, then [ceil(S/RG)][n] folded partials when S > RG
size_t wgrad_ws_floats(int S, long n) { return (size_t)S * n + (S > RG ? (size_t)((S + RG - 1) / RG) * n : 0); }
int wgrad_reduce(float* part, float* dw, int C, int K, int S, hipStream_t st) {
    const long n = (long)9 * C * K;
    const float* src = part;
    int rows = S;
    if (S > RG) {
        float* part2 = part + (size_t)S * n;
        rows = (S + RG - 1) / RG;
        hipLaunchKernelGGL(wgrad_fold_kernel, dim3(egz_cdiv(n, 256), rows), dim3(256), 0, st, part, part2, n, S);
        EGZ_CHECK_LAUNCH("egz_conv3x3_wgrad(fold)");
        src = part2;
    }
    if (C % 8 == 0 && K % 32 == 0) {
        hipLaunchKernelGGL(wgrad_reduce_tile_kernel, dim3(C / 8, K / 32), dim3(256), 0, st, src, dw, C, K, rows);
    } else {
        const int g = egz_cdiv(n, 256) > 4096 ? 4096 : egz_cdiv(n, 256);
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(g), dim3(256), 0, st, src, dw, C, K, rows);
    }
    EGZ_CHECK_LAUNCH("egz_conv3x3_wgrad(reduce)");
    return 0;
}

int pick_splits(long M, int C, int K, int BT) {
    const long tiles = (long)((C + BT - 1) / BT) * ((K + BT - 1) / BT) * 9;
    long s = (768 + tiles - 1) / tiles;           // aim at ~3 blocks per CU
    const long smax = (M + 1023) / 1024;          // at least ~1k pixels per split
    if (s > smax) s = smax;
    if (s < 1) s = 1;
    return (int)s;
}

// segment length of the upsample-phase kernel on the LOW-res row (0 = not applicable)
int pick_seg_ups(int W, int C, int K, int flags) {
    if ((flags & 0x1800) || C % 64 != 0 || K % 64 != 0 || W % 2) return 0;   // 0x1000: force the folded 9-tap form
    const int Wl = W / 2;
    if (Wl % 32 == 0) return 32;
    if (Wl % 28 == 0) return 28;
    if (Wl % 14 == 0) return 14;
    return 0;
}
int pick_splits_ups(long nseg, int C, int K) {
    const long tiles = (long)(C / 64) * (K / 64) * 4;
    long s = (1024 + tiles - 1) / tiles;
    const long smax = (nseg + 7) / 8;
    if (s > smax) s = smax;
    if (s < 1) s = 1;
    return (int)s;
}

// segment length of the 9-tap fused kernel (0 = not applicable -> per-tap kernel)
int pick_seg(int W, int C, int K, int flags) {
    if (flags & 0x800) return 0;                      // force the per-tap kernel (A/B benchmarking)
    if (C % 64 != 0 || K % 64 != 0) return 0;
    if (W % 32 == 0) return 32;
    if (W % 28 == 0) return 28;
    if (W % 14 == 0) return 14;
    return 0;
}
int pick_splits9(long nseg, int C, int K, int target = 1024) {
    const long tiles = (long)((C + 63) / 64) * ((K + 63) / 64);
    long s = (target + tiles - 1) / tiles;            // 1024: ~4 blocks per CU (two rounds of 2 resident)
    const long smax = (nseg + 7) / 8;                 // at least 8 segments per split
    if (s > smax) s = smax;
    if (s < 1) s = 1;
    return (int)s;
}

// split-half 9-tap kernel (flags 0x2000): patch width (0 = not applicable); rows narrower than the patch are masked
int pick_patch_x3(int W, int C, int K, int flags) {
    if (!(flags & 0x2000) || (flags & 0x800) || C % 32 != 0) return 0;
    if (K % 64 != 0 && ((flags & 1) || K > 64 || K % 4 != 0)) return 0;   // K < 64 (late-fusion widths): plain form, masked k-tile
    if (C % 64 != 0 && ((flags & 1) || C != 32)) return 0;       // C = 32 (padded first conv): plain 9-tap form, half c-tile
    // Large images: the 4 x 8 patch.  A stage stages the patch's halo, (R + 2) x (WD + 2) pixels for 32 outputs: 3.2x for 1 x 32,
    // 2.25x for 2 x 16, 1.9x for 4 x 8 -- and every staged element costs a fetch from L2 and an f16 split.  On the 224- and
    // 112-wide layers the squarer patch is 5-10 % faster (64 -> 64 @ 224: 474 -> 441 us, @ 224 decoder: 397 -> 358, 128 -> 128
    // @ 112: 368 -> 351; profiles/r03_ab_notes.txt); at 56 and below it makes no difference.
    if (W >= 112 && W % 8 == 0) return 8;
    // ... and on rows of 17-31 pixels (the 28 x 28 layers): four 8-wide patches with the last one partly masked waste the same
    // 14 % as one masked 32-wide run, with the smaller halo (512 -> 512 @ 28: 388 -> 353 us, 256 -> 512: 210 -> 190).  At 14 x 14
    // the 2 x 16 patch stays (4 x 8: 116 -> 122 us).
    if (W > 16 && W < 32) return 8;
    if (W % 32 == 0 || (W > 16 && W < 32)) return 32;
    if (W % 16 == 0 || (W > 8 && W < 16)) return 16;
    if (W % 8 == 0) return 8;
    return 0;
}
// narrow split-half kernel (C, K <= 32, plain conv): run width of its 64-pixel patch (0 = not applicable)
int pick_narrow_x3(int W, int C, int K, int flags) {
    if (!(flags & 0x2000) || (flags & 0x801) || C > 32 || K > 32) return 0;
    // (a 4 x 16 patch where a 2 x 32 one fits measured the same on the late-fusion step: 1.30 vs 1.31 ms)
    return (W % 32 == 0) ? 32 : (W % 16 == 0) ? 16 : 0;
}
long npatch_x3n(int B, int H, int W, int WD) { return (long)B * ((H + 64 / WD - 1) / (64 / WD)) * (W / WD); }
// blocks per launch of the split-half kernel: one round of 2 resident blocks per CU.  (Round 5 measured fewer: 448 / 384 / 320 /
// 256 blocks -> step +0.23 / +0.33 / +0.78 / +0.55 ms, the weight gradients alone 9.65 -> 11.2 ms at 384: a CU with ONE
// resident block loses its latency hiding.  What does NOT cost anything is fewer CUs at two blocks each -- tools/micro/
// cu_share_probe.py -- but the launch cannot choose that; profiles/r05_ab_notes.txt.)
constexpr int X3_BLOCKS = 512;
long npatch_x3(int B, int H, int W, int WD) { return (long)B * ((H + 32 / WD - 1) / (32 / WD)) * ((W + WD - 1) / WD); }

int pick_bt(int C, int K, int flags) {
    if (C % 64 != 0 || K % 64 != 0) return 32;
    if (flags & 0x100) return 64;
    return (C % 128 == 0 && K % 128 == 0) ? 128 : 64;
}

}  // namespace

EGZ_API size_t egz_conv3x3_wgrad_ws_bytes(int B, int H, int W, int C, int K, int flags) {
    if (flags & 0x2000) {   // same size rule as egz_conv3x3_wgrad
        const bool ups = flags & 1;
        const unsigned long long xb = 4ull * B * (ups ? H / 2 : H) * (ups ? W / 2 : W) * C + 4ull * (W + 1) * C;
        const unsigned long long db = 4ull * B * H * W * K;
        if (xb >= (1ull << 32) || db >= (1ull << 32)) flags &= ~0x2000;
    }
    const int L = pick_seg(W, C, K, flags);
    const long n = (long)9 * C * K;
    if ((flags & 1) && !(flags & 0x1000) && H % 2 == 0 && W % 2 == 0) {
        if (const int WD = pick_patch_x3(W / 2, C, K, flags))       // phase form on the low-res grid, 16 partial tiles
            return wgrad_ws_floats(pick_splits9(npatch_x3(B, H / 2, W / 2, WD), C, K, X3_BLOCKS / 2), 16L * C * K) * sizeof(float);
    }
    if (const int WD = pick_narrow_x3(W, C, K, flags))
        return wgrad_ws_floats(pick_splits9(npatch_x3n(B, H, W, WD), C, K, X3_BLOCKS), n) * sizeof(float);
    if (const int WD = pick_patch_x3(W, C, K, flags))
        return wgrad_ws_floats(pick_splits9(npatch_x3(B, H, W, WD), C, K, X3_BLOCKS), n) * sizeof(float);
    if (flags & 1) {
        const int Lu = pick_seg_ups(W, C, K, flags);
        if (Lu) return wgrad_ws_floats(pick_splits_ups((long)B * (H / 2) * (W / 2 / Lu), C, K), 16L * C * K) * sizeof(float);
    }
    if (L) return wgrad_ws_floats(pick_splits9((long)B * H * (W / L), C, K), n) * sizeof(float);
    const int bt = pick_bt(C, K, flags);
    const int S = pick_splits((long)B * H * W, C, K, bt);
    return wgrad_ws_floats(S, n) * sizeof(float);
}

// flags: bit0 = the conv input was the nearest-x2 upsampling of x ([B][H/2][W/2][C]); 0x100 forces 64 tiles;
//        0x800 forces the per-tap kernel, 0x1000 the folded 9-tap form of an upsampled conv (A/B benchmarking);
//        0x2000 = split-half arithmetic on the 16-bit MFMA path (C, K multiples of 64; else exact f32): bf16 x3 (16 bits,
//        no scaling) when dy_absmax is NULL, f16 x3 (22 bits) with dy scaled by absmax_scale(*dy_absmax) when it is given;
//        x_absmax (optional, f16 x3 only): max |x| -- x is scaled the same way (activations outside [2^-3, 6e4] otherwise
//        leave the f16 pair's 22-bit domain).
//        0x20000 = with f16 split halves: TWO products per MAC instead of three -- x's lo half is not multiplied (11 significant
//        bits of x, rounded to nearest; 22 of dy; per-element error ~2^-12 instead of 2^-22) -- on conv3x3_wgrad9_x3_kernel / conv3x3_wgrad_ups_x3_kernel.
// x: conv input (NHWC), dy: gradient of the conv output ([B][H][W][K]), dw: (K, C, 3, 3) like the reference.
// 1 when a plain split-half weight gradient of this geometry runs on the narrow kernel (C, K <= 32, W % 16 == 0, 32-bit
// buffer offsets) -- the only one that takes a deferred-BatchNorm activation operand (x_bn)
// 1 when egz_conv3x3_wgrad(flags 0x2000 [| 0x8000]) of a plain conv runs on conv3x3_wgrad9_x3_kernel, i.e. can take a pre-split
// x operand: C, K multiples of 64, a patch geometry for this width, operands below 4 GiB.
EGZ_API int egz_conv3x3_wgrad_presplit_ok(int B, int H, int W, int C, int K) {
    if (B <= 0 || H <= 0 || W <= 0 || C % 64 != 0 || K % 64 != 0) return 0;
    const unsigned long long xb = 4ull * B * H * W * C + 4ull * (W + 1) * C, db = 4ull * B * H * W * K;
    if (xb >= (1ull << 32) || db >= (1ull << 32)) return 0;
    return pick_patch_x3(W, C, K, 0x2000) != 0;
}

EGZ_API int egz_conv3x3_wgrad_narrow_ok(int B, int H, int W, int C, int K) {
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || K <= 0 || C % 4 || K % 4) return 0;
    const unsigned long long xb = 4ull * B * H * W * C + 4ull * (W + 1) * C, db = 4ull * B * H * W * K;
    if (xb >= (1ull << 32) || db >= (1ull << 32)) return 0;
    return pick_narrow_x3(W, C, K, 0x2000) ? 1 : 0;
}

// x_bn (optional, narrow geometry only): x is the PRE-BatchNorm output of the block below and x_bn that BatchNorm's 4 x C
// coefficient rows (mean, 1/std, scale, shift); relu(x * scale + shift) is applied while x is staged, x_absmax = its max.
EGZ_API int egz_conv3x3_wgrad(const float* x, const float* dy, float* dw, int B, int H, int W, int C, int K,
                              int flags, void* workspace, size_t ws_bytes, const unsigned int* dy_absmax,
                              const unsigned int* x_absmax, const float* x_bn, hipStream_t st) {
    EGZ_CHECK_ARG(x && dy && dw && workspace, "egz_conv3x3_wgrad: null pointer");
    EGZ_CHECK_ARG(!x_bn || ((flags & 0x2000) && egz_conv3x3_wgrad_narrow_ok(B, H, W, C, K) && pick_narrow_x3(W, C, K, flags)),
                  "egz_conv3x3_wgrad: a deferred-BatchNorm operand (x_bn) exists on the narrow split-half kernel only "
                  "(C, K <= 32, W %% 16 == 0, plain conv)");
    EGZ_CHECK_ARG(C % 4 == 0 && K % 4 == 0 && C > 0 && K > 0, "egz_conv3x3_wgrad: C=%d K=%d must be multiples of 4", C, K);
    const bool ups = flags & 1;
    EGZ_CHECK_ARG(!ups || (H % 2 == 0 && W % 2 == 0), "egz_conv3x3_wgrad: upsampled output must be even");
    const long M = (long)B * H * W;
    if (flags & 0x2000) {   // split-half kernels fetch through 32-bit buffer offsets
        const unsigned long long xb = 4ull * B * (ups ? H / 2 : H) * (ups ? W / 2 : W) * C + 4ull * (W + 1) * C;
        const unsigned long long db = 4ull * B * H * W * K;
        if (xb >= (1ull << 32) || db >= (1ull << 32)) flags &= ~0x2000;      // too large: exact-f32 kernels (64-bit addressing)
    }
    // flags 0x8000: x holds pre-split activations (see conv3x3_wgrad9_x3_kernel) -- only where that kernel runs
    const bool xpre = (flags & 0x8000) != 0, dpre = (flags & 0x10000) != 0;
    // 0x20000: two products per MAC on the split-half f16 kernels of the wide layers (x_hi dy_hi + x_hi dy_lo: x enters with its hi
    // half only, rounded to nearest; egz_f16p2 in egz_common.h).  Also on the narrow (late-fusion) kernels; ignored by bf16 and exact-f32 launches.
    const bool p2 = (flags & 0x20000) != 0 && dy_absmax;
    EGZ_CHECK_ARG(!(xpre || dpre) || (egz_conv3x3_wgrad_presplit_ok(B, H, W, C, K) && (flags & 0x2000) && !ups && dy_absmax && x_absmax && !x_bn),
                  "egz_conv3x3_wgrad: a pre-split x / dy operand (flags 0x8000 / 0x10000) needs the split-half 9-tap kernel's geometry "
                  "(egz_conv3x3_wgrad_presplit_ok), f16 x3 (dy_absmax, x_absmax) and a plain conv");
    float* part = static_cast<float*>(workspace);
    const long nred = (long)9 * C * K;
    if (ups && !(flags & 0x1000)) {
        if (const int WD = pick_patch_x3(W / 2, C, K, flags)) {     // split-half phase form of an upsampled conv
            const long n16 = 16L * C * K;
            const long np = npatch_x3(B, H / 2, W / 2, WD);
            const int S = pick_splits9(np, C, K, X3_BLOCKS / 2);    // two row-phase blocks per (tile, split)
            EGZ_CHECK_ARG(ws_bytes >= wgrad_ws_floats(S, n16) * sizeof(float), "egz_conv3x3_wgrad: workspace too small");
            const int pps = (int)((np + S - 1) / S);
            dim3 grid((C / 64) * (K / 64), S, 2);
#define EGZ_WUX(TT, RR, WW) hipLaunchKernelGGL((conv3x3_wgrad_ups_x3_kernel<TT, RR, WW>), grid, dim3(256), 0, st, x, dy, part, B, H, W, C, K, pps, dy_absmax, dy_absmax ? x_absmax : nullptr)
            if (dy_absmax && p2) { if (WD == 32) EGZ_WUX(egz_f16p2, 1, 32); else if (WD == 16) EGZ_WUX(egz_f16p2, 2, 16); else EGZ_WUX(egz_f16p2, 4, 8); }
            else if (dy_absmax) { if (WD == 32) EGZ_WUX(_Float16, 1, 32); else if (WD == 16) EGZ_WUX(_Float16, 2, 16); else EGZ_WUX(_Float16, 4, 8); }
            else           { if (WD == 32) EGZ_WUX(__bf16, 1, 32); else if (WD == 16) EGZ_WUX(__bf16, 2, 16); else EGZ_WUX(__bf16, 4, 8); }
#undef EGZ_WUX
            EGZ_CHECK_LAUNCH("egz_conv3x3_wgrad(ups-phase split)");
            const float* src = part;
            int rows = S;
            if (S > RG) {
                float* part2 = part + (size_t)S * n16;
                rows = (S + RG - 1) / RG;
                hipLaunchKernelGGL(wgrad_fold_kernel, dim3(egz_cdiv(n16, 256), rows), dim3(256), 0, st, part, part2, n16, S);
                EGZ_CHECK_LAUNCH("egz_conv3x3_wgrad(fold)");
                src = part2;
            }
            const int g = egz_cdiv(nred, 256) > 4096 ? 4096 : egz_cdiv(nred, 256);
            hipLaunchKernelGGL(wgrad_reduce_ups_kernel, dim3(g), dim3(256), 0, st, src, dw, C, K, rows);
            EGZ_CHECK_LAUNCH("egz_conv3x3_wgrad(reduce-ups)");
            return 0;
        }
    }
    if (const int WD = pick_narrow_x3(W, C, K, flags)) {  // late-fusion widths: one 32 x 32 tile, waves split the pixels
        const long np = npatch_x3n(B, H, W, WD);
        const int S = pick_splits9(np, C, K, X3_BLOCKS);
        EGZ_CHECK_ARG(ws_bytes >= wgrad_ws_floats(S, nred) * sizeof(float), "egz_conv3x3_wgrad: workspace too small");
        const int pps = (int)((np + S - 1) / S);
        dim3 grid(1, S);
#define EGZ_W9N(TT, RR, WW)                                                                                                   \
        do {                                                                                                                  \
            if (x_bn) hipLaunchKernelGGL((conv3x3_wgrad9_x3n_kernel<TT, RR, WW, true>), grid, dim3(256), 0, st, x, dy, part, B, H, W, C, K, pps, dy_absmax, dy_absmax ? x_absmax : nullptr, x_bn); \
            else      hipLaunchKernelGGL((conv3x3_wgrad9_x3n_kernel<TT, RR, WW, false>), grid, dim3(256), 0, st, x, dy, part, B, H, W, C, K, pps, dy_absmax, dy_absmax ? x_absmax : nullptr, x_bn); \
        } while (0)
#define EGZ_W9T(TT, RR, WW)                                                                                                   \
        do {                                                                                                                  \
            if (x_bn) hipLaunchKernelGGL((conv3x3_wgrad9_x3t_kernel<TT, RR, WW, true>), grid, dim3(256), 0, st, x, dy, part, B, H, W, C, K, pps, dy_absmax, dy_absmax ? x_absmax : nullptr, x_bn); \
            else      hipLaunchKernelGGL((conv3x3_wgrad9_x3t_kernel<TT, RR, WW, false>), grid, dim3(256), 0, st, x, dy, part, B, H, W, C, K, pps, dy_absmax, dy_absmax ? x_absmax : nullptr, x_bn); \
        } while (0)
        // few filters: (tap, k) pairs as GEMM columns (0x4000 keeps the nine-tile form: A/B runs).  The kernel fetches dy as one float4
        // per k-quad at a pixel stride of 4 K bytes: K is 4 or 8 here (K % 4 == 0 is an argument check above)
        if ((K == 4 || K == 8) && !(flags & 0x4000)) {
            if (dy_absmax && p2) { if (WD == 32) EGZ_W9T(egz_f16p2, 2, 32); else EGZ_W9T(egz_f16p2, 4, 16); }
            else if (dy_absmax) { if (WD == 32) EGZ_W9T(_Float16, 2, 32); else EGZ_W9T(_Float16, 4, 16); }
            else           { if (WD == 32) EGZ_W9T(__bf16, 2, 32); else EGZ_W9T(__bf16, 4, 16); }
        } else
        if (dy_absmax && p2) { if (WD == 32) EGZ_W9N(egz_f16p2, 2, 32); else EGZ_W9N(egz_f16p2, 4, 16); }
        else if (dy_absmax) { if (WD == 32) EGZ_W9N(_Float16, 2, 32); else EGZ_W9N(_Float16, 4, 16); }
        else           { if (WD == 32) EGZ_W9N(__bf16, 2, 32); else EGZ_W9N(__bf16, 4, 16); }
#undef EGZ_W9T
#undef EGZ_W9N
        EGZ_CHECK_LAUNCH("egz_conv3x3_wgrad(9-tap split, narrow)");
        return wgrad_reduce(part, dw, C, K, S, st);
    }
    if (const int WD = pick_patch_x3(W, C, K, flags)) {   // split-half (bf16 x3 / f16 x3) on the 16-bit MFMA path
        const long np = npatch_x3(B, H, W, WD);
        const int S = pick_splits9(np, C, K, X3_BLOCKS);
        EGZ_CHECK_ARG(ws_bytes >= wgrad_ws_floats(S, nred) * sizeof(float), "egz_conv3x3_wgrad: workspace too small");
        const int pps = (int)((np + S - 1) / S);
        dim3 grid(((C + 63) / 64) * ((K + 63) / 64), S);
#define EGZ_W9X(TT, U, RR, WW) hipLaunchKernelGGL((conv3x3_wgrad9_x3_kernel<TT, U, RR, WW>), grid, dim3(256), 0, st, x, dy, part, B, H, W, C, K, pps, dy_absmax, dy_absmax ? x_absmax : nullptr)
#define EGZ_W9QT(TT, RR, WW, XP, DP) hipLaunchKernelGGL((conv3x3_wgrad9_x3_kernel<TT, false, RR, WW, XP, DP>), grid, dim3(256), 0, st, x, dy, part, B, H, W, C, K, pps, dy_absmax, x_absmax)
#define EGZ_W9Q(RR, WW, XP, DP) do { if (p2) EGZ_W9QT(egz_f16p2, RR, WW, XP, DP); else EGZ_W9QT(_Float16, RR, WW, XP, DP); } while (0)
#define EGZ_W9P(RR, WW) do { if (xpre && dpre) EGZ_W9Q(RR, WW, true, true); else if (xpre) EGZ_W9Q(RR, WW, true, false); else EGZ_W9Q(RR, WW, false, true); } while (0)
#define EGZ_W9T(TT)                                                                                                    \
        if (ups) { if (WD == 32) EGZ_W9X(TT, true, 1, 32); else if (WD == 16) EGZ_W9X(TT, true, 2, 16); else EGZ_W9X(TT, true, 4, 8); } \
        else     { if (WD == 32) EGZ_W9X(TT, false, 1, 32); else if (WD == 16) EGZ_W9X(TT, false, 2, 16); else EGZ_W9X(TT, false, 4, 8); }
        if (xpre || dpre) {        // pre-split x (flags 0x8000) and / or dy (0x10000) operand: f16 x3, plain conv
            if (WD == 32) EGZ_W9P(1, 32); else if (WD == 16) EGZ_W9P(2, 16); else EGZ_W9P(4, 8);
        } else
        if (dy_absmax && p2) { EGZ_W9T(egz_f16p2) } else if (dy_absmax) { EGZ_W9T(_Float16) } else { EGZ_W9T(__bf16) }
#undef EGZ_W9P
#undef EGZ_W9Q
#undef EGZ_W9QT
#undef EGZ_W9T
#undef EGZ_W9X
        EGZ_CHECK_LAUNCH("egz_conv3x3_wgrad(9-tap split)");
        return wgrad_reduce(part, dw, C, K, S, st);
    }
    const int Lu = ups ? pick_seg_ups(W, C, K, flags) : 0;
    if (Lu) {      // phase-decomposed upsample: 16 (phase, tap) partial tiles per split, 4/9 of the MACs
        const long n16 = 16L * C * K;
        const long nseg = (long)B * (H / 2) * (W / 2 / Lu);
        const int S = pick_splits_ups(nseg, C, K);
        EGZ_CHECK_ARG(ws_bytes >= wgrad_ws_floats(S, n16) * sizeof(float), "egz_conv3x3_wgrad: workspace too small");
        const int sps = (int)((nseg + S - 1) / S);
        dim3 grid((C / 64) * (K / 64), S, 4);
        if (Lu == 32)      hipLaunchKernelGGL(conv3x3_wgrad_ups_kernel<32>, grid, dim3(256), 0, st, x, dy, part, B, H, W, C, K, sps);
        else if (Lu == 28) hipLaunchKernelGGL(conv3x3_wgrad_ups_kernel<28>, grid, dim3(256), 0, st, x, dy, part, B, H, W, C, K, sps);
        else               hipLaunchKernelGGL(conv3x3_wgrad_ups_kernel<14>, grid, dim3(256), 0, st, x, dy, part, B, H, W, C, K, sps);
        EGZ_CHECK_LAUNCH("egz_conv3x3_wgrad(ups-phase)");
        const float* src = part;
        int rows = S;
        if (S > RG) {
            float* part2 = part + (size_t)S * n16;
            rows = (S + RG - 1) / RG;
            hipLaunchKernelGGL(wgrad_fold_kernel, dim3(egz_cdiv(n16, 256), rows), dim3(256), 0, st, part, part2, n16, S);
            EGZ_CHECK_LAUNCH("egz_conv3x3_wgrad(fold)");
            src = part2;
        }
        const int g = egz_cdiv(nred, 256) > 4096 ? 4096 : egz_cdiv(nred, 256);
        hipLaunchKernelGGL(wgrad_reduce_ups_kernel, dim3(g), dim3(256), 0, st, src, dw, C, K, rows);
        EGZ_CHECK_LAUNCH("egz_conv3x3_wgrad(reduce-ups)");
        return 0;
    }
    const int L = pick_seg(W, C, K, flags);
    if (L) {
        const long nseg = (long)B * H * (W / L);
        const int S = pick_splits9(nseg, C, K);
        EGZ_CHECK_ARG(ws_bytes >= wgrad_ws_floats(S, nred) * sizeof(float), "egz_conv3x3_wgrad: workspace too small");
        const int sps = (int)((nseg + S - 1) / S);
        dim3 grid((C / 64) * (K / 64), S);
#define EGZ_W9(U, LL) hipLaunchKernelGGL((conv3x3_wgrad9_kernel<U, LL>), grid, dim3(256), 0, st, x, dy, part, B, H, W, C, K, sps)
        if (ups) { if (L == 32) EGZ_W9(true, 32); else if (L == 28) EGZ_W9(true, 28); else EGZ_W9(true, 14); }
        else     { if (L == 32) EGZ_W9(false, 32); else if (L == 28) EGZ_W9(false, 28); else EGZ_W9(false, 14); }
#undef EGZ_W9
        EGZ_CHECK_LAUNCH("egz_conv3x3_wgrad(9-tap)");
        return wgrad_reduce(part, dw, C, K, S, st);
    }
    const int bt = pick_bt(C, K, flags);
    const int S = pick_splits(M, C, K, bt);
    EGZ_CHECK_ARG(ws_bytes >= wgrad_ws_floats(S, nred) * sizeof(float), "egz_conv3x3_wgrad: workspace too small");
    long pps = (M + S - 1) / S;
    pps = (pps + PK - 1) / PK * PK;
    dim3 grid(((C + bt - 1) / bt) * ((K + bt - 1) / bt) * 9, S);
    if (bt == 32) {
        if (ups) hipLaunchKernelGGL(conv3x3_wgrad32_kernel<true>, grid, dim3(256), 0, st, x, dy, part, B, H, W, C, K, pps);
        else     hipLaunchKernelGGL(conv3x3_wgrad32_kernel<false>, grid, dim3(256), 0, st, x, dy, part, B, H, W, C, K, pps);
    } else if (bt == 128) {
        if (ups) hipLaunchKernelGGL((conv3x3_wgrad_kernel<128, true>), grid, dim3(256), 0, st, x, dy, part, B, H, W, C, K, pps);
        else     hipLaunchKernelGGL((conv3x3_wgrad_kernel<128, false>), grid, dim3(256), 0, st, x, dy, part, B, H, W, C, K, pps);
    } else {
        if (ups) hipLaunchKernelGGL((conv3x3_wgrad_kernel<64, true>), grid, dim3(256), 0, st, x, dy, part, B, H, W, C, K, pps);
        else     hipLaunchKernelGGL((conv3x3_wgrad_kernel<64, false>), grid, dim3(256), 0, st, x, dy, part, B, H, W, C, K, pps);
    }
    EGZ_CHECK_LAUNCH("egz_conv3x3_wgrad");
    return wgrad_reduce(part, dw, C, K, S, st);
}
