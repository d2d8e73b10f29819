// Error-compensated split-half variant of the 3x3 implicit GEMM (conv3x3_igemm.hip) for layers whose GEMM output
// channel count is a multiple of 64: every fp32 operand is represented as hi + lo in a 16-bit type and the product is
// accumulated in fp32 as  a*b ~= a_hi*b_hi + a_hi*b_lo + a_lo*b_hi  -- three v_mfma_f32_32x32x16_{f16,bf16}
// (2.5 PFLOP/s class, 16x the rate of the exact-f32 MFMA) per 16 k instead of eight v_mfma_f32_32x32x2_f32:
// 5.3x less matrix-pipe time at fp32-class accuracy ("error-compensated split-half").
//   * f16 x3 (forward; gradients with EGAZE_GRAD_SPLIT=f16): 22 significant bits per operand; measured |err| 2e-7 of
//     max|ref| on K = 4608 dot products (the exact-f32 MFMA kernel: 8.6e-7).  Weights are pre-scaled by 2^10 at pack time
//     (keeps the lo half normal), undone exactly in the epilogue; hi = round-toward-zero (saturating), lo = RNE residual;
//     a gradient operand is first multiplied by absmax_scale(max |x|) (egz_common.h).
//   * bf16 x3 (data gradient, default): fp32 exponent range for tiny gradients, 16 significant bits, |err| 5e-6.
// The activation operand is split on the fly while it is staged into LDS (it stays fp32 in HBM); the weight operand is
// split once per optimizer step (egz_pack_w3x3_split).  Same gather modes and epilogues as the fp32 kernel.
// Two kernels:
//   conv3x3_igemm_x3_kernel   per-tap gather (all modes: plain, upsample fold / phase, upsampled dgrad): block tile
//       128 x {128, 64}, K-slice = one tap x 32 channels, 4 waves as 2 x 2; software-pipelined (double-buffered LDS, two
//       register sets, conversion in the MFMA shadow), buffer-load gathers with per-row tap masks, swizzled 64-byte rows.
//   conv3x3_igemm_x3h_kernel  halo tile (plain convs): the input halo of a compact 128-pixel tile is staged ONCE per
//       32-channel block and the nine taps read it at shifted addresses; weights by LDS-DMA; fragments one k-step ahead.
#include "egz_common.h"
#include "x3_split.h"
#include <type_traits>

EGZ_API int egz_absmax(const float* x, long n, unsigned int* absmax, hipStream_t st);

namespace {

constexpr int XBM = 128, XBK = 32;
constexpr int XLD = 32;                 // row stride in 16-bit elements: 64 B, no padding -- the four 16-byte chunks
                                        // of a row are XOR-swizzled with (row >> 2) & 3, which makes the ds_read_b128
                                        // fragment reads conflict-free in every 16-lane service group
constexpr int NSET = 2;                 // staging register sets (slice s+1 being converted, slice s+2 in flight)

enum { EPI_BIAS = 0, EPI_BIAS_RELU = 1, EPI_BIAS_STATS = 2 };
enum { PLAIN = 0, UPS_FOLD = 1, UPS_PHASE = 2, UPS_DGRAD = 3 };

using namespace x3;

// wp: [2 planes (hi, lo)][taps][Kp][Cp] 16-bit.  H, W: hi-res (conv output) dims for the UPS_* modes.
template <typename T, int XBN, int MODE, int EPI>   // XBN = 128 (Cout % 128 == 0) or 64 (Cout = 64 layers)
__global__ __launch_bounds__(256, 2) void conv3x3_igemm_x3_kernel(
    const float* __restrict__ x, const unsigned short* __restrict__ wp, const float* __restrict__ bias,
    float* __restrict__ y, double* __restrict__ stat, int B, int H, int W, int C, int K, int Cp, int Kp,
    float out_scale, int mt, int tile_base, int nsplit, int pass, float* __restrict__ ws,
    const unsigned int* __restrict__ a_absmax, unsigned int* __restrict__ absmax_out) {
    // pass 0: whole tile (all K-slices + epilogue).  pass 1: split-K part blockIdx.y of nsplit -> raw accumulators to ws.
    // pass 2: sum the nsplit partials of the tile in a fixed order, then the normal epilogue (launch_x3: tail tiles).
    const float a_scale = absmax_scale(a_absmax);
    out_scale /= a_scale;
    constexpr int NTAP = (MODE == UPS_PHASE) ? 4 : (MODE == UPS_DGRAD) ? 16 : 9;
    constexpr int NR = XBN / 64;            // 32-wide n-tiles per wave (2 x 2 waves)
    constexpr int WN = XBN / 2;
    constexpr int BLD = XBN / 64;           // 16-byte B chunks per thread per plane
    constexpr int ABUF = 2 * XBM * XLD, BBUF = 2 * XBN * XLD;                    // elements per LDS buffer
    __shared__ __attribute__((aligned(16))) unsigned short As[2 * ABUF];         // [buffer][plane][row][k swizzled]
    __shared__ __attribute__((aligned(16))) unsigned short Bs[2 * BBUF];
    __shared__ long Ro[XBM];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, hl = lane >> 5, l31 = lane & 31;
    const int ntn = Kp / XBN;
    const int gb = blockIdx.x + tile_base, tpp = mt * ntn;       // global tile id over [phase][tile_m][tile_n]
    const int phase = (MODE == UPS_PHASE) ? gb / tpp : 0, py = phase >> 1, px = phase & 1;
    const int tl = gb - phase * tpp;
    const int tile_n = tl % ntn, tile_m = tl / ntn;
    const int m0 = tile_m * XBM, n0 = tile_n * XBN;
    const int Hr = (MODE >= UPS_PHASE) ? (H >> 1) : H, Wr = (MODE >= UPS_PHASE) ? (W >> 1) : W;
    const int Hg = (MODE == UPS_FOLD || MODE == UPS_PHASE) ? (H >> 1) : H;
    const int Wg = (MODE == UPS_FOLD || MODE == UPS_PHASE) ? (W >> 1) : W;
    const long HWr = (long)Hr * Wr;
    const long M = (long)B * HWr;
    const long plane = (long)((MODE == UPS_PHASE) ? 16 : NTAP) * Kp * Cp;    // elements per weight plane

    const int a_c4 = tid & 7, r0 = tid >> 3;            // A: rows r0 + 32 j, channels 4*a_c4 .. +3
    const int b_ch = tid & 3, b_r0 = tid >> 2;          // B: rows b_r0 + 64 j, 16-byte chunk b_ch (8 k)
    int a_y[4], a_x[4];
    long a_img[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
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
    // Operand fetch through buffer loads (all modes but UPS_FOLD, whose >> 1 gather is not affine in the tap): per row a
    // 32-bit byte offset and a bit mask of the taps whose source pixel exists are fixed for the whole tile; per slice the
    // tap / channel-block displacement is one scalar offset for all lanes, and a row whose tap falls outside the image
    // gets offset 0xFFFFFFFF, which the buffer bounds check turns into zeros.  ~3 VALU per row per slice instead of ~14.
    // The resource base sits (Wg + 1) * C floats before x so that scalar displacements are never negative.
    constexpr bool BUFA = (MODE != UPS_FOLD);
    const unsigned a_bias = (unsigned)(Wg + 1) * (unsigned)C * 4u;
    const __amdgpu_buffer_rsrc_t a_rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(x)) - a_bias, 0,
        (int)((unsigned)B * Hg * Wg * C * 4u + a_bias), 0x00020000);
    const __amdgpu_buffer_rsrc_t b_rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned short*>(wp), 0, (int)(4 * plane), 0x00020000);
    unsigned a_vo[4], a_mask[4], b_vo[2][BLD];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int by = (MODE == UPS_DGRAD) ? 2 * a_y[j] : a_y[j], bx = (MODE == UPS_DGRAD) ? 2 * a_x[j] : a_x[j];
        a_vo[j] = (unsigned)(((a_img[j] + (long)by * Wg + bx) * C + a_c4 * 4) * 4);
        unsigned mk = 0;
#pragma unroll
        for (int t = 0; t < NTAP; ++t) {
            const int dy = (MODE == UPS_PHASE) ? (t >> 1) + py - 1 : (MODE == UPS_DGRAD) ? (t >> 2) - 1 : t / 3 - 1;
            const int dx = (MODE == UPS_PHASE) ? (t & 1) + px - 1 : (MODE == UPS_DGRAD) ? (t & 3) - 1 : t % 3 - 1;
            mk |= ((unsigned)(by + dy) < (unsigned)Hg && (unsigned)(bx + dx) < (unsigned)Wg) ? (1u << t) : 0u;
        }
        a_mask[j] = mk;
    }
#pragma unroll
    for (int pl = 0; pl < 2; ++pl)
#pragma unroll
        for (int j = 0; j < BLD; ++j)
            b_vo[pl][j] = (unsigned)((pl * plane + (long)(n0 + b_r0 + 64 * j) * Cp + b_ch * 8) * 2);

    f32x4 ra[NSET][4];
    u32x4 rb[NSET][2][BLD];   // [set][plane][j]
    auto gload = [&](int s, const int set) {
        const int cblk = s / NTAP, tap = s - cblk * NTAP;
        const int c0 = cblk * XBK;
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
        if constexpr (BUFA) {
            const unsigned so_a = (unsigned)(((dy * Wg + dx) * C + c0) * 4) + a_bias;
            const unsigned so_b = (unsigned)((((long)(phase * NTAP + tap) * Kp) * Cp + c0) * 2);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned vo = a_vo[j] | (((a_mask[j] >> tap) & 1u) - 1u);      // valid ? offset : 0xFFFFFFFF
                ra[set][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(a_rs, vo, so_a, 0));
            }
#pragma unroll
            for (int pl = 0; pl < 2; ++pl)
#pragma unroll
                for (int j = 0; j < BLD; ++j)
                    rb[set][pl][j] = __builtin_amdgcn_raw_buffer_load_b128(b_rs, b_vo[pl][j], so_b, 0);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int iy = a_y[j] + dy, ix = a_x[j] + dx;
                const bool ok = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
                const float* p = x + ((a_img[j] + (long)(iy >> 1) * Wg + (ix >> 1)) * C + c0 + a_c4 * 4);
                ra[set][j] = ok ? *reinterpret_cast<const f32x4*>(p) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int pl = 0; pl < 2; ++pl)
#pragma unroll
                for (int j = 0; j < BLD; ++j) {
                    const unsigned short* p = wp + pl * plane +
                                              ((long)((phase * NTAP + tap) * Kp + n0 + b_r0 + 64 * j) * Cp + c0 + b_ch * 8);
                    rb[set][pl][j] = *reinterpret_cast<const u32x4*>(p);
                }
        }
    };
    // LDS staging of register set `set` into buffer `buf`, in five pieces (four A rows with their hi/lo conversion, then
    // the pre-split B chunks) so the main loop can spread them between the MFMAs of the slice being multiplied
    const int a_sw = (r0 >> 2) & 3, b_sw = (b_r0 >> 2) & 3;     // (row >> 2) & 3 of every row this thread stages
    const int a_off = r0 * XLD + (((a_c4 >> 1) ^ a_sw) << 3) + (a_c4 & 1) * 4;
    const int b_off = b_r0 * XLD + ((b_ch ^ b_sw) << 3);
    auto lstore_a = [&](const int set, const int buf, const int j) {
        u32x2 hi, lo;
        Half<T>::split4(ra[set][j] * a_scale, hi, lo);
        unsigned short* d = As + buf * ABUF + a_off + 32 * j * XLD;
        *reinterpret_cast<u32x2*>(d) = hi;
        *reinterpret_cast<u32x2*>(d + XBM * XLD) = lo;
    };
    auto lstore_b = [&](const int set, const int buf) {
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int j = 0; j < BLD; ++j)
                *reinterpret_cast<u32x4*>(Bs + buf * BBUF + pl * XBN * XLD + b_off + 64 * j * XLD) = rb[set][pl][j];
    };

    f32x16 acc[2][NR];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NR; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int S = (Cp / XBK) * NTAP;
    // one K-slice: 2 k-steps x (fragment reads, 3 * 2 * NR MFMAs).  The conversion + staging of the NEXT slice (register
    // set `set` -> LDS buffer buf ^ 1) is issued between the MFMAs, whose 32-cycle matrix-pipe occupancy leaves ~7 issue
    // slots each: the VALU / LDS-write work disappears into that shadow instead of forming a phase of its own.
    const int f_sw = (l31 >> 2) & 3;
    const int fo0 = ((hl ^ f_sw) << 3), fo1 = fo0 ^ 16;          // swizzled chunk of k-step 0 / 1 for this lane
    auto slice = [&](const int buf, const int set, const bool stage) {
        const unsigned short* Ab = As + buf * ABUF + (wm * 64 + l31) * XLD;
        const unsigned short* Bb = Bs + buf * BBUF + (wn * WN + l31) * XLD;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int fo = ks ? fo1 : fo0;
            u32x4 ah[2], al[2], bh[NR], bl[NR];
#pragma unroll
            for (int mr = 0; mr < 2; ++mr) {
                ah[mr] = *reinterpret_cast<const u32x4*>(Ab + mr * 32 * XLD + fo);
                al[mr] = *reinterpret_cast<const u32x4*>(Ab + XBM * XLD + mr * 32 * XLD + fo);
            }
#pragma unroll
            for (int nr = 0; nr < NR; ++nr) {
                bh[nr] = *reinterpret_cast<const u32x4*>(Bb + nr * 32 * XLD + fo);
                bl[nr] = *reinterpret_cast<const u32x4*>(Bb + XBN * XLD + nr * 32 * XLD + fo);
            }
            // small terms first; the three products of one accumulator are 2 * NR MFMAs apart (no dependent back-to-back)
#pragma unroll
            for (int term = 0; term < 3; ++term) {
#pragma unroll
                for (int nr = 0; nr < NR; ++nr)
#pragma unroll
                    for (int mr = 0; mr < 2; ++mr)
                        if (!(egz_drop_alo<T>::value && term == 0)) acc[mr][nr] = Half<T>::mfma(term == 0 ? al[mr] : ah[mr], term == 1 ? bl[nr] : bh[nr], acc[mr][nr]);
                if (stage) {
                    if (term < 2) lstore_a(set, buf ^ 1, ks * 2 + term);
                    else if (ks == 1) lstore_b(set, buf ^ 1);
                }
            }
            // pin the interleave: one MFMA, then a few VALU (conversion) and an LDS write
#pragma unroll
            for (int i = 0; i < 6 * NR; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
            }
        }
    };
    // software pipeline over the K-slices [s_lo, s_hi): LDS is double buffered, registers hold two slices.  In iteration s
    // the MFMAs read buffer s & 1 while slice s + 1 (loaded during iteration s - 1) is converted into the other buffer
    // and slice s + 2 is fetched from L2 / HBM; one barrier per slice.
    int s_lo = 0, s_hi = S;
    if (pass == 1) {
        s_lo = (int)((long)S * blockIdx.y / nsplit);
        s_hi = (int)((long)S * (blockIdx.y + 1) / nsplit);
    }
    if (pass != 2) {
        gload(s_lo, 0);
        gload(s_lo + 1 < s_hi ? s_lo + 1 : s_hi - 1, 1);
#pragma unroll
        for (int j = 0; j < 4; ++j) lstore_a(0, 0, j);
        lstore_b(0, 0);
        __syncthreads();
        for (int s = s_lo; s < s_hi; s += 2) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (s + u < s_hi) {      // block-uniform
                    const int sp = (s + u + 2 < s_hi) ? s + u + 2 : s_hi - 1;   // branch-free prefetch (clamped past the end)
                    // set u held slice s + u, staged during the previous iteration -> free for slice s + u + 2;
                    // set u ^ 1 holds slice s + u + 1 -> staged into buffer (u ^ 1) now
                    gload(sp, u);
                    slice(u, u ^ 1, true);
                    __syncthreads();
                }
            }
        }
    } else {
        __syncthreads();                          // Ro
    }
    if (pass) {      // raw accumulators, thread-major: element ((mr * NR + nr) * 16 + r) * 256 + tid -> coalesced both ways
        float* wt = ws + (long)blockIdx.x * nsplit * (XBM * XBN);
        if (pass == 1) {
            wt += (long)blockIdx.y * (XBM * XBN);
#pragma unroll
            for (int mr = 0; mr < 2; ++mr)
#pragma unroll
                for (int nr = 0; nr < NR; ++nr)
#pragma unroll
                    for (int r = 0; r < 16; ++r) wt[((mr * NR + nr) * 16 + r) * 256 + tid] = acc[mr][nr][r];
            return;
        }
        for (int sp = 0; sp + 1 < nsplit; sp += 2, wt += 2 * XBM * XBN)     // pairs: twice the loads in flight
#pragma unroll
            for (int mr = 0; mr < 2; ++mr)
#pragma unroll
                for (int nr = 0; nr < NR; ++nr)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int o = ((mr * NR + nr) * 16 + r) * 256 + tid;
                        acc[mr][nr][r] = (acc[mr][nr][r] + wt[o]) + wt[XBM * XBN + o];
                    }
        if (nsplit & 1)
#pragma unroll
            for (int mr = 0; mr < 2; ++mr)
#pragma unroll
                for (int nr = 0; nr < NR; ++nr)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[mr][nr][r] += wt[((mr * NR + nr) * 16 + r) * 256 + tid];
    }

    // ---- epilogue (same as the fp32 kernel; out_scale undoes the weight pre-scaling of the f16 path exactly).  Outputs below
    // 4 GiB leave through buffer stores with an out-of-range offset for invalid rows / columns instead of a per-lane branch
    // around each store (see conv3x3_igemm_x3h_kernel).
    double* red = reinterpret_cast<double*>(As);   // [2 (wm)][2 (sum, sumsq)][128] doubles = 4 KB
    __shared__ float samax[4];
    float amx = 0.f;                               // EPI_BIAS_RELU: max of the activation this thread wrote
    const bool bufst = (unsigned long long)B * H * W * K * 4ull < (1ull << 32);        // block-uniform (H, W: the output image)
    const __amdgpu_buffer_rsrc_t y_rs = __builtin_amdgcn_make_buffer_rsrc(y, 0, bufst ? (int)((unsigned)B * H * W * K * 4u) : 0, 0x00020000);
#pragma unroll
    for (int nr = 0; nr < NR; ++nr) {
        const int col = wn * WN + nr * 32 + l31;
        const bool nok = n0 + col < K;
        const float bz = (bias && nok) ? bias[n0 + col] : 0.f;
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int mr = 0; mr < 2; ++mr) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long off = Ro[wm * 64 + mr * 32 + egz_acc_row(r, lane)];
                if (bufst) {
                    const bool ok = off >= 0 && nok;
                    float v = acc[mr][nr][r] * out_scale + bz;
                    if (EPI == EPI_BIAS_RELU) v = fmaxf(v, 0.f);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), y_rs,
                                                          ok ? (unsigned)(off + n0 + col) * 4u : 0xFFFFFFFFu, 0, 0);
                    const float vs = ok ? v : 0.f;
                    if (EPI == EPI_BIAS_RELU) amx = fmaxf(amx, vs);
                    if (EPI == EPI_BIAS_STATS) {
                        s1 += (double)vs;
                        s2 += (double)vs * (double)vs;
                    }
                    continue;
                }
                if (off >= 0 && nok) {
                    float v = acc[mr][nr][r] * out_scale + bz;
                    if (EPI == EPI_BIAS_RELU) {
                        v = fmaxf(v, 0.f);
                        amx = fmaxf(amx, v);
                    }
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
                red[(wm * 2 + 0) * XBN + col] = s1;
                red[(wm * 2 + 1) * XBN + col] = s2;
            }
        }
    }
    if (EPI == EPI_BIAS_RELU && absmax_out && pass == 0) {                    // block-uniform: per-block partial, folded by the launcher
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) amx = fmaxf(amx, __shfl_xor(amx, o));
        if (lane == 0) samax[wave] = amx;
        __syncthreads();
        if (tid == 0) absmax_commit(absmax_out, blockIdx.x, fmaxf(fmaxf(samax[0], samax[1]), fmaxf(samax[2], samax[3])));
    }
    if (EPI == EPI_BIAS_STATS) {
        __syncthreads();
        if (tid < XBN && n0 + tid < K) {
            const double t1 = red[(0 * 2 + 0) * XBN + tid] + red[(1 * 2 + 0) * XBN + tid];
            const double t2 = red[(0 * 2 + 1) * XBN + tid] + red[(1 * 2 + 1) * XBN + tid];
            const long srow = (long)phase * mt + tile_m;
            stat[(srow * 2 + 0) * K + n0 + tid] = t1;
            stat[(srow * 2 + 1) * K + n0 + tid] = t2;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Halo-tile variant of the plain (stride 1, pad 1) 3x3 implicit GEMM.  The kernel above fetches and converts every
// activation element once per tap and per n-tile; here the 128 output pixels of a tile are a compact set -- a raster run
// of 128 consecutive pixels (W <= 56) or an 8 x 16 patch (W % 16 == 0, H % 8 == 0) -- whose input halo (run +- (W + 1)
// pixels = 130 + 2W slots, or 10 x 18 pixels in a 10 x 20 grid) is fetched, split into hi / lo halves and written to LDS
// ONCE per 32-channel block; the nine taps then read their fragments from that one image at shifted addresses.  Per
// K-slice that is 6.4x (patch) / 4.8-8x (run) fewer activation loads and conversions, and no per-slice address work:
// the 18 fragment addresses (2 row groups x 9 taps) are computed once per tile and live in registers.
//   * out-of-image halo slots are filled with zeros by the buffer bounds check, so a patch needs no tap masks at all;
//     in a raster run a tap that wraps around a row end or an image border reads the all-zero slot 255 instead;
//   * LDS A image: [plane hi/lo][256 slots][32 ch] 16-bit, 64-byte rows, 16-byte chunks XOR-swizzled with (col >> 2) & 3
//     where col = the slot (run) or the halo column (patch, row pitch 20): conflict-free ds_read_b128 for 32 consecutive
//     slots at any start offset and for two 16-runs one grid row apart (brute-forced over the gfx950 service groups);
//   * B (pre-split weights) as in the kernel above: one tap x 32 channels per slice, double buffered, two register sets;
//     one barrier per slice, two more per channel block around the A restaging (single A buffer: 33 + 32 KB LDS, 2 blocks/CU).
constexpr int HSLOTS = 256, HZERO = 255, HPITCH = 20;

template <typename T, int XBN, int EPI>
__global__ __launch_bounds__(256, (XBN == 64) ? 3 : 2) void conv3x3_igemm_x3h_kernel(
    const float* __restrict__ x, const unsigned short* __restrict__ wp, const float* __restrict__ bias,
    float* __restrict__ y, double* __restrict__ stat, int B, int H, int W, int C, int K, int Cp, int Kp,
    float out_scale, int mt, int patch, const unsigned int* __restrict__ a_absmax, unsigned int* __restrict__ absmax_out) {
    const float a_scale = absmax_scale(a_absmax);
    out_scale /= a_scale;
    constexpr int NR = XBN / 64, WN = XBN / 2, BLD = XBN / 64;
    constexpr int APL = HSLOTS * XLD;                       // elements per A plane
    constexpr int BBUF = 2 * XBN * XLD;
    constexpr int NJ = HSLOTS / 32;                         // halo slots per thread (8 threads x 4 channels per slot)
    __shared__ __attribute__((aligned(16))) unsigned short Ah[2 * APL];
    __shared__ __attribute__((aligned(16))) unsigned short Bs[2 * BBUF];
    __shared__ long Ro[XBM];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, hl = lane >> 5, l31 = lane & 31;
    const int ntn = Kp / XBN;
    const int tile_n = blockIdx.x % ntn, tile_m = blockIdx.x / ntn;
    const int n0 = tile_n * XBN;
    const long HW = (long)H * W, M = (long)B * HW;
    const long plane = (long)9 * Kp * Cp;
    // tile origin: raster run = flat pixel m0; patch = image b0, top-left (y0, x0)
    const long m0 = (long)tile_m * XBM;
    const int pw = W >> 4, ppi = (H >> 3) * pw;             // patches per row / per image (patch geometry only)
    const int b0 = patch ? tile_m / ppi : 0;
    const int y0 = patch ? ((tile_m - b0 * ppi) / pw) * 8 : 0, x0 = patch ? ((tile_m - b0 * ppi) % pw) * 16 : 0;

    // output offsets of the 128 tile rows
    if (tid < XBM) {
        long off = -1;
        if (patch) {
            off = (((long)b0 * H + y0 + (tid >> 4)) * W + x0 + (tid & 15)) * K;
        } else if (m0 + tid < M) {
            off = (m0 + tid) * K;
        }
        Ro[tid] = off;
    }

    // ---- A halo staging map: slot q = (tid >> 3) + 32 j holds 4 channels (tid & 7) of one input pixel
    const __amdgpu_buffer_rsrc_t a_rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(x), 0, (int)((unsigned)B * H * W * C * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t b_rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned short*>(wp), 0, (int)(4 * plane), 0x00020000);
    const int a_c4 = tid & 7;
    unsigned a_vo[NJ];
    int a_lds[NJ];                                          // element offset of the slot's 8-byte piece in a plane
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int q = (tid >> 3) + 32 * j;
        long pix = -1;
        int col = q;
        if (patch) {
            const int hy = q / HPITCH, hx = q - hy * HPITCH;
            const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
            col = hx;
            if (hy < 10 && hx < 18 && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
                pix = ((long)b0 * H + iy) * W + ix;
        } else {
            const long g = m0 - W - 1 + q;
            if (q < XBM + 2 * W + 2 && g >= 0 && g < M) pix = g;
        }
        a_vo[j] = (pix >= 0) ? (unsigned)((pix * C + a_c4 * 4) * 4) : 0xFFFFFFFFu;
        a_lds[j] = q * XLD + (((a_c4 >> 1) ^ ((col >> 2) & 3)) << 3) + (a_c4 & 1) * 4;
    }
    // ---- B: the pre-split weight slice goes HBM/L2 -> LDS directly (buffer_load ... lds, 1 KB = 16 rows x 64 B per wave
    // instruction, no VGPRs, no ds_write).  The LDS image is lane-linear, so the chunk swizzle is applied on the global
    // side: lane (row r = lane >> 2, position p = lane & 3) fetches chunk p ^ ((r >> 2) & 3) of its row.
    constexpr int NPIECE = 2 * (XBN / 16) / 4;              // 1 KB pieces per wave per slice (planes x row blocks / waves)
    unsigned b_vo[NPIECE];
    int b_lds[NPIECE];                                       // element offset of the piece inside a B buffer
#pragma unroll
    for (int k = 0; k < NPIECE; ++k) {
        const int pi = wave + 4 * k, pl = pi / (XBN / 16), rb16 = pi % (XBN / 16);
        const int r = lane >> 2, p = lane & 3;
        b_vo[k] = (unsigned)((pl * plane + (long)(n0 + rb16 * 16 + r) * Cp + ((p ^ ((r >> 2) & 3)) << 3)) * 2);
        b_lds[k] = pl * XBN * XLD + rb16 * 16 * XLD;
    }

    // ---- fragment addresses: lane row i = wm * 64 + mr * 32 + l31, tap t -> byte address of the k-step-0 chunk in plane 0
    int fa[2][9];
#pragma unroll
    for (int mr = 0; mr < 2; ++mr) {
        const int i = wm * 64 + mr * 32 + l31;
        int py = 0, px = 0;
        bool rowok = true;
        if (!patch) {
            const long m = m0 + i;
            rowok = m < M;
            const int rem = (int)(m % HW);
            py = rem / W;
            px = rem - py * W;
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int dy = t / 3 - 1, dx = t % 3 - 1;
            int slot, col;
            if (patch) {
                slot = ((i >> 4) + 1 + dy) * HPITCH + (i & 15) + 1 + dx;
                col = (i & 15) + 1 + dx;
            } else {
                const bool ok = rowok && (unsigned)(py + dy) < (unsigned)H && (unsigned)(px + dx) < (unsigned)W;
                slot = ok ? i + W + 1 + dy * W + dx : HZERO;
                col = slot;
            }
            fa[mr][t] = (slot * XLD + ((hl ^ ((col >> 2) & 3)) << 3)) * 2;
        }
    }
    const int fb = ((wn * WN + l31) * XLD + ((hl ^ ((l31 >> 2) & 3)) << 3)) * 2;     // B fragment, k-step 0, plane 0

    f32x4 ra[NJ];
    auto gload_a = [&](int cblk) {
        const unsigned so = (unsigned)(cblk * XBK * 4);
#pragma unroll
        for (int j = 0; j < NJ; ++j)
            ra[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(a_rs, a_vo[j], so, 0));
    };
    auto lstore_a = [&]() {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            u32x2 hi, lo;
            Half<T>::split4s(ra[j], a_scale, hi, lo);
            *reinterpret_cast<u32x2*>(Ah + a_lds[j]) = hi;
            *reinterpret_cast<u32x2*>(Ah + APL + a_lds[j]) = lo;
        }
    };
    auto dma_so = [&](int s) {                           // scalar offset of slice s = cblk * 9 + tap in a weight plane
        const int cblk = s / 9, tap = s - cblk * 9;
        return (unsigned)((((long)tap * Kp) * Cp + cblk * XBK) * 2);
    };
    auto dma_piece = [&](const unsigned so, const int buf, const int k) {
#if defined(__HIP_DEVICE_COMPILE__)      // the host pass of hipcc rejects the LDS-DMA builtin (and then drops the kernel stub)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(
            b_rs, (__attribute__((address_space(3))) void*)(Bs + buf * BBUF + b_lds[k]), 16, b_vo[k], so, 0, 0);
#else
        (void)so; (void)buf; (void)k;
#endif
    };
    auto dma_b = [&](int s, const int buf) {             // slice s -> B buffer buf
        const unsigned so = dma_so(s);
#pragma unroll
        for (int k = 0; k < NPIECE; ++k) dma_piece(so, buf, k);
    };

    f32x16 acc[2][NR];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NR; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const char* Ab = reinterpret_cast<const char*>(Ah);
    const char* Bb = reinterpret_cast<const char*>(Bs);
    // Fragment pipeline.  An LDS read needs ~130 cycles to land and a k-step is only 12 MFMAs (384 cycles), so fragments
    // are fetched one k-step ahead: k-step 1 of the slice before the MFMAs of k-step 0, the A fragments of the next tap
    // (same halo image, no hazard) before the MFMAs of k-step 1, the B fragments of the next slice right after the barrier.
    // sched_barrier(0) pins those points (hipcc otherwise sinks each read next to its use and every wave stalls ~6 x 100
    // cycles per slice: MFMA pipe 54 % busy).
    u32x4 ah0[2], al0[2], bh0[NR], bl0[NR];               // k-step 0 fragments of the slice about to run
    auto read_a0 = [&](const int t) {
#pragma unroll
        for (int mr = 0; mr < 2; ++mr) {
            ah0[mr] = *reinterpret_cast<const u32x4*>(Ab + fa[mr][t]);
            al0[mr] = *reinterpret_cast<const u32x4*>(Ab + APL * 2 + fa[mr][t]);
        }
    };
    auto read_b0 = [&](const int buf) {
#pragma unroll
        for (int nr = 0; nr < NR; ++nr) {
            const int b = fb + buf * BBUF * 2 + nr * 32 * XLD * 2;
            bh0[nr] = *reinterpret_cast<const u32x4*>(Bb + b);
            bl0[nr] = *reinterpret_cast<const u32x4*>(Bb + XBN * XLD * 2 + b);
        }
    };
    // one K-slice (tap t of the staged channel block) on B buffer `buf`
    auto slice = [&](const int t, const int buf, const bool next_a, const unsigned dso) {
        u32x4 ah1[2], al1[2], bh1[NR], bl1[NR];
#pragma unroll
        for (int mr = 0; mr < 2; ++mr) {
            const int a = fa[mr][t] ^ 32;
            ah1[mr] = *reinterpret_cast<const u32x4*>(Ab + a);
            al1[mr] = *reinterpret_cast<const u32x4*>(Ab + APL * 2 + a);
        }
#pragma unroll
        for (int nr = 0; nr < NR; ++nr) {
            const int b = (fb ^ 32) + buf * BBUF * 2 + nr * 32 * XLD * 2;
            bh1[nr] = *reinterpret_cast<const u32x4*>(Bb + b);
            bl1[nr] = *reinterpret_cast<const u32x4*>(Bb + XBN * XLD * 2 + b);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int term = 0; term < 3; ++term)
#pragma unroll
            for (int nr = 0; nr < NR; ++nr)
#pragma unroll
                for (int mr = 0; mr < 2; ++mr) {
                    if (!(egz_drop_alo<T>::value && term == 0)) acc[mr][nr] = Half<T>::mfma(term == 0 ? al0[mr] : ah0[mr], term == 1 ? bl0[nr] : bh0[nr], acc[mr][nr]);
                }
        __builtin_amdgcn_sched_barrier(0);
        if (next_a) read_a0(t + 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int term = 0; term < 3; ++term)
#pragma unroll
            for (int nr = 0; nr < NR; ++nr)
#pragma unroll
                for (int mr = 0; mr < 2; ++mr)
                    if (!(egz_drop_alo<T>::value && term == 0)) acc[mr][nr] = Half<T>::mfma(term == 0 ? al1[mr] : ah1[mr], term == 1 ? bl1[nr] : bh1[nr], acc[mr][nr]);
    };

    const int ncb = Cp / XBK, S = ncb * 9;
    gload_a(0);
    dma_b(0, 0);
    lstore_a();
    __builtin_amdgcn_s_waitcnt(0);                            // vmcnt(0): this wave's B pieces have landed
    __syncthreads();
    read_a0(0);
    read_b0(0);
    // Two channel blocks per trip so that the B buffer index is compile-time: slice s uses buffer s & 1, and 9 taps per
    // block flip the parity.  The DMA of slice s + 1 is issued at the top of slice s (its buffer was last read in slice
    // s - 1, all waves are past that barrier) and has the 24 MFMAs of the slice to land.
    for (int cb = 0; cb < ncb; cb += 2) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int c = cb + h;
            if (c < ncb) {                                   // block-uniform
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int s = c * 9 + t, par = (h + t) & 1;          // s & 1 == par (cb is even)
                    const unsigned dso = 0;
                    dma_b(s + 1 < S ? s + 1 : S - 1, par ^ 1);
                    if (t == 6 && c + 1 < ncb) gload_a(c + 1);            // lands during slices 6..8
                    slice(t, par, t < 8, dso);
                    __builtin_amdgcn_s_waitcnt(0);            // vmcnt(0) (+ lgkmcnt(0)): B pieces of slice s + 1 in LDS
                    __syncthreads();
                    read_b0(par ^ 1);                         // next slice's B, k-step 0
                }
                if (c + 1 < ncb) {                            // restage the halo image for the next channel block
                    lstore_a();
                    __syncthreads();
                }
                read_a0(0);
            }
        }
    }

    // ---- epilogue (as conv3x3_igemm_x3_kernel).  Outputs below 4 GiB leave through BUFFER stores with an out-of-range offset
    // for rows / columns that do not exist instead of a per-lane branch around each store: with the branch every store sat in
    // its own basic block behind an s_waitcnt vmcnt(0), i.e. waited for the previous one to be acknowledged.
    double* red = reinterpret_cast<double*>(Ah);
    __shared__ float samax[4];
    float amx = 0.f;
    const bool bufst = (unsigned long long)B * H * W * K * 4ull < (1ull << 32);        // block-uniform (H, W: the output image)
    const __amdgpu_buffer_rsrc_t y_rs = __builtin_amdgcn_make_buffer_rsrc(y, 0, bufst ? (int)((unsigned)B * H * W * K * 4u) : 0, 0x00020000);
#pragma unroll
    for (int nr = 0; nr < NR; ++nr) {
        const int col = wn * WN + nr * 32 + l31;
        const bool nok = n0 + col < K;
        const float bz = (bias && nok) ? bias[n0 + col] : 0.f;
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int mr = 0; mr < 2; ++mr) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long off = Ro[wm * 64 + mr * 32 + egz_acc_row(r, lane)];
                if (bufst) {
                    const bool ok = off >= 0 && nok;
                    float v = acc[mr][nr][r] * out_scale + bz;
                    if (EPI == EPI_BIAS_RELU) v = fmaxf(v, 0.f);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), y_rs,
                                                          ok ? (unsigned)(off + n0 + col) * 4u : 0xFFFFFFFFu, 0, 0);
                    const float vs = ok ? v : 0.f;
                    if (EPI == EPI_BIAS_RELU) amx = fmaxf(amx, vs);
                    if (EPI == EPI_BIAS_STATS) {
                        s1 += (double)vs;
                        s2 += (double)vs * (double)vs;
                    }
                    continue;
                }
                if (off >= 0 && nok) {
                    float v = acc[mr][nr][r] * out_scale + bz;
                    if (EPI == EPI_BIAS_RELU) {
                        v = fmaxf(v, 0.f);
                        amx = fmaxf(amx, v);
                    }
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
                red[(wm * 2 + 0) * XBN + col] = s1;
                red[(wm * 2 + 1) * XBN + col] = s2;
            }
        }
    }
    if (EPI == EPI_BIAS_RELU && absmax_out) {                    // block-uniform: per-block partial, folded by the launcher
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) amx = fmaxf(amx, __shfl_xor(amx, o));
        if (lane == 0) samax[wave] = amx;
        __syncthreads();
        if (tid == 0) absmax_commit(absmax_out, blockIdx.x, fmaxf(fmaxf(samax[0], samax[1]), fmaxf(samax[2], samax[3])));
    }
    if (EPI == EPI_BIAS_STATS) {
        __syncthreads();
        if (tid < XBN && n0 + tid < K) {
            const double t1 = red[(0 * 2 + 0) * XBN + tid] + red[(1 * 2 + 0) * XBN + tid];
            const double t2 = red[(0 * 2 + 1) * XBN + tid] + red[(1 * 2 + 1) * XBN + tid];
            stat[((long)tile_m * 2 + 0) * K + n0 + tid] = t1;
            stat[((long)tile_m * 2 + 1) * K + n0 + tid] = t2;
        }
    }
}

// ---------------------------------------------------------------------------------------------- split packing
__device__ __forceinline__ float weff9(const float* __restrict__ w9, int py, int a, int px, int b) {
    const int rlo = (py == 0) ? (a == 0 ? 0 : 1) : (a == 0 ? 0 : 2), rhi = (py == 0) ? (a == 0 ? 0 : 2) : (a == 0 ? 1 : 2);
    const int slo = (px == 0) ? (b == 0 ? 0 : 1) : (b == 0 ? 0 : 2), shi = (px == 0) ? (b == 0 ? 0 : 2) : (b == 0 ? 1 : 2);
    float s = 0.f;
    for (int r = rlo; r <= rhi; ++r)
        for (int q = slo; q <= shi; ++q) s += w9[r * 3 + q];
    return s;
}
// kind 0 fwd [9][Kp][Cp], 1 dgrad [9][Cp][Kp], 2 ups_fwd [16][Kp][Cp], 3 ups_dgrad [16][Cp][Kp]  (same value
// definitions as the fp32 pack kernels of conv3x3_igemm.hip); output = hi plane then lo plane, 16-bit each.
// value of packed element i of layout `kind` (0 fwd [9][Kp][Cp], 1 dgrad [9][Cp][Kp], 2 ups_fwd [16][Kp][Cp], 3 ups_dgrad [16][Cp][Kp])
__device__ __forceinline__ float pack_value(const float* __restrict__ w, long i, int C, int K, int Cp, int Kp, int kind) {
    int c, k, tap;
    if (kind == 0 || kind == 2) {
        c = (int)(i % Cp);
        const long t = i / Cp;
        k = (int)(t % Kp);
        tap = (int)(t / Kp);
    } else {
        k = (int)(i % Kp);
        const long t = i / Kp;
        c = (int)(t % Cp);
        tap = (int)(t / Cp);
    }
    float v = 0.f;
    if (c < C && k < K) {
        const float* w9 = w + ((long)k * C + c) * 9;
        if (kind == 0) v = w9[tap];
        else if (kind == 1) v = w9[8 - tap];
        else if (kind == 2) v = weff9(w9, (tap >> 2) >> 1, (tap & 3) >> 1, (tap >> 2) & 1, tap & 1);
        else {
            const int oy = (tap >> 2) - 1, ox = (tap & 3) - 1;
            v = weff9(w9, (oy == -1 || oy == 1) ? 1 : 0, (oy <= 0) ? 1 : 0, (ox == -1 || ox == 1) ? 1 : 0, (ox <= 0) ? 1 : 0);
        }
    }
    return v;
}

template <typename T>
__global__ void pack_split_kernel(const float* __restrict__ w, unsigned short* __restrict__ wp, int C, int K, int Cp,
                                  int Kp, int kind, float scale) {
    const int taps = (kind >= 2) ? 16 : 9;
    const long n = (long)taps * Cp * Kp;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        unsigned short h, l;
        Half<T>::split(pack_value(w, i, C, K, Cp, Kp, kind) * scale, h, l);
        wp[i] = h;
        wp[n + i] = l;
    }
}

// All split packings of a model in ONE launch (after the optimizer step): 74 tiny launches per step become one.
// table: n rows of 8 int64 = {w, wp, C, K, kind, dtype, nelem, first block}; a block packs PACK_PER_BLOCK elements of
// the row that owns it (binary search over the first-block column).
constexpr int PACK_PER_BLOCK = 2048;

// Tile schedule.  Every block of this kernel does the same work, so a launch runs in rounds of R = (resident blocks per
// CU) x (CUs) tiles; the SP shapes give 196 / 784 / 1568 / 3136 tiles, i.e. 0.4 / 1.5 / 3.06 / 6.1 rounds of 512.  With
// flag 0x4000 the tiles beyond the last full round (when they fill at most a quarter of a round) are run split-K -- each
// tile's K-slices divided over nsplit blocks -- as raw partial accumulators through the workspace, and a third small
// launch sums them in a fixed order and applies the epilogue.  Measured on MI355X this does NOT pay for the SP shapes
// (the blocks of a thin last round run ~1.7x faster than co-resident ones, while the fixup pass reads nsplit x 64 KB per
// tile from few CUs): enc7 263 vs 240 us, enc10 433 vs 415 us.  The default is therefore the plain single launch.
struct X3Plan { int mt, ntn, nph, total, main, tail, nsplit; };
int x3_slots(int XBN) {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
            cus = 256;
    }
    // resident blocks per CU: 128 x 128 tile = 204 VGPRs / 65 KB LDS -> 2; 128 x 64 tile = 129 VGPRs / 49 KB LDS -> 3
    return (XBN == 128 ? 2 : 3) * cus;
}
X3Plan x3_plan(long M, int Cp, int Kp, int XBN, int mode, int flags) {
    X3Plan p;
    p.mt = egz_cdiv(M, XBM);
    p.ntn = Kp / XBN;
    p.nph = (mode == UPS_PHASE) ? 4 : 1;
    p.total = p.mt * p.ntn * p.nph;
    p.main = p.total;
    p.tail = 0;
    p.nsplit = 1;
    const int ntap = (mode == UPS_PHASE) ? 4 : (mode == UPS_DGRAD) ? 16 : 9;
    const int nslices = (Cp / XBK) * ntap, R = x3_slots(XBN);
    const int rem = p.total % R;
    if ((flags & 0x4000) && rem != 0 && rem * 4 <= R) {
        int ns = R / rem;
        if (ns > 16) ns = 16;
        if (ns > nslices / 2) ns = nslices / 2;
        if (ns >= 2) {
            p.main = p.total - rem;
            p.tail = rem;
            p.nsplit = ns;
        }
    }
    return p;
}
int x3_mode(int flags) { return (flags & 4) ? UPS_DGRAD : ((flags & 3) == 3) ? UPS_PHASE : ((flags & 3) == 1) ? UPS_FOLD : PLAIN; }
long x3_rows(int mode, int B, int H, int W) { return (mode >= UPS_PHASE) ? (long)B * (H / 2) * (W / 2) : (long)B * H * W; }

template <typename T, int XBN, int MODE>
int launch_x3(int epi, const float* x, const unsigned short* wp, const float* bias, float* y, double* stat, int B, int H,
              int W, int C, int K, float out_scale, int flags, float* ws, size_t ws_bytes, const unsigned int* a_absmax,
              unsigned int* absmax_out, hipStream_t st) {
    const long M = x3_rows(MODE, B, H, W);
    if (epi != EPI_BIAS_RELU || MODE == UPS_DGRAD) absmax_out = nullptr;
    // max |y| of a bias + ReLU launch (the f16 x3 scaling of the next convolution): per-block partials from the epilogue when
    // the launch is one plain grid of at most 16384 blocks, else one extra pass over y
    const long ny = (long)B * H * W * K;
    auto absmax_pass = [&]() -> int { return absmax_out ? egz_absmax(y, ny, absmax_out, st) : 0; };
    const int Cp = (C + 31) / 32 * 32, Kp = (K + 31) / 32 * 32;
    if constexpr (MODE == PLAIN) {
        // halo-tile kernel: 8 x 16 patches, or raster runs for narrow images; flag 0x2000 forces the per-tap gather kernel
        const bool patch = (W % 16 == 0) && (H % 8 == 0);
        if (!(flags & 0x2000) && (patch || (W <= 56 && XBM + 2 * W + 2 <= HZERO))) {
            const int mt = egz_cdiv(M, XBM);
            dim3 grid(mt * (Kp / XBN));
            unsigned int* amo = (grid.x <= 16384) ? absmax_out : nullptr;
#define EGZ_X3H(E) hipLaunchKernelGGL((conv3x3_igemm_x3h_kernel<T, XBN, E>), grid, dim3(256), 0, st, x, wp, bias, y, stat, B, H, W, C, K, Cp, Kp, out_scale, mt, patch ? 1 : 0, a_absmax, amo)
            if (epi == EPI_BIAS) EGZ_X3H(EPI_BIAS);
            else if (epi == EPI_BIAS_RELU) EGZ_X3H(EPI_BIAS_RELU);
            else EGZ_X3H(EPI_BIAS_STATS);
#undef EGZ_X3H
            EGZ_CHECK_LAUNCH("egz_conv3x3_fwd_split(halo)");
            return absmax_pass();
        }
    }
    const X3Plan p = x3_plan(M, Cp, Kp, XBN, MODE, flags);
    EGZ_CHECK_ARG(!p.tail || (ws && ws_bytes >= (size_t)p.tail * p.nsplit * XBM * XBN * sizeof(float)),
                  "egz_conv3x3_fwd_split: workspace too small (%zu bytes; see egz_conv3x3_fwd_split_ws_bytes)", ws_bytes);
    unsigned int* amo = (!p.tail && p.main <= 16384) ? absmax_out : nullptr;
#define EGZ_X3L(E, GRID, BASE, NS, PASS) hipLaunchKernelGGL((conv3x3_igemm_x3_kernel<T, XBN, MODE, E>), GRID, dim3(256), 0, st, x, wp, bias, y, stat, B, H, W, C, K, Cp, Kp, out_scale, p.mt, BASE, NS, PASS, ws, a_absmax, amo)
#define EGZ_X3(E)                                                                  \
    do {                                                                           \
        if (p.main) EGZ_X3L(E, dim3(p.main), 0, 1, 0);                             \
        if (p.tail) {                                                              \
            EGZ_X3L(E, dim3(p.tail, p.nsplit), p.main, p.nsplit, 1);               \
            EGZ_X3L(E, dim3(p.tail), p.main, p.nsplit, 2);                         \
        }                                                                          \
    } while (0)
    if (MODE == UPS_DGRAD || epi == EPI_BIAS) EGZ_X3(EPI_BIAS);
    else if (epi == EPI_BIAS_RELU) EGZ_X3(EPI_BIAS_RELU);
    else EGZ_X3(EPI_BIAS_STATS);
#undef EGZ_X3
#undef EGZ_X3L
    EGZ_CHECK_LAUNCH("egz_conv3x3_fwd_split");
    return absmax_pass();
}

}  // namespace

// Split a (K, C, 3, 3) weight into hi / lo 16-bit planes in the packed layout `kind`
// (0 fwd, 1 dgrad, 2 ups_fwd, 3 ups_dgrad); dtype 1 = f16 (values pre-scaled by 2^10), 2 = bf16.
// wp needs egz_pack_w3x3_elems(C, K, kind >= 2) * 4 bytes (two 16-bit planes = one fp32 plane).
EGZ_API int egz_pack_w3x3_split(const float* w, void* wp, int C, int K, int kind, int dtype, hipStream_t st) {
    EGZ_CHECK_ARG(w && wp && C > 0 && K > 0 && kind >= 0 && kind <= 3 && (dtype == 1 || dtype == 2),
                  "egz_pack_w3x3_split: bad arguments");
    const int Cp = (C + 31) / 32 * 32, Kp = (K + 31) / 32 * 32;
    const long n = (long)(kind >= 2 ? 16 : 9) * Cp * Kp;
    const int g = egz_cdiv(n, 256) > 4096 ? 4096 : egz_cdiv(n, 256);
    unsigned short* o = static_cast<unsigned short*>(wp);
    if (dtype == 1) hipLaunchKernelGGL(pack_split_kernel<_Float16>, dim3(g), dim3(256), 0, st, w, o, C, K, Cp, Kp, kind, F16_WSCALE);
    else            hipLaunchKernelGGL(pack_split_kernel<__bf16>, dim3(g), dim3(256), 0, st, w, o, C, K, Cp, Kp, kind, 1.f);
    EGZ_CHECK_LAUNCH("egz_pack_w3x3_split");
    return 0;
}

// Same contract as egz_conv3x3_fwd (flags: bit0/bit1 upsample forms, bits 4-5 epilogue) / egz_conv3x3_ups_dgrad
// (flags bit 2 = 0x4 selects the 16-tap data gradient of an upsampled conv), computed with split-half operands.
// dtype 1 = f16 x3, 2 = bf16 x3; wp from egz_pack_w3x3_split with the same dtype.  Needs Cout % 64 == 0, Cin % 32 == 0
// (tile 128 x 128, or 128 x 64 when Cout is not a multiple of 128).
// Workspace bytes egz_conv3x3_fwd_split needs for these arguments (0 when the tile count fills whole rounds).
EGZ_API size_t egz_conv3x3_fwd_split_ws_bytes(int B, int H, int W, int C, int K, int flags) {
    if (K % 64 != 0 || C % 32 != 0 || C <= 0) return 0;
    const int mode = x3_mode(flags), XBN = (K % 128 == 0) ? 128 : 64;
    const X3Plan p = x3_plan(x3_rows(mode, B, H, W), (C + 31) / 32 * 32, (K + 31) / 32 * 32, XBN, mode, flags);
    return (size_t)p.tail * p.nsplit * XBM * XBN * sizeof(float);
}

// x_absmax (optional, device): max |x| as a float bit pattern (egz_absmax or a gradient producer).  When given, x is
// multiplied by the power of two that brings that maximum into [2^12, 2^13) before the split and the result is divided by
// it again -- this is what lets the f16 x3 form (22 bits) carry gradients whose magnitude is 1e-3 .. 1e-9.
// flags bit 14 (0x4000): run the tiles beyond the last full round split-K (see x3_plan; measured slower than the plain
// launch on MI355X for the SP shapes -- lone tail blocks already run ~1.7x faster -- so it is opt-in).
EGZ_API int egz_conv3x3_fwd_split(const float* x, const void* wp, const float* bias, float* y, double* stat_partial,
                                  int B, int H, int W, int C, int K, int flags, int dtype, void* workspace,
                                  size_t ws_bytes, const unsigned int* x_absmax, unsigned int* absmax_out, hipStream_t st) {
    EGZ_CHECK_ARG(x && wp && y, "egz_conv3x3_fwd_split: null pointer");
    EGZ_CHECK_ARG(K % 64 == 0 && C % 32 == 0 && C > 0, "egz_conv3x3_fwd_split: needs Cout %% 64 == 0 and Cin %% 32 == 0 (got %d, %d)", K, C);
    EGZ_CHECK_ARG(dtype == 1 || dtype == 2, "egz_conv3x3_fwd_split: dtype must be 1 (f16) or 2 (bf16)");
    const int ups = flags & 3, epi = (flags >> 4) & 3;
    EGZ_CHECK_ARG(ups != 2 && epi <= 2, "egz_conv3x3_fwd_split: bad flags");
    EGZ_CHECK_ARG(!(ups || (flags & 4)) || (H % 2 == 0 && W % 2 == 0), "egz_conv3x3_fwd_split: upsampled dims must be even");
    EGZ_CHECK_ARG(epi != EPI_BIAS_STATS || stat_partial, "egz_conv3x3_fwd_split: stats epilogue needs stat_partial");
    {   // the operand fetch uses 32-bit buffer offsets: the gathered tensor (+ its one-row bias) must stay below 4 GiB
        const int hs = ((flags & 3) && !(flags & 4)) ? H / 2 : H, wsrc = ((flags & 3) && !(flags & 4)) ? W / 2 : W;
        const unsigned long long bytes = 4ull * B * hs * wsrc * C + 4ull * (wsrc + 1) * C;
        EGZ_CHECK_ARG(bytes < (1ull << 32), "egz_conv3x3_fwd_split: input of %llu bytes exceeds the 4 GiB buffer-offset range "
                      "(use the exact-f32 kernels, EGAZE_PRECISION=f32, or a smaller per-GPU batch)", bytes);
    }
    const unsigned short* w16 = static_cast<const unsigned short*>(wp);
    const float os = (dtype == 1) ? 1.f / F16_WSCALE : 1.f;
    float* ws = static_cast<float*>(workspace);
#define EGZ_MODE(T, N)                                                                                          \
    if (flags & 4) return launch_x3<T, N, UPS_DGRAD>(epi, x, w16, bias, y, stat_partial, B, H, W, C, K, os, flags, ws, ws_bytes, x_absmax, absmax_out, st); \
    if (ups == 3) return launch_x3<T, N, UPS_PHASE>(epi, x, w16, bias, y, stat_partial, B, H, W, C, K, os, flags, ws, ws_bytes, x_absmax, absmax_out, st);  \
    if (ups == 1) return launch_x3<T, N, UPS_FOLD>(epi, x, w16, bias, y, stat_partial, B, H, W, C, K, os, flags, ws, ws_bytes, x_absmax, absmax_out, st);   \
    return launch_x3<T, N, PLAIN>(epi, x, w16, bias, y, stat_partial, B, H, W, C, K, os, flags, ws, ws_bytes, x_absmax, absmax_out, st)
    if (K % 128 == 0) {
        if (dtype == 1) { EGZ_MODE(_Float16, 128); }
        EGZ_MODE(__bf16, 128);
    }
    if (dtype == 1) { EGZ_MODE(_Float16, 64); }
    EGZ_MODE(__bf16, 64);
#undef EGZ_MODE
}
