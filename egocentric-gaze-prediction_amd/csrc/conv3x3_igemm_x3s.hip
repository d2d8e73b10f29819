// Streamed-weight halo-tile 3x3 implicit GEMM in split-half arithmetic (f16 x3 / bf16 x3, see conv3x3_igemm_x3.hip):
// the default kernel of every plain (stride 1, pad 1) convolution forward and data gradient of the SP path
// (utils.py:64-76 encoders, models/model_SP.py:13-31 decoder; Conv2d + its autograd backward in the reference).
//
// What differs from conv3x3_igemm_x3h_kernel: the weight operand never touches LDS.
//   * Activations: as there -- the input halo of a compact pixel tile is fetched, split into hi / lo 16-bit halves and
//     written to LDS once per 32-channel block; the nine taps read their MFMA fragments from that image at shifted
//     addresses (patch geometry: 8 x 16 or 16 x 16 pixels; raster-run geometry for narrow images).
//   * Weights: packed ONCE per optimizer step in MFMA *fragment order* (egz_pack_w3x3_split kinds 4 / 5):
//     [slice = channel block x tap][32-column tile][k-step][hi | lo][lane][8 halves], so that the B fragment of a wave is
//     one fully coalesced 1 KB buffer_load_b128 per (k-step, plane) straight from L2 into the registers the MFMA reads.
//     No LDS-DMA, no LDS write bandwidth, no LDS read for B, and -- because nothing about B is shared through LDS -- no
//     barrier per K-slice: the only barriers left guard the activation image (one per channel block).
//     Measured motivation (profiles/r02_x3h_diag.txt): removing the weight LDS-DMA from the x3h kernel was worth 12 %,
//     DMA + barriers + halo restaging 20 %; re-placing the DMA issue between the MFMAs changed nothing.
//   * Wave layout WM x (4 / WM): every wave owns 128 pixel rows x 32 output columns (4 accumulator tiles), so the four
//     waves of a block load DISJOINT weight columns (no duplicate L2 traffic) and share the activation fragments.
//     WM = 1: tile 128 x 128 (GEMM N % 128 == 0), activation image double buffered (64 KB, 2 blocks / CU);
//     WM = 2: tile 256 x 64 (the 64-channel layers), single image of 384 slots (48 KB).
//   * B fragments are prefetched two slices ahead into a ring of three register sets (9 taps per channel block keep the
//     ring index compile-time); the activation fragments one k-step ahead, as in the x3h kernel.
//   * blockIdx -> tile map: the column tiles of one pixel tile run on the same XCD (b % 8 is the XCD), adjacent in time,
//     so the activation halo is fetched into ONE L2 instead of up to four.
#include "egz_common.h"
#include "x3_split.h"
#include <type_traits>



namespace {
using namespace x3;

// EPI_MASK_SUMS: the launch is a data gradient whose result is the gradient w.r.t. a post-ReLU activation `mask_src` (same
// layout as y): the epilogue applies the ReLU mask (mask_src > 0), accumulates the per-channel sums of the masked gradient
// (= the bias gradient of the layer below, plane 0 of the stat rows; plane 1 = 0) and the per-tile max |value| (for the
// f16 scaling of the next backward kernels) -- the separate ReLU-backward pass of that layer disappears.
enum { EPI_BIAS = 0, EPI_BIAS_RELU = 1, EPI_BIAS_STATS = 2, EPI_MASK_SUMS = 3, EPI_PARTIAL = 4, EPI_BNSUMS = 5 };
// EPI_BNSUMS (persistent narrow kernel, data gradients): the result is the gradient w.r.t. the OUTPUT of a train-mode
// BatchNorm + ReLU below (late_fusion.py:10-12: conv -> BN -> ReLU -> conv).  The epilogue also reads that layer's pre-BN conv
// output `bn_y` (same layout as y) and accumulates the two per-channel sums its BatchNorm backward needs -- sum dz and
// sum dz * xhat with dz = (bn_y * scale + shift > 0) ? v : 0 -- into the stat rows (planes 0 / 1, one row per block):
// the separate reduce pass of that BatchNorm (two reads of 205 MB tensors at 224 x 224 x 32) disappears.
// EPI_PARTIAL: split-K launch (small pixel counts: batch 1 inference, the 14 x 14 / 28 x 28 layers at small batches).  The
// channel blocks of a tile are divided over `nsplit` blocks; each writes its raw scaled accumulators to
// y[split][pixel][column] (y = the workspace) and splitk_fixup_kernel sums them in split order and applies the epilogue.
constexpr int XLD = 32;                 // 16-bit elements per LDS row: 64 B = 32 channels of one plane of one pixel
constexpr int XBK = 32;                 // channels per block of the reduction
constexpr int HPITCH = 20;              // patch geometry: halo columns per LDS grid row (18 used)

// Tile configurations (template parameter WM):
//   1: 4 waves 1 x 4, tile 128 x 128, activation image double buffered (2 blocks / CU)
//   2: 4 waves 2 x 2, tile 256 x 64 (the 64-channel layers)
//   4: 4 waves 4 x 1, tile 256 x 32, every wave 64 rows (the narrow layers of the late-fusion stack, late_fusion.py:10-12)
// (Round 2 / 3 also carried an 8-wave 256 x 128 tile and a one-wave-per-SIMD 256 x 128 tile with eight accumulator tiles per wave:
// both measured neutral-to-slower on the step -- profiles/r02_x3s_tile8.txt, r03_x3s_tile16.txt -- and were removed in round 4.)
template <int WM> struct Geo {
    static constexpr int WMM = WM;                             // waves along the pixel dimension
    static constexpr int NWN = 4 / WM;                         // waves along the columns
    static constexpr int NTHR = 64 * WMM * NWN;
    static constexpr int MR = (WM == 4) ? 2 : 4;               // 32-row groups (accumulator tiles) per wave
    static constexpr int RPW = 32 * MR;                        // pixel rows per wave
    static constexpr int BM = RPW * WMM, BN = 32 * NWN;
    static constexpr int HSLOTS = (WM == 1) ? 256 : 384, HZERO = HSLOTS - 1;
    static constexpr int NABUF = (WM == 1) ? 2 : 1;
    static constexpr int PROWS = BM / 16;                      // patch: PROWS x 16 pixels
    static constexpr int SPP = NTHR / 8;                       // halo slots staged per pass (8 threads x 4 channels per slot)
    static constexpr int NJ = HSLOTS / SPP;                    // halo slots per thread
    static constexpr int OCC = (WM == 4) ? 3 : 2;   // waves per SIMD the register budget is cut for
};

// workgroup barrier that leaves this wave's global loads (the weight prefetch ring) in flight: __syncthreads() would
// wait for vmcnt(0).  lgkmcnt(0) = this wave's LDS reads / writes are done; the "memory" clobber pins the compiler.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

enum { PLAIN = 0, UPSD = 1, UPSF = 2 };

// wq: fragment-ordered split weights (see the header); x: [B][H][W][C] fp32 (the gathered operand);
// MODE PLAIN: y [B][H][W][K] = conv3x3(x);  MODE UPSD: y [B][H/2][W/2][K] = the data gradient of [nearest x2 upsample ->
// conv3x3] w.r.t. the LOW-res input, x = the hi-res dy (see the UPSD notes in front of the image loop).
// MODE UPSF: y [B][H][W][K] = conv3x3(nearest x2 upsample(x)), x [B][H/2][W/2][C], as four phase convolutions: output pixel
// (2 yy + p, 2 xx + q) is a 2 x 2-tap convolution of the low-res image with the pre-summed weights of phase (p, q) (kind-7
// packing) -- 4/9 of the MACs.  A tile is (phase, low-res pixel tile, column tile), the phase innermost in the tile order so that
// the four phases of a pixel tile meet in one L2; the tap shifts ((a + p - 1, b + q - 1), a, b in {0, 1}) are block-uniform
// run-time values, the staged image is the plain low-res halo, and the rows go out to the strided hi-res positions.
// PRE (round 5): x holds PRE-SPLIT activations -- per 4-channel quad the 16 bytes [4 hi halves | 4 lo halves] of the f16 pair of
// (value * absmax_scale(a_absmax)), written by the pass that produced the tensor (egz_bn_relu_pool_fwd, presplit form; same
// footprint as fp32).  The staging then moves the quad into the hi / lo planes of the LDS image without touching the vector
// ALU: ~3.5 VALU per staged float gone from a kernel that sits at the chip's power limit (profiles/r05_presplit_gonogo.txt).
// The pair is bit-identical to what the split-at-staging form computes, so is the result.
// mm_out (EPI_BIAS_STATS, wide tiles): 1024 uints, ZERO-FILLED by the caller = 1024 / (2 K) sets of 2 K: order-preserving integer images
// of the per-channel max of y (slot k of a set) and of -y (slot K + k), folded in with atomic max (exact and order independent: deterministic) -- what
// egz_bn_finalize needs to bound the [BatchNorm -> ReLU] output BEFORE the pass that writes it runs.
__device__ __forceinline__ unsigned int ordered_bits(float f) {
    const unsigned int b = __float_as_uint(f);
    return b ^ (((int)b < 0) ? 0xffffffffu : 0x80000000u);
}
template <typename T, int WM, int EPI, bool PATCH, int MODE, bool PRE = false>
__global__ __launch_bounds__(Geo<WM>::NTHR, Geo<WM>::OCC) void conv3x3_igemm_x3s_kernel(
    const float* __restrict__ x, const unsigned short* __restrict__ wq, const float* __restrict__ bias,
    float* __restrict__ y, double* __restrict__ stat, int B, int H, int W, int C, int K, int Cp, int Kp, float out_scale,
    int mt, int total, const unsigned int* __restrict__ a_absmax, const float* __restrict__ mask_src,
    unsigned int* __restrict__ absmax_out, int nsplit, unsigned int* __restrict__ mm_out) {
    static_assert(!PRE || (MODE == PLAIN && IS_F16<T>), "pre-split operands: plain convolutions, f16 x3");
    using G = Geo<WM>;
    constexpr int BM = G::BM, NWN = G::NWN, BN = G::BN, HSLOTS = G::HSLOTS, HZERO = G::HZERO, NJ = G::NJ;
    constexpr int MR = G::MR, RPW = G::RPW, NTHR = G::NTHR, SPP = G::SPP;
    constexpr int APL = HSLOTS * XLD;                          // elements per plane of an activation image
    constexpr int ABUF = 2 * APL;                              // elements per image (hi + lo)
    constexpr int NIMG = (MODE == UPSD) ? 4 : 1;               // staged images per channel block
    constexpr int NT = (MODE == PLAIN) ? 9 : 4;                // taps per staged image
    constexpr int NRING = (MODE == PLAIN) ? 3 : 2;             // weight-fragment register sets (NIMG * NT % NRING == 0)
    static_assert(MODE != UPSD || WM == 1, "the upsample data gradient is built for the 128-column tile only");
    static_assert(MODE != UPSF || ((WM == 1 || WM == 2) && EPI != EPI_PARTIAL && EPI != EPI_BIAS_STATS && EPI != EPI_MASK_SUMS && EPI != EPI_BNSUMS),
                  "the upsample forward is built for the 4-wave 128- / 64-column tiles, bias / bias + ReLU epilogues");
    __shared__ __attribute__((aligned(16))) unsigned short Ah[G::NABUF * ABUF];
    __shared__ long Ro[BM];
    __shared__ double sred[(RPW < 128) ? 4 * 2 * 32 : 1];       // BN partial sums of the waves that share a 128-row stat row
    __shared__ float samax[NTHR / 64];

    const float a_scale = absmax_scale(a_absmax);
    out_scale /= a_scale;
    // the wave index is wave-uniform: telling the compiler (readfirstlane -> SGPR) keeps everything derived from it scalar --
    // in particular the soffset of the weight-fragment loads, which otherwise is wrapped in a waterfall loop per load
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / NWN, wn = wave % NWN, hl = lane >> 5, l31 = lane & 31;
    const int ntn = Kp / BN;
    // XCD-aware tile id: block b runs on XCD b % 8; give every XCD a contiguous range of tiles
    const int per = (total + 7) >> 3;
#ifdef EGZ_TILE_LOOP
    // Timing-only variant (stage (i); profiles/r06_tile_loop_gonogo.txt): the launch is capped at EGZ_GRID_CAP
    // blocks (a fixed share of the CUs x 2 resident blocks) and every block walks its XCD's tile range with that stride, so that
    // the kernels of other streams find CUs free for the whole duration of the launch instead of queueing behind it.
    for (unsigned vb = blockIdx.x; vb < (unsigned)(per * 8); vb += gridDim.x) {
    if (vb != blockIdx.x) __syncthreads();
    const int gts = (int)(vb & 7) * per + (int)(vb >> 3);
    if (gts >= total) continue;
#else
    const int gts = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);      // total = tiles x splits
    if (gts >= total) return;
#endif
    const int split = (EPI == EPI_PARTIAL) ? gts % nsplit : 0;
    const int gt = (EPI == EPI_PARTIAL) ? gts / nsplit : gts;
    const int phase = (MODE == UPSF) ? (gt & 3) : 0, ph_p = phase >> 1, ph_q = phase & 1;    // UPSF: block-uniform
    const int gtl = (MODE == UPSF) ? (gt >> 2) : gt;
    const int tile_n = gtl % ntn, tile_m = gtl / ntn;
    const int n0 = tile_n * BN;
    const int Ho = (MODE != PLAIN) ? (H >> 1) : H, Wo = (MODE != PLAIN) ? (W >> 1) : W;   // image the tile's pixel rows live in
    const long HW = (long)Ho * Wo, M = (long)B * HW;
    const long m0 = (long)tile_m * BM;
    const int pw = Wo >> 4, ppi = (Ho / G::PROWS) * pw;         // patches per row / per image
    const int b0 = PATCH ? tile_m / ppi : 0;
    const int y0 = PATCH ? ((tile_m - b0 * ppi) / pw) * G::PROWS : 0, x0 = PATCH ? ((tile_m - b0 * ppi) % pw) * 16 : 0;

    for (int i = tid; i < BM; i += NTHR) {
        long off = -1;
        if (MODE == UPSF) {                                    // low-res pixel (b, yy, xx) -> hi-res (2 yy + p, 2 xx + q)
            int ob = b0, yy = y0 + (i >> 4), xx = x0 + (i & 15);
            bool ok = true;
            if (!PATCH) {
                ok = m0 + i < M;
                ob = (int)((m0 + i) / HW);
                const int rem = (int)((m0 + i) - (long)ob * HW);
                yy = rem / Wo;
                xx = rem - yy * Wo;
            }
            if (ok) off = (((long)ob * H + 2 * yy + ph_p) * W + 2 * xx + ph_q) * K;
        } else if (PATCH) off = (((long)b0 * Ho + y0 + (i >> 4)) * Wo + x0 + (i & 15)) * K;
        else if (m0 + i < M) off = (m0 + i) * K;
        Ro[i] = off;
    }

    // ---- activation halo staging map: slot q = (tid >> 3) + SPP j holds 4 channels (tid & 7) of one pixel of the staged
    // image.  PLAIN: the image is x itself.  UPSD: the image is one of the four polyphase components of the hi-res dy,
    // D_pq[yy][xx] = dy[2 yy + p][2 xx + q] -- an Ho x Wo image whose pixel (yy, xx) sits at a byte offset that is affine
    // in (p, q): the map below is built for (p, q) = (0, 0) and a component is selected by ONE scalar offset.
    const __amdgpu_buffer_rsrc_t a_rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(x), 0, (int)((unsigned)B * ((MODE == UPSF) ? Ho * Wo : H * W) * C * 4u), 0x00020000);
    const int a_c4 = tid & 7;
    // The single-image configurations (64- and 32-column tiles) run at the register limit: their LDS staging addresses are
    // recomputed at each use from an opaque copy of the slot base (12 registers less; ~10 integer ops per staged slot, twelve
    // slots per channel block) instead of living in registers across the MFMA loop.
    constexpr bool LDS_RECOMP = (G::NABUF == 1);
    unsigned a_vo[NJ];
    int a_lds[LDS_RECOMP ? 1 : NJ];
    auto lds_slot = [&](const int j) -> int {
        if constexpr (!LDS_RECOMP) {
            return a_lds[j];
        } else {
            int tq = tid >> 3;
            asm volatile("" : "+v"(tq));                       // keeps the compiler from hoisting the twelve addresses again
            const int q = tq + SPP * j;
            const int col = PATCH ? q - (q / HPITCH) * HPITCH : q;
            return q * XLD + (((a_c4 >> 1) ^ ((col >> 2) & 3)) << 3) + (a_c4 & 1) * 4;
        }
    };
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int q = (tid >> 3) + SPP * j;
        long pix = -1;                                         // source pixel index in x (class (0, 0) for UPSD)
        int col = q;
        int iy = -1, ix = -1, ib = 0;
        if (PATCH) {
            const int hy = q / HPITCH, hx = q - hy * HPITCH;
            iy = y0 - 1 + hy;
            ix = x0 - 1 + hx;
            ib = b0;
            col = hx;
            if (!(hy < G::PROWS + 2 && hx < 18)) iy = -1;
        } else {
            const long g = m0 - Wo - 1 + q;
            if (q < BM + 2 * Wo + 2 && g >= 0 && g < M) {
                ib = (int)(g / HW);
                const int rem = (int)(g - (long)ib * HW);
                iy = rem / Wo;
                ix = rem - iy * Wo;
            }
        }
        if ((unsigned)iy < (unsigned)Ho && (unsigned)ix < (unsigned)Wo)
            pix = (MODE == UPSD) ? ((long)ib * H + 2 * iy) * W + 2 * ix : ((long)ib * Ho + iy) * Wo + ix;
        if (a_c4 * 4 >= C) pix = -1;                           // C < 32 (one zero-padded channel block): channels past C read zeros
        a_vo[j] = (pix >= 0) ? (unsigned)((pix * C + a_c4 * 4) * 4) : 0xFFFFFFFFu;
        if constexpr (!LDS_RECOMP) a_lds[j] = q * XLD + (((a_c4 >> 1) ^ ((col >> 2) & 3)) << 3) + (a_c4 & 1) * 4;
    }

    // ---- weight fragments: lane-linear 1 KB pieces, [slice][ntile32][ks][plane][lane][8 halves]
    const int nt32 = Kp >> 5, ncb = Cp / XBK, S = ncb * NIMG * NT;
    // this block's channel blocks [c_lo, c_hi) (all of them unless the launch is split-K) and slice range [., S_hi)
    const int c_lo = (EPI == EPI_PARTIAL) ? (split * ncb) / nsplit : 0;
    const int c_hi = (EPI == EPI_PARTIAL) ? ((split + 1) * ncb) / nsplit : ncb;
    const int S_lo = c_lo * NIMG * NT, S_hi = c_hi * NIMG * NT;
    // (UPSF: the four phases' packings follow each other, S slices each)
    const __amdgpu_buffer_rsrc_t b_rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned short*>(wq) + (long)phase * S * nt32 * 2048, 0, (int)((unsigned)S * nt32 * 4096u), 0x00020000);
    const unsigned b_vo = (unsigned)lane * 16u;
    const unsigned b_tile = (unsigned)(tile_n * NWN + wn) * 4096u;
    u32x4 bq[NRING][4];                                         // [ring][ks * 2 + plane]
    auto gload_b = [&](int s, const int ring) {
        const unsigned so = (unsigned)s * (unsigned)nt32 * 4096u + b_tile;
#pragma unroll
        for (int c = 0; c < 4; ++c) bq[ring][c] = __builtin_amdgcn_raw_buffer_load_b128(b_rs, b_vo + c * 1024, so, 0);
    };

    // ---- activation fragment addresses (bytes, plane 0, k-step 0, image 0) of the wave's four 32-row groups for one tap
    // shift t9 = (dy + 1) * 3 + (dx + 1).
    // patch: one table entry per shift, the row groups are 2 grid rows = 2560 B apart (immediate offsets);
    // run: slot = row + Wo + 1 + shift; the 32-slot row groups share the swizzle key, so one address + 2048 B steps,
    // and a row whose shifted pixel falls outside the image (per-group validity mask) reads the all-zero slot instead.
    // The per-tap value is recomputed from an opaque base every slice: hoisting all 36 of them costs more registers than
    // the kernel has.
    // (dy + 1, dx + 1) of table entry t: the nine shifts of a 3 x 3 window, or -- UPSF -- the four taps of this block's phase
    auto tap_dy1 = [&](const int t) -> int { return (MODE == UPSF) ? (t >> 1) + ph_p : t / 3; };
    auto tap_dx1 = [&](const int t) -> int { return (MODE == UPSF) ? (t & 1) + ph_q : t % 3; };
    constexpr int NTAB = (MODE == UPSF) ? 4 : 9;
    int fa9[9];
    int sl0 = wm * RPW + l31 + Wo + 1;
    unsigned vmask[MR];
#pragma unroll
    for (int mr = 0; mr < MR; ++mr) vmask[mr] = 0;
    if (PATCH) {
        const int i = wm * RPW + l31;
#pragma unroll
        for (int t = 0; t < NTAB; ++t) {
            const int dy = tap_dy1(t) - 1, dx = tap_dx1(t) - 1;
            const int slot = ((i >> 4) + 1 + dy) * HPITCH + (i & 15) + 1 + dx, col = (i & 15) + 1 + dx;
            fa9[t] = (slot * XLD + ((hl ^ ((col >> 2) & 3)) << 3)) * 2;
        }
    } else {
#pragma unroll
        for (int mr = 0; mr < MR; ++mr) {
            const long m = m0 + wm * RPW + mr * 32 + l31;
            const int rem = (int)(m % HW), py = rem / Wo, px = rem - py * Wo;
            unsigned mk = 0;
#pragma unroll
            for (int t = 0; t < NTAB; ++t) {
                const int dy = tap_dy1(t) - 1, dx = tap_dx1(t) - 1;
                mk |= (m < M && (unsigned)(py + dy) < (unsigned)Ho && (unsigned)(px + dx) < (unsigned)Wo) ? (1u << t) : 0u;
            }
            vmask[mr] = mk;
        }
    }
    constexpr int MRSTEP = PATCH ? 2 * HPITCH * XLD * 2 : 32 * XLD * 2;      // bytes between the wave's row groups
    // addresses of the four row groups for shift t9 in image `abuf` -> fa[0..3]
    auto tap_addr = [&](const int t9, const int abuf, int* fa) {
        if (PATCH) {
            int f;
            if constexpr (LDS_RECOMP) {                        // (register-limited configurations: the table entry is recomputed)
                int i = wm * RPW + l31;
                asm volatile("" : "+v"(i));
                const int dy = tap_dy1(t9) - 1, dx = tap_dx1(t9) - 1;
                const int col = (i & 15) + 1 + dx, slot = ((i >> 4) + 1 + dy) * HPITCH + col;
                f = (slot * XLD + ((hl ^ ((col >> 2) & 3)) << 3)) * 2 + abuf * (ABUF * 2);
            } else {
                f = fa9[t9] + abuf * (ABUF * 2);
            }
            asm volatile("" : "+v"(f));
#pragma unroll
            for (int mr = 0; mr < MR; ++mr) fa[mr] = f + mr * MRSTEP;
        } else {
            int base = sl0;
            asm volatile("" : "+v"(base));
            const int dy = tap_dy1(t9) - 1, dx = tap_dx1(t9) - 1;
            const int slot = base + dy * Wo + dx;
            const int a = slot * (XLD * 2) + ((hl ^ ((slot >> 2) & 3)) << 4) + abuf * (ABUF * 2);
            const int z = HZERO * (XLD * 2) + (hl << 4) + abuf * (ABUF * 2);          // the all-zero slot
#pragma unroll
            for (int mr = 0; mr < MR; ++mr) fa[mr] = ((vmask[mr] >> t9) & 1u) ? a + mr * MRSTEP : z;
        }
    };
    // UPSD: image `img` = polyphase component (p, q) = (img >> 1, img & 1); its tap (a, b) = (tap >> 1, tap & 1) is dy row
    // 2 (yy + a - p) + p: the shift in the component image is (a - p, b - q) -- the same nine shifts as a plain conv.
    auto shift_of = [&](const int img, const int tap) -> int {
        if (MODE == UPSD) return ((tap >> 1) - (img >> 1) + 1) * 3 + ((tap & 1) - (img & 1) + 1);
        return tap;
    };

    // staging registers: one half of an image at a time (PLAIN: its nine taps leave two taps between a fetch and its split),
    // or -- UPSD, where an image only lasts four taps -- both halves in flight at once, fetched at tap 0 and split at taps 2
    // and 3: one tap of distance (~0.4 us per wave) did not cover the HBM latency of the strided polyphase fetch, and the
    // four upsample data gradients ran at half the rate of the plain convolutions (profiles/r02_conv_microbench.txt)
    constexpr bool FINE = (G::NABUF == 2);   // instruction-level interleave of a slice (main loop)
    // (the interleaved loop fetches half 1 of an upsample-dgrad image in the same tap that splits half 0: two register sets)
    constexpr int RAOFF = (NT == 4 && FINE) ? NJ / 2 : 0;
    f32x4 ra[NJ / 2 + RAOFF];
    auto gload_a = [&](int cblk, const int img, const int half) {
        const unsigned so = (unsigned)(cblk * XBK * 4) +
                            ((MODE == UPSD) ? (unsigned)(((img >> 1) * W + (img & 1)) * C * 4) : 0u);
#pragma unroll
        for (int j = 0; j < NJ / 2; ++j)
            ra[half * RAOFF + j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(a_rs, a_vo[half * (NJ / 2) + j], so, 0));
    };
    // one staged quad -> its hi / lo halves (PRE: the quad already holds them)
    auto split_q = [&](const f32x4 v, u32x2& hi, u32x2& lo) {
        if constexpr (PRE) {
            const u32x4 b = __builtin_bit_cast(u32x4, v);
            hi = u32x2{b[0], b[1]};
            lo = u32x2{b[2], b[3]};
            if constexpr (egz_drop_alo<T>::value) hi = Half<T>::pair_rne(hi, lo);
        } else if constexpr (egz_drop_alo<T>::value) {
            hi = Half<T>::rne4s(v, a_scale);
        } else {
            Half<T>::split4s(v, a_scale, hi, lo);
        }
    };
    auto lstore_a = [&](const int abuf, const int half) {
#pragma unroll
        for (int j = 0; j < NJ / 2; ++j) {
            u32x2 hi, lo;
            split_q(ra[half * RAOFF + j], hi, lo);
            unsigned short* d = Ah + abuf * ABUF + lds_slot(half * (NJ / 2) + j);
            *reinterpret_cast<u32x2*>(d) = hi;
            if constexpr (!egz_drop_alo<T>::value) *reinterpret_cast<u32x2*>(d + APL) = lo;
        }
    };

    f32x16 acc[MR];
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    const char* Ab = reinterpret_cast<const char*>(Ah);
    u32x4 ah0[MR], al0[MR];                                   // k-step 0 fragments of the tap about to run
    int cur[MR];                                              // their addresses (k-step 1 = address ^ 32)
    auto read_a0 = [&](const int t9, const int abuf) {
        tap_addr(t9, abuf, cur);
#pragma unroll
        for (int mr = 0; mr < MR; ++mr) {
            ah0[mr] = *reinterpret_cast<const u32x4*>(Ab + cur[mr]);
            al0[mr] = *reinterpret_cast<const u32x4*>(Ab + APL * 2 + cur[mr]);
        }
    };
    auto mfma12 = [&](const u32x4* ah, const u32x4* al, const u32x4 bh, const u32x4 bl) {
        // small terms first; the three products of one accumulator are four MFMAs apart (no dependent back-to-back issue)
#pragma unroll
        for (int term = 0; term < 3; ++term)
#pragma unroll
            for (int mr = 0; mr < MR; ++mr)
                if (!(egz_drop_alo<T>::value && term == 0)) acc[mr] = Half<T>::mfma(term == 0 ? al[mr] : ah[mr], term == 1 ? bl : bh, acc[mr]);
    };

    // ---- prologue
    gload_b(S_lo, 0);
    if (NRING > 2) gload_b(S_hi - S_lo > 1 ? S_lo + 1 : S_lo, 1);
    // both halves of the first image in flight together (the second into registers that are dead until the main loop): one
    // exposed global round trip per tile instead of two -- on the 64-channel layers, whose tiles only run 18 K-slices, the
    // second one was a tenth of a tile's time
    f32x4 rp[NJ / 2];
    {
        const unsigned so0 = (unsigned)(c_lo * XBK * 4);
        gload_a(c_lo, 0, 0);
#pragma unroll
        for (int j = 0; j < NJ / 2; ++j)
            rp[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(a_rs, a_vo[NJ / 2 + j], so0, 0));
        lstore_a(0, 0);
#pragma unroll
        for (int j = 0; j < NJ / 2; ++j) {
            u32x2 hi, lo;
            split_q(rp[j], hi, lo);
            unsigned short* d = Ah + lds_slot(NJ / 2 + j);
            *reinterpret_cast<u32x2*>(d) = hi;
            if constexpr (!egz_drop_alo<T>::value) *reinterpret_cast<u32x2*>(d + APL) = lo;
        }
    }
    lds_barrier();
    read_a0(shift_of(0, 0), 0);

    // staging schedule of the NEXT image inside the NT taps of the current one (double-buffered images)
    // (fetching both halves of an upsample-gradient image at tap 0 into two register sets measured no gain: r03_upsd_pipe_ab.txt)
    constexpr int G0 = (NT == 9) ? 1 : 0, L0 = (NT == 9) ? 3 : 1, G1 = (NT == 9) ? 4 : 1, L1 = (NT == 9) ? 6 : 2;
    for (int c = c_lo; c < c_hi; ++c) {
#pragma unroll
        for (int img = 0; img < NIMG; ++img) {
            const int abuf = (G::NABUF == 2) ? ((NIMG == 1) ? ((c - c_lo) & 1) : (img & 1)) : 0;
            const bool more = (img + 1 < NIMG) || (c + 1 < c_hi);       // block-uniform
            const int nimg = (img + 1 < NIMG) ? img + 1 : 0;
            const int ncblk = (img + 1 < NIMG) ? c : c + 1;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int s = (c * NIMG + img) * NT + t, ring = (img * NT + t) % NRING;
                if constexpr (FINE) {
                    // ---- one wave per SIMD: nothing else fills the matrix pipe while this wave fetches, so every non-MFMA
                    // instruction of the slice is issued in the ~24 free issue cycles behind ONE MFMA (sched_barrier after each
                    // pair pins the order).  Group 0 = the 3 MR MFMAs of k-step 0, carrying the k-step-1 fragment reads, the
                    // weight prefetch and the halo fetch; group 1 = k-step 1, carrying the next tap's fragment reads, the halo
                    // split + LDS stores and (last tap of an image) the barrier.
                    constexpr int NM = 3 * MR, NH = NJ / 2;
                    const int ringn = (img * NT + t + NRING - 1) % NRING;
                    const int sn = s + NRING - 1 < S_hi ? s + NRING - 1 : S_hi - 1;
                    const unsigned bso = (unsigned)sn * (unsigned)nt32 * 4096u + b_tile;
                    const unsigned aso = (unsigned)(ncblk * XBK * 4) +
                                         ((MODE == UPSD) ? (unsigned)(((nimg >> 1) * W + (nimg & 1)) * C * 4) : 0u);
                    const int ghalf = (t == G0) ? 0 : 1;
                    const bool gl = more && (t == G0 || t == G1);
                    u32x4 ah1[MR], al1[MR];
#pragma unroll
                    for (int i = 0; i < NM; ++i) {
                        const int term = i / MR, mr = i % MR;
                        if (!(egz_drop_alo<T>::value && term == 0)) acc[mr] = Half<T>::mfma(term == 0 ? al0[mr] : ah0[mr], term == 1 ? bq[ring][1] : bq[ring][0], acc[mr]);
                        if (i < MR) {
                            ah1[i] = *reinterpret_cast<const u32x4*>(Ab + (cur[i] ^ 32));
                            al1[i] = *reinterpret_cast<const u32x4*>(Ab + APL * 2 + (cur[i] ^ 32));
                        } else if (i < MR + 4) {
                            bq[ringn][i - MR] = __builtin_amdgcn_raw_buffer_load_b128(b_rs, b_vo + (i - MR) * 1024, bso, 0);
                        } else if (gl && i - MR - 4 < NH) {
                            const int j = i - MR - 4;
                            ra[ghalf * RAOFF + j] = __builtin_bit_cast(
                                f32x4, __builtin_amdgcn_raw_buffer_load_b128(a_rs, a_vo[ghalf * NH + j], aso, 0));
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    static_assert(!FINE || (MR + 4 + NJ / 2 <= 3 * MR), "group 0 has too few MFMAs for its side work");
                    static_assert(!FINE || (L1 < NT - 1 && L0 < NT - 1), "the last tap of an image carries the barrier, not a split");
                    const int lhalf = (t == L0) ? 0 : 1;
                    const bool ls = more && (t == L0 || t == L1);
                    const bool last = (t == NT - 1);
                    if (!last) tap_addr(shift_of(img, t + 1), abuf, cur);
#pragma unroll
                    for (int i = 0; i < NM; ++i) {
                        const int term = i / MR, mr = i % MR;
                        if (!(egz_drop_alo<T>::value && term == 0)) acc[mr] = Half<T>::mfma(term == 0 ? al1[mr] : ah1[mr], term == 1 ? bq[ring][3] : bq[ring][2], acc[mr]);
                        if (!last && i < MR) {
                            ah0[i] = *reinterpret_cast<const u32x4*>(Ab + cur[i]);
                            al0[i] = *reinterpret_cast<const u32x4*>(Ab + APL * 2 + cur[i]);
                        } else if (ls && i >= MR && i - MR < NH) {
                            const int j = i - MR;
                            u32x2 hi, lo;
                            split_q(ra[lhalf * RAOFF + j], hi, lo);
                            unsigned short* d = Ah + (abuf ^ 1) * ABUF + lds_slot(lhalf * NH + j);
                            *reinterpret_cast<u32x2*>(d) = hi;
                            if constexpr (!egz_drop_alo<T>::value) *reinterpret_cast<u32x2*>(d + APL) = lo;
                        } else if (last && more) {
                            // the image boundary: every wave has staged its share (taps L0, L1) and read its last fragments
                            // (group 0); the first fragments of the next image ride on the remaining MFMAs of this group
                            if (i == 1) {
                                lds_barrier();
                                tap_addr(shift_of(nimg, 0), abuf ^ 1, cur);
                            }
                            if (i >= 2 && i - 2 < MR) {
                                ah0[i - 2] = *reinterpret_cast<const u32x4*>(Ab + cur[i - 2]);
                                al0[i - 2] = *reinterpret_cast<const u32x4*>(Ab + APL * 2 + cur[i - 2]);
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    continue;
                }
                // the set being refilled was last read by slice s - 1
                gload_b(s + NRING - 1 < S_hi ? s + NRING - 1 : S_hi - 1, (img * NT + t + NRING - 1) % NRING);
                u32x4 ah1[MR], al1[MR];
#pragma unroll
                for (int mr = 0; mr < MR; ++mr) {
                    ah1[mr] = *reinterpret_cast<const u32x4*>(Ab + (cur[mr] ^ 32));
                    al1[mr] = *reinterpret_cast<const u32x4*>(Ab + APL * 2 + (cur[mr] ^ 32));
                }
                __builtin_amdgcn_sched_barrier(0);
                mfma12(ah0, al0, bq[ring][0], bq[ring][1]);
                __builtin_amdgcn_sched_barrier(0);
                if (G::NABUF == 2) {
                    // the next image goes into the OTHER buffer while this one is being multiplied
                    if (more) {
                        if (t == L0) lstore_a(abuf ^ 1, 0);
                        if (t == G0) gload_a(ncblk, nimg, 0);
                        if (t == L1) lstore_a(abuf ^ 1, 1);
                        if (t == G1) gload_a(ncblk, nimg, 1);
                    }
                    if (t < NT - 1) {
                        read_a0(shift_of(img, t + 1), abuf);
                    } else if (more) {
                        lds_barrier();                         // every wave has staged its share and is done reading
                        read_a0(shift_of(nimg, 0), abuf ^ 1);
                    }
                } else {
                    if (more && t == NT - 3) gload_a(ncblk, nimg, 0);
                    if (t < NT - 1) read_a0(shift_of(img, t + 1), 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                mfma12(ah1, al1, bq[ring][2], bq[ring][3]);
                if (G::NABUF == 1 && t == NT - 1 && more) {
                    // single image: everyone finishes reading, then the two halves are restaged in place.  (Fetching the second
                    // half into spare registers half a tap early, so that its round trip overlaps the first half's, was tried in
                    // round 3: the 64-column configuration sits at 256 VGPRs and the extra live range spilled inside the loop --
                    // enc3 forward 520 -> 595 us.)
                    lds_barrier();
                    lstore_a(0, 0);
                    gload_a(ncblk, nimg, 1);
                    lstore_a(0, 1);
                    lds_barrier();
                    read_a0(shift_of(nimg, 0), 0);
                }
            }
        }
    }

    // ---- epilogue: every wave owns RPW rows x 32 columns; out_scale undoes the f16 weight pre-scaling exactly
    const int col = n0 + wn * 32 + l31;
    const bool nok = col < K;
    const float bz = (EPI != EPI_BNSUMS && bias && nok) ? bias[col] : 0.f;
    // EPI_BNSUMS (data gradient w.r.t. the output of a train-mode [BatchNorm -> ReLU]): `bias` carries that BatchNorm's 4 x K
    // coefficient rows (mean, 1/std, scale, shift) and mask_src its pre-BN conv output; the epilogue also accumulates the two
    // per-channel sums of the BatchNorm backward (see the enum) -- the reduce pass of that layer disappears.
    const float cmu = (EPI == EPI_BNSUMS && nok) ? bias[col] : 0.f, cis = (EPI == EPI_BNSUMS && nok) ? bias[K + col] : 0.f;
    const float csc = (EPI == EPI_BNSUMS && nok) ? bias[2 * K + col] : 0.f, csh = (EPI == EPI_BNSUMS && nok) ? bias[3 * K + col] : 0.f;
    constexpr int SR = (RPW > 128) ? RPW / 128 : 1;            // 128-row stat rows a wave owns (2 on the 256-row waves)
    double s1 = 0.0, s2 = 0.0, s1b[SR], s2b[SR];
    float amx = 0.f;
    float cmx = -INFINITY, cmn = INFINITY;                      // mm_out: this lane's column max / min of y
    // mm_out holds 1024 / (2 K) slot sets of 2 K uints (K = 64: eight): a tile folds into set (pixel tile % sets), so the
    // blocks of a round spread their atomics over several lines per channel; the current slot values are requested HERE, in
    // front of the store loop, and compared behind it -- read at the end they cost every tile one exposed L2 round trip
    // (a tenth of a 64-channel tile's time).
    unsigned int* mm_p = nullptr;
    unsigned int seen_mx = 0, seen_mn = 0;
    if (EPI == EPI_BIAS_STATS && mm_out) {                      // block-uniform
        const int nsl = 512 / K > 0 ? 512 / K : 1;
        mm_p = mm_out + (tile_m % nsl) * 2 * K + (nok ? col : 0);
        if (hl == 0 && nok) {
            seen_mx = __hip_atomic_load(mm_p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            seen_mn = __hip_atomic_load(mm_p + K, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // The result goes out through BUFFER stores whose per-lane byte offset is out of range for rows / columns that do not exist
    // (dropped by the hardware): no per-lane branch around a store.  With `if (valid) y[...] = v` every store sat in its own
    // basic block and the wait-count pass put s_waitcnt vmcnt(0) in front of each one -- 32-64 stores per wave, each waiting
    // for the previous one to be acknowledged.  (The output is < 4 GiB: egz_conv3x3_streamed_ok.)
    constexpr bool BUFST = EPI != EPI_PARTIAL;
    constexpr unsigned OUTMUL = (MODE == UPSF) ? 4u : 1u;       // UPSF: y is the hi-res image, four output pixels per tile-image pixel
    const __amdgpu_buffer_rsrc_t y_rs = __builtin_amdgcn_make_buffer_rsrc(y, 0, BUFST ? (int)(OUTMUL * (unsigned)M * (unsigned)K * 4u) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t mk_rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>((EPI == EPI_MASK_SUMS || EPI == EPI_BNSUMS) ? mask_src : y), 0, BUFST ? (int)((unsigned)M * (unsigned)K * 4u) : 0, 0x00020000);
#pragma unroll
    for (int mr = 0; mr < MR; ++mr) {
        if (SR > 1 && (mr & 3) == 0 && mr) {                    // a 128-row group is complete: park its sums
            s1b[mr / 4 - 1] = s1;
            s2b[mr / 4 - 1] = s2;
            s1 = 0.0;
            s2 = 0.0;
        }
        unsigned mko[16];                                       // EPI_MASK_SUMS: the 16 mask values of this row group, requested
        float mkv[16];                                          // together (branch-free buffer loads), consumed below
        if (BUFST && (EPI == EPI_MASK_SUMS || EPI == EPI_BNSUMS)) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long off = Ro[wm * RPW + mr * 32 + egz_acc_row(r, lane)];
                mko[r] = (off >= 0 && nok) ? (unsigned)(off + col) * 4u : 0xFFFFFFFFu;
                mkv[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(mk_rs, mko[r], 0, 0));
            }
        }
        if (BUFST && EPI == EPI_BNSUMS) {                      // 16 rows in fp32, then one fp64 add per row group
            float q1 = 0.f, q2 = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = acc[mr][r] * out_scale;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), y_rs, mko[r], 0, 0);
                const float yp = mkv[r];
                const float dz = (mko[r] != 0xFFFFFFFFu && yp * csc + csh > 0.f) ? v : 0.f;
                amx = fmaxf(amx, fabsf(v));                      // (rows / columns that do not exist hold zeros)
                q1 += dz;
                q2 += dz * ((yp - cmu) * cis);
            }
            s1 += (double)q1;
            s2 += (double)q2;
            continue;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const long off = Ro[wm * RPW + mr * 32 + egz_acc_row(r, lane)];
            if (BUFST && EPI == EPI_MASK_SUMS) {                // (out-of-range lanes read 0: masked, not stored, not summed)
                const float v = (mkv[r] > 0.f) ? acc[mr][r] * out_scale : 0.f;
                amx = fmaxf(amx, fabsf(v));
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), y_rs, mko[r], 0, 0);
                s1 += (double)v;
                continue;
            }
            if (EPI == EPI_PARTIAL) {                          // raw partial sums of this split; epilogue in the fix-up pass
                if (off >= 0 && nok) y[(long)split * M * K + off + col] = acc[mr][r] * out_scale;
                continue;
            }
            if (BUFST) {
                const bool ok = off >= 0 && nok;
                float v = acc[mr][r] * out_scale + bz;
                if (EPI == EPI_BIAS_RELU) v = fmaxf(v, 0.f);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), y_rs,
                                                      ok ? (unsigned)(off + col) * 4u : 0xFFFFFFFFu, 0, 0);
                const float vs = ok ? v : 0.f;
                if (EPI == EPI_BIAS_RELU) amx = fmaxf(amx, vs);                  // (post-ReLU: vs >= 0)
                if (EPI == EPI_BIAS) amx = fmaxf(amx, fabsf(vs));                // (data gradients: max |dx| bounds the next BatchNorm backward)
                if (EPI == EPI_BIAS_STATS) {                   // (fp64 per element: the variance is a difference of these two
                    s1 += (double)vs;                          //  sums, and fp32 partial sums over 16 rows already cost the
                    s2 += (double)vs * (double)vs;             //  gradients their fp32-class accuracy -- test_model_sp_grads_vs_fp64)
                    cmx = fmaxf(cmx, v);                       // (rows that do not exist: only in a raster run's last tile, redone below;
                    cmn = fminf(cmn, v);                       //  the column test is at the commit)
                }
                continue;
            }
            if (off >= 0 && nok) {
                float v = acc[mr][r] * out_scale + bz;
                if (EPI == EPI_BIAS_RELU) v = fmaxf(v, 0.f);
                if (EPI == EPI_MASK_SUMS) v = (mask_src[off + col] > 0.f) ? v : 0.f;
                if (EPI == EPI_MASK_SUMS || EPI == EPI_BIAS_RELU) amx = fmaxf(amx, fabsf(v));
                y[off + col] = v;
                if (EPI == EPI_BIAS_STATS || EPI == EPI_MASK_SUMS) s1 += (double)v;
                if (EPI == EPI_BIAS_STATS) s2 += (double)v * (double)v;
            }
        }
    }
    if ((EPI == EPI_MASK_SUMS || EPI == EPI_BIAS_RELU || EPI == EPI_BIAS || EPI == EPI_BNSUMS) && absmax_out) {   // block-uniform
        // per-tile max |value| -> the abs-max buffer (egz_common.h: one atomic max per tile at most).  EPI_BIAS_RELU:
        // the result is a post-ReLU activation that the next convolution splits into f16 halves -- its abs-max scales that split
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) amx = fmaxf(amx, __shfl_xor(amx, o));
        if (lane == 0) samax[wave] = amx;
        lds_barrier();
        if (tid == 0) {
            float m = samax[0];
#pragma unroll
            for (int w = 1; w < NTHR / 64; ++w) m = fmaxf(m, samax[w]);
            absmax_commit(absmax_out, (unsigned)gt, m);
        }
    }
    if (EPI == EPI_BIAS_STATS && mm_out) {                      // block-uniform
        if (!PATCH && m0 + BM > M) {
            // the last tile of a raster run has rows past the image: the maxima above saw their (bias-only) values -- take them
            // again over the rows that exist (a per-element select in the store loop cost 17 spilled registers)
            cmx = -INFINITY;
            cmn = INFINITY;
#pragma unroll
            for (int mr = 0; mr < MR; ++mr)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const bool ok = Ro[wm * RPW + mr * 32 + egz_acc_row(r, lane)] >= 0;
                    const float v = acc[mr][r] * out_scale + bz;
                    cmx = fmaxf(cmx, ok ? v : -INFINITY);
                    cmn = fminf(cmn, ok ? v : INFINITY);
                }
        }
        cmx = fmaxf(cmx, __shfl_xor(cmx, 32));
        cmn = fminf(cmn, __shfl_xor(cmn, 32));
        if (hl == 0 && nok) {
            // (seen_mx / seen_mn were requested before the store loop: an atomic is issued only where this tile raises the slot)
            const unsigned int umx = ordered_bits(cmx), umn = ordered_bits(-cmn);
            if (umx > seen_mx) __hip_atomic_fetch_max(mm_p, umx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (umn > seen_mn) __hip_atomic_fetch_max(mm_p + K, umn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (EPI == EPI_BIAS_STATS || EPI == EPI_MASK_SUMS || EPI == EPI_BNSUMS) {
        // one partial row per 128 pixel rows (the granularity egz_conv3x3_stat_rows promises)
        s1 += __shfl_xor(s1, 32);
        s2 += __shfl_xor(s2, 32);
        if (RPW > 128) {                                       // the wave's own SR rows
            s1b[SR - 1] = s1;
            s2b[SR - 1] = s2;
#pragma unroll
            for (int q = 0; q + 1 < SR; ++q) {
                s1b[q] += __shfl_xor(s1b[q], 32);
                s2b[q] += __shfl_xor(s2b[q], 32);
            }
#pragma unroll
            for (int q = 0; q < SR; ++q) {
                const long srow = ((long)tile_m * G::WMM + wm) * SR + q;
                if (hl == 0 && nok && srow * 128 < M) {
                    stat[(srow * 2 + 0) * K + col] = s1b[q];
                    stat[(srow * 2 + 1) * K + col] = s2b[q];
                }
            }
        } else if (RPW == 128) {                               // the wave's own row
            const long srow = (long)tile_m * G::WMM + wm;
            if (hl == 0 && nok && srow * 128 < M) {
                stat[(srow * 2 + 0) * K + col] = s1;
                stat[(srow * 2 + 1) * K + col] = s2;
            }
        } else {                                               // two 64-row waves per stat row: combine through LDS
            if (hl == 0) {
                sred[(wave * 2 + 0) * 32 + l31] = s1;
                sred[(wave * 2 + 1) * 32 + l31] = s2;
            }
            lds_barrier();
            const long srow = (long)tile_m * (BM / 128) + (wm >> 1);
            if ((wm & 1) == 0 && hl == 0 && nok && srow * 128 < M) {
                stat[(srow * 2 + 0) * K + col] = s1 + sred[((wave + NWN) * 2 + 0) * 32 + l31];
                stat[(srow * 2 + 1) * K + col] = s2 + sred[((wave + NWN) * 2 + 1) * 32 + l31];
            }
        }
    }
#ifdef EGZ_TILE_LOOP
    }
#endif
}

// pre-summed 3x3 weights of the upsample-fused forms (same definition as weff9 in conv3x3_igemm_x3.hip)
__device__ __forceinline__ float weff9s(const float* __restrict__ w9, int py, int a, int px, int b) {
    const int rlo = (py == 0) ? (a == 0 ? 0 : 1) : (a == 0 ? 0 : 2), rhi = (py == 0) ? (a == 0 ? 0 : 2) : (a == 0 ? 1 : 2);
    const int slo = (px == 0) ? (b == 0 ? 0 : 1) : (b == 0 ? 0 : 2), shi = (px == 0) ? (b == 0 ? 0 : 2) : (b == 0 ? 1 : 2);
    float s = 0.f;
    for (int r = rlo; r <= rhi; ++r)
        for (int q = slo; q <= shi; ++q) s += w9[r * 3 + q];
    return s;
}

// value of GEMM-view weight element (slice-in-block si) from the nine taps w9 of its (output channel, input channel) pair:
// kind 4 = forward (si = tap), kind 5 = data gradient (taps flipped), kind 6 = data gradient of [upsample x2 -> conv] w.r.t.
// the low-res input: si = component (p, q) * 4 + tap (a, b)  <->  the 4x4 / stride-2 gather tap (ty, tx) = (2a + 1 - p,
// 2b + 1 - q) of the 'ups_dgrad' packing (egz_pack_w3x3_split kind 3); kind 7 = forward of [upsample x2 -> conv]: si = phase
// (p, q) * 4 + tap (a, b), the 'ups_fwd' pre-summed taps.
__device__ __forceinline__ float frag_value9(const float* w9, int kind, int si) {
    if (kind == 4) return w9[si];
    if (kind == 5) return w9[8 - si];
    if (kind == 7) return weff9s(w9, (si >> 2) >> 1, (si & 3) >> 1, (si >> 2) & 1, si & 1);
    const int img = si >> 2, tap = si & 3;
    const int oy = 2 * (tap >> 1) - (img >> 1), ox = 2 * (tap & 1) - (img & 1);        // ty - 1, tx - 1
    return weff9s(w9, (oy == -1 || oy == 1) ? 1 : 0, (oy <= 0) ? 1 : 0, (ox == -1 || ox == 1) ? 1 : 0, (ox <= 0) ? 1 : 0);
}

// All slices of one (GEMM column, reduction element) pair: j = (((channel block * nt32 + ntile) * 2 + ks) * 64 + lane) * 8 + e.
// The nine taps of the pair are read ONCE (36 contiguous bytes; one thread per packed VALUE read each weight nine times from
// nine launches' worth of distance -- 9x the weight bytes out of L2 per packing, 0.6 ms per step over the ~80 packings) and
// every slice's hi / lo halves are derived from them.  Fragment order: value (slice s, ntile, ks, lane, e) sits at
// (((s * nt32 + ntile) * 2 + ks) * 2 + plane) * 512 + lane * 8 + e.
template <typename T, int KIND>
__device__ __forceinline__ void frag_pack_pair(const float* __restrict__ w, unsigned short* __restrict__ wq, int C, int K,
                                               int Np, int Rp, float scale, long j) {
    constexpr int NS = (KIND >= 6) ? 16 : 9;                   // slices per channel block
    const int nt32 = Np >> 5, ncb = Rp / XBK;
    const int e = (int)(j & 7), lane = (int)((j >> 3) & 63), ks = (int)((j >> 9) & 1);
    const long r2 = j >> 10;
    const int ntile = (int)(r2 % nt32), cblk = (int)(r2 / nt32);
    const int col = ntile * 32 + (lane & 31), k = cblk * XBK + ks * 16 + (lane >> 5) * 8 + e;
    float w9[9];
    const bool fwd = (KIND == 4 || KIND == 7);                 // columns = output channels, reduction = input channels
    const bool ok = fwd ? (col < K && k < C) : (col < C && k < K);
    const float* src = w + (fwd ? ((long)col * C + k) : ((long)k * C + col)) * 9;
#pragma unroll
    for (int t = 0; t < 9; ++t) w9[t] = ok ? src[t] : 0.f;
#pragma unroll
    for (int si = 0; si < NS; ++si) {
        // kind 7: [phase][channel block][tap] -- one phase's slices are contiguous (a tile of the kernel runs one phase)
        const long s = (KIND == 7) ? (long)(si >> 2) * (ncb * 4) + cblk * 4 + (si & 3) : (long)cblk * NS + si;
        const long r = s * nt32 + ntile;
        unsigned short h, l;
        Half<T>::split(frag_value9(w9, KIND, si) * scale, h, l);
        const long base = ((r * 2 + ks) * 2) * 512 + lane * 8 + e;     // plane 0; plane 1 is 512 elements on
        wq[base] = h;
        wq[base + 512] = l;
    }
}
template <typename T>
__device__ __forceinline__ void frag_pack_pair_k(const float* __restrict__ w, unsigned short* __restrict__ wq, int C, int K,
                                                 int kind, int Np, int Rp, float scale, long j) {
    if (kind == 4) frag_pack_pair<T, 4>(w, wq, C, K, Np, Rp, scale, j);
    else if (kind == 5) frag_pack_pair<T, 5>(w, wq, C, K, Np, Rp, scale, j);
    else if (kind == 6) frag_pack_pair<T, 6>(w, wq, C, K, Np, Rp, scale, j);
    else frag_pack_pair<T, 7>(w, wq, C, K, Np, Rp, scale, j);
}

// one thread per (channel block, ntile32, ks, lane, e)
template <typename T>
__global__ __launch_bounds__(256) void pack_split_frag_kernel(const float* __restrict__ w, unsigned short* __restrict__ wq,
                                                             int C, int K, int kind, int Np, int Rp, float scale) {
    const long n = (long)Np * Rp;                              // (column, reduction element) pairs
    for (long j = blockIdx.x * (long)blockDim.x + threadIdx.x; j < n; j += (long)gridDim.x * blockDim.x)
        frag_pack_pair_k<T>(w, wq, C, K, kind, Np, Rp, scale, j);
}


// All fragment-ordered packings an optimizer step made stale, in ONE launch (egz_pack_w3x3_frag_batch).  The per-layer kernel
// above is launch-bound (~80 launches of 4-15 us per SP step, each in front of the convolution that needs it) and its 2-byte
// stores keep it at ~0.7 TB/s.  Here a block owns one 32 (output channels) x 32 (input channels) x 9 tile of ONE weight tensor:
// 32 contiguous 1152-byte runs of the OIHW tensor go through LDS (row pitch 289 words: the forward orientation walks rows, the
// data-gradient orientation walks 9-word columns -- both conflict-free), and every thread item gathers the 8 reduction elements
// of one lane of one fragment and writes its hi and lo halves as two 16-byte stores (1 KB contiguous per wave).  Same values,
// same split as pack_split_frag_kernel (bit-identical packings: test_pack_frag_batch_matches_per_layer).
// table: rows of 8 x int64 [w, wq, C, K, kind, dtype, first block, -] in device memory, sorted by first block.
constexpr int PT_LD = 289;
template <typename T>
__device__ __forceinline__ void pack_frag_tile(const float* __restrict__ w, unsigned short* __restrict__ wq, int C, int K,
                                               int kind, float scale, int tile, float* __restrict__ sw) {
    const int Cp = (C + 31) & ~31, Kp = (K + 31) & ~31;
    const int ncbc = Cp >> 5;
    const int kb = tile / ncbc, cb = tile - kb * ncbc;
    const int tid = threadIdx.x;
    if (C % 32 == 0 && (reinterpret_cast<unsigned long long>(w) & 15) == 0) {
        for (int idx = tid; idx < 32 * 72; idx += 256) {
            const int i = idx / 72, q = idx - i * 72;
            const int k = kb * 32 + i;
            f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
            if (k < K) v = *reinterpret_cast<const f32x4*>(w + ((long)k * C + cb * 32) * 9 + q * 4);
            float* d = sw + i * PT_LD + q * 4;
            d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
        }
    } else {
        for (int idx = tid; idx < 32 * 288; idx += 256) {
            const int i = idx / 288, q = idx - i * 288;
            const int k = kb * 32 + i, c = cb * 32 + q / 9;
            sw[i * PT_LD + q] = (k < K && c < C) ? w[(long)k * C * 9 + cb * 288 + q] : 0.f;
        }
    }
    __syncthreads();
    const bool fwd = (kind == 4 || kind == 7);
    const int NS = (kind >= 6) ? 16 : 9;
    const int nt32 = (fwd ? Kp : Cp) >> 5, ncb = (fwd ? Cp : Kp) >> 5;
    const int ntile = fwd ? kb : cb, cblk = fwd ? cb : kb;
    for (int it = tid; it < NS * 128; it += 256) {
        const int lane = it & 63, ks = (it >> 6) & 1, si = it >> 7;
        const int col = lane & 31, kk0 = ks * 16 + (lane >> 5) * 8;
        const float* p = fwd ? sw + col * PT_LD + kk0 * 9 : sw + kk0 * PT_LD + col * 9;
        const int estep = fwd ? 9 : PT_LD;
        unsigned short h[8], l[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float* pe = p + e * estep;
            float v;
            if (kind == 4) v = pe[si];
            else if (kind == 5) v = pe[8 - si];
            else {
                float w9[9];
#pragma unroll
                for (int t = 0; t < 9; ++t) w9[t] = pe[t];
                v = frag_value9(w9, kind, si);
            }
            Half<T>::split(v * scale, h[e], l[e]);
        }
        const long s = (kind == 7) ? (long)(si >> 2) * (ncb * 4) + cblk * 4 + (si & 3) : (long)cblk * NS + si;
        const long r = s * nt32 + ntile;
        const long base = ((r * 2 + ks) * 2) * 512 + lane * 8;
        u32x4 hv, lv;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            hv[e] = (unsigned)h[2 * e] | ((unsigned)h[2 * e + 1] << 16);
            lv[e] = (unsigned)l[2 * e] | ((unsigned)l[2 * e + 1] << 16);
        }
        *reinterpret_cast<u32x4*>(wq + base) = hv;
        *reinterpret_cast<u32x4*>(wq + base + 512) = lv;
    }
}

__global__ __launch_bounds__(256) void pack_frag_batch_kernel(const long long* __restrict__ table, int n) {
    __shared__ float sw[32 * PT_LD];
    const int b = (int)blockIdx.x;
    int lo = 0, hi = n - 1;
    while (lo < hi) {                                          // last row whose first block is <= b
        const int mid = (lo + hi + 1) >> 1;
        if (table[mid * 8 + 6] <= b) lo = mid; else hi = mid - 1;
    }
    const long long* d = table + lo * 8;
    const float* w = reinterpret_cast<const float*>(d[0]);
    unsigned short* wq = reinterpret_cast<unsigned short*>(d[1]);
    const int C = (int)d[2], K = (int)d[3], kind = (int)d[4], tile = b - (int)d[6];
    if (d[5] == 1) pack_frag_tile<_Float16>(w, wq, C, K, kind, F16_WSCALE, tile, sw);
    else pack_frag_tile<__bf16>(w, wq, C, K, kind, 1.f, tile, sw);
}


// ---------------------------------------------------------------------------------------------------------
// Persistent form for the narrowest layers (late_fusion.py:10-12: at most 32 reduction channels AND at most 32 GEMM columns,
// patch geometry).  With one channel block the kernel above has nothing to pipeline inside a tile (fetch -> split -> barrier ->
// 108 MFMAs -> store, serially) and every tile re-streams all 36 KB of weight fragments from L2.  Here
//   * one 256-thread block per CU walks a contiguous range of 16 x 16 pixel tiles of its XCD (so neighbouring halos meet in
//     one L2);
//   * the 36 weight fragments (9 taps x 2 k-steps x hi / lo) are loaded ONCE into registers -- 144 VGPRs; the block is
//     alone on its CU, so a wave has the whole 512-register file;
//   * the halo of tile i + 1 is fetched (buffer loads with fixed per-thread offsets + one scalar origin + border masks)
//     before the MFMAs of tile i, split and written to the OTHER LDS image after them: one barrier per tile.
// Same LDS image, fragment addressing, arithmetic and epilogues (bias / bias + ReLU / bias + BN statistics) as the 32-column
// configuration (WM = 4) of the kernel above, which stays the path for images that are not multiples of 16.
// BNIN: x is the PRE-BatchNorm output of the block below and bn_coef that BatchNorm's (mean, 1/std, scale, shift) rows: the
// halo staging applies relu(x * scale + shift) per channel on the way into LDS (zero padding stays zero), so the normalised
// activation tensor is never written or read (late_fusion.py:10-12; a_absmax = max of the normalised values,
// egz_bn_finalize_deferred).  mm_out (EPI_BIAS_STATS): per-block rows [2][K] of the per-channel max / min of y -- what the
// next layer's deferred BatchNorm needs to bound its output.
template <typename T, int EPI, bool BNIN>
__global__ __launch_bounds__(256, 1) void conv3x3_x3p_narrow_kernel(
    const float* __restrict__ x, const unsigned short* __restrict__ wq, const float* __restrict__ bias,
    float* __restrict__ y, double* __restrict__ stat, int B, int H, int W, int C, int K, float out_scale, int total,
    const unsigned int* __restrict__ a_absmax, const float* __restrict__ bn_y, const float* __restrict__ bn_coef,
    float* __restrict__ mm_out) {
    using G = Geo<4>;
    constexpr int BM = G::BM, HSLOTS = G::HSLOTS, NJ = G::NJ, SPP = G::SPP, MR = G::MR, RPW = G::RPW;
    constexpr int APL = HSLOTS * XLD, ABUF = 2 * APL;
    static_assert(BM == 256 && MR == 2 && RPW == 64 && G::PROWS == 16, "16 x 16 patch, 64 rows per wave");
    __shared__ __attribute__((aligned(16))) unsigned short Ah[2 * ABUF];
    __shared__ double sred[4 * 2 * 32];

    const float a_scale = absmax_scale(a_absmax);
    out_scale /= a_scale;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hl = lane >> 5, l31 = lane & 31;
    const int pw = W >> 4, ppi = (H >> 4) * pw;                 // patches per row / per image

    // this block's tiles: XCD (blockIdx % 8) owns the contiguous range [xcd * per, (xcd + 1) * per)
    const int per = (total + 7) >> 3, xcd = (int)(blockIdx.x & 7), nbx = (int)(gridDim.x >> 3);
    const int t_end = ((xcd + 1) * per < total) ? (xcd + 1) * per : total;
    int tile = xcd * per + (int)(blockIdx.x >> 3);
    if (tile >= t_end) {                                        // (a block without tiles still owns its row of partial sums)
        if ((EPI == EPI_BIAS_STATS || EPI == EPI_BNSUMS) && tid < 2 * K) stat[(long)blockIdx.x * 2 * K + tid] = 0.0;
        if (EPI == EPI_BIAS_STATS && mm_out && tid < 2 * K) mm_out[(long)blockIdx.x * 2 * K + tid] = (tid < K) ? -INFINITY : INFINITY;
        return;
    }

    // ---- weight fragments: [tap][ks][plane][lane][8 halves], 1 KB pieces, resident in registers for the whole launch
    const __amdgpu_buffer_rsrc_t b_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(wq), 0, 9 * 4096, 0x00020000);
    u32x4 bq[9][4];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int c = 0; c < 4; ++c) bq[t][c] = __builtin_amdgcn_raw_buffer_load_b128(b_rs, (unsigned)lane * 16u + c * 1024, t * 4096, 0);

    // ---- halo staging map: slot q = (tid >> 3) + SPP j = grid position (hy, hx) of the 18 x 18 halo (pitch 20), 4 channels
    // (tid & 7) each.  The x resource starts (W + 1) pixels early so that the (-1, -1) corner keeps lane offsets >= 0.
    const unsigned x_bias = (unsigned)(W + 1) * (unsigned)C * 4u;
    const __amdgpu_buffer_rsrc_t a_rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(x)) - x_bias, 0, (int)((unsigned)B * H * W * C * 4u + x_bias), 0x00020000);
    const int a_c4 = tid & 7;
    unsigned a_vo[NJ], a_rc[NJ];
    int a_lds[NJ];
    unsigned a_in = 0, ra_ok = 0;                               // BNIN: bit j = slot j holds a real element / was inside the image
    f32x4 in_sc = {0.f, 0.f, 0.f, 0.f}, in_sh = {0.f, 0.f, 0.f, 0.f};
    if (BNIN && a_c4 * 4 < C) {
        in_sc = *reinterpret_cast<const f32x4*>(bn_coef + 2 * C + a_c4 * 4);
        in_sh = *reinterpret_cast<const f32x4*>(bn_coef + 3 * C + a_c4 * 4);
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int q = (tid >> 3) + SPP * j;
        const int hy = q / HPITCH, hx = q - hy * HPITCH;
        const bool in = hy < 18 && hx < 18 && a_c4 * 4 < C;
        if (in) a_in |= 1u << j;
        a_vo[j] = in ? (unsigned)(((hy * W + hx) * C + a_c4 * 4) * 4) : 0xFFFFFFFFu;
        a_rc[j] = (unsigned)(hy << 8 | hx);
        a_lds[j] = q * XLD + (((a_c4 >> 1) ^ ((hx >> 2) & 3)) << 3) + (a_c4 & 1) * 4;
    }
    f32x4 ra[NJ];
    auto gload_a = [&](const int tl) {
        const int b0 = tl / ppi, rem = tl - b0 * ppi;
        const int y0 = (rem / pw) * 16, x0 = (rem % pw) * 16;
        // valid halo rows hy in [rlo, rlo + rn], columns hx in [clo, clo + cn] (source pixel = (y0 - 1 + hy, x0 - 1 + hx))
        const unsigned rlo = (y0 == 0) ? 1u : 0u, rn = (unsigned)((H - y0 < 17) ? (H - y0) : 17) - rlo;
        const unsigned clo = (x0 == 0) ? 1u : 0u, cn = (unsigned)((W - x0 < 17) ? (W - x0) : 17) - clo;
        const unsigned so = (unsigned)((((long)b0 * H + y0) * W + x0) * C * 4);          // + x_bias - x_bias
        unsigned okm = 0;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const bool ok = ((a_rc[j] >> 8) - rlo <= rn) && ((a_rc[j] & 255u) - clo <= cn);
            if (BNIN && ok) okm |= 1u << j;
            ra[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(a_rs, ok ? a_vo[j] : 0xFFFFFFFFu, so, 0));
        }
        ra_ok = okm & a_in;
    };
    auto lstore_a = [&](const int abuf) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            u32x2 hi, lo;
            if (BNIN) {                                         // BatchNorm + ReLU of the block below, on the way into LDS;
                const float ms = ((ra_ok >> j) & 1u) ? a_scale : 0.f;      // the zero-padding mask rides on the scale factor
#pragma unroll
                for (int e = 0; e < 4; ++e) ra[j][e] = fmaxf(__builtin_fmaf(ra[j][e], in_sc[e], in_sh[e]), 0.f) * ms;
                if constexpr (egz_drop_alo<T>::value) hi = Half<T>::rne4s(ra[j], 1.f);
                else Half<T>::split4(ra[j], hi, lo);
            } else if constexpr (egz_drop_alo<T>::value) {
                hi = Half<T>::rne4s(ra[j], a_scale);
            } else {
                Half<T>::split4s(ra[j], a_scale, hi, lo);
            }
            unsigned short* d = Ah + abuf * ABUF + a_lds[j];
            *reinterpret_cast<u32x2*>(d) = hi;
            if constexpr (!egz_drop_alo<T>::value) *reinterpret_cast<u32x2*>(d + APL) = lo;
        }
    };

    // ---- activation fragment addresses (bytes, plane 0, k-step 0, image 0) of the wave's first 32-row group, per tap shift
    int fa9[9];
    {
        const int i = wave * RPW + l31;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int dy = t / 3 - 1, dx = t % 3 - 1;
            const int slot = ((i >> 4) + 1 + dy) * HPITCH + (i & 15) + 1 + dx, col = (i & 15) + 1 + dx;
            fa9[t] = (slot * XLD + ((hl ^ ((col >> 2) & 3)) << 3)) * 2;
        }
    }
    constexpr int MRSTEP = 2 * HPITCH * XLD * 2;               // bytes between the wave's two row groups (2 grid rows)
    const char* Ab = reinterpret_cast<const char*>(Ah);
    const bool nok = l31 < K;
    const float bz = (bias && nok) ? bias[l31] : 0.f;
    // EPI_BNSUMS: (mean, invstd, scale, shift) of this lane's channel of the BatchNorm below (bn_coef: 4 rows of K floats)
    const float bn_mu = (EPI == EPI_BNSUMS && nok) ? bn_coef[l31] : 0.f, bn_is = (EPI == EPI_BNSUMS && nok) ? bn_coef[K + l31] : 0.f;
    const float bn_sc = (EPI == EPI_BNSUMS && nok) ? bn_coef[2 * K + l31] : 0.f, bn_sh = (EPI == EPI_BNSUMS && nok) ? bn_coef[3 * K + l31] : 0.f;

    double s1 = 0.0, s2 = 0.0;                                  // partial sums of the block's tiles: ONE stat row per block
    float vmx = -INFINITY, vmn = INFINITY;                      // EPI_BIAS_STATS + mm_out: this lane's channel max / min of y
    // output (and EPI_BNSUMS: bn_y) addressing: buffer stores / loads with a fixed per-thread byte offset inside the 16 x 16
    // patch + one scalar patch origin; lanes beyond K get an out-of-range offset (dropped / zero).  No per-lane branches: the
    // tile body stays ONE basic block, so the bn_y requests issued in front of the MFMAs are not sunk to their use and the
    // stores are not serialised by per-block wait counts.
    const int obytes = (int)((unsigned)B * H * W * K * 4u);
    const __amdgpu_buffer_rsrc_t y_rs = __builtin_amdgcn_make_buffer_rsrc(y, 0, obytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t bn_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(EPI == EPI_BNSUMS ? bn_y : y), 0, obytes, 0x00020000);
    unsigned o_vo[MR][16];
#pragma unroll
    for (int mr = 0; mr < MR; ++mr)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = wave * RPW + mr * 32 + egz_acc_row(r, lane);
            o_vo[mr][r] = nok ? (unsigned)((((i >> 4) * W + (i & 15)) * K + l31) * 4) : 0xFFFFFFFFu;
        }

    // ---- software-pipelined tile loop.  The block is alone on its CU with one wave per SIMD, so nothing overlaps a wave's own
    // phases: fetch -> MFMAs -> epilogue (32 stores, statistics) -> split + LDS stores of the next halo -> barrier ran back to
    // back.  Here the 108 MFMAs of tile i are issued in 18 groups of six, and behind each group goes a piece of the OTHER work:
    // two epilogue elements of tile i - 1 (its accumulators were copied aside) and, in the second half, one piece of the halo
    // of tile i + 1 (requested at the top of the iteration).  The first iteration has no previous tile: its side stores / loads
    // go through a zero-length buffer resource (dropped / zero) and its statistics terms are zero.  Measured: -10 % on the
    // 32 -> 8 layer, -3..5 % on the data gradients, nothing on the 32 -> 32 forward (profiles/r03_ab_notes.txt) -- the launches
    // sit at 3.3-3.7 TB/s with one tile's loads in flight per CU; a second halo register set (two tiles in flight) spilled and
    // was slower.
    auto lstore_piece = [&](const int abuf, const int j) {
        u32x2 hi, lo;
        if (BNIN) {
            const float ms = ((ra_ok >> j) & 1u) ? a_scale : 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) ra[j][e] = fmaxf(__builtin_fmaf(ra[j][e], in_sc[e], in_sh[e]), 0.f) * ms;
            if constexpr (egz_drop_alo<T>::value) hi = Half<T>::rne4s(ra[j], 1.f);
            else Half<T>::split4(ra[j], hi, lo);
        } else if constexpr (egz_drop_alo<T>::value) {
            hi = Half<T>::rne4s(ra[j], a_scale);
        } else {
            Half<T>::split4s(ra[j], a_scale, hi, lo);
        }
        unsigned short* d = Ah + abuf * ABUF + a_lds[j];
        *reinterpret_cast<u32x2*>(d) = hi;
        if constexpr (!egz_drop_alo<T>::value) *reinterpret_cast<u32x2*>(d + APL) = lo;
    };
    static_assert(NJ <= 12, "one halo piece per MFMA group of the second half");
    f32x16 pacc[MR];
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) pacc[i][r] = 0.f;
    unsigned pso = 0;
    bool have_prev = false;                                     // block-uniform
    gload_a(tile);
    lstore_a(0);
    lds_barrier();
    int buf = 0;
    for (;;) {
        const int nxt = tile + nbx;
        const bool more = nxt < t_end;                          // block-uniform
        if (more) gload_a(nxt);
        const int pbytes = have_prev ? obytes : 0;
        const __amdgpu_buffer_rsrc_t py_rs = __builtin_amdgcn_make_buffer_rsrc(y, 0, pbytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t pbn_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(EPI == EPI_BNSUMS ? bn_y : y), 0, pbytes, 0x00020000);
        const float pbz = have_prev ? bz : 0.f;
        float byp[MR][16];
        if (EPI == EPI_BNSUMS) {                                // the previous tile's values of the pre-BN tensor below
#pragma unroll
            for (int mr = 0; mr < MR; ++mr)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    byp[mr][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(pbn_rs, o_vo[mr][r], pso, 0));
        }
        __builtin_amdgcn_sched_barrier(0);
        // EPI_BNSUMS: the 32 elements of a tile are summed in fp32 and added to the fp64 block sums once per tile -- 4
        // double-precision instructions per tile instead of 128 (the kernel is issue-bound: ~980 vector instructions per 108
        // MFMAs, SQ counters).  The forward statistics stay fp64 per element (their difference is the variance).
        float q1 = 0.f, q2 = 0.f;
        auto epi_elem = [&](const int mr, const int r) {
            float v = pacc[mr][r] * out_scale + pbz;
            if (EPI == EPI_BIAS_RELU) v = fmaxf(v, 0.f);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), py_rs, o_vo[mr][r], pso, 0);
            if (EPI == EPI_BIAS_STATS) {                       // (fp64 per element: the variance is a difference of the two sums)
                s1 += (double)v;
                s2 += (double)v * (double)v;
                vmx = fmaxf(vmx, have_prev ? v : -INFINITY);
                vmn = fminf(vmn, have_prev ? v : INFINITY);
            }
            if (EPI == EPI_BNSUMS) {
                const float yp = byp[mr][r];
                const float dz = (yp * bn_sc + bn_sh > 0.f) ? v : 0.f;
                q1 += dz;
                q2 += dz * ((yp - bn_mu) * bn_is);
            }
        };
        f32x16 acc[MR];
#pragma unroll
        for (int i = 0; i < MR; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        // the activation fragments of group g + 1 are read from LDS before the MFMAs of group g (two register sets): a lone wave
        // has nobody to hide its own LDS latency behind
        u32x4 fh[2][MR], fl[2][MR];
        auto frag_read = [&](const int g, u32x4 (&h)[MR], u32x4 (&l)[MR]) {
            const int t = g >> 1, ks = g & 1;
#pragma unroll
            for (int mr = 0; mr < MR; ++mr) {
                const int a = (fa9[t] + buf * (ABUF * 2) + mr * MRSTEP) ^ (ks * 32);
                h[mr] = *reinterpret_cast<const u32x4*>(Ab + a);
                l[mr] = *reinterpret_cast<const u32x4*>(Ab + APL * 2 + a);
            }
        };
        frag_read(0, fh[0], fl[0]);
#pragma unroll
        for (int g = 0; g < 18; ++g) {
            const int t = g >> 1, ks = g & 1;
            if (g + 1 < 18) frag_read(g + 1, fh[(g + 1) & 1], fl[(g + 1) & 1]);
            u32x4 (&ah)[MR] = fh[g & 1];
            u32x4 (&al)[MR] = fl[g & 1];
#pragma unroll
            for (int term = 0; term < 3; ++term)
#pragma unroll
                for (int mr = 0; mr < MR; ++mr)
                    if (!(egz_drop_alo<T>::value && term == 0)) acc[mr] = Half<T>::mfma(term == 0 ? al[mr] : ah[mr], term == 1 ? bq[t][ks * 2 + 1] : bq[t][ks * 2], acc[mr]);
            if (g < 16) {                                       // two epilogue elements of the previous tile
                epi_elem((2 * g) >> 4, (2 * g) & 15);
                epi_elem((2 * g + 1) >> 4, (2 * g + 1) & 15);
            }
            // (all pieces in the last NJ / 2 groups, two per group, measured the same: the halo fetch is not what the wave waits for)
            if (g >= 18 - NJ) lstore_piece(buf ^ 1, g - (18 - NJ));      // (stale registers when there is no next tile: harmless)
            __builtin_amdgcn_sched_barrier(0);
        }
        if (EPI == EPI_BNSUMS) {
            s1 += (double)q1;
            s2 += (double)q2;
        }
#pragma unroll
        for (int i = 0; i < MR; ++i) pacc[i] = acc[i];
        {
            const int b0 = tile / ppi, rem = tile - b0 * ppi;
            const int y0 = (rem / pw) * 16, x0 = (rem % pw) * 16;
            pso = (unsigned)((((long)b0 * H + y0) * W + x0) * K * 4);
        }
        have_prev = true;
        if (!more) break;
        lds_barrier();                                          // everyone has staged its share and is done reading `buf`
        buf ^= 1;
        tile = nxt;
    }
    {   // ---- epilogue of the last tile
        float byp[MR][16];
        if (EPI == EPI_BNSUMS) {
#pragma unroll
            for (int mr = 0; mr < MR; ++mr)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    byp[mr][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(bn_rs, o_vo[mr][r], pso, 0));
        }
        float q1 = 0.f, q2 = 0.f;
#pragma unroll
        for (int mr = 0; mr < MR; ++mr)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = pacc[mr][r] * out_scale + bz;
                if (EPI == EPI_BIAS_RELU) v = fmaxf(v, 0.f);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), y_rs, o_vo[mr][r], pso, 0);
                if (EPI == EPI_BIAS_STATS) {
                    s1 += (double)v;
                    s2 += (double)v * (double)v;
                    vmx = fmaxf(vmx, v);
                    vmn = fminf(vmn, v);
                }
                if (EPI == EPI_BNSUMS) {
                    const float yp = byp[mr][r];
                    const float dz = (yp * bn_sc + bn_sh > 0.f) ? v : 0.f;
                    q1 += dz;
                    q2 += dz * ((yp - bn_mu) * bn_is);
                }
            }
        if (EPI == EPI_BNSUMS) {
            s1 += (double)q1;
            s2 += (double)q2;
        }
    }
    if (EPI == EPI_BIAS_STATS || EPI == EPI_BNSUMS) {            // row blockIdx.x: the four waves' sums in wave order
        s1 += __shfl_xor(s1, 32);
        s2 += __shfl_xor(s2, 32);
        if (hl == 0) {
            sred[(wave * 2 + 0) * 32 + l31] = s1;
            sred[(wave * 2 + 1) * 32 + l31] = s2;
        }
        lds_barrier();
        if (wave == 0 && hl == 0 && nok) {
            double t1 = 0.0, t2 = 0.0;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                t1 += sred[(w * 2 + 0) * 32 + l31];
                t2 += sred[(w * 2 + 1) * 32 + l31];
            }
            stat[((long)blockIdx.x * 2 + 0) * K + l31] = t1;
            stat[((long)blockIdx.x * 2 + 1) * K + l31] = t2;
        }
        if (EPI == EPI_BIAS_STATS && mm_out) {                   // (max / min are order-independent: no fixed order needed)
            vmx = fmaxf(vmx, __shfl_xor(vmx, 32));
            vmn = fminf(vmn, __shfl_xor(vmn, 32));
            float* smm = reinterpret_cast<float*>(Ah);         // the LDS images are dead by now
            lds_barrier();
            if (hl == 0) {
                smm[(wave * 2 + 0) * 32 + l31] = vmx;
                smm[(wave * 2 + 1) * 32 + l31] = vmn;
            }
            lds_barrier();
            if (wave == 0 && hl == 0 && nok) {
                float a = smm[l31], b = smm[32 + l31];
#pragma unroll
                for (int w = 1; w < 4; ++w) {
                    a = fmaxf(a, smm[(w * 2 + 0) * 32 + l31]);
                    b = fminf(b, smm[(w * 2 + 1) * 32 + l31]);
                }
                mm_out[((long)blockIdx.x * 2 + 0) * K + l31] = a;
                mm_out[((long)blockIdx.x * 2 + 1) * K + l31] = b;
            }
        }
    }
}

// geometry of the persistent narrow kernel (the output is addressed through a 32-bit buffer resource)
bool x3p_narrow_ok(int B, int H, int W, int C, int K) {
    return C <= 32 && K <= 32 && H % 16 == 0 && W % 16 == 0 && 4ull * B * H * W * K < (1ull << 32);
}

// blocks of a persistent narrow launch = rows of its partial-sum buffer (epi 2 / 5): one block per CU, a multiple of 8 (XCDs)
int x3p_narrow_blocks(int B, int H, int W) {
    const int total = (int)((long)B * H * W / 256);
    int cus = 256, dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
        cus = 256;
    int blocks = (cus / 8) * 8;
    if (blocks > ((total + 7) / 8) * 8) blocks = ((total + 7) / 8) * 8;
    return blocks;
}

template <typename T>
int launch_x3p_narrow(int epi, const float* x, const unsigned short* wq, const float* bias, float* y, double* stat, int B, int H,
                      int W, int C, int K, float out_scale, const unsigned int* a_absmax, const float* bn_y, const float* bn_coef,
                      float* mm_out, hipStream_t st) {
    const int total = (int)((long)B * H * W / 256);
    const int blocks = x3p_narrow_blocks(B, H, W);
#define EGZ_X3P(E, BN) hipLaunchKernelGGL((conv3x3_x3p_narrow_kernel<T, E, BN>), dim3(blocks), dim3(256), 0, st, x, wq, bias, y, stat, B, H, W, C, K, out_scale, total, a_absmax, bn_y, bn_coef, mm_out)
    const bool bnin = bn_coef && epi != EPI_BNSUMS;             // forward epilogues: bn_coef = the INPUT's deferred BatchNorm
    if (epi == EPI_BNSUMS) EGZ_X3P(EPI_BNSUMS, false);
    else if (epi == EPI_BIAS) { if (bnin) EGZ_X3P(EPI_BIAS, true); else EGZ_X3P(EPI_BIAS, false); }
    else if (epi == EPI_BIAS_RELU) { if (bnin) EGZ_X3P(EPI_BIAS_RELU, true); else EGZ_X3P(EPI_BIAS_RELU, false); }
    else { if (bnin) EGZ_X3P(EPI_BIAS_STATS, true); else EGZ_X3P(EPI_BIAS_STATS, false); }
#undef EGZ_X3P
    EGZ_CHECK_LAUNCH("egz_conv3x3_fwd_streamed(narrow)");
    return 0;
}

// Sum of the nsplit partial results of a split-K launch in split order + the epilogue of the unsplit kernel.
// One block = 32 pixel rows x 64 columns: thread = 4 columns (one 16-byte load per split) x 2 rows; BN partial sums: one stat
// row per 32 pixels (egz_conv3x3_fwd_streamed_splitk_stat_rows), reduced over the block's rows through LDS in a fixed order.
constexpr int FIX_ROWS = 32;
template <int EPI>
__global__ __launch_bounds__(256) void splitk_fixup_kernel(const float* __restrict__ part, const float* __restrict__ bias,
                                                           float* __restrict__ y, double* __restrict__ stat, long M, int K,
                                                           int nsplit, unsigned int* __restrict__ absmax_out) {
    __shared__ double sred[2][16][64];
    __shared__ float samax[4];
    float amx = 0.f;
    const int c4 = threadIdx.x & 15, rr = threadIdx.x >> 4;
    const int col = blockIdx.y * 64 + c4 * 4;
    const bool cok = col < K;                                  // K % 4 == 0: a float4 is inside or outside as a whole
    f32x4 bz = {0.f, 0.f, 0.f, 0.f};
    if (bias && cok) bz = *reinterpret_cast<const f32x4*>(bias + col);
    double s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int p = 0; p < FIX_ROWS / 16; ++p) {
        const long m = (long)blockIdx.x * FIX_ROWS + p * 16 + rr;
        if (m < M && cok) {
            f32x4 v = *reinterpret_cast<const f32x4*>(part + m * K + col);
            for (int sp = 1; sp < nsplit; ++sp) v += *reinterpret_cast<const f32x4*>(part + ((long)sp * M + m) * K + col);
            v += bz;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (EPI == EPI_BIAS_RELU) {
                    v[e] = fmaxf(v[e], 0.f);
                    amx = fmaxf(amx, v[e]);
                }
                if (EPI == EPI_BIAS_STATS) {
                    s1[e] += (double)v[e];
                    s2[e] += (double)v[e] * (double)v[e];
                }
            }
            *reinterpret_cast<f32x4*>(y + m * K + col) = v;
        }
    }
    if (EPI == EPI_BIAS_RELU && absmax_out) {                   // block-uniform: per-block max of the activation written
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) amx = fmaxf(amx, __shfl_xor(amx, o));
        if ((threadIdx.x & 63) == 0) samax[threadIdx.x >> 6] = amx;
        __syncthreads();
        if (threadIdx.x == 0)
            absmax_commit(absmax_out, blockIdx.y * gridDim.x + blockIdx.x, fmaxf(fmaxf(samax[0], samax[1]), fmaxf(samax[2], samax[3])));
    }
    if (EPI == EPI_BIAS_STATS) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            sred[0][rr][c4 * 4 + e] = s1[e];
            sred[1][rr][c4 * 4 + e] = s2[e];
        }
        __syncthreads();
        if (threadIdx.x < 128) {
            const int which = threadIdx.x >> 6, c = threadIdx.x & 63;
            double t = 0.0;
#pragma unroll
            for (int r = 0; r < 16; ++r) t += sred[which][r][c];
            if (blockIdx.y * 64 + c < K) stat[((long)blockIdx.x * 2 + which) * K + blockIdx.y * 64 + c] = t;
        }
    }
}

// split count for a plain launch on the 128 x 128 tile: > 1 when the tiles fill less than a quarter of one round of resident
// blocks (2 per CU) and every split keeps at least two channel blocks.  (Half-filled rounds -- the 14 x 14 layers at batch
// 32 -- measured neutral, 35.4 vs 35.4 ms per step, for 14 MB more HBM traffic per launch: not split.)
int x3s_splits(int B, int H, int W, int C, int K) {
    if (K % 128 != 0 || C % 32 != 0) return 1;
    const long M = (long)B * H * W;
    const bool patch = (W % 16 == 0) && (H % 8 == 0);
    const long tiles = (patch ? M / 128 : (M + 127) / 128) * (K / 128);
    const int ncb = C / 32;
    int cus = 256, dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
        cus = 256;
    const long slots = 2L * cus;
    if (tiles * 4 > slots || ncb < 4) return 1;
    long ns = slots / tiles;
    if (ns > ncb / 2) ns = ncb / 2;
    if (ns > 16) ns = 16;
    return ns < 2 ? 1 : (int)ns;
}

template <typename T>
int launch_x3s_splitk(int epi, const float* x, const unsigned short* wq, const float* bias, float* y, double* stat, int B, int H,
                      int W, int C, int K, float out_scale, const unsigned int* a_absmax, float* part, int nsplit,
                      unsigned int* absmax_out, hipStream_t st) {
    using G = Geo<1>;
    const long M = (long)B * H * W;
    const int Cp = (C + 31) / 32 * 32, Kp = (K + 31) / 32 * 32;
    const bool patch = (W % 16 == 0) && (H % G::PROWS == 0);
    const int mt = patch ? (int)(M / G::BM) : egz_cdiv(M, G::BM);
    const int total = mt * (Kp / G::BN) * nsplit;
    const dim3 grid(((total + 7) / 8) * 8);
    if (patch) hipLaunchKernelGGL((conv3x3_igemm_x3s_kernel<T, 1, EPI_PARTIAL, true, PLAIN>), grid, dim3(G::NTHR), 0, st, x, wq, nullptr, part, nullptr, B, H, W, C, K, Cp, Kp, out_scale, mt, total, a_absmax, nullptr, nullptr, nsplit, nullptr);
    else       hipLaunchKernelGGL((conv3x3_igemm_x3s_kernel<T, 1, EPI_PARTIAL, false, PLAIN>), grid, dim3(G::NTHR), 0, st, x, wq, nullptr, part, nullptr, B, H, W, C, K, Cp, Kp, out_scale, mt, total, a_absmax, nullptr, nullptr, nsplit, nullptr);
    EGZ_CHECK_LAUNCH("egz_conv3x3_fwd_streamed_splitk");
    const dim3 fg(egz_cdiv(M, FIX_ROWS), egz_cdiv(K, 64));
    if (epi == EPI_BIAS) hipLaunchKernelGGL(splitk_fixup_kernel<EPI_BIAS>, fg, dim3(256), 0, st, part, bias, y, stat, M, K, nsplit, nullptr);
    else if (epi == EPI_BIAS_RELU) hipLaunchKernelGGL(splitk_fixup_kernel<EPI_BIAS_RELU>, fg, dim3(256), 0, st, part, bias, y, stat, M, K, nsplit, absmax_out);
    else hipLaunchKernelGGL(splitk_fixup_kernel<EPI_BIAS_STATS>, fg, dim3(256), 0, st, part, bias, y, stat, M, K, nsplit, nullptr);
    EGZ_CHECK_LAUNCH("egz_conv3x3_fwd_streamed_splitk(fixup)");
    return 0;
}

template <typename T, int WM, int MODE>
int launch_x3s(int epi, const float* x, const unsigned short* wq, const float* bias, float* y, double* stat, int B, int H,
               int W, int C, int K, float out_scale, const unsigned int* a_absmax, const float* mask_src,
               unsigned int* absmax_out, hipStream_t st, unsigned int* mm_out = nullptr, bool pre = false) {
    using G = Geo<WM>;
    const int Ho = (MODE != PLAIN) ? H / 2 : H, Wo = (MODE != PLAIN) ? W / 2 : W;
    const long M = (long)B * Ho * Wo;
    const int Cp = (C + 31) / 32 * 32, Kp = (K + 31) / 32 * 32;
    const bool patch = (Wo % 16 == 0) && (Ho % G::PROWS == 0);
    const int mt = patch ? (int)(M / G::BM) : egz_cdiv(M, G::BM);
    const int total = mt * (Kp / G::BN) * ((MODE == UPSF) ? 4 : 1);
    dim3 grid(((total + 7) / 8) * 8);
#ifdef EGZ_TILE_LOOP
    {
        static const int cap = getenv("EGZ_GRID_CAP") ? atoi(getenv("EGZ_GRID_CAP")) / 8 * 8 : 0;
        if (cap > 0 && (int)grid.x > cap) grid.x = cap;
    }
#endif
#define EGZ_X3S(E, P) hipLaunchKernelGGL((conv3x3_igemm_x3s_kernel<T, WM, E, P, MODE>), grid, dim3(G::NTHR), 0, st, x, wq, bias, y, stat, B, H, W, C, K, Cp, Kp, out_scale, mt, total, a_absmax, mask_src, absmax_out, 1, mm_out)
    if (epi != EPI_BIAS_RELU && epi != EPI_MASK_SUMS && epi != EPI_BIAS && epi != EPI_BNSUMS) absmax_out = nullptr;
    if (pre) {                         // pre-split activation operand: the training forward of the wide encoder layers
        if constexpr (MODE == PLAIN && (WM == 1 || WM == 2) && IS_F16<T>) {
#define EGZ_X3P(E, P) hipLaunchKernelGGL((conv3x3_igemm_x3s_kernel<T, WM, E, P, PLAIN, true>), grid, dim3(G::NTHR), 0, st, x, wq, bias, y, stat, B, H, W, C, K, Cp, Kp, out_scale, mt, total, a_absmax, mask_src, absmax_out, 1, mm_out)
            // epi 2: the training forward over pre-split activations; epi 0 / 5: data gradients over a pre-split gradient
            if (epi == EPI_BIAS_STATS) { if (patch) EGZ_X3P(EPI_BIAS_STATS, true); else EGZ_X3P(EPI_BIAS_STATS, false); }
            else if (epi == EPI_BIAS)  { if (patch) EGZ_X3P(EPI_BIAS, true); else EGZ_X3P(EPI_BIAS, false); }
            else if (epi == EPI_BNSUMS) { if (patch) EGZ_X3P(EPI_BNSUMS, true); else EGZ_X3P(EPI_BNSUMS, false); }
            else {
                egz_set_error("egz_conv3x3_fwd_streamed: a pre-split operand is taken by epi 0 / 2 / 5 only");
                return (int)hipErrorInvalidValue;
            }
#undef EGZ_X3P
            EGZ_CHECK_LAUNCH("egz_conv3x3_fwd_streamed(pre-split)");
            return 0;
        } else {
            egz_set_error("egz_conv3x3_fwd_streamed: pre-split operands exist for plain convs on the 128- / 64-column tiles");
            return (int)hipErrorInvalidValue;
        }
    }
    if constexpr (MODE == UPSF) {                                   // decoder blocks: bias + ReLU (or plain bias)
        if (epi != EPI_BIAS && epi != EPI_BIAS_RELU) {
            egz_set_error("egz_conv3x3_fwd_streamed: the upsample forward has the bias and bias + ReLU epilogues only");
            return (int)hipErrorInvalidValue;
        }
        if (patch) { if (epi == EPI_BIAS) EGZ_X3S(EPI_BIAS, true); else EGZ_X3S(EPI_BIAS_RELU, true); }
        else       { if (epi == EPI_BIAS) EGZ_X3S(EPI_BIAS, false); else EGZ_X3S(EPI_BIAS_RELU, false); }
    } else if (epi == EPI_BNSUMS) {
        if constexpr ((WM == 1 || WM == 2) && MODE == PLAIN) {      // data gradients of the encoders: 128- / 64-column 4-wave tiles
            if (patch) EGZ_X3S(EPI_BNSUMS, true); else EGZ_X3S(EPI_BNSUMS, false);
        } else {
            egz_set_error("egz_conv3x3_fwd_streamed: the BatchNorm-sums epilogue is built for the 128- / 64-column tiles of plain convs");
            return (int)hipErrorInvalidValue;
        }
    } else if (epi == EPI_MASK_SUMS) {
        if constexpr ((WM == 1 || WM == 2) && MODE != UPSF) {      // data gradients of the SP decoder: 128- and 64-column 4-wave tiles
            if (patch) EGZ_X3S(EPI_MASK_SUMS, true); else EGZ_X3S(EPI_MASK_SUMS, false);
        } else {
            egz_set_error("egz_conv3x3_fwd_streamed: the mask epilogue is not built for 32-column tiles");
            return (int)hipErrorInvalidValue;
        }
    } else if (patch) {
        if (epi == EPI_BIAS) EGZ_X3S(EPI_BIAS, true);
        else if (epi == EPI_BIAS_RELU) EGZ_X3S(EPI_BIAS_RELU, true);
        else EGZ_X3S(EPI_BIAS_STATS, true);
    } else {
        if (epi == EPI_BIAS) EGZ_X3S(EPI_BIAS, false);
        else if (epi == EPI_BIAS_RELU) EGZ_X3S(EPI_BIAS_RELU, false);
        else EGZ_X3S(EPI_BIAS_STATS, false);
    }
#undef EGZ_X3S
    EGZ_CHECK_LAUNCH("egz_conv3x3_fwd_streamed");
    return 0;
}

}  // namespace

// 1 when the streamed-weight kernel covers this geometry (C = reduction channels: a multiple of 32, or a multiple of 4
// below 32; K = GEMM columns: any).
// mode 0: plain conv over an H x W image; mode 1: data gradient of [nearest x2 upsample -> conv3x3] w.r.t. the low-res
// input, H x W = the hi-res gradient image (both even), 128-column tiles only.  Needs the split-half channel constraints
// and either the patch geometry or a raster run whose halo fits the LDS image (on the OUTPUT image: H/2 x W/2 in mode 1).
// mode 2: forward of [nearest x2 upsample -> conv3x3] as four phase convolutions, H x W = the hi-res output image (both even),
// K % 64 == 0, C % 32 == 0, 4-wave tiles on the low-res image.
EGZ_API int egz_conv3x3_streamed_ok(int B, int H, int W, int C, int K, int mode) {
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || K <= 0 || mode < 0 || mode > 2) return 0;
    if (mode == 2) {     // forward of [upsample x2 -> conv]: H x W = the hi-res OUTPUT image; tiles live on the low-res image
        if (K % 64 != 0 || C % 32 != 0 || (H & 1) || (W & 1)) return 0;
        if (4ull * B * H * W * K >= (1ull << 32)) return 0;
        H >>= 1;
        W >>= 1;
        if (4ull * B * H * W * C >= (1ull << 32)) return 0;
        const bool small2 = (K % 128 == 0);
        const int prow2 = small2 ? 8 : 16, bm2 = small2 ? 128 : 256, hzero2 = small2 ? 255 : 383;
        if (W % 16 == 0 && H % prow2 == 0) return 1;
        return (bm2 + 2 * W + 2 <= hzero2) ? 1 : 0;
    }
    if (!(C % 32 == 0 || (C < 32 && C % 4 == 0))) return 0;            // whole channel blocks, or one zero-padded block
    if (4ull * B * H * W * C >= (1ull << 32)) return 0;
    if (mode == 1) {
        if (K % 128 != 0 || C % 32 != 0 || (H & 1) || (W & 1)) return 0;
        H >>= 1;
        W >>= 1;
    }
    if (4ull * B * H * W * K >= (1ull << 32)) return 0;                // the result leaves through 32-bit buffer offsets
    // column tile: 128 (K % 128 == 0), 64 (K % 64 == 0), else 32-column tiles padded up to K (narrow layers)
    const bool small = (K % 128 == 0);                                  // 128 x 128 tile: 8 x 16 patches, 256 slots
    const int prow = small ? 8 : 16, bm = small ? 128 : 256, hzero = small ? 255 : 383;
    if (W % 16 == 0 && H % prow == 0) return 1;
    return (bm + 2 * W + 2 <= hzero) ? 1 : 0;
}

// rows of the fp64 partial-sum buffer (stat_partial) of an epi 2 / 5 launch of egz_conv3x3_fwd_streamed (mode 0): one per
// 128 output pixels, except on the persistent narrow kernel (C, K <= 32, H and W multiples of 16), whose blocks each
// write one row.
EGZ_API int egz_conv3x3_streamed_stat_rows(int B, int H, int W, int C, int K) {
    if (x3p_narrow_ok(B, H, W, C, K)) return x3p_narrow_blocks(B, H, W);
    return (int)(((long)B * H * W + 127) / 128);
}

// Fragment-ordered split packing for the streamed kernel.  kind 4: forward of a (K, C, 3, 3) weight (GEMM columns = K,
// reduction = C); kind 5: its data gradient (columns = C, reduction = K, taps flipped); kind 6: the data gradient of
// [upsample x2 -> conv] w.r.t. the low-res input (columns = C, reduction = K, 16 polyphase taps).  dtype 1 = f16 (values
// pre-scaled by 2^10), 2 = bf16.  kind 7: the forward of [upsample x2 -> conv] as four phase convolutions (columns = K,
// reduction = C, [phase][channel block][2 x 2 pre-summed taps]).  wq needs egz_pack_w3x3_elems(C, K, kind >= 6) * 4 bytes, like
// the plane-ordered packings.
EGZ_API int egz_pack_w3x3_split_frag(const float* w, void* wq, int C, int K, int kind, int dtype, hipStream_t st) {
    EGZ_CHECK_ARG(w && wq && C > 0 && K > 0 && kind >= 4 && kind <= 7 && (dtype == 1 || dtype == 2),
                  "egz_pack_w3x3_split_frag: bad arguments");
    const int Cp = (C + 31) / 32 * 32, Kp = (K + 31) / 32 * 32;
    const int Np = (kind == 4 || kind == 7) ? Kp : Cp, Rp = (kind == 4 || kind == 7) ? Cp : Kp;
    const long n = (long)Np * Rp;
    const int g = egz_cdiv(n, 256) > 4096 ? 4096 : egz_cdiv(n, 256);
    unsigned short* o = static_cast<unsigned short*>(wq);
    if (dtype == 1) hipLaunchKernelGGL(pack_split_frag_kernel<_Float16>, dim3(g), dim3(256), 0, st, w, o, C, K, kind, Np, Rp, F16_WSCALE);
    else            hipLaunchKernelGGL(pack_split_frag_kernel<__bf16>, dim3(g), dim3(256), 0, st, w, o, C, K, kind, Np, Rp, 1.f);
    EGZ_CHECK_LAUNCH("egz_pack_w3x3_split_frag");
    return 0;
}

// Blocks one weight tensor takes in an egz_pack_w3x3_frag_batch launch: one per 32 x 32 (output, input channel) tile.
EGZ_API int egz_pack_w3x3_frag_blocks(int C, int K) { return ((C + 31) / 32) * ((K + 31) / 32); }

// Rebuild many fragment-ordered packings in one launch (what an optimizer step leaves stale: ~80 packings per SP step).
// table: n rows of 8 x int64 in DEVICE memory: [w pointer, wq pointer, C, K, kind (4..7), dtype (1 f16 / 2 bf16), first block,
// unused]; first block = running sum of egz_pack_w3x3_frag_blocks over the rows before; total_blocks = the sum over all rows.
// Every wq as for egz_pack_w3x3_split_frag, 16-byte aligned; the result is bit-identical to n calls of it.
EGZ_API int egz_pack_w3x3_frag_batch(const long long* table, int n, int total_blocks, hipStream_t st) {
    EGZ_CHECK_ARG(table && n > 0 && total_blocks > 0, "egz_pack_w3x3_frag_batch: bad arguments");
    hipLaunchKernelGGL(pack_frag_batch_kernel, dim3(total_blocks), dim3(256), 0, st, table, n);
    EGZ_CHECK_LAUNCH("egz_pack_w3x3_frag_batch");
    return 0;
}

// 3x3 conv in split-half arithmetic with streamed fragment-ordered weights.  GEMM view as egz_conv3x3_fwd_split:
// x [B][H][W][C] (the gathered operand), y [B][H'][W'][K].  mode 0: plain conv (forward: kind-4 packing; data gradient: the
// caller passes dy as x, C = Cout, K = Cin and the kind-5 packing), H' x W' = H x W.  mode 1: data gradient of an
// upsample-fused conv w.r.t. its low-res input (x = hi-res dy, kind-6 packing, H' x W' = H/2 x W/2).
// epi: 0 bias, 1 bias + ReLU, 2 bias + per-channel (sum, sumsq) partials in egz_conv3x3_streamed_stat_rows(B, H, W, C, K)
// rows (one per 128 output pixels; one per block on the persistent narrow kernel); 3 (data gradients, 128- / 64-column tiles): y = result where
// mask_src > 0 else 0 (mask_src: [B][H'][W'][K], the post-ReLU activation whose gradient this is), stat rows = per-channel
// sums of the masked result (plane 0; the bias gradient of the layer below), absmax_out (zero-filled by the caller) = max |y|.  Only for geometries egz_conv3x3_streamed_ok accepts.
EGZ_API int egz_conv3x3_fwd_streamed(const float* x, const void* wq, const float* bias, float* y, double* stat_partial,
                                     int B, int H, int W, int C, int K, int epi, int dtype, int mode,
                                     const unsigned int* x_absmax, const float* mask_src, unsigned int* absmax_out,
                                     const float* bn_coef, float* minmax_out, hipStream_t st) {
    EGZ_CHECK_ARG(x && wq && y, "egz_conv3x3_fwd_streamed: null pointer");
    const bool p2 = (dtype & 0x10) != 0;      // two products per MAC (egz_f16p2: x enters hi-only): f16 only, wide tiles; elsewhere three
    dtype &= 0xf;
    EGZ_CHECK_ARG(!p2 || dtype == 1, "egz_conv3x3_fwd_streamed: dtype 0x10 (two products) goes with f16 (dtype 0x11)");
    // mode | 0x100 (mode 0, f16 x3, epi 2, K % 64 == 0, C % 32 == 0): x holds PRE-SPLIT activations (egz_bn_relu_pool_fwd's
    // presplit form; x_absmax = the abs-max the pairs were scaled with)
    const bool pre = (mode & 0x100) != 0;
    mode &= 0xff;
    EGZ_CHECK_ARG(!pre || (mode == 0 && dtype == 1 && (epi == EPI_BIAS_STATS || epi == EPI_BIAS || epi == EPI_BNSUMS) && K % 64 == 0 &&
                           C % 32 == 0 && x_absmax && (!bn_coef || epi == EPI_BNSUMS)),
                  "egz_conv3x3_fwd_streamed: a pre-split operand needs mode 0, dtype 1, epi 0 / 2 / 5, K %% 64 == 0, C %% 32 == 0 and x_absmax");
    // epi 0 / 1 / 2 with bn_coef: x is a pre-BatchNorm tensor, normalised + ReLU'd while it is staged (narrow geometry only);
    // minmax_out (epi 2): narrow geometry: [egz_conv3x3_streamed_stat_rows][2][K] per-channel max / min rows of y;
    // 64- / 128-column tiles (K % 64 == 0): 2 K uints, zero-filled by the caller: order-preserving integer images of the
    // per-channel max of y and of -y (atomic max; egz_bn_finalize's `minmax` argument)
    const bool wide_mm = minmax_out && K % 64 == 0 && K <= 512 && mode == 0 && epi == EPI_BIAS_STATS && !bn_coef;
    EGZ_CHECK_ARG(wide_mm || !((bn_coef && epi != EPI_BNSUMS) || minmax_out) || (mode == 0 && x3p_narrow_ok(B, H, W, C, K) && (dtype == 1 || dtype == 2) &&
                  epi != EPI_MASK_SUMS && (!minmax_out || epi == EPI_BIAS_STATS)),
                  "egz_conv3x3_fwd_streamed: a deferred-BatchNorm input (bn_coef) exists on the narrow persistent kernel only (C, K <= 32, "
                  "H and W multiples of 16); minmax_out needs epi 2 and that geometry or K %% 64 == 0");
    unsigned int* mmw = wide_mm ? reinterpret_cast<unsigned int*>(minmax_out) : nullptr;
    if (epi == EPI_BNSUMS) {       // data gradient + the BatchNorm-backward sums of the layer below
        EGZ_CHECK_ARG((dtype == 1 || dtype == 2) && mode == 0 && mask_src && bn_coef && stat_partial && !bias &&
                      egz_conv3x3_streamed_ok(B, H, W, C, K, 0) && (x3p_narrow_ok(B, H, W, C, K) || K % 64 == 0),
                      "egz_conv3x3_fwd_streamed: epi 5 (BatchNorm sums) needs mode 0, the narrow geometry (C, K <= 32, H and W "
                      "multiples of 16) or K %% 64 == 0, mask_src = the pre-BN conv output of the layer below, bn_coef, "
                      "stat_partial and no bias");
        const unsigned short* w16b = static_cast<const unsigned short*>(wq);
        const float osb = (dtype == 1) ? 1.f / F16_WSCALE : 1.f;
        if (x3p_narrow_ok(B, H, W, C, K)) {
            if (dtype == 1 && p2) return launch_x3p_narrow<egz_f16p2>(epi, x, w16b, nullptr, y, stat_partial, B, H, W, C, K, osb, x_absmax, mask_src, bn_coef, nullptr, st);
            if (dtype == 1) return launch_x3p_narrow<_Float16>(epi, x, w16b, nullptr, y, stat_partial, B, H, W, C, K, osb, x_absmax, mask_src, bn_coef, nullptr, st);
            return launch_x3p_narrow<__bf16>(epi, x, w16b, nullptr, y, stat_partial, B, H, W, C, K, osb, x_absmax, mask_src, bn_coef, nullptr, st);
        }
        // wide tiles: the coefficient rows travel in the kernel's (otherwise unused) bias argument
        // (absmax_out, optional: max |y| -- the gradient's abs-max bounds the BatchNorm backward of the block below)
        if (K % 128 == 0) {
            if (dtype == 1) return p2 ? launch_x3s<egz_f16p2, 1, PLAIN>(epi, x, w16b, bn_coef, y, stat_partial, B, H, W, C, K, osb, x_absmax, mask_src, absmax_out, st, nullptr, pre) : launch_x3s<_Float16, 1, PLAIN>(epi, x, w16b, bn_coef, y, stat_partial, B, H, W, C, K, osb, x_absmax, mask_src, absmax_out, st, nullptr, pre);
            return launch_x3s<__bf16, 1, PLAIN>(epi, x, w16b, bn_coef, y, stat_partial, B, H, W, C, K, osb, x_absmax, mask_src, absmax_out, st);
        }
        if (dtype == 1) return p2 ? launch_x3s<egz_f16p2, 2, PLAIN>(epi, x, w16b, bn_coef, y, stat_partial, B, H, W, C, K, osb, x_absmax, mask_src, absmax_out, st, nullptr, pre) : launch_x3s<_Float16, 2, PLAIN>(epi, x, w16b, bn_coef, y, stat_partial, B, H, W, C, K, osb, x_absmax, mask_src, absmax_out, st, nullptr, pre);
        return launch_x3s<__bf16, 2, PLAIN>(epi, x, w16b, bn_coef, y, stat_partial, B, H, W, C, K, osb, x_absmax, mask_src, absmax_out, st);
    }
    EGZ_CHECK_ARG(egz_conv3x3_streamed_ok(B, H, W, C, K, mode), "egz_conv3x3_fwd_streamed: geometry B=%d H=%d W=%d C=%d K=%d "
                  "mode=%d is not covered (see egz_conv3x3_streamed_ok)", B, H, W, C, K, mode);
    EGZ_CHECK_ARG((dtype == 1 || dtype == 2) && epi >= 0 && epi <= 3, "egz_conv3x3_fwd_streamed: bad dtype / epilogue");
    EGZ_CHECK_ARG(epi < EPI_BIAS_STATS || stat_partial, "egz_conv3x3_fwd_streamed: stats / mask epilogue needs stat_partial");
    EGZ_CHECK_ARG(epi != EPI_MASK_SUMS || (mask_src && absmax_out && !bias), "egz_conv3x3_fwd_streamed: mask epilogue needs "
                  "mask_src and absmax_out and takes no bias");
    const unsigned short* w16 = static_cast<const unsigned short*>(wq);
    const float os = (dtype == 1) ? 1.f / F16_WSCALE : 1.f;
    if (mode == 2) {     // forward of [upsample x2 -> conv] (kind-7 packing): f16 x3 only (the forward arithmetic)
        EGZ_CHECK_ARG(dtype == 1, "egz_conv3x3_fwd_streamed: the upsample forward runs in f16 x3 (dtype 1)");
        if (K % 128 == 0) return launch_x3s<_Float16, 1, UPSF>(epi, x, w16, bias, y, stat_partial, B, H, W, C, K, os, x_absmax, mask_src, absmax_out, st);
        return launch_x3s<_Float16, 2, UPSF>(epi, x, w16, bias, y, stat_partial, B, H, W, C, K, os, x_absmax, mask_src, absmax_out, st);
    }
    if (mode == 1) {
        if (dtype == 1) return p2 ? launch_x3s<egz_f16p2, 1, UPSD>(epi, x, w16, bias, y, stat_partial, B, H, W, C, K, os, x_absmax, mask_src, absmax_out, st) : launch_x3s<_Float16, 1, UPSD>(epi, x, w16, bias, y, stat_partial, B, H, W, C, K, os, x_absmax, mask_src, absmax_out, st);
        return launch_x3s<__bf16, 1, UPSD>(epi, x, w16, bias, y, stat_partial, B, H, W, C, K, os, x_absmax, mask_src, absmax_out, st);
    }
    if (K % 128 == 0) {
        if (dtype == 1) return p2 ? launch_x3s<egz_f16p2, 1, PLAIN>(epi, x, w16, bias, y, stat_partial, B, H, W, C, K, os, x_absmax, mask_src, absmax_out, st, mmw, pre) : launch_x3s<_Float16, 1, PLAIN>(epi, x, w16, bias, y, stat_partial, B, H, W, C, K, os, x_absmax, mask_src, absmax_out, st, mmw, pre);
        return launch_x3s<__bf16, 1, PLAIN>(epi, x, w16, bias, y, stat_partial, B, H, W, C, K, os, x_absmax, mask_src, absmax_out, st, mmw);
    }
    if (K % 64 == 0) {
        if (dtype == 1) return p2 ? launch_x3s<egz_f16p2, 2, PLAIN>(epi, x, w16, bias, y, stat_partial, B, H, W, C, K, os, x_absmax, mask_src, absmax_out, st, mmw, pre) : launch_x3s<_Float16, 2, PLAIN>(epi, x, w16, bias, y, stat_partial, B, H, W, C, K, os, x_absmax, mask_src, absmax_out, st, mmw, pre);
        return launch_x3s<__bf16, 2, PLAIN>(epi, x, w16, bias, y, stat_partial, B, H, W, C, K, os, x_absmax, mask_src, absmax_out, st, mmw);
    }
    EGZ_CHECK_ARG(!(absmax_out && epi == EPI_BIAS_RELU), "egz_conv3x3_fwd_streamed: the abs-max epilogue exists for 64- and "
                  "128-column tiles only (K %% 64 == 0)");
    if (x3p_narrow_ok(B, H, W, C, K) && epi != EPI_MASK_SUMS) {   // persistent narrow form
        if (dtype == 1 && p2 && epi == EPI_BIAS && !bn_coef) return launch_x3p_narrow<egz_f16p2>(epi, x, w16, bias, y, stat_partial, B, H, W, C, K, os, x_absmax, nullptr, bn_coef, minmax_out, st);
        if (dtype == 1) return launch_x3p_narrow<_Float16>(epi, x, w16, bias, y, stat_partial, B, H, W, C, K, os, x_absmax, nullptr, bn_coef, minmax_out, st);
        return launch_x3p_narrow<__bf16>(epi, x, w16, bias, y, stat_partial, B, H, W, C, K, os, x_absmax, nullptr, bn_coef, minmax_out, st);
    }
    if (dtype == 1) return p2 ? launch_x3s<egz_f16p2, 4, PLAIN>(epi, x, w16, bias, y, stat_partial, B, H, W, C, K, os, x_absmax, mask_src, absmax_out, st) : launch_x3s<_Float16, 4, PLAIN>(epi, x, w16, bias, y, stat_partial, B, H, W, C, K, os, x_absmax, mask_src, absmax_out, st);
    return launch_x3s<__bf16, 4, PLAIN>(epi, x, w16, bias, y, stat_partial, B, H, W, C, K, os, x_absmax, mask_src, absmax_out, st);
}

// Split-K form of a PLAIN launch for small pixel counts (batch-1 inference, the 14 x 14 / 28 x 28 layers at small batches):
// egz_conv3x3_streamed_splits recommends the split count (1 = use egz_conv3x3_fwd_streamed) and the workspace holds
// nsplit x B x H x W x K floats.  epi 0 / 1 / 2 as egz_conv3x3_fwd_streamed; K % 128 == 0, C % 32 == 0.
EGZ_API int egz_conv3x3_streamed_splits(int B, int H, int W, int C, int K) {
    if (!egz_conv3x3_streamed_ok(B, H, W, C, K, 0)) return 1;
    return x3s_splits(B, H, W, C, K);
}
// rows of the stat_partial buffer a split-K launch with the BN-statistics epilogue fills (one per 32 output pixels)
EGZ_API int egz_conv3x3_fwd_streamed_splitk_stat_rows(int B, int H, int W) { return egz_cdiv((long)B * H * W, FIX_ROWS); }
EGZ_API size_t egz_conv3x3_fwd_streamed_splitk_ws_bytes(int B, int H, int W, int K, int nsplit) {
    return (size_t)nsplit * B * H * W * K * sizeof(float);
}
EGZ_API int egz_conv3x3_fwd_streamed_splitk(const float* x, const void* wq, const float* bias, float* y, double* stat_partial,
                                            int B, int H, int W, int C, int K, int epi, int dtype,
                                            const unsigned int* x_absmax, void* workspace, size_t ws_bytes, int nsplit,
                                            unsigned int* absmax_out, hipStream_t st) {
    dtype &= 0xf;      // (the two-product bit 0x10 is honoured by egz_conv3x3_fwd_streamed only: split-K launches stay three-product)
    EGZ_CHECK_ARG(x && wq && y && workspace, "egz_conv3x3_fwd_streamed_splitk: null pointer");
    EGZ_CHECK_ARG(egz_conv3x3_streamed_ok(B, H, W, C, K, 0) && K % 128 == 0 && C % 32 == 0,
                  "egz_conv3x3_fwd_streamed_splitk: geometry B=%d H=%d W=%d C=%d K=%d is not covered", B, H, W, C, K);
    EGZ_CHECK_ARG((dtype == 1 || dtype == 2) && epi >= 0 && epi <= 2, "egz_conv3x3_fwd_streamed_splitk: bad dtype / epilogue");
    EGZ_CHECK_ARG(epi != EPI_BIAS_STATS || stat_partial, "egz_conv3x3_fwd_streamed_splitk: stats epilogue needs stat_partial");
    EGZ_CHECK_ARG(nsplit >= 2 && nsplit <= C / 32, "egz_conv3x3_fwd_streamed_splitk: nsplit=%d outside [2, C/32]", nsplit);
    EGZ_CHECK_ARG(ws_bytes >= egz_conv3x3_fwd_streamed_splitk_ws_bytes(B, H, W, K, nsplit),
                  "egz_conv3x3_fwd_streamed_splitk: workspace too small");
    const unsigned short* w16 = static_cast<const unsigned short*>(wq);
    const float os = (dtype == 1) ? 1.f / F16_WSCALE : 1.f;
    float* part = static_cast<float*>(workspace);
    if (dtype == 1) return launch_x3s_splitk<_Float16>(epi, x, w16, bias, y, stat_partial, B, H, W, C, K, os, x_absmax, part, nsplit, absmax_out, st);
    return launch_x3s_splitk<__bf16>(epi, x, w16, bias, y, stat_partial, B, H, W, C, K, os, x_absmax, part, nsplit, absmax_out, st);
}
