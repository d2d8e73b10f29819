// Go / no-go microbenchmark for a FUSED Winograd F(2x2, 3x3) form of the split-half (f16 x3) 3x3 convolution
// ("next round" item 1): stride-1 / pad-1 Conv2d forward of utils.py:64-76 and models/model_SP.py:13-31
// on NHWC fp32 activations.  Standalone: builds its own inputs, packs the transformed weights on the host, checks
// the kernel against an fp64 direct convolution on sampled output pixels and times it with HIP events.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/wino_f23.hip -o tools/micro/wino_f23
//   tools/micro/wino_f23 [B H C K] ...
//
// Kernel design (one 512-thread block per CU, 8 waves = 2 per SIMD):
//   * block tile = R x CC Winograd tiles (<= 64; a tile = 2 x 2 output pixels) x 64 output channels, all 16 transform
//     positions: 16 x 64 x 64 fp32 accumulators = half of the CU's register file.
//   * wave (a, bp), a = transform row 0..3, bp = column pair: positions (a, 2 bp) and (a, 2 bp + 1), each a 64-tile x
//     64-column GEMM over the input channels (2 x 2 MFMA tiles of 32 x 32, f16 x3 = 12 MFMAs per 16 channels).
//   * the raw fp32 input halo of the block goes HBM/L2 -> LDS by LDS-DMA (no registers), one 32-channel block per
//     buffer, double buffered, laid out [4-channel chunk][column parity][row][column / 2][16 B]: the 32 tiles of an MFMA
//     row group read conflict-free ds_read_b128 at stride-2 pixel positions.
//   * every wave builds ITS OWN V = B^T d B fragments from that image: 2 rows x 3 columns of the 4 x 4 patch per lane
//     (the row combination is shared by the wave's two positions), fp32 adds, then the f16 hi / lo split -- the
//     transformed tile never round-trips through LDS.
//   * transformed weights U = G g G^T (x 2^10, split hi / lo) are packed in MFMA fragment order and streamed
//     L2 -> registers like conv3x3_igemm_x3s does.
//   * epilogue: output transform A^T m A through LDS (the 16 positions of a tile live in 8 different waves).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

#define CK(x)                                                                                   \
    do {                                                                                        \
        hipError_t e__ = (x);                                                                   \
        if (e__ != hipSuccess) {                                                                \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e__)); \
            exit(1);                                                                            \
        }                                                                                       \
    } while (0)

#ifndef WINO_DIAG      // timing diagnostics (WRONG RESULTS): 1 no weight loads in the loop, 2 no raw LDS reads, 4 no transform / split,
#define WINO_DIAG 0    // 8 no LDS-DMA in the loop, 16 no barrier in the loop, 32 no MFMAs
#endif
constexpr float WSCALE = 1024.f;
constexpr int PLANE = 256 * 16;            // bytes per (chunk, parity) plane: 256 slots of 16 B
constexpr int IMG = 16 * PLANE;            // one 32-channel image: 8 chunks x 2 parities = 64 KB

__device__ __forceinline__ int acc_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

// 4 floats -> packed hi / lo f16 halves (hi = RTZ so that the residual is exact, lo = RNE of the residual).
// WINO_MIX: the residual v - (float)hi as ONE v_fma_mix_f32 per value (f16 operand widened inside the FMA) instead of
// v_cvt_f32_f16 + v_sub_f32: 2 instead of 3 VALU per value.
#ifndef WINO_RAW2
#define WINO_RAW2 0
#endif
#ifndef WINO_XCDN
#define WINO_XCDN 0
#endif
#ifndef WINO_MIX
#define WINO_MIX 1
#endif
__device__ __forceinline__ void split4(const f32x4 v, u32x2& hi, u32x2& lo) {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const unsigned hp = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(v[2 * e], v[2 * e + 1]));
#if WINO_MIX
        float r0, r1;
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hp), "v"(v[2 * e]));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hp), "v"(v[2 * e + 1]));
        const f16x2 l = __builtin_convertvector(f32x2{r0, r1}, f16x2);
#else
        const f16x2 h = __builtin_bit_cast(f16x2, hp);
        const f16x2 l = __builtin_convertvector(f32x2{v[2 * e] - (float)h[0], v[2 * e + 1] - (float)h[1]}, f16x2);
#endif
        hi[e] = hp;
        lo[e] = __builtin_bit_cast(unsigned, l);
    }
}
__device__ __forceinline__ f32x16 mfma(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// All vector-memory operations of the main loop are inline asm with hand-placed s_waitcnt vmcnt(N): hipcc does not count the
// LDS-DMA operations when it sizes the waits of the weight loads around them (observed: vmcnt(7) ... vmcnt(0) in front of the
// MFMAs of the unit that follows a DMA batch = every weight use waited for the whole 64 KB image to land, once per channel block).
__device__ __forceinline__ u32x4 make_rsrc(const void* p, unsigned bytes) {
    const unsigned long long a = (unsigned long long)p;
    u32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((unsigned)a);
    r[1] = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xFFFFu);
    r[2] = __builtin_amdgcn_readfirstlane(bytes);
    r[3] = 0x00020000u;
    return r;
}
#define VLOAD(dst, vo, rs, so, imm) \
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:" #imm : "=v"(dst) : "v"(vo), "s"(rs), "s"(so))
#define VWAIT2(n, r0, r1) asm volatile("s_waitcnt vmcnt(" #n ")" : "+v"(r0), "+v"(r1))

struct Params {
    const float* x;             // [B][H][W][C]
    const unsigned short* uq;   // [C/32][16 pos][2 ks][K/32][2 planes][64 lanes][8 halves]
    const float* bias;          // [K]
    float* y;                   // [B][H][W][K]
    int B, H, W, C, K;
    int nbands, ncolb, ntn;     // row bands (R stacked tile rows), column blocks (CC tiles), 64-column tiles
    int total;
    float out_scale;
    int relu;
};

// One wave's main loop + epilogue, specialised on the column pair (the only thing that changes the instruction stream).
template <int RP, int CC, int R, int BP>
__device__ __forceinline__ void wino_body(const Params& p, char* lds, const int tid, const int wave) {
    const int lane = tid & 63, kg = lane >> 5, l31 = lane & 31;
    const int a = wave >> 1;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
    const int H = p.H, W = p.W, C = p.C, K = p.K, TH = H >> 1, TW = W >> 1;
    constexpr int NROWS = 2 * R + 4;
    // rows of the block's SECOND image (stacked bands may straddle an image boundary) sit DELTA slots further, so that the slot
    // index keeps following the linear tile index (mod 16) across the two padded rows between the images: conflict-free reads
    constexpr int DELTA = (32 - (2 * RP) % 16) % 16;
    static_assert(NROWS * RP + DELTA <= 256, "image plane holds 256 slots");
    static_assert((2 * RP) % 16 == CC % 16, "row pitch must make the slot index follow the linear tile index (mod 16)");

    // ---- block -> (band, column block, column tile); XCD-aware: block b runs on XCD b % 8, contiguous range per XCD
    const int per = (p.total + 7) >> 3;
    const int gt = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if (gt >= p.total) return;
#if WINO_XCDN          // column-tile-major: the CUs of an XCD work on the same column tile (its U slice stays in that XCD's L2)
    const int npb = p.nbands * p.ncolb;
    const int tile_n = gt / npb;
    const int gm = gt - tile_n * npb;
#else
    const int tile_n = gt % p.ntn;
    const int gm = gt / p.ntn;
#endif
    const int colb = gm % p.ncolb, band = gm / p.ncolb;
    const int srow0 = band * R, b0 = srow0 / TH;
    const int P0 = 2 * srow0 + 2 * b0;                // first stacked padded input row of the block (= LDS row 0)
    const int tx0 = colb * CC;
    const int n0 = tile_n * 64;
    const int ncb = C >> 5;

    // ---- LDS-DMA staging map: wave -> (parity, quarter of the 256 slots); lane -> slot
    const u32x4 x_rs = make_rsrc(p.x, (unsigned)p.B * H * W * C * 4u);
    unsigned dma_vo;
    const int dpar = wave & 1, dq = wave >> 1;
    {
        const int Lb = (b0 + 1) * (H + 2) - P0;               // first LDS row of the second image
        int s = dq * 64 + lane;
        const bool gap = s >= Lb * RP && s < Lb * RP + DELTA;
        if (s >= Lb * RP) s -= DELTA;
        const int row = s / RP, xh = s - row * RP;
        const int Pp = P0 + row, bb = Pp / (H + 2), yy = Pp - bb * (H + 2) - 1;
        const int xx = 2 * tx0 - 1 + 2 * xh + dpar;
        const bool ok = !gap && row < NROWS && xh <= CC && bb < p.B && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
        dma_vo = ok ? (unsigned)((((long)bb * H + yy) * W + xx) * C * 4) : 0xFFFFFFFFu;
    }
    auto dma_image = [&](const int cb, const int buf, const bool on) {
        const unsigned vo = on ? dma_vo : 0xFFFFFFFFu;            // off: every lane out of range -> no memory traffic
#pragma unroll
        for (int ch = 0; ch < 8; ++ch) {
            const unsigned m0v = (unsigned)(size_t)(lds_base + buf * IMG + (ch * 2 + dpar) * PLANE + dq * 1024);
            asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                         :: "s"(m0v), "v"(vo), "s"(x_rs), "s"((unsigned)(cb * 128 + ch * 16)) : "memory");
        }
    };

    // ---- lane -> tile of the two MFMA row groups; LDS byte offsets of patch rows P / Q of transform row a
    //   a = 0: d0 - d2, a = 1: d2 + d1, a = 2: d2 - d1, a = 3: d1 - d3   (t = dP + sgn dQ)
    const int iP = (a == 0) ? 0 : (a == 3) ? 1 : 2, iQ = (a == 0) ? 2 : (a == 3) ? 3 : 1;
    const float sgn = (a == 1) ? 1.f : -1.f;
    int offP[2], offQ[2];
    bool tvalid[2];
    long yoff[2];                                     // output offset of pixel (2 ty, 2 tx) of the lane's tile (epilogue readers use their own)
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        int t = rt * 32 + l31;
        const bool inblk = t < R * CC;
        if (!inblk) t = 0;
        const int r = t / CC, c = t - r * CC;
        const int srow = srow0 + r, bb = srow / TH;
        tvalid[rt] = inblk && srow < p.B * TH && tx0 + c < TW;
        const int lrow = 2 * r + 2 * (bb - b0);
        const int base = kg * (4 * PLANE) + (c + (bb > b0 ? DELTA : 0)) * 16;
        offP[rt] = base + (lrow + iP) * RP * 16;
        offQ[rt] = base + (lrow + iQ) * RP * 16;
        (void)yoff;
    }

    // ---- weight fragments
    const int nt32 = K >> 5;
    const u32x4 u_rs = make_rsrc(p.uq, (unsigned)C * 16u * (unsigned)K * 4u);
    const unsigned u_vo = (unsigned)lane * 16u;
    // byte offset of (cb, pos, ks, ct32 = 2 tile_n): 2 KB per ct32
    auto u_so = [&](const int cb, const int pos, const int ks) -> unsigned {
        return (unsigned)((((cb * 16 + pos) * 2 + ks) * nt32 + tile_n * 2) * 2048);
    };
    u32x4 bh[2][2], bl[2][2];                         // [position][column tile]
    const int pos0 = a * 4 + 2 * BP;
    auto load_b = [&](const int cb, const int ks, const int pp) {
        const unsigned so = u_so(cb, pos0 + pp, ks);
        VLOAD(bh[pp][0], u_vo, u_rs, so, 0);
        VLOAD(bh[pp][1], u_vo, u_rs, so, 2048);
        VLOAD(bl[pp][0], u_vo, u_rs, so, 1024);
        VLOAD(bl[pp][1], u_vo, u_rs, so, 3072);
    };

    f32x16 acc[2][2][2];                              // [position][row group][column tile]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][k][r] = 0.f;

    // ---- V fragments of one unit (k-step ks, row group rt) for both positions
    struct Frag { u32x4 h, l; };
    // raw reads of half `hf` (4 channels): 2 rows x 3 columns j = BP + jj
    auto read_raw = [&](const int bufoff, const int ks, const int rt, const int hf, f32x4* dP, f32x4* dQ) {
#pragma unroll
        for (int jj = 0; jj < 3; ++jj) {
            const int j = BP + jj;
            const int imm = (ks * 4 + hf) * (2 * PLANE) + (j & 1) * PLANE + (j >> 1) * 16;
            dP[jj] = *reinterpret_cast<const f32x4*>(lds + bufoff + offP[rt] + imm);
            dQ[jj] = *reinterpret_cast<const f32x4*>(lds + bufoff + offQ[rt] + imm);
        }
    };
    // transform + split of half hf -> halves [2 hf, 2 hf + 1] of the two positions' fragments, in four pieces that the main
    // loop spreads over the MFMA slots (a bunched transform stalls both waves of a SIMD at the same point of their streams)
    auto xf_t = [&](const f32x4* dP, const f32x4* dQ, f32x4* t) {
#pragma unroll
        for (int jj = 0; jj < 3; ++jj)
#pragma unroll
            for (int e = 0; e < 4; ++e) t[jj][e] = __builtin_fmaf(dQ[jj][e], sgn, dP[jj][e]);
    };
    auto xf_v = [&](const f32x4* t, f32x4& v0, f32x4& v1) {
        if (BP == 0) {                                // b = 0: t0 - t2, b = 1: t1 + t2
            v0 = t[0] - t[2];
            v1 = t[1] + t[2];
        } else {                                      // b = 2: t2 - t1 (cols 1, 2 = jj 0, 1), b = 3: t1 - t3 (jj 0, 2)
            v0 = t[1] - t[0];
            v1 = t[0] - t[2];
        }
    };
    auto xf_s = [&](const f32x4 v, const int hf, Frag& f) {
        u32x2 h, l;
        split4(v, h, l);
        f.h[2 * hf] = h[0]; f.h[2 * hf + 1] = h[1]; f.l[2 * hf] = l[0]; f.l[2 * hf + 1] = l[1];
    };
    auto xform = [&](const f32x4* dP, const f32x4* dQ, const int hf, Frag& f0, Frag& f1) {
        f32x4 t[3], v0, v1;
        xf_t(dP, dQ, t);
        xf_v(t, v0, v1);
        xf_s(v0, hf, f0);
        xf_s(v1, hf, f1);
    };

    // ---- prologue
    load_b(0, 0, 0);
    load_b(0, 0, 1);
    dma_image(0, 0, true);
    dma_image(1, 1, ncb > 1);
    Frag cur0, cur1;
    {
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");       // image 1's DMAs may stay in flight
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        f32x4 dP[3], dQ[3];
        read_raw(0, 0, 0, 0, dP, dQ);
        xform(dP, dQ, 0, cur0, cur1);
        read_raw(0, 0, 0, 1, dP, dQ);
        xform(dP, dQ, 1, cur0, cur1);
    }

    // one channel block = 4 units (k-step, row group); LAST = the final block (nothing to prepare / reload / stage after it)
    auto do_cb = [&](const int cb, auto last_tag) {
        constexpr bool LAST = decltype(last_tag)::value;
        const int bufoff = (cb & 1) * IMG;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int ks = u >> 1, rt = u & 1;
            // the unit being prepared: (cb, ks, rt) + 1
            const int nu = (u + 1) & 3, nks = nu >> 1, nrt = nu & 1;
            const bool wrap = (u == 3);                          // next unit belongs to channel block cb + 1
            const bool more = !(wrap && LAST);
            const int nbufoff = wrap ? (IMG - bufoff) : bufoff;
            if (u == 3) {
                // every wave is done reading image cb & 1 (its last preparation ran in unit 2) and has its share of image
                // cb + 1 in LDS (those DMAs are older than the weight loads waited for since)
#if !(WINO_DIAG & 16)
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");       // (the image's DMAs landed before unit 2's weights)
#endif
            }
            Frag nx0, nx1;
            f32x4 dP[3], dQ[3], tt[3], v0, v1;
#if WINO_RAW2
            f32x4 dP2[3], dQ2[3];
#endif
#if WINO_DIAG & 2
            for (int jj = 0; jj < 3; ++jj) { dP[jj] = __builtin_bit_cast(f32x4, cur0.h); dQ[jj] = __builtin_bit_cast(f32x4, cur1.l); }
#endif
            // weights of the k-step after this one are reloaded in place during the second row group's unit
            const bool reload = (rt == 1) && (ks == 0 || !LAST);
            const int rcb = (ks == 0) ? cb : cb + 1, rks = ks ^ 1;
#pragma unroll
            for (int s = 0; s < 12; ++s) {
                const int pp = s / 6, q = s % 6, ct = q & 1, term = q >> 1;       // term 0: hi hi, 1: lo(A) hi(B), 2: hi(A) lo(B)
                const Frag& f = pp ? cur1 : cur0;
#if !(WINO_DIAG & 1)
                if (rt == 0) {
                    // weight loads of this k-step were issued in the previous unit in the order bh0 bh0 bl0 bl0 bh1 bh1 bl1 bl1, followed
                    // (unit 3 -> next block's unit 0) by the 8 DMAs of the next image, which may stay in flight
                    if (u == 0) {
                        if (s == 0) VWAIT2(14, bh[0][0], bh[0][1]);
                        if (s == 4) VWAIT2(12, bl[0][0], bl[0][1]);
                        if (s == 6) VWAIT2(10, bh[1][0], bh[1][1]);
                        if (s == 10) VWAIT2(8, bl[1][0], bl[1][1]);
                    } else {
                        if (s == 0) VWAIT2(6, bh[0][0], bh[0][1]);
                        if (s == 4) VWAIT2(4, bl[0][0], bl[0][1]);
                        if (s == 6) VWAIT2(2, bh[1][0], bh[1][1]);
                        if (s == 10) VWAIT2(0, bl[1][0], bl[1][1]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
#endif
#if !(WINO_DIAG & 32)
                acc[pp][rt][ct] = mfma(term == 1 ? f.l : f.h, term == 2 ? bl[pp][ct] : bh[pp][ct], acc[pp][rt][ct]);
#endif
                if (more) {
                    // LDS read -> first use: >= 4 MFMA slots (a batch of 6 reads from each of the 8 waves takes ~200 LDS cycles to drain)
#if WINO_RAW2            // two raw register sets: both halves' reads up front
#if !(WINO_DIAG & 2)
                    if (s == 0) read_raw(nbufoff, nks, nrt, 0, dP, dQ);
                    if (s == 1) read_raw(nbufoff, nks, nrt, 1, dP2, dQ2);
#endif
#if !(WINO_DIAG & 4)
                    if (s == 5) xf_t(dP, dQ, tt);
                    if (s == 6) xf_v(tt, v0, v1);
                    if (s == 7) { xf_s(v0, 0, nx0); xf_t(dP2, dQ2, tt); }
                    if (s == 8) xf_s(v1, 0, nx1);
                    if (s == 9) xf_v(tt, v0, v1);
                    if (s == 10) xf_s(v0, 1, nx0);
                    if (s == 11) xf_s(v1, 1, nx1);
#endif
#else
#if !(WINO_DIAG & 2)
                    if (s == 0) read_raw(nbufoff, nks, nrt, 0, dP, dQ);
                    if (s == 5) read_raw(nbufoff, nks, nrt, 1, dP, dQ);
#endif
#if !(WINO_DIAG & 4)
                    if (s == 4 || s == 9) xf_t(dP, dQ, tt);
                    if (s == 5 || s == 10) xf_v(tt, v0, v1);
                    if (s == 6 || s == 10) xf_s(v0, s > 6, nx0);
                    if (s == 7 || s == 11) xf_s(v1, s > 6, nx1);
#endif
#endif
                }
                if (reload && !(WINO_DIAG & 1)) {
                    const unsigned so0 = u_so(rcb, pos0, rks), so1 = u_so(rcb, pos0 + 1, rks);
                    if (s == 4) { VLOAD(bh[0][0], u_vo, u_rs, so0, 0); VLOAD(bh[0][1], u_vo, u_rs, so0, 2048); }
                    if (s == 6) { VLOAD(bl[0][0], u_vo, u_rs, so0, 1024); VLOAD(bl[0][1], u_vo, u_rs, so0, 3072); }
                    if (s == 10) { VLOAD(bh[1][0], u_vo, u_rs, so1, 0); VLOAD(bh[1][1], u_vo, u_rs, so1, 2048); }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (reload && !(WINO_DIAG & 1)) {
                const unsigned so1 = u_so(rcb, pos0 + 1, rks);
                VLOAD(bl[1][0], u_vo, u_rs, so1, 1024);
                VLOAD(bl[1][1], u_vo, u_rs, so1, 3072);
            }
            // image cb + 2 replaces image cb (free since the barrier above).  Issued BEHIND this unit's weight loads: loads
            // complete in order, so a DMA in front of them would have to land before the next unit may touch its weights.
            if (u == 3 && !LAST && !(WINO_DIAG & 8)) dma_image(cb + 2, cb & 1, cb + 2 < ncb);
#if !(WINO_DIAG & 4)
            if (more) { cur0 = nx0; cur1 = nx1; }
#elif !(WINO_DIAG & 2)
            if (more) { cur0.h = __builtin_bit_cast(u32x4, dP[0] + dQ[1]); cur1.l = __builtin_bit_cast(u32x4, dP[2] + dQ[0]); cur0.l = __builtin_bit_cast(u32x4, dP[1]); cur1.h = __builtin_bit_cast(u32x4, dQ[2]); }
#endif
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    for (int cb = 0; cb + 1 < ncb; ++cb) do_cb(cb, std::false_type{});
    do_cb(ncb - 1, std::true_type{});

    // ---- epilogue: output transform through LDS.  Round j (output column parity): every wave reduces its two positions to
    // its share of P[a][j] = sum_b A^T[j][b] M[a][b], writes it as Ex[wave][tile][64 columns]; the readers then sum
    // Y[0][j] = P0 + P1 + P2, Y[1][j] = P1 - P2 - P3 (each Pa = the two column-pair waves of row a).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float* Ex = reinterpret_cast<float*>(lds);
    const __amdgpu_buffer_rsrc_t y_rs = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (int)((unsigned)p.B * H * W * K * 4u), 0x00020000);
#pragma unroll 1
    for (int j = 0; j < 2; ++j) {
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // main-loop reads / previous round's reads done
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float m0 = acc[0][rt][ct][r], m1 = acc[1][rt][ct][r];
                    float v;
                    if (BP == 0) v = (j == 0) ? m0 + m1 : m1;            // b = 0, 1:  j0: M0 + M1,  j1: M1
                    else v = (j == 0) ? m0 : -m0 - m1;                   // b = 2, 3:  j0: M2,       j1: -M2 - M3
                    const int tile = rt * 32 + acc_row(r, lane);
                    Ex[(wave * 64 + tile) * 64 + ct * 32 + l31] = v;
                }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int item = tid + 512 * it, tile = item >> 4, cq = item & 15;
            const int r = tile / CC, c = tile - r * CC;
            const int srow = srow0 + r, bb = srow / TH, ty = srow - bb * TH;
            const bool ok = tile < R * CC && srow < p.B * TH && tx0 + c < TW;
            f32x4 pa[4];
#pragma unroll
            for (int aa = 0; aa < 4; ++aa) {
                const f32x4 e0 = *reinterpret_cast<const f32x4*>(Ex + ((2 * aa) * 64 + tile) * 64 + cq * 4);
                const f32x4 e1 = *reinterpret_cast<const f32x4*>(Ex + ((2 * aa + 1) * 64 + tile) * 64 + cq * 4);
                pa[aa] = e0 + e1;
            }
            const f32x4 bz = *reinterpret_cast<const f32x4*>(p.bias + n0 + cq * 4);
            f32x4 y0 = (pa[0] + pa[1] + pa[2]) * p.out_scale + bz;
            f32x4 y1 = (pa[1] - pa[2] - pa[3]) * p.out_scale + bz;
            if (p.relu) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { y0[e] = fmaxf(y0[e], 0.f); y1[e] = fmaxf(y1[e], 0.f); }
            }
            const long o0 = ((((long)bb * H + 2 * ty) * W + 2 * (tx0 + c) + j) * K + n0 + cq * 4) * 4;
            const unsigned vo0 = ok ? (unsigned)o0 : 0xFFFFFFFFu;
            const unsigned vo1 = ok ? (unsigned)(o0 + (long)W * K * 4) : 0xFFFFFFFFu;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, y0), y_rs, vo0, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, y1), y_rs, vo1, 0, 0);
        }
    }
    (void)tvalid;
}

template <int RP, int CC, int R>
__global__ __launch_bounds__(512, 2) void wino_f23_kernel(const Params p) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (wave & 1) wino_body<RP, CC, R, 1>(p, lds, tid, wave);
    else wino_body<RP, CC, R, 0>(p, lds, tid, wave);
}

// ------------------------------------------------------------------------------------------------ host side
static void split_host(float x, unsigned short& h, unsigned short& l) {
    // RTZ hi (like v_cvt_pkrtz), RNE lo
    _Float16 hi = (_Float16)x;
    if (std::fabs((float)hi) > std::fabs(x)) {        // RNE rounded away from zero: step one ulp toward zero
        unsigned short b;
        memcpy(&b, &hi, 2);
        b -= 1;
        memcpy(&hi, &b, 2);
    }
    const _Float16 lo = (_Float16)(x - (float)hi);
    memcpy(&h, &hi, 2);
    memcpy(&l, &lo, 2);
}

// w [K][C][3][3] -> U fragments [C/32][16][2][K/32][2][64][8]
static void pack_u(const std::vector<float>& w, int C, int K, std::vector<unsigned short>& uq) {
    static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    uq.assign((size_t)C * 16 * K * 2, 0);
    for (int k = 0; k < K; ++k)
        for (int c = 0; c < C; ++c) {
            const float* g = &w[((size_t)k * C + c) * 9];
            for (int a = 0; a < 4; ++a)
                for (int b = 0; b < 4; ++b) {
                    double u = 0;
                    for (int i = 0; i < 3; ++i)
                        for (int j = 0; j < 3; ++j) u += G[a][i] * G[b][j] * (double)g[i * 3 + j];
                    unsigned short h, l;
                    split_host((float)(u * WSCALE), h, l);
                    const int cb = c >> 5, ks = (c >> 4) & 1, kgrp = (c >> 3) & 1, e = c & 7;
                    const int ct32 = k >> 5, col = k & 31, lane = kgrp * 32 + col;
                    const size_t base = ((((size_t)(cb * 16 + a * 4 + b) * 2 + ks) * (K / 32) + ct32) * 2) * 512;
                    uq[base + (size_t)lane * 8 + e] = h;
                    uq[base + 512 + (size_t)lane * 8 + e] = l;
                }
        }
}

struct Geo { int RP, CC, R; };
static Geo pick_geo(int W) {
    const int TW = W / 2;
    if (TW % 8 == 0 || TW < 8) return {12, 8, 8};
    return {15, 14, 4};
}

static void launch(const Params& p, const Geo& g, hipStream_t st) {
    const int grid = ((p.total + 7) / 8) * 8;
    const size_t shm = 2 * IMG;
    if (g.CC == 8) {
        hipLaunchKernelGGL((wino_f23_kernel<12, 8, 8>), dim3(grid), dim3(512), shm, st, p);
    } else {
        hipLaunchKernelGGL((wino_f23_kernel<15, 14, 4>), dim3(grid), dim3(512), shm, st, p);
    }
}

static double run_case(int B, int H, int C, int K, int iters, bool zero) {
    const int W = H;
    const size_t nx = (size_t)B * H * W * C, ny = (size_t)B * H * W * K, nw = (size_t)K * C * 9;
    std::vector<float> hx(nx), hw(nw), hb(K);
    uint64_t s = 0x9E3779B97F4A7C15ull;
    auto rnd = [&]() {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        return (float)((double)(s >> 11) / 9007199254740992.0 * 2.0 - 1.0);
    };
    for (auto& v : hx) v = zero ? 0.f : rnd() * 1.5f;
    for (auto& v : hw) v = zero ? 0.f : rnd() * 0.05f;
    for (auto& v : hb) v = rnd();
    std::vector<unsigned short> huq;
    pack_u(hw, C, K, huq);

    float *dx, *dy, *db;
    unsigned short* du;
    CK(hipMalloc(&dx, nx * 4));
    CK(hipMalloc(&dy, ny * 4));
    CK(hipMalloc(&db, K * 4));
    CK(hipMalloc(&du, huq.size() * 2));
    CK(hipMemcpy(dx, hx.data(), nx * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, hb.data(), K * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(du, huq.data(), huq.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemset(dy, 0xFF, ny * 4));

    const Geo g = pick_geo(W);
    Params p{};
    p.x = dx; p.uq = du; p.bias = db; p.y = dy;
    p.B = B; p.H = H; p.W = W; p.C = C; p.K = K;
    const int TH = H / 2, TW = W / 2;
    p.nbands = (B * TH + g.R - 1) / g.R;
    p.ncolb = (TW + g.CC - 1) / g.CC;
    p.ntn = K / 64;
    p.total = p.nbands * p.ncolb * p.ntn;
    p.out_scale = 1.f / WSCALE;
    p.relu = 0;
    static bool attr_set[2] = {false, false};
    if (!attr_set[g.CC == 8]) {
        if (g.CC == 8) CK(hipFuncSetAttribute((const void*)wino_f23_kernel<12, 8, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * IMG));
        else CK(hipFuncSetAttribute((const void*)wino_f23_kernel<15, 14, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * IMG));
        attr_set[g.CC == 8] = true;
    }
    launch(p, g, 0);
    CK(hipDeviceSynchronize());

    // ---- check sampled pixels against an fp64 direct convolution
    std::vector<float> hy(ny);
    CK(hipMemcpy(hy.data(), dy, ny * 4, hipMemcpyDeviceToHost));
    double maxerr = 0, maxref = 0;
    size_t nbad = 0;
    for (auto v : hy) if (!(v == v)) ++nbad;
    std::vector<int> pix;
    const int cand[] = {0, 1, 2, H / 2 - 1, H / 2, H - 3, H - 2, H - 1, 15, 16, 17, 27, 28};
    for (int b : {0, B / 2, B - 1})
        for (int yy : cand)
            for (int xx : cand)
                if (yy >= 0 && yy < H && xx >= 0 && xx < W) { pix.push_back(b); pix.push_back(yy); pix.push_back(xx); }
    if (!zero)
        for (size_t q = 0; q < pix.size(); q += 3) {
            const int b = pix[q], yy = pix[q + 1], xx = pix[q + 2];
            for (int k = 0; k < K; ++k) {
                double r = hb[k];
                for (int ky = 0; ky < 3; ++ky)
                    for (int kx = 0; kx < 3; ++kx) {
                        const int iy = yy + ky - 1, ix = xx + kx - 1;
                        if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
                        const float* xp = &hx[(((size_t)b * H + iy) * W + ix) * C];
                        const float* wp = &hw[(size_t)k * C * 9 + ky * 3 + kx];
                        for (int c = 0; c < C; ++c) r += (double)xp[c] * (double)wp[(size_t)c * 9];
                    }
                const double got = hy[(((size_t)b * H + yy) * W + xx) * K + k];
                maxerr = std::fmax(maxerr, std::fabs(got - r));
                maxref = std::fmax(maxref, std::fabs(r));
            }
        }

    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) launch(p, g, 0);
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) launch(p, g, 0);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1000.0 / iters;
    const double gflop = 2.0 * B * H * W * (double)K * 9 * C * 1e-9;
    printf("wino B%d %dx%d %d->%d%s  blocks %d (%.2f rounds)  %8.1f us  %7.1f TF algorithmic  rel err %.2e (max|ref| %.3g, NaN %zu)\n",
           B, H, W, C, K, zero ? " ZERO" : "", p.total, p.total / 256.0, us, gflop / us * 1e3, maxref > 0 ? maxerr / maxref : 0.0,
           maxref, nbad);
    fflush(stdout);
    CK(hipFree(dx)); CK(hipFree(dy)); CK(hipFree(db)); CK(hipFree(du));
    return us;
}

int main(int argc, char** argv) {
    if (argc >= 5) {
        for (int i = 1; i + 3 < argc; i += 4) run_case(atoi(argv[i]), atoi(argv[i + 1]), atoi(argv[i + 2]), atoi(argv[i + 3]), 20, false);
        return 0;
    }
    // correctness on awkward shapes first (image boundaries inside a band, partial column blocks)
    run_case(3, 12, 32, 64, 2, false);
    run_case(2, 28, 64, 64, 2, false);
    run_case(5, 14, 64, 128, 2, false);
    run_case(2, 56, 32, 64, 2, false);
    // the go / no-go shapes at B = 32 (direct f16 x3 kernel, r03: 301 / 327 / 351 us)
    run_case(32, 112, 128, 128, 20, false);
    run_case(32, 28, 512, 512, 20, false);
    run_case(32, 224, 64, 64, 20, false);
    run_case(32, 56, 256, 256, 20, false);
    run_case(32, 14, 512, 512, 20, false);
    run_case(32, 28, 512, 512, 20, true);
    return 0;
}
