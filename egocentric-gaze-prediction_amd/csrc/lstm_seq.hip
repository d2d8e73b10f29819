// The recurrence of the AT module's nn.LSTM (models/LSTMnet.py:18,26-35; gate order i,f,g,o; torch semantics) as ONE
// launch per time step that fuses the recurrent product with the cell:
//
//   forward   gates_t = gx_t + h_{t-1} W_hh^T  ->  (i,f,g,o)  ->  c_t, h_t                       lstm_step_fwd_kernel
//   backward  dh_t = dh_out_t + dgates_{t+1} W_hh ; cell backward -> dgates_t, dc_{t-1}          lstm_step_bwd_kernel
//
// The step is latency-bound (M = batch rows <= 32, 67 MFLOP): what counts is how many CUs pull the 4 MB of W_hh out of
// L2 in parallel and how few dependent launches / passes a step costs.  A dependent kernel boundary costs ~1.5 us on
// MI355X, a device-wide barrier inside a persistent kernel 4-7 us (MI355X_MICROARCH.md, persistent-kernel price list),
// so the sequence is T plain launches issued back to back by ONE C-ABI call (no host round trip per step):
//   * forward: block = 4 hidden units x all four gates = 16 rows of W_hh (N = 16), batch tile M = 16 (x2), K = H split
//     over the 4 waves; 128 blocks for H = 512.  The block owns everything the cell of its 4 units needs, so the
//     point-wise part runs in the epilogue and the pre-activations never touch memory.
//   * backward: block = 16 batch rows x 16 hidden units, K = 4H split over 8 waves; the reduced dh tile feeds the cell
//     backward of exactly those (row, unit) pairs in the epilogue, which emits dgates_t for the next (earlier) step.
// Exact f32 on v_mfma_f32_16x16x4_f32.  Operands go global/L2 -> registers as 16-byte loads: within a group of 16
// reduction indices lane (r = l & 15, q = l >> 4) takes k = 4q .. 4q+3 of ITS row and feeds them to 4 MFMAs; the k <-> MFMA
// pairing is the same for A and B, and a reduction index may be visited in any order.
#include "egz_common.h"

namespace {

__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + expf(-x)); }

constexpr int FW_UNITS = 4;        // hidden units per forward block -> N = 16 gate rows

// gx: [B][4H] of step t (input projection + b_ih + b_hh); w_hh: [4H][H]; h_prev, c_prev: [B][H];
// h_out, c_out: [B][H]; act: [B][4H] activated gates (null in no-grad runs).
template <int MT>
__global__ __launch_bounds__(256) void lstm_step_fwd_kernel(const float* __restrict__ gx, const float* __restrict__ w_hh,
                                                            const float* __restrict__ h_prev,
                                                            const float* __restrict__ c_prev, float* __restrict__ h_out,
                                                            float* __restrict__ c_out, float* __restrict__ act, int B,
                                                            int H) {
    __shared__ float red[4][MT][256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int j0 = blockIdx.x * FW_UNITS, b0 = blockIdx.y * (16 * MT);
    const int kw = H / 4, k0 = wave * kw;                      // this wave's share of the reduction
    const int wrow = (r >> 2) * H + j0 + (r & 3);             // n = gate*4 + unit  ->  row of W_hh
    const float* wp = w_hh + (long)wrow * H + k0 + 4 * q;
    const float* hp[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        int b = b0 + 16 * m + r;
        b = b < B ? b : B - 1;                                 // rows past the batch: valid address, result unused
        hp[m] = h_prev + (long)b * H + k0 + 4 * q;
    }
    f32x4 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int g0 = 0; g0 < kw; g0 += 64) {                      // 4 groups of 16 k per trip: 12 loads in flight per lane
        f32x4 wv[4], hv[MT][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            wv[u] = *reinterpret_cast<const f32x4*>(wp + g0 + 16 * u);
#pragma unroll
            for (int m = 0; m < MT; ++m) hv[m][u] = *reinterpret_cast<const f32x4*>(hp[m] + g0 + 16 * u);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int m = 0; m < MT; ++m)
                    acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(hv[m][u][e], wv[u][e], acc[m], 0, 0, 0);
    }
    // D[row = 4*(lane>>4) + reg][col = lane & 15]: row = batch row of the tile, col = n
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int e = 0; e < 4; ++e) red[wave][m][(4 * q + e) * 16 + r] = acc[m][e];
    __syncthreads();
    // cell: one thread per (batch row, unit)
    for (int i = tid; i < 16 * MT * FW_UNITS; i += 256) {
        const int u = i & 3, row = i >> 2, m = row >> 4, rr = row & 15;
        const int b = b0 + row;
        if (b >= B) continue;
        float pre[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int n = g * 4 + u;
            pre[g] = gx[(long)b * 4 * H + g * H + j0 + u] +
                     ((red[0][m][rr * 16 + n] + red[1][m][rr * 16 + n]) + (red[2][m][rr * 16 + n] + red[3][m][rr * 16 + n]));
        }
        const float gi = sigm(pre[0]), gf = sigm(pre[1]), gg = tanhf(pre[2]), go = sigm(pre[3]);
        const long o = (long)b * H + j0 + u;
        const float c = gf * c_prev[o] + gi * gg;
        c_out[o] = c;
        h_out[o] = go * tanhf(c);
        if (act) {
            float* a = act + (long)b * 4 * H + j0 + u;
            a[0] = gi; a[H] = gf; a[2 * H] = gg; a[3 * H] = go;
        }
    }
}

// One backward step.  dg_next: [B][4H] = dgates_{t+1} (null at the last step -> dh_rec = dh_last or 0);
// w_hh_t: [H][4H] = W_hh transposed; dh_out: [B][H] gradient w.r.t. h_t from the layer above (null: none);
// dh_last: [B][H] gradient w.r.t. the returned h_n (used only when dg_next is null; may be null);
// dc: [B][H] running cell-state gradient, updated in place (dc_init: its value before the first step, may be null = 0);
// act, c, c_prev: saved forward state of step t; dgates: [B][4H] out.
// cell == 0: only the product is formed and written to dh_rec_out (gradient w.r.t. h_0 after the last step).
__global__ __launch_bounds__(512) void lstm_step_bwd_kernel(const float* __restrict__ dg_next,
                                                            const float* __restrict__ w_hh_t,
                                                            const float* __restrict__ dh_out,
                                                            const float* __restrict__ dh_last,
                                                            const float* __restrict__ dc_init, float* __restrict__ dc,
                                                            const float* __restrict__ act, const float* __restrict__ c,
                                                            const float* __restrict__ c_prev, float* __restrict__ dgates,
                                                            float* __restrict__ dh_rec_out, int B, int H, int cell) {
    __shared__ float red[8][256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int j0 = blockIdx.x * 16, b0 = blockIdx.y * 16;
    const int K = 4 * H, kw = K / 8, k0 = wave * kw;
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    if (dg_next) {
        int b = b0 + r;
        b = b < B ? b : B - 1;
        const float* ap = dg_next + (long)b * K + k0 + 4 * q;
        const float* bp = w_hh_t + (long)(j0 + r) * K + k0 + 4 * q;
        for (int g0 = 0; g0 < kw; g0 += 128) {                 // 8 groups of 16 k per trip: 16 loads in flight per lane
            f32x4 av[8], bv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                av[u] = *reinterpret_cast<const f32x4*>(ap + g0 + 16 * u);
                bv[u] = *reinterpret_cast<const f32x4*>(bp + g0 + 16 * u);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][e], bv[u][e], acc, 0, 0, 0);
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) red[wave][(4 * q + e) * 16 + r] = acc[e];
    __syncthreads();
    if (tid < 256) {
        const int row = tid >> 4, j = tid & 15;
        const int b = b0 + row;
        if (b < B) {
            float rec = 0.f;
            if (dg_next) {
                rec = ((red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid])) +
                      ((red[4][tid] + red[5][tid]) + (red[6][tid] + red[7][tid]));
            } else if (dh_last) {
                rec = dh_last[(long)b * H + j0 + j];
            }
            const long o = (long)b * H + j0 + j;
            if (!cell) {
                dh_rec_out[o] = rec;
            } else {
                const float dhv = (dh_out ? dh_out[o] : 0.f) + rec;
                const float dcin = dg_next ? dc[o] : (dc_init ? dc_init[o] : 0.f);
                const float* a = act + (long)b * K + j0 + j;
                const float gi = a[0], gf = a[H], gg = a[2 * H], go = a[3 * H];
                const float tc = tanhf(c[o]);
                const float dct = dcin + dhv * go * (1.f - tc * tc);
                float* d = dgates + (long)b * K + j0 + j;
                d[0] = dct * gg * gi * (1.f - gi);
                d[H] = dct * c_prev[o] * gf * (1.f - gf);
                d[2 * H] = dct * gi * (1.f - gg * gg);
                d[3 * H] = dhv * tc * go * (1.f - go);
                dc[o] = dct * gf;
            }
        }
    }
}

}  // namespace

// nn.LSTM layer forward over T steps.  gx: [T][B][4H] = x_t W_ih^T + b_ih + b_hh for every step (one batched GEMM by the
// caller); w_hh: [4H][H]; h0, c0: [B][H]; hs, cs: [T][B][H] out; acts: [T][B][4H] out (activated gates for the backward
// pass) or null.  T launches on `stream`, no host synchronisation.  H must be a multiple of 256.
EGZ_API int egz_lstm_seq_fwd(const float* gx, const float* w_hh, const float* h0, const float* c0, float* hs, float* cs,
                             float* acts, int T, int B, int H, hipStream_t st) {
    EGZ_CHECK_ARG(gx && w_hh && h0 && c0 && hs && cs, "egz_lstm_seq_fwd: null pointer");
    EGZ_CHECK_ARG(T > 0 && B > 0 && H > 0 && H % 256 == 0, "egz_lstm_seq_fwd: T=%d B=%d H=%d (H must be a multiple of 256)", T, B, H);
    const long bh = (long)B * H;
    for (int t = 0; t < T; ++t) {
        const float* hp = t ? hs + (t - 1) * bh : h0;
        const float* cp = t ? cs + (t - 1) * bh : c0;
        float* a = acts ? acts + t * 4 * bh : nullptr;
        if (B <= 16) {
            hipLaunchKernelGGL(lstm_step_fwd_kernel<1>, dim3(H / FW_UNITS, 1), dim3(256), 0, st, gx + t * 4 * bh, w_hh, hp, cp,
                               hs + t * bh, cs + t * bh, a, B, H);
        } else {
            hipLaunchKernelGGL(lstm_step_fwd_kernel<2>, dim3(H / FW_UNITS, egz_cdiv(B, 32)), dim3(256), 0, st, gx + t * 4 * bh,
                               w_hh, hp, cp, hs + t * bh, cs + t * bh, a, B, H);
        }
    }
    EGZ_CHECK_LAUNCH("egz_lstm_seq_fwd");
    return 0;
}

// Backward through time of the same layer.  dh_out: [T][B][H] gradient w.r.t. every h_t from above (null: none);
// dhn, dcn: [B][H] gradients w.r.t. the returned final state (null: none); acts, cs, c0: forward state;
// w_hh_t: [H][4H] (W_hh transposed, e.g. by egz_nhwc_to_nchw(w_hh, w_hh_t, 1, H, 4H, 1));
// dgates: [T][B][4H] out (pre-activation gradients: the caller forms dW_ih, dW_hh, db and dx from them with batched
// GEMMs); dh0, dc0: [B][H] out.  T + 1 launches on `stream`.
EGZ_API int egz_lstm_seq_bwd(const float* dh_out, const float* dhn, const float* dcn, const float* acts, const float* cs,
                             const float* c0, const float* w_hh_t, float* dgates, float* dh0, float* dc0, int T, int B,
                             int H, hipStream_t st) {
    EGZ_CHECK_ARG(acts && cs && c0 && w_hh_t && dgates && dh0 && dc0, "egz_lstm_seq_bwd: null pointer");
    EGZ_CHECK_ARG(T > 0 && B > 0 && H > 0 && H % 256 == 0, "egz_lstm_seq_bwd: T=%d B=%d H=%d (H must be a multiple of 256)", T, B, H);
    const long bh = (long)B * H;
    const dim3 grid(H / 16, egz_cdiv(B, 16));
    for (int t = T - 1; t >= 0; --t) {
        const float* dgn = (t == T - 1) ? nullptr : dgates + (t + 1) * 4 * bh;
        hipLaunchKernelGGL(lstm_step_bwd_kernel, grid, dim3(512), 0, st, dgn, w_hh_t, dh_out ? dh_out + t * bh : nullptr, dhn,
                           dcn, dc0, acts + t * 4 * bh, cs + t * bh, t ? cs + (t - 1) * bh : c0, dgates + t * 4 * bh,
                           (float*)nullptr, B, H, 1);
    }
    hipLaunchKernelGGL(lstm_step_bwd_kernel, grid, dim3(512), 0, st, dgates, w_hh_t, (const float*)nullptr,
                       (const float*)nullptr, (const float*)nullptr, dc0, (const float*)nullptr, (const float*)nullptr,
                       (const float*)nullptr, (float*)nullptr, dh0, B, H, 0);
    EGZ_CHECK_LAUNCH("egz_lstm_seq_bwd");
    return 0;
}
