// The recurrence of the AT module's nn.LSTM (models/LSTMnet.py:18,26-35; gate order i,f,g,o; torch semantics): every launch
// fuses the recurrent product(s) of a time step with its cell,
//
//   forward   gates_t = gx_t (or b + h_below,t W_ih^T) + h_{t-1} W_hh^T  ->  (i,f,g,o)  ->  c_t, h_t         lstm_wave_fwd_kernel
//   backward  dh_t = dh_out_t (or dgates_above,t W_ih_above) + dgates_{t+1} W_hh ; cell backward -> dgates_t, dc_{t-1}
//                                                                                                          lstm_wave_bwd_kernel
// and the launches of the L stacked layers run as a WAVEFRONT (see the kernels).
// A step is latency- and per-CU-bandwidth-bound (M = batch rows <= 32, 67 MFLOP per product, a block pulls 64-512 KB of
// weights and state through one CU's L2 port): what counts is how many CUs pull the 4 MB of W out of L2 in parallel and how few
// dependent launches a sequence costs.  A dependent kernel boundary costs ~1.5 us on MI355X, a device-wide barrier inside a
// persistent kernel 4-7 us (MI355X_MICROARCH.md, persistent-kernel price list), so the sequence is plain launches issued back to
// back by ONE C-ABI call (no host round trip per step):
//   * forward: block = 4 hidden units x all four gates = 16 rows of W (N = 16), batch tile M = 16 (x2), K split over the 4
//     waves; 128 blocks per layer for H = 512.  The block owns everything the cell of its 4 units needs, so the point-wise
//     part runs in the epilogue and the pre-activations never touch memory.  An upper layer's block reduces over
//     [h_below | h] (K = 2H): a forward launch costs ~9.2 us whether its blocks carry K = H or 2H (round 5 measured the
//     projection as blocks of its own one launch ahead: 18 x 9.2 us instead of 17 x 9.4 -- not kept).
//   * backward: block = 16 batch rows x 16 hidden units, K = 4H split over 8 waves; the reduced dh tile feeds the cell backward
//     of exactly those (row, unit) pairs in the epilogue, which emits dgates_t.  Round 5: the product that carries the
//     gradient from the layer above runs in blocks of its own one launch ahead ("product role") -- round 4's lower-layer
//     blocks carried it as a second segment (K = 8H, 512 KB per block) and set the length of every launch: 15.9 -> 10.1 us.
// Exact f32 on v_mfma_f32_16x16x4_f32.  Operands go global/L2 -> registers as 16-byte loads: within a group of 16
// reduction indices lane (r = l & 15, q = l >> 4) takes k = 4q .. 4q+3 of ITS row and feeds them to 4 MFMAs; the k <-> MFMA
// pairing is the same for A and B, and a reduction index may be visited in any order.
// (Rounds 2-3 ran the layers one after the other -- T step launches per layer plus one batched GEMM for each upper layer's input
// projection / input gradient: 0.55 ms of step launches per AT step at T = 16, B = 32, L = 2; the wavefront: 0.50 ms in half
// the launches -- the lower layer's blocks now stream two weight matrices, and a launch lasts as long as its longest block.)
#include "egz_common.h"

namespace {

__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + expf(-x)); }

constexpr int FW_UNITS = 4;        // hidden units per forward block -> N = 16 gate rows

// ---------------------------------------------------------------------------------------------------------------------------
// Wavefront form of the stacked recurrence (nn.LSTM(512, 512, num_layers = L), models/LSTMnet.py:18): launch s runs step
// t = s - l of EVERY layer l at once -- layer l's step t needs h of the layer below at the same t (launch s - 1) and its own
// h at t - 1 (launch s - 1) -- so the T x L dependent step launches of the layer-by-layer form become T + L - 1, and the
// batched input projection of the upper layers (one GEMM each) disappears: a block of layer l > 0 reduces over
// [h_{l-1,t} | h_{l,t-1}] against [W_ih | W_hh] (K = 2H).  Same block shape, MFMA pairing and summation order per segment as
// lstm_step_fwd_kernel; the two segments' partial sums are added segment 0 (W_hh) first.
constexpr int WAVE_MAX_L = 4;
struct WaveFwdLayer {
    const float* gx;       // layer 0: [T][B][4H] = x W_ih^T + b_ih + b_hh of every step; upper layers: null
    const float* bsum;     // upper layers: [4H] = b_ih + b_hh
    const float* w_ih;     // upper layers: [4H][H]
    const float* w_hh;     // [4H][H]
    const float* h0;       // [B][H]
    const float* c0;
    float* hs;             // [T][B][H]; the [B][H] slot in FRONT of it receives a copy of h0 (h_{t-1} of every step is then one
                           // contiguous [T][B][H] block: d W_hh is ONE batched product)
    float* cs;
    float* acts;           // [T][B][4H] or null
    float* hn;             // [B][H]: the returned final state (written by step T - 1)
    float* cn;
};
struct WaveFwd { WaveFwdLayer l[WAVE_MAX_L]; };

template <int MT>
__global__ __launch_bounds__(256) void lstm_wave_fwd_kernel(const WaveFwd a, int s, int T, int B, int H) {
    __shared__ float red[4][MT][256];
    const int layer = blockIdx.z, t = s - layer;
    if (t < 0 || t >= T) return;
    const WaveFwdLayer& p = a.l[layer];
    const long bh = (long)B * H;
    const float* h_prev = t ? p.hs + (t - 1) * bh : p.h0;
    const float* c_prev = t ? p.cs + (t - 1) * bh : p.c0;
    const float* x_in = layer ? a.l[layer - 1].hs + t * bh : nullptr;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int j0 = blockIdx.x * FW_UNITS, b0 = blockIdx.y * (16 * MT);
    const int kw = H / 4, k0 = wave * kw;
    const int wrow = (r >> 2) * H + j0 + (r & 3);
    f32x4 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nseg = layer ? 2 : 1;
    for (int seg = 0; seg < nseg; ++seg) {
        const float* wp = (seg ? p.w_ih : p.w_hh) + (long)wrow * H + k0 + 4 * q;
        const float* src = seg ? x_in : h_prev;
        const float* hp[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            int b = b0 + 16 * m + r;
            b = b < B ? b : B - 1;
            hp[m] = src + (long)b * H + k0 + 4 * q;
        }
        for (int g0 = 0; g0 < kw; g0 += 64) {
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
    }
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int e = 0; e < 4; ++e) red[wave][m][(4 * q + e) * 16 + r] = acc[m][e];
    __syncthreads();
    float* h_out = p.hs + t * bh;
    float* c_out = p.cs + t * bh;
    for (int i = tid; i < 16 * MT * FW_UNITS; i += 256) {
        const int u = i & 3, row = i >> 2, m = row >> 4, rr = row & 15;
        const int b = b0 + row;
        if (b >= B) continue;
        float pre[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int n = g * 4 + u;
            const float base = layer ? p.bsum[g * H + j0 + u] : p.gx[((long)t * B + b) * 4 * H + g * H + j0 + u];
            pre[g] = base + ((red[0][m][rr * 16 + n] + red[1][m][rr * 16 + n]) + (red[2][m][rr * 16 + n] + red[3][m][rr * 16 + n]));
        }
        const float gi = sigm(pre[0]), gf = sigm(pre[1]), gg = tanhf(pre[2]), go = sigm(pre[3]);
        const long o = (long)b * H + j0 + u;
        const float c = gf * c_prev[o] + gi * gg;
        const float hv_ = go * tanhf(c);
        c_out[o] = c;
        h_out[o] = hv_;
        if (t == T - 1) {
            p.hn[o] = hv_;
            p.cn[o] = c;
        }
        if (t == 0) p.hs[o - bh] = h_prev[o];                  // slot -1 = h0
        if (p.acts) {
            float* aa = p.acts + ((long)t * B + b) * 4 * H + j0 + u;
            aa[0] = gi; aa[H] = gf; aa[2 * H] = gg; aa[3 * H] = go;
        }
    }
}

// Backward wavefront (round 5 form).  Every block reduces over K = 4H only: the product that carries the gradient from the layer
// above, dgates_{l+1,t} W_ih_{l+1}, no longer rides in the lower layer's step block (round 4: K = 8H there, 512 KB per block
// through one CU, and a launch lasts as long as its largest block: 15.9-17.6 us against 9.8 for a K = 4H launch) but in blocks of
// its own ("product role") one launch earlier, which park it in dhin[l][t]; the lower layer lags the upper one by TWO launches
// instead of one.  Launch s runs, for every layer l,
//   cell role     step t  = T - 1 - s + 2 (L - 1 - l):  dh_t = [top: dh_out_t | below: dhin_l[t]] + dgates_{l,t+1} W_hh_l, cell backward;
//                 t = -1 is that layer's final product dh_0 = dgates_{l,0} W_hh_l;
//   product role  step t' = T - s + 2 (L - 2 - l)  (l < L - 1):  dhin_l[t'] = dgates_{l+1,t'} W_ih_{l+1}
// -- both read what launch s - 1 wrote.  T + 2 L - 1 launches (19 for T = 16, L = 2) of uniform K = 4H blocks on
// (H / 16) x (B / 16) x (2 L - 1) blocks instead of T + L (18) launches with K = 8H blocks on x L.
struct WaveBwdLayer {
    const float* w_hh_t;     // [H][4H]
    const float* w_ih_t_up;  // W_ih of the layer ABOVE, transposed [H][4H]; null for the top layer
    const float* dh_out;     // top layer: [T][B][H] gradient w.r.t. every h_t (or null); lower layers: null
    const float* dhn;        // [B][H] or null
    const float* dcn;        // [B][H] or null
    const float* acts;       // [T][B][4H]
    const float* cs;         // [T][B][H]
    const float* c0;         // [B][H]
    float* dgates;           // [T][B][4H] out
    float* dh0;              // [B][H] out
    float* dc;               // [B][H] running cell-state gradient = dc0 out
    float* dhin;             // lower layers: [T][B][H] = dgates_above,t W_ih_above (written by the product role); top: null
};
struct WaveBwd { WaveBwdLayer l[WAVE_MAX_L]; };

__global__ __launch_bounds__(512) void lstm_wave_bwd_kernel(const WaveBwd a, int s, int L, int T, int B, int H) {
    __shared__ float red[8][256];
    const bool prod = (int)blockIdx.z >= L;                    // block-uniform role
    const int layer = prod ? (int)blockIdx.z - L : (int)blockIdx.z;
    const int t = prod ? T - s + 2 * (L - 2 - layer) : T - 1 - s + 2 * (L - 1 - layer);
    if (prod ? (t < 0 || t >= T) : (t < -1 || t >= T)) return;
    const WaveBwdLayer& p = a.l[layer];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int j0 = blockIdx.x * 16, b0 = blockIdx.y * 16;
    const int K = 4 * H, kw = K / 8, k0 = wave * kw;
    const long bk = (long)B * K, bh = (long)B * H;
    const bool cell = t >= 0;
    // the one product of this block: [16 batch rows][4H] x [16 hidden units][4H]^T
    const float* src = prod ? a.l[layer + 1].dgates + (long)t * bk                       // dgates of the layer above, step t
                            : ((t < T - 1) ? p.dgates + (long)(t + 1) * bk : nullptr);     // own layer, step t + 1 (t = -1: step 0)
    const float* wsrc = prod ? p.w_ih_t_up : p.w_hh_t;
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    if (src) {
        int b = b0 + r;
        b = b < B ? b : B - 1;
        const float* ap = src + (long)b * K + k0 + 4 * q;
        const float* bp = wsrc + (long)(j0 + r) * K + k0 + 4 * q;
        for (int g0 = 0; g0 < kw; g0 += 128) {
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
            float rec = ((red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid])) +
                        ((red[4][tid] + red[5][tid]) + (red[6][tid] + red[7][tid]));
            const long o = (long)b * H + j0 + j;
            if (prod) {
                p.dhin[(long)t * bh + o] = rec;
                return;
            }
            if (!src && p.dhn) rec += p.dhn[o];                     // last step: the gradient w.r.t. the returned h_n
            if (!cell) {
                p.dh0[o] = rec;
            } else {
                const float dhv = (p.dh_out ? p.dh_out[(long)t * bh + o] : 0.f) + (p.dhin ? p.dhin[(long)t * bh + o] : 0.f) + rec;
                const float dcin = src ? p.dc[o] : (p.dcn ? p.dcn[o] : 0.f);
                const float* aa = p.acts + ((long)t * B + b) * K + j0 + j;
                const float gi = aa[0], gf = aa[H], gg = aa[2 * H], go = aa[3 * H];
                const float cv = p.cs[(long)t * bh + o];
                const float cpv = t ? p.cs[(long)(t - 1) * bh + o] : p.c0[o];
                const float tc = tanhf(cv);
                const float dct = dcin + dhv * go * (1.f - tc * tc);
                float* d = p.dgates + ((long)t * B + b) * K + j0 + j;
                d[0] = dct * gg * gi * (1.f - gi);
                d[H] = dct * cpv * gf * (1.f - gf);
                d[2 * H] = dct * gi * (1.f - gg * gg);
                d[3 * H] = dhv * tc * go * (1.f - go);
                p.dc[o] = dct * gf;
            }
        }
    }
}

}  // namespace

// The stacked recurrence of nn.LSTM(H, H, num_layers = L) as a wavefront over (layer, step): T + L - 1 launches (see
// lstm_wave_fwd_kernel).  Host arrays of L device pointers: w_ih[l] / bsum[l] ([4H][H] / [4H] = b_ih + b_hh; entry 0 unused:
// layer 0's input projection arrives as gx0 [T][B][4H], bias included), w_hh[l] [4H][H]; h0, c0: [L][B][H];
// hs: [L][T + 1][B][H] out -- slot 0 of a layer = a copy of its h0, slots 1 .. T = h_1 .. h_T (so h_{t-1} for t = 1 .. T is the
// contiguous block of slots 0 .. T - 1); cs: [L][T][B][H] out; acts: [L][T][B][4H] out or null; hn, cn: [L][B][H] out (the final
// state).  1 <= L <= 4, H % 256 == 0.
EGZ_API int egz_lstm_wave_fwd(const float* gx0, const float* const* w_ih, const float* const* w_hh, const float* const* bsum,
                              const float* h0, const float* c0, float* hs, float* cs, float* acts, float* hn, float* cn, int L,
                              int T, int B, int H, hipStream_t st) {
    EGZ_CHECK_ARG(gx0 && w_ih && w_hh && bsum && h0 && c0 && hs && cs && hn && cn, "egz_lstm_wave_fwd: null pointer");
    EGZ_CHECK_ARG(L >= 1 && L <= WAVE_MAX_L && T > 0 && B > 0 && H > 0 && H % 256 == 0,
                  "egz_lstm_wave_fwd: L=%d T=%d B=%d H=%d (1 <= L <= 4, H a multiple of 256)", L, T, B, H);
    WaveFwd a;
    const long bh = (long)B * H;
    for (int l = 0; l < L; ++l) {
        EGZ_CHECK_ARG(w_hh[l] && (l == 0 || (w_ih[l] && bsum[l])), "egz_lstm_wave_fwd: null weight pointer (layer %d)", l);
        a.l[l] = WaveFwdLayer{l ? nullptr : gx0, l ? bsum[l] : nullptr, l ? w_ih[l] : nullptr, w_hh[l], h0 + l * bh, c0 + l * bh,
                              hs + ((long)l * (T + 1) + 1) * bh, cs + (long)l * T * bh, acts ? acts + (long)l * T * 4 * bh : nullptr,
                              hn + l * bh, cn + l * bh};
    }
    for (int s = 0; s < T + L - 1; ++s) {
        if (B <= 16) hipLaunchKernelGGL(lstm_wave_fwd_kernel<1>, dim3(H / FW_UNITS, 1, L), dim3(256), 0, st, a, s, T, B, H);
        else         hipLaunchKernelGGL(lstm_wave_fwd_kernel<2>, dim3(H / FW_UNITS, egz_cdiv(B, 32), L), dim3(256), 0, st, a, s, T, B, H);
    }
    EGZ_CHECK_LAUNCH("egz_lstm_wave_fwd");
    return 0;
}

// Backward through time of the same stack: T + 2 L - 1 launches (see lstm_wave_bwd_kernel).  dh_top: [T][B][H] gradient w.r.t. the top
// layer's outputs (or null); dhn, dcn: [L][B][H] gradients w.r.t. the returned final state (or null); acts, cs, c0 as written by /
// passed to the forward; host arrays of L device pointers: w_hh_t[l] = W_hh_l transposed [H][4H], w_ih_t[l] = W_ih_l transposed
// [H][4H] (entry 0 unused); dgates: [L][T][B][4H] out (the caller forms dW_ih, dW_hh, db and the input gradient from them); dh0,
// dc0: [L][B][H] out; dhin: [L - 1][T][B][H] scratch (the gradient each lower layer receives from the layer above; unused at L = 1).
EGZ_API int egz_lstm_wave_bwd(const float* dh_top, const float* dhn, const float* dcn, const float* acts, const float* cs,
                              const float* c0, const float* const* w_hh_t, const float* const* w_ih_t, float* dgates, float* dh0,
                              float* dc0, float* dhin, int L, int T, int B, int H, hipStream_t st) {
    EGZ_CHECK_ARG(acts && cs && c0 && w_hh_t && w_ih_t && dgates && dh0 && dc0 && (dhin || L == 1), "egz_lstm_wave_bwd: null pointer");
    EGZ_CHECK_ARG(L >= 1 && L <= WAVE_MAX_L && T > 0 && B > 0 && H > 0 && H % 256 == 0,
                  "egz_lstm_wave_bwd: L=%d T=%d B=%d H=%d (1 <= L <= 4, H a multiple of 256)", L, T, B, H);
    WaveBwd a;
    const long bh = (long)B * H;
    for (int l = 0; l < L; ++l) {
        EGZ_CHECK_ARG(w_hh_t[l] && (l == L - 1 || w_ih_t[l + 1]), "egz_lstm_wave_bwd: null weight pointer (layer %d)", l);
        a.l[l] = WaveBwdLayer{w_hh_t[l], l < L - 1 ? w_ih_t[l + 1] : nullptr, l == L - 1 ? dh_top : nullptr,
                              dhn ? dhn + l * bh : nullptr, dcn ? dcn + l * bh : nullptr, acts + (long)l * T * 4 * bh,
                              cs + (long)l * T * bh, c0 + l * bh, dgates + (long)l * T * 4 * bh, dh0 + l * bh, dc0 + l * bh,
                              l < L - 1 ? dhin + (long)l * T * bh : nullptr};
    }
    const dim3 grid(H / 16, egz_cdiv(B, 16), 2 * L - 1);
    for (int s = 0; s < T + 2 * L - 1; ++s) hipLaunchKernelGGL(lstm_wave_bwd_kernel, grid, dim3(512), 0, st, a, s, L, T, B, H);
    EGZ_CHECK_LAUNCH("egz_lstm_wave_bwd");
    return 0;
}

namespace {
// ---------------------------------------------------------------------------------------------------------------------------
// Persistent, weight-stationary form of the forward wavefront for the AT network's geometry (nn.LSTM(512, 512, num_layers = 2),
// models/LSTMnet.py:18; batch <= 32): ONE launch for the whole (layer, step) wavefront instead of T + 1.
//   * grid = 128 unit slices x (1 or 2) batch tiles of 16 rows; a block owns 4 hidden units (x 4 gates = 16 weight rows) of
//     BOTH layers for its 16 batch rows.  Its 16 x (512 + 1024) weights live in REGISTERS for the whole launch: wave w of the 8
//     holds reduction indices 64 w .. 64 w + 63 of each 512-wide segment, 48 VGPRs per lane -- the MFMA B operand never moves
//     again (12 MB of weights = 48 KB per block, x 2 batch tiles).
//   * global step s runs layer 0's step t = s and layer 1's step t = s - 1; both read what step s - 1 published: h0_{s-1}
//     (feeds W_hh_l0 and W_ih_l1: loaded once) and h1_{s-2}.  16 rows x 512 x 2 = 64 KB per block and step, the all-to-all
//     that a step launch performs through the kernel boundary.
//   * hand-off inside the launch (cdna_hip_programming.md, Guideline 16, R1): h is stored write-through (sc1, 16 bytes per
//     lane), every storing wave drains vmcnt, one lane adds 1 to the arrival counter of its (batch tile, shard) -- 8 shards per
//     tile, one per XCD under round-robin dispatch, 16 arrivals each -- consumers poll the 8 counters from 8 lanes (relaxed,
//     s_sleep), then read h with sc1 loads (no fence on either side).  The two batch tiles never talk to each other.
//   * the cell state stays in a register of the epilogue lane that owns the cell; cs / acts (only read after the launch) are
//     stored AFTER the arrival.
// Every block must be resident for the counters to fill: 128 or 256 blocks of 512 threads at 142 VGPRs = one block per CU, so the
// launch needs that many CUs to come free (work of other streams drains by itself; a SECOND persistent launch of another process
// on the same device could interleave with this one).  A poll that sees no progress for 0.5 s of wall clock therefore gives up, raises the
// error word and turns everything the block produces from then on into NaN (which reaches h_n / c_n, every later h_t and, in the
// backward kernel, every later gradient) -- loud, instead of hanging the queue.
typedef unsigned int u32x4p __attribute__((ext_vector_type(4)));
constexpr int PF_H = 512, PF_SHARDS = 8, PF_LINE = 32;       // counters one per 128-byte line
// A poll that sees no progress for PF_TIMEOUT_TICKS of the 100 MHz wall clock (0.5 s; the clock, not a spin count: a poll iteration
// takes as long as the memory system lets it) gives up.  Besides the per-launch error word at sync[1024] (zeroed by every call)
// the block raises a STICKY word behind the scratch -- sync[PF_STICKY_OFF] for a forward launch, [PF_STICKY_OFF + 1] for a
// backward launch (atomic max of 1 + step) -- that no call ever clears: the owner of the scratch zeroes it once and reads it at a
// synchronisation point of its own (hipops.lstm_persist_check: the AT loss-ring drain), so a failed forward launch is not
// overwritten by the backward launch that follows it, nor by the next replay of a captured step.
constexpr unsigned long long PF_TIMEOUT_TICKS = 50000000ull;
// -DEGZ_PERSIST_TRACE (tools/lstm_persist_trace.py, a variant build): thread 0 of block (0, 0) stamps the 100 MHz wall clock at the
// phase boundaries of every global step into the tail of the sync scratch ([step][8] x 64 bit from word 1280).
// -DEGZ_PERSIST_ACQ (A/B only): the consumer side as "one agent-scope acquire after the poll, then plain loads" instead of sc1 loads.
#ifdef EGZ_PERSIST_ACQ
#define PF_LOAD_AUX 0
#define PF_ACQUIRE() __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent")
#else
#define PF_LOAD_AUX 16
#define PF_ACQUIRE() do {} while (0)
#endif
#ifdef EGZ_PERSIST_TRACE
constexpr int PF_TRACE_WORDS = 64 * 8 * 2;
#define PF_TRACE(ph)                                                                                              \
    do {                                                                                                          \
        if (tid == 0 && blockIdx.x == 0 && blockIdx.y == 0 && s < 64)                                             \
            reinterpret_cast<unsigned long long*>(a.sync + 5 * PF_SHARDS * PF_LINE)[s * 8 + (ph)] = wall_clock64(); \
    } while (0)
#else
constexpr int PF_TRACE_WORDS = 0;
#define PF_TRACE(ph) do {} while (0)
#endif
constexpr int PF_ZERO_WORDS = 5 * PF_SHARDS * PF_LINE + PF_TRACE_WORDS;      // counters, error word, tickets (+ trace): zeroed by every call
constexpr int PF_STICKY_OFF = PF_ZERO_WORDS + 4 * 2 * 4 * PF_H;             // behind the backward form's partial sums (4 tiles x 8 shards; the forward uses the first half)
struct PersistFwd {
    const float* gx0;                       // [T][B][4H] layer 0's input projection x W_ih_l0^T, WITHOUT bias
    const float* w_hh0; const float* w_ih1; const float* w_hh1;      // [4H][H]
    const float* b_ih[2]; const float* b_hh[2];                      // [4H] each; their sum is formed once per cell lane
    const float* h0; const float* c0;       // [2][B][H]
    float* hs;                              // [2][T + 1][B][H], slot 0 of a layer receives its h0
    float* cs;                              // [2][T][B][H]
    float* acts;                            // [2][T][B][4H] or null
    float* hn; float* cn;                   // [2][B][H]
    unsigned int* sync;                     // [2 tiles][8 shards][32] arrival counters; the error word at [1024]
};

__global__ __launch_bounds__(512) void lstm_persist_fwd_kernel(const PersistFwd a, int T, int B) {
    constexpr int H = PF_H;
    __shared__ float red[2][8][256];
    __shared__ int dead;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int unit_blk = blockIdx.x, tile = blockIdx.y;
    const int j0 = unit_blk * 4, b0 = tile * 16;
    const long bh = (long)B * H;
    if (tid == 0) dead = 0;
    // --- the block's weights: row n = r of the 16 (gate r >> 2, unit r & 3), k = 64 wave + 16 u + 4 q .. + 3
    const long wrow = (long)((r >> 2) * H + j0 + (r & 3)) * H + 64 * wave + 4 * q;
    f32x4 w0[4], wi[4], wh[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        w0[u] = *reinterpret_cast<const f32x4*>(a.w_hh0 + wrow + 16 * u);
        wi[u] = *reinterpret_cast<const f32x4*>(a.w_ih1 + wrow + 16 * u);
        wh[u] = *reinterpret_cast<const f32x4*>(a.w_hh1 + wrow + 16 * u);
    }
    // --- A operand addressing: row b0 + r of a [B][H] slot of hs, same k as the weights
    int brow = b0 + r;
    brow = brow < B ? brow : B - 1;
    const unsigned a_vo = (unsigned)((brow * H + 64 * wave + 4 * q) * 4);
    const long hs_layer = (long)(T + 1) * bh;                  // floats per layer of hs
    const __amdgpu_buffer_rsrc_t hs_rs = __builtin_amdgcn_make_buffer_rsrc(a.hs, 0, (int)(2 * hs_layer * 4), 0x00020000);
    // --- epilogue role: threads 0..127 own one cell each: layer = tid >> 6, row = (tid & 63) >> 2, unit = tid & 3
    const bool epi = tid < 128;
    const int elayer = tid >> 6, erow = (tid & 63) >> 2, eu = tid & 3;
    const int eb = b0 + erow;
    const bool evalid = epi && eb < B;
    const long eo = (long)(eb < B ? eb : B - 1) * H + j0 + eu;
    float c_state = 0.f, bs[4] = {0.f, 0.f, 0.f, 0.f};
    if (epi) {
        c_state = a.c0[elayer * bh + eo];
#pragma unroll
        for (int g = 0; g < 4; ++g) bs[g] = a.b_ih[elayer][g * H + j0 + eu] + a.b_hh[elayer][g * H + j0 + eu];
        // slot 0 of each layer = its h0: published like a step (arrival below), so step 0 reads it like any other
        const float hv = a.h0[elayer * bh + eo];
        if (evalid) __hip_atomic_store(a.hs + elayer * hs_layer + eo, hv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    unsigned int* cnt = a.sync + (tile * PF_SHARDS) * PF_LINE;
    unsigned int* my_cnt = cnt + (unit_blk & (PF_SHARDS - 1)) * PF_LINE;
    unsigned int* err = a.sync + 4 * PF_SHARDS * PF_LINE;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_fetch_add(my_cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

    for (int s = 0; s <= T; ++s) {
        const bool run0 = s < T, run1 = s >= 1;
        PF_TRACE(0);
        // layer 0's pre-activation base of this step: in flight while the block waits
        float base[4] = {bs[0], bs[1], bs[2], bs[3]};
        if (epi && elayer == 0 && run0)
#pragma unroll
            for (int g = 0; g < 4; ++g) base[g] = a.gx0[((long)s * B + (eb < B ? eb : B - 1)) * 4 * H + g * H + j0 + eu] + bs[g];
        // --- wait until all 128 blocks of this tile have published step s - 1 (16 arrivals per shard and step, + the h0 one)
        if (wave == 0 && !dead) {
            const unsigned int target = 16u * (unsigned)(s + 1);
            unsigned int spins = 0;
            unsigned long long t_wait = 0;
            bool ok;
            do {
                const unsigned int v = lane < PF_SHARDS ? __hip_atomic_load(cnt + lane * PF_LINE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : target;
                ok = __all(v >= target);
                if (!ok) {
                    __builtin_amdgcn_s_sleep(2);
                    if ((++spins & 255u) == 0) {
                        const unsigned long long now = wall_clock64();
                        if (!t_wait) t_wait = now;
                        else if (now - t_wait > PF_TIMEOUT_TICKS) {
                            if (lane == 0) {
                                dead = 1;
                                __hip_atomic_store(err, 1u + (unsigned)s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                atomicMax(a.sync + PF_STICKY_OFF, 1u + (unsigned)s);
                            }
                            break;
                        }
                    }
                }
            } while (!ok);
            PF_ACQUIRE();
        }
        __syncthreads();
        PF_TRACE(1);
        // --- operands: h0_{s-1} = slot s of layer 0, h1_{s-2} = slot s - 1 of layer 1 (sc1: served from the coherent level)
        f32x4 x0[4], x1[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            x0[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(hs_rs, a_vo + 64u * u, (unsigned)((long)s * bh * 4), PF_LOAD_AUX));
            x1[u] = run1 ? __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                               hs_rs, a_vo + 64u * u, (unsigned)((hs_layer + (long)(s - 1) * bh) * 4), PF_LOAD_AUX))
                         : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0[u][e], w0[u][e], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0[u][e], wi[u][e], acc1, 0, 0, 0);
            }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x1[u][e], wh[u][e], acc1, 0, 0, 0);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            red[0][wave][(4 * q + e) * 16 + r] = acc0[e];
            red[1][wave][(4 * q + e) * 16 + r] = acc1[e];
        }
        __syncthreads();
        PF_TRACE(2);
        // --- cells
        float gi = 0.f, gf = 0.f, gg = 0.f, go = 0.f, hv = 0.f;
        const int t = elayer ? s - 1 : s;
        const bool act = epi && (elayer ? run1 : run0);
        if (act) {
            float pre[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float* p = &red[elayer][0][erow * 16 + g * 4 + eu];
                pre[g] = base[g] + (((p[0] + p[256]) + (p[512] + p[768])) + ((p[1024] + p[1280]) + (p[1536] + p[1792])));
            }
            gi = sigm(pre[0]); gf = sigm(pre[1]); gg = tanhf(pre[2]); go = sigm(pre[3]);
            c_state = gf * c_state + gi * gg;
            hv = go * tanhf(c_state);
            if (dead) hv = c_state = __builtin_nanf("");          // a hand-off never arrived: poison what this launch returns
        }
        PF_TRACE(3);
        // publish h_t: the 4 units of a row sit in 4 neighbouring lanes -> one 16-byte write-through store per row
        if (epi) {
            f32x4 h4;
            h4[0] = __shfl(hv, (lane & ~3) + 0); h4[1] = __shfl(hv, (lane & ~3) + 1);
            h4[2] = __shfl(hv, (lane & ~3) + 2); h4[3] = __shfl(hv, (lane & ~3) + 3);
            if (act && evalid && eu == 0)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4p, h4), hs_rs, (unsigned)((eb * H + j0) * 4),
                                                       (unsigned)((elayer * hs_layer + (long)(t + 1) * bh) * 4), 16);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        PF_TRACE(4);
        if (tid == 0 && s < T) __hip_atomic_fetch_add(my_cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        PF_TRACE(5);
        // what only the backward pass / the caller reads: after the arrival
        if (act && evalid) {
            a.cs[(elayer * (long)T + t) * bh + eo] = c_state;
            if (a.acts) {
                float* aa = a.acts + ((elayer * (long)T + t) * B + eb) * 4 * H + j0 + eu;
                aa[0] = gi; aa[H] = gf; aa[2 * H] = gg; aa[3 * H] = go;
            }
            if (t == T - 1) {
                a.hn[elayer * bh + eo] = hv;
                a.cn[elayer * bh + eo] = c_state;
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------------------
// The backward wavefront of the same network in ONE persistent launch.  The reduction of a backward step runs over the GATE index
// (K = 4H = 2048 per product), its outputs are hidden units -- so what the blocks exchange is dgates, 4x wider than the forward's
// h.  To halve what a block has to pull per step the batch is cut into tiles of 8 rows:
//   * grid = 64 unit slices x ceil(B / 8) batch tiles = 256 blocks at B = 32; a block owns 8 hidden units of BOTH layers for its 8
//     rows: rows 8 units x 2048 of W_hh_l1^T, W_ih_l1^T and W_hh_l0^T = 192 KB of weights in registers (96 VGPRs per lane; one
//     block of 512 threads per CU).
//   * MFMA shape: 8 rows x 8 units per block would waste a 16x16 tile, so the products run on v_mfma_f32_4x4x1 (16 independent
//     4x4 blocks per instruction): block index = 16 consecutive reduction indices, A = 4 batch rows, B = 4 units; a lane
//     (ks = lane >> 2, i = lane & 3) feeds row i / unit i at k = 256 wave + 64 c + 4 ks + e.  The 16 blocks' partial sums are
//     folded across lanes (two DPP row shifts, then LDS), then across the 8 waves.
//   * global step s: layer 1's cell at t = T - 1 - s, layer 0's cell at t = T - s (one behind: it needs dgates_{1,t}); both read
//     what step s - 1 published: G1 = dgates_{1,T-s} (feeds W_hh_l1 for layer 1 AND W_ih_l1 for layer 0: loaded once) and
//     G0 = dgates_{0,T-s+1}: 8 rows x 2048 x 2 = 128 KB per block and step.  Steps T and T + 1 form dh0 of layer 1 / layer 0.
//   * hand-off as in the forward kernel: dgates stored write-through (16 bytes per lane), drained, one arrival per block on the
//     counter of its (tile, shard); the running dc of a cell stays in its epilogue lane's register.
constexpr int PB_SHARDS = 8;
constexpr int PB_TICKET_OFF = 4 * PB_SHARDS * PF_LINE + PF_LINE;      // 64 fold tickets behind the error word's line
constexpr int PB_PART_OFF = 5 * PB_SHARDS * PF_LINE + PF_TRACE_WORDS;  // [4 tiles][2][4H] bias-gradient partials (not zeroed)
struct PersistBwd {
    const float* dh_top;                    // [T][B][H] gradient w.r.t. the top layer's outputs, or null
    const float* dhn; const float* dcn;     // [2][B][H] or null
    const float* acts;                      // [2][T][B][4H]
    const float* cs;                        // [2][T][B][H]
    const float* c0;                        // [2][B][H]
    const float* w_hh0; const float* w_ih1; const float* w_hh1;            // the weights as the module holds them: [4H][H]
    float* dgates;                          // [2][T][B][4H] out
    float* dh0; float* dc0;                 // [2][B][H] out
    float* db[4];                           // bias gradients (b_ih_l0, b_hh_l0, b_ih_l1, b_hh_l1), [4H] each, null = not wanted
    unsigned int* sync;                     // [4 tiles][8 shards][32] arrival counters, the error word's line, fold tickets, partials
};

__device__ __forceinline__ float dpp_row_shr_add(float v, const int shift4) {        // v + (v of the lane `4` or `8` below in the row of 16)
    const int t = shift4 ? __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xf, 0xf, true)
                         : __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xf, 0xf, true);
    return v + __builtin_bit_cast(float, t);
}

__global__ __launch_bounds__(512) void lstm_persist_bwd_kernel(const PersistBwd a, int T, int B) {
    constexpr int H = PF_H, K = 4 * PF_H, RS = 132;            // RS: row stride of the reduction buffer (128 outputs + pad)
    __shared__ float red[32 * RS];
    __shared__ float dbl[4][128];           // running sums over the steps of this block's dgates (bias gradients), per cell lane
    __shared__ int dead;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ks = lane >> 2, li = lane & 3;
    if (tid < 128) dbl[0][tid] = dbl[1][tid] = dbl[2][tid] = dbl[3][tid] = 0.f;
    const int unit_blk = blockIdx.x, tile = blockIdx.y;
    const int j0 = unit_blk * 8, r0 = tile * 8;
    const long bh = (long)B * H, bk = (long)B * K;
    if (tid == 0) dead = 0;
    // --- the block's weights: unit j0 + 4 ug + li, k = 256 wave + 64 c + 4 ks .. + 3
    f32x4 whh1[2][4], wih1[2][4], whh0[2][4];
#pragma unroll
    for (int ug = 0; ug < 2; ++ug)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                // W^T[unit][k] = W[k][unit], gathered once per launch (4 neighbouring lanes read 16 contiguous bytes of a row)
                const long o = (long)(256 * wave + 64 * c + 4 * ks + e) * H + j0 + 4 * ug + li;
                whh1[ug][c][e] = a.w_hh1[o];
                wih1[ug][c][e] = a.w_ih1[o];
                whh0[ug][c][e] = a.w_hh0[o];
            }
    // --- A operand addressing: rows r0 + 4 g + li of a [B][4H] slot of dgates
    unsigned a_vo[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        int b = r0 + 4 * g + li;
        b = b < B ? b : B - 1;
        a_vo[g] = (unsigned)((b * K + 256 * wave + 4 * ks) * 4);
    }
    const __amdgpu_buffer_rsrc_t dg_rs = __builtin_amdgcn_make_buffer_rsrc(a.dgates, 0, (int)(2 * T * bk * 4), 0x00020000);
    // --- epilogue role: thread o < 128 owns one (layer, row, unit): o = layer * 64 + g * 32 + ug * 16 + e * 4 + j
    const bool epi = tid < 128;
    const int el = tid >> 6, erow = 4 * ((tid >> 5) & 1) + ((tid >> 2) & 3), eunit = 4 * ((tid >> 4) & 1) + (tid & 3);
    const int eb = r0 + erow;
    const bool evalid = epi && eb < B;
    const int ebc = eb < B ? eb : B - 1;
    const long eo = (long)ebc * H + j0 + eunit;
    float dc_run = 0.f;
    unsigned int* cnt = a.sync + (tile * PB_SHARDS) * PF_LINE;
    unsigned int* my_cnt = cnt + (unit_blk & (PB_SHARDS - 1)) * PF_LINE;
    unsigned int* err = a.sync + 4 * PB_SHARDS * PF_LINE;
    __syncthreads();

    for (int s = 0; s <= T + 1; ++s) {
        // this thread's cell (if any) of the step, and everything about it that does not depend on the hand-off
        PF_TRACE(0);
        const int t = el ? T - 1 - s : T - s;
        const bool cell = epi && t >= 0 && t < T;
        const bool fin = epi && t == -1;
        float gi = 0.f, gf = 0.f, gg = 0.f, go = 0.f, cv = 0.f, cpv = 0.f, dh_add = 0.f, dc_in = 0.f;
        if (cell) {
            const float* aa = a.acts + ((long)(el * T + t) * B + ebc) * K + j0 + eunit;
            gi = aa[0]; gf = aa[H]; gg = aa[2 * H]; go = aa[3 * H];
            cv = a.cs[(long)(el * T + t) * bh + eo];
            cpv = t ? a.cs[(long)(el * T + t - 1) * bh + eo] : a.c0[el * bh + eo];
            if (el && a.dh_top) dh_add = a.dh_top[(long)t * bh + eo];
            if (t == T - 1) {
                if (a.dhn) dh_add += a.dhn[el * bh + eo];
                dc_in = a.dcn ? a.dcn[el * bh + eo] : 0.f;
            }
        }
        // --- wait until the 64 blocks of this tile have published step s - 1 (8 arrivals per shard and step)
        if (s && wave == 0 && !dead) {
            const unsigned int target = 8u * (unsigned)s;
            unsigned int spins = 0;
            unsigned long long t_wait = 0;
            bool ok;
            do {
                const unsigned int v = lane < PB_SHARDS ? __hip_atomic_load(cnt + lane * PF_LINE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : target;
                ok = __all(v >= target);
                if (!ok) {
                    __builtin_amdgcn_s_sleep(2);
                    if ((++spins & 255u) == 0) {
                        const unsigned long long now = wall_clock64();
                        if (!t_wait) t_wait = now;
                        else if (now - t_wait > PF_TIMEOUT_TICKS) {
                            if (lane == 0) {
                                dead = 1;
                                __hip_atomic_store(err, 1u + (unsigned)s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                atomicMax(a.sync + PF_STICKY_OFF + 1, 1u + (unsigned)s);
                            }
                            break;
                        }
                    }
                }
            } while (!ok);
            PF_ACQUIRE();
        }
        __syncthreads();
        PF_TRACE(1);
        // --- operands (sc1 loads): G1 = dgates_{1, T - s} for 1 <= s <= T, G0 = dgates_{0, T - s + 1} for 2 <= s <= T + 1
        const bool has1 = s >= 1 && s <= T, has0 = s >= 2;
        f32x4 x1[2][4], x0[2][4];
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                x1[g][c] = has1 ? __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                      dg_rs, a_vo[g] + 256u * c, (unsigned)(((long)(T + T - s)) * bk * 4), PF_LOAD_AUX))
                                : f32x4{0.f, 0.f, 0.f, 0.f};
                x0[g][c] = has0 ? __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                      dg_rs, a_vo[g] + 256u * c, (unsigned)(((long)(T - s + 1)) * bk * 4), PF_LOAD_AUX))
                                : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        f32x4 acc1[2][2], acc0[2][2];
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int ug = 0; ug < 2; ++ug) acc1[g][ug] = acc0[g][ug] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (s) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int g = 0; g < 2; ++g)
#pragma unroll
                        for (int ug = 0; ug < 2; ++ug) {
                            acc1[g][ug] = __builtin_amdgcn_mfma_f32_4x4x1f32(x1[g][c][e], whh1[ug][c][e], acc1[g][ug], 0, 0, 0);
                            acc0[g][ug] = __builtin_amdgcn_mfma_f32_4x4x1f32(x1[g][c][e], wih1[ug][c][e], acc0[g][ug], 0, 0, 0);
                            acc0[g][ug] = __builtin_amdgcn_mfma_f32_4x4x1f32(x0[g][c][e], whh0[ug][c][e], acc0[g][ug], 0, 0, 0);
                        }
        }
        // fold the 16 reduction blocks of the instruction: ks & 3 by two DPP row shifts (lanes 12..15 of every row of 16 end up with
        // the sum of their 4), ks >> 2 and the 8 waves through LDS
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int ug = 0; ug < 2; ++ug)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v1 = dpp_row_shr_add(dpp_row_shr_add(acc1[g][ug][e], 1), 0);
                    float v0 = dpp_row_shr_add(dpp_row_shr_add(acc0[g][ug][e], 1), 0);
                    if ((lane & 15) >= 12) {
                        float* rr = red + (wave * 4 + (lane >> 4)) * RS + g * 32 + ug * 16 + e * 4 + li;
                        rr[64] = v1;
                        rr[0] = v0;
                    }
                }
        __syncthreads();
        PF_TRACE(2);
        float val = 0.f;
        if (epi) {
            float p[32];
#pragma unroll
            for (int i = 0; i < 32; ++i) p[i] = red[i * RS + tid];
#pragma unroll
            for (int i = 0; i < 16; ++i) p[i] += p[i + 16];
#pragma unroll
            for (int i = 0; i < 8; ++i) p[i] += p[i + 8];
#pragma unroll
            for (int i = 0; i < 4; ++i) p[i] += p[i + 4];
            val = (p[0] + p[2]) + (p[1] + p[3]);
            if (dead) val = __builtin_nanf("");                   // a hand-off never arrived: poison every gradient from here on
        }
        float d4[4] = {0.f, 0.f, 0.f, 0.f};
        if (cell) {
            const float dhv = dh_add + val;
            const float dcin = (t == T - 1) ? dc_in : dc_run;
            const float tc = tanhf(cv);
            const float dct = dcin + dhv * go * (1.f - tc * tc);
            d4[0] = dct * gg * gi * (1.f - gi);
            d4[1] = dct * cpv * gf * (1.f - gf);
            d4[2] = dct * gi * (1.f - gg * gg);
            d4[3] = dhv * tc * go * (1.f - go);
            dc_run = dct * gf;
            if (evalid)
#pragma unroll
                for (int g = 0; g < 4; ++g) dbl[g][tid] += d4[g];
        }
        PF_TRACE(3);
        // publish dgates_t: the 4 units tid & 3 of a (row, gate) sit in 4 neighbouring lanes -> one 16-byte write-through store
        if (epi) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v4;
                v4[0] = __shfl(d4[g], (lane & ~3) + 0); v4[1] = __shfl(d4[g], (lane & ~3) + 1);
                v4[2] = __shfl(d4[g], (lane & ~3) + 2); v4[3] = __shfl(d4[g], (lane & ~3) + 3);
                if (cell && evalid && (tid & 3) == 0)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4p, v4), dg_rs,
                                                           (unsigned)((eb * K + g * H + j0 + (eunit & ~3)) * 4),
                                                           (unsigned)(((long)(el * T + t)) * bk * 4), 16);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        PF_TRACE(4);
        if (tid == 0 && s <= T) __hip_atomic_fetch_add(my_cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        PF_TRACE(5);
        if (evalid) {
            if (cell && t == 0) a.dc0[el * bh + eo] = dc_run;
            if (fin) a.dh0[el * bh + eo] = val;
        }
    }
    // --- bias gradients = the sums of dgates over steps and batch rows: the steps are summed in dbl, the block's 8 rows here, the
    // (<= 4) batch tiles by whichever block of the unit slice draws the last ticket -- in tile order, so the result does not
    // depend on who that is.  Partials travel write-through like every other hand-off of the launch.
    if (a.db[0] || a.db[1] || a.db[2] || a.db[3]) {
        float* part = reinterpret_cast<float*>(a.sync + PB_PART_OFF);
        unsigned int* ticket = a.sync + PB_TICKET_OFF + unit_blk;
        const int fl = tid >> 5, fg = (tid >> 3) & 3, fu = tid & 7;            // tid < 64: (layer, gate, unit)
        const int fidx = fl * K + fg * H + j0 + fu;
        __syncthreads();
        if (tid < 64) {
            float sum = 0.f;
#pragma unroll
            for (int r = 0; r < 8; ++r) sum += dbl[fg][fl * 64 + (r >> 2) * 32 + (fu >> 2) * 16 + (r & 3) * 4 + (fu & 3)];
            __hip_atomic_store(part + tile * 2 * K + fidx, sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        if (tid == 0) dead = (int)__hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (dead == (int)gridDim.y - 1 && tid < 64) {
            float sum = 0.f;
            for (int tl = 0; tl < (int)gridDim.y; ++tl)
                sum += __hip_atomic_load(part + tl * 2 * K + fidx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (a.db[2 * fl]) a.db[2 * fl][fg * H + j0 + fu] = sum;
            if (a.db[2 * fl + 1]) a.db[2 * fl + 1][fg * H + j0 + fu] = sum;
        }
    }
}

}  // namespace

// Words of the `sync` scratch of egz_lstm_persist_fwd / _bwd.  The first part is zeroed by every call; word [1024] is the per-launch error word: 0 = every
// hand-off arrived, 1 + s = a block gave up waiting in global step s.  The LAST 32 words are never written by a call except on such a
// time-out (word 0 of them: forward launches, word 1: backward launches; atomic max of 1 + s): the caller zeroes the scratch once
// when it allocates it and reads them when it synchronises.
// every block of a persistent launch needs a CU of its own (the backward kernel's 252 VGPRs leave room for nothing else): refuse a
// device with fewer CUs than blocks instead of relying on the hand-off time-out
static int persist_cus_ok(int blocks) {
    static int cus = -1;
    if (cus < 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 0;
        cus = n;
    }
    return cus >= blocks;
}
constexpr int PF_SYNC_WORDS = PF_STICKY_OFF + PF_LINE;      // ... + the line of the two sticky error words (never zeroed by a call)
EGZ_API int egz_lstm_persist_sync_words(void) { return PF_SYNC_WORDS; }

// The recurrence of egz_lstm_wave_fwd for L = 2, H = 512, B <= 32 in ONE persistent launch (lstm_persist_fwd_kernel).  Differences in
// the arguments: gx0 [T][B][4H] = x W_ih_l0^T WITHOUT bias, and the biases as the module holds them -- b_ih / b_hh: HOST arrays of
// L device pointers ([4H] each) -- summed inside the kernel (no separate b_ih + b_hh launches); w_ih[0] unused.  Outputs as
// egz_lstm_wave_fwd.  `sync`: egz_lstm_persist_sync_words() uints of device scratch; after the launch sync[1024] is 0, or 1 + s if
// a block's wait timed out in step s (outputs then undefined).  Returns hipErrorNotSupported for any other geometry (the caller
// launches the wavefront instead).
EGZ_API int egz_lstm_persist_fwd(const float* gx0, const float* const* w_ih, const float* const* w_hh, const float* const* b_ih,
                                 const float* const* b_hh, const float* h0, const float* c0, float* hs, float* cs, float* acts,
                                 float* hn, float* cn, unsigned int* sync, int L, int T, int B, int H, hipStream_t st) {
    EGZ_CHECK_ARG(gx0 && w_ih && w_hh && b_ih && b_hh && h0 && c0 && hs && cs && hn && cn && sync, "egz_lstm_persist_fwd: null pointer");
    if (L != 2 || H != PF_H || B < 1 || B > 32 || T < 1) {
        egz_set_error("egz_lstm_persist_fwd: L=%d T=%d B=%d H=%d (built for L = 2, H = 512, B <= 32)", L, T, B, H);
        return (int)hipErrorNotSupported;
    }
    if ((long)T * B * 4 * H * 4 * 2 >= (1l << 31)) {        // 32-bit buffer-resource sizes / offsets over acts and gx0
        egz_set_error("egz_lstm_persist_fwd: T=%d B=%d: sequence too long for the 32-bit offsets of the persistent form", T, B);
        return (int)hipErrorNotSupported;
    }
    if (!persist_cus_ok(128 * egz_cdiv(B, 16))) {
        egz_set_error("egz_lstm_persist_fwd: the device has fewer compute units than the launch has blocks (%d)", 128 * egz_cdiv(B, 16));
        return (int)hipErrorNotSupported;
    }
    EGZ_CHECK_ARG(w_hh[0] && w_hh[1] && w_ih[1] && b_ih[0] && b_ih[1] && b_hh[0] && b_hh[1], "egz_lstm_persist_fwd: null weight pointer");
    PersistFwd a{gx0, w_hh[0], w_ih[1], w_hh[1], {b_ih[0], b_ih[1]}, {b_hh[0], b_hh[1]}, h0, c0, hs, cs, acts, hn, cn, sync};
    hipError_t e = hipMemsetAsync(sync, 0, PF_ZERO_WORDS * sizeof(unsigned int), st);
    if (e != hipSuccess) { egz_set_error("egz_lstm_persist_fwd: memset failed: %s", hipGetErrorString(e)); return (int)e; }
    hipLaunchKernelGGL(lstm_persist_fwd_kernel, dim3(128, egz_cdiv(B, 16)), dim3(512), 0, st, a, T, B);
    EGZ_CHECK_LAUNCH("egz_lstm_persist_fwd");
    return 0;
}

// Its backward for the same geometry in ONE persistent launch (lstm_persist_bwd_kernel): egz_lstm_wave_bwd's inputs and outputs
// without dhin, except that the weights come AS THE MODULE HOLDS THEM -- w_hh / w_ih: HOST arrays of L device pointers [4H][H]
// (w_ih[0] unused; no transposed copies) -- and that the launch also forms the bias gradients: db = HOST array of 2 L device
// pointers (b_ih_l0, b_hh_l0, b_ih_l1, b_hh_l1; [4H] each, = the sums of dgates_l over steps and batch rows; null entries are
// skipped, db itself may be null).  `sync` as above.  hipErrorNotSupported for any other geometry.
EGZ_API int egz_lstm_persist_bwd(const float* dh_top, const float* dhn, const float* dcn, const float* acts, const float* cs,
                                 const float* c0, const float* const* w_hh, const float* const* w_ih, float* dgates, float* dh0,
                                 float* dc0, float* const* db, unsigned int* sync, int L, int T, int B, int H, hipStream_t st) {
    EGZ_CHECK_ARG(acts && cs && c0 && w_hh && w_ih && dgates && dh0 && dc0 && sync, "egz_lstm_persist_bwd: null pointer");
    if (L != 2 || H != PF_H || B < 1 || B > 32 || T < 1) {
        egz_set_error("egz_lstm_persist_bwd: L=%d T=%d B=%d H=%d (built for L = 2, H = 512, B <= 32)", L, T, B, H);
        return (int)hipErrorNotSupported;
    }
    if ((long)T * B * 4 * H * 4 * 2 >= (1l << 31)) {
        egz_set_error("egz_lstm_persist_bwd: T=%d B=%d: sequence too long for the 32-bit offsets of the persistent form", T, B);
        return (int)hipErrorNotSupported;
    }
    if (!persist_cus_ok(64 * egz_cdiv(B, 8))) {
        egz_set_error("egz_lstm_persist_bwd: the device has fewer compute units than the launch has blocks (%d)", 64 * egz_cdiv(B, 8));
        return (int)hipErrorNotSupported;
    }
    EGZ_CHECK_ARG(w_hh[0] && w_hh[1] && w_ih[1], "egz_lstm_persist_bwd: null weight pointer");
    PersistBwd a{dh_top, dhn, dcn, acts, cs, c0, w_hh[0], w_ih[1], w_hh[1], dgates, dh0, dc0,
                 {db ? db[0] : nullptr, db ? db[1] : nullptr, db ? db[2] : nullptr, db ? db[3] : nullptr}, sync};
    hipError_t e = hipMemsetAsync(sync, 0, PF_ZERO_WORDS * sizeof(unsigned int), st);
    if (e != hipSuccess) { egz_set_error("egz_lstm_persist_bwd: memset failed: %s", hipGetErrorString(e)); return (int)e; }
    hipLaunchKernelGGL(lstm_persist_bwd_kernel, dim3(64, egz_cdiv(B, 8)), dim3(512), 0, st, a, T, B);
    EGZ_CHECK_LAUNCH("egz_lstm_persist_bwd");
    return 0;
}
