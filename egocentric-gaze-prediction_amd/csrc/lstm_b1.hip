// The AT module in the reference's OWN use: one fixation sample at a time, T = 1, B = 1 (AT.py:127-145 trains on
// (1, 1, 512) inputs; AT.py:246 / vis_features.py step the same way at inference).  A step is a chain of matrix-VECTOR
// products over 17.9 MB of weights -- launch- and latency-bound, not arithmetic-bound -- so the whole network step is ONE
// C-ABI call that issues a few plain launches back to back (a dependent kernel boundary costs ~1.5 us, the generic path's
// ~55 launches with a host round trip each cost ~0.9 ms per sample):
//
//   forward   tanh(x) -> per layer [all four gate rows of a hidden unit . (x, h) -> cell] -> Linear + ReLU       L + 1 launches
//   backward  Linear backward (+ dW as an outer product, + the transposed product as per-strip partial columns)
//             -> per layer (top down) [sum the partials -> cell backward -> dgates] , [dW_ih, dW_hh outer products
//                + partial columns of W_ih^T dgates for the layer below]                                    2L + 1 launches
//
// Weight gradients of a batch-1 step are rank-1 (dgates x input), written straight to their destination (the optimizer's
// flat gradient buffer through hipops.GradSink).  Exact fp32 on the vector ALUs; dot products are reduced lane-wise then
// by wave shuffles, the transposed products by fixed-order partial sums (deterministic, no atomics).
// Gate order i, f, g, o and torch's nn.LSTM semantics, as lstm_seq.hip.
#include "egz_common.h"

namespace {

__device__ __forceinline__ float sigm1(float x) { return 1.f / (1.f + expf(-x)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// One block per hidden unit j, one wave per gate: pre[g] = W_ih[g*H + j] . x + W_hh[g*H + j] . h + b_ih + b_hh.
// x_raw != null: layer 0, x = tanh(x_raw) (block 0 also stores it to xt for the backward pass).
__global__ __launch_bounds__(256) void lstm_b1_layer_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ x_raw, float* __restrict__ xt, const float* __restrict__ w_ih,
    const float* __restrict__ w_hh, const float* __restrict__ b_ih, const float* __restrict__ b_hh,
    const float* __restrict__ h_prev, const float* __restrict__ c_prev, float* __restrict__ h_out,
    float* __restrict__ c_out, float* __restrict__ act, int C, int H) {
    __shared__ float pre[4];
    const int j = blockIdx.x, lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    const float* wi = w_ih + (long)(g * H + j) * C;
    const float* wh = w_hh + (long)(g * H + j) * H;
    float s = 0.f;
    for (int k = lane * 4; k < C; k += 256) {
        f32x4 xv;
        if (x_raw) {
            const f32x4 r = *reinterpret_cast<const f32x4*>(x_raw + k);
            xv = f32x4{tanhf(r[0]), tanhf(r[1]), tanhf(r[2]), tanhf(r[3])};
            if (j == 0 && g == 0) *reinterpret_cast<f32x4*>(xt + k) = xv;
        } else {
            xv = *reinterpret_cast<const f32x4*>(x + k);
        }
        const f32x4 wv = *reinterpret_cast<const f32x4*>(wi + k);
        s += wv[0] * xv[0] + wv[1] * xv[1] + wv[2] * xv[2] + wv[3] * xv[3];
    }
    for (int k = lane * 4; k < H; k += 256) {
        const f32x4 hv = *reinterpret_cast<const f32x4*>(h_prev + k);
        const f32x4 wv = *reinterpret_cast<const f32x4*>(wh + k);
        s += wv[0] * hv[0] + wv[1] * hv[1] + wv[2] * hv[2] + wv[3] * hv[3];
    }
    s = wave_sum(s);
    if (lane == 0) pre[g] = s + b_ih[g * H + j] + b_hh[g * H + j];
    __syncthreads();
    if (threadIdx.x == 0) {
        const float ig = sigm1(pre[0]), fg = sigm1(pre[1]), gg = tanhf(pre[2]), og = sigm1(pre[3]);
        const float c = fg * c_prev[j] + ig * gg;
        c_out[j] = c;
        h_out[j] = og * tanhf(c);
        if (act) {
            act[j] = ig;
            act[H + j] = fg;
            act[2 * H + j] = gg;
            act[3 * H + j] = og;
        }
    }
}

// out[i] = relu(W[i] . h + b[i]); one wave per row
__global__ __launch_bounds__(256) void lin_relu_b1_kernel(const float* __restrict__ w, const float* __restrict__ b,
                                                          const float* __restrict__ h, float* __restrict__ out, int N,
                                                          int H) {
    const int lane = threadIdx.x & 63, i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= N) return;
    const float* wr = w + (long)i * H;
    float s = 0.f;
    for (int k = lane * 4; k < H; k += 256) {
        const f32x4 hv = *reinterpret_cast<const f32x4*>(h + k);
        const f32x4 wv = *reinterpret_cast<const f32x4*>(wr + k);
        s += wv[0] * hv[0] + wv[1] * hv[1] + wv[2] * hv[2] + wv[3] * hv[3];
    }
    s = wave_sum(s);
    if (lane == 0) out[i] = fmaxf(s + b[i], 0.f);
}

constexpr int STRIP = 8;      // rows of a weight matrix per backward block (128 threads x 4 columns each)

// Rows r0 .. r0 + STRIP - 1 of a (R x Ccols) weight W with the row gradient d[r] (times the mask out[r] > 0 when `relu_out`
// is given: the Linear + ReLU head):
//   dW[r][j]  = d[r] * vin[j]                      (and dW2[r][j] = d[r] * vin2[j] for the second matrix fed by the same d)
//   part[strip][j] = sum_r W[r][j] * d[r]          (the strip's share of W^T d; skipped when part == null)
//   db[r] = db2[r] = d[r]
__global__ __launch_bounds__(128) void outer_b1_kernel(const float* __restrict__ w, const float* __restrict__ d,
                                                       const float* __restrict__ relu_out, const float* __restrict__ vin,
                                                       const float* __restrict__ vin2, float* __restrict__ dw,
                                                       float* __restrict__ dw2, float* __restrict__ db,
                                                       float* __restrict__ db2, float* __restrict__ part, int R, int Ccols,
                                                       int C2cols) {
    __shared__ float ds[STRIP];
    const int r0 = blockIdx.x * STRIP, tid = threadIdx.x;
    if (tid < STRIP) {
        const int r = r0 + tid;
        float v = 0.f;
        if (r < R) {
            v = d[r];
            if (relu_out && !(relu_out[r] > 0.f)) v = 0.f;
            if (db) db[r] = v;
            if (db2) db2[r] = v;
        }
        ds[tid] = v;
    }
    __syncthreads();
    const int nr = (R - r0 < STRIP) ? (R - r0) : STRIP;
    for (int j = tid * 4; j < Ccols; j += 512) {
        const f32x4 xv = *reinterpret_cast<const f32x4*>(vin + j);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int rr = 0; rr < nr; ++rr) {
            const float dv = ds[rr];
            const long o = (long)(r0 + rr) * Ccols + j;
            if (dw) *reinterpret_cast<f32x4*>(dw + o) = f32x4{dv * xv[0], dv * xv[1], dv * xv[2], dv * xv[3]};
            if (part) {
                const f32x4 wv = *reinterpret_cast<const f32x4*>(w + o);
                acc[0] += wv[0] * dv; acc[1] += wv[1] * dv; acc[2] += wv[2] * dv; acc[3] += wv[3] * dv;
            }
        }
        if (part) *reinterpret_cast<f32x4*>(part + (long)blockIdx.x * Ccols + j) = acc;
    }
    if (dw2) {
        for (int j = tid * 4; j < C2cols; j += 512) {
            const f32x4 xv = *reinterpret_cast<const f32x4*>(vin2 + j);
            for (int rr = 0; rr < nr; ++rr) {
                const float dv = ds[rr];
                *reinterpret_cast<f32x4*>(dw2 + (long)(r0 + rr) * C2cols + j) = f32x4{dv * xv[0], dv * xv[1], dv * xv[2], dv * xv[3]};
            }
        }
    }
}

// dh[j] = sum_s part[s][j] (+ dhn[j]); cell backward of one step with no later step: dc = dcn[j] + dh * o * (1 - tanh(c)^2)
//   di = dc * g * i (1 - i), df = dc * c_prev * f (1 - f), dg = dc * i * (1 - g^2), do = dh * tanh(c) * o (1 - o)
// dh_prev / dc_prev are not produced here: the caller's hidden state carries no gradient on this path.
// Block = 32 hidden units x 8 strip groups: group q sums strips q, q + 8, ... (independent loads in flight), the eight group
// sums are combined in group order through LDS (fixed order: deterministic), then 32 threads do the cell arithmetic.
__global__ __launch_bounds__(256) void cell_bwd_b1_kernel(const float* __restrict__ part, int nstrips,
                                                          const float* __restrict__ dhn, const float* __restrict__ dcn,
                                                          const float* __restrict__ act, const float* __restrict__ c_new,
                                                          const float* __restrict__ c_prev, float* __restrict__ dgates,
                                                          int H) {
    __shared__ float red[8][32];
    const int jl = threadIdx.x & 31, q = threadIdx.x >> 5;
    const int j = blockIdx.x * 32 + jl;
    float a0 = 0.f, a1 = 0.f;
    if (j < H) {
        int s = q;
        for (; s + 8 < nstrips; s += 16) {
            a0 += part[(long)s * H + j];
            a1 += part[(long)(s + 8) * H + j];
        }
        if (s < nstrips) a0 += part[(long)s * H + j];
    }
    red[q][jl] = a0 + a1;
    __syncthreads();
    if (q != 0 || j >= H) return;
    float dh = ((red[0][jl] + red[1][jl]) + (red[2][jl] + red[3][jl])) + ((red[4][jl] + red[5][jl]) + (red[6][jl] + red[7][jl]));
    if (dhn) dh += dhn[j];
    const float ig = act[j], fg = act[H + j], gg = act[2 * H + j], og = act[3 * H + j];
    const float tc = tanhf(c_new[j]);
    float dc = dh * og * (1.f - tc * tc);
    if (dcn) dc += dcn[j];
    dgates[j] = dc * gg * ig * (1.f - ig);
    dgates[H + j] = dc * c_prev[j] * fg * (1.f - fg);
    dgates[2 * H + j] = dc * ig * (1.f - gg * gg);
    dgates[3 * H + j] = dh * tc * og * (1.f - og);
}

}  // namespace

#define P_ALIGNED(p) ((reinterpret_cast<uintptr_t>(p) & 15u) == 0)

// Workspace floats of the backward call: per-strip partial columns of the widest transposed product + the gate gradients.
EGZ_API size_t egz_lstm_b1_ws_bytes(int L, int C, int H, int N) {
    const int rows = 4 * H > N ? 4 * H : N;                  // the Linear head's launch writes ceil(N / STRIP) strips of H floats
    const int strips = (rows + STRIP - 1) / STRIP;
    const int wide = C > H ? C : H;
    return ((size_t)strips * wide + (size_t)L * 4 * H) * sizeof(float);
}

// params: HOST array of 4L + 2 device pointers in state-dict order (w_ih, w_hh, b_ih, b_hh per layer, lin.weight [N][H],
// lin.bias).  inp [C] raw input (tanh is applied here); h0, c0 [L][H]; xt [C] = tanh(inp) (saved); acts [L][4H] activated
// gates (null in no-grad runs); hn, cn [L][H]; out [N].
EGZ_API int egz_lstm_b1_fwd(const void* const* params, int L, const float* inp, const float* h0, const float* c0, float* xt,
                            float* acts, float* hn, float* cn, float* out, int C, int H, int N, hipStream_t st) {
    EGZ_CHECK_ARG(params && inp && h0 && c0 && xt && hn && cn && out && L >= 1, "egz_lstm_b1_fwd: null pointer");
    EGZ_CHECK_ARG(C % 4 == 0 && H % 4 == 0 && C > 0 && H > 0 && N > 0, "egz_lstm_b1_fwd: C=%d H=%d must be multiples of 4", C, H);
    for (int i = 0; i < 4 * L + 2; ++i) {                   // the kernels use 16-byte loads on the parameter rows
        EGZ_CHECK_ARG(P_ALIGNED(reinterpret_cast<const float* const*>(params)[i]), "egz_lstm_b1_fwd: parameter %d is not 16-byte aligned", i);
    }
    EGZ_CHECK_ARG(P_ALIGNED(inp) && P_ALIGNED(h0) && P_ALIGNED(xt) && P_ALIGNED(hn), "egz_lstm_b1_fwd: inp / h0 / xt / hn must be 16-byte aligned");
    for (int l = 0; l < L; ++l) {
        const float* const* p = reinterpret_cast<const float* const*>(params) + 4 * l;
        const float* xin = l ? hn + (long)(l - 1) * H : nullptr;
        hipLaunchKernelGGL(lstm_b1_layer_fwd_kernel, dim3(H), dim3(256), 0, st, xin, l ? nullptr : inp, xt, p[0], p[1], p[2], p[3],
                           h0 + (long)l * H, c0 + (long)l * H, hn + (long)l * H, cn + (long)l * H,
                           acts ? acts + (long)l * 4 * H : nullptr, l ? H : C, H);
        EGZ_CHECK_LAUNCH("egz_lstm_b1_fwd(layer)");
    }
    const float* const* pl = reinterpret_cast<const float* const*>(params) + 4 * L;
    hipLaunchKernelGGL(lin_relu_b1_kernel, dim3((N + 3) / 4), dim3(256), 0, st, pl[0], pl[1], hn + (long)(L - 1) * H, out, N, H);
    EGZ_CHECK_LAUNCH("egz_lstm_b1_fwd(lin)");
    return 0;
}

// Backward of egz_lstm_b1_fwd.  dout [N] (gradient of the ReLU output); dhn / dcn [L][H] or null; grads: HOST array of
// 4L + 2 device pointers (same order as params; a null entry skips that gradient).  The gradient with respect to the input
// and the initial state is not produced (the AT loop detaches them, AT.py:143; callers that need it use the sequence path).
EGZ_API int egz_lstm_b1_bwd(const void* const* params, void* const* grads, int L, const float* dout, const float* dhn,
                            const float* dcn, const float* xt, const float* acts, const float* h0, const float* c0,
                            const float* hn, const float* cn, const float* out, int C, int H, int N, void* workspace,
                            size_t ws_bytes, hipStream_t st) {
    EGZ_CHECK_ARG(params && grads && dout && xt && acts && h0 && c0 && hn && cn && out && workspace && L >= 1,
                  "egz_lstm_b1_bwd: null pointer");
    EGZ_CHECK_ARG(C % 4 == 0 && H % 4 == 0, "egz_lstm_b1_bwd: C=%d H=%d must be multiples of 4", C, H);
    EGZ_CHECK_ARG(ws_bytes >= egz_lstm_b1_ws_bytes(L, C, H, N), "egz_lstm_b1_bwd: workspace too small");
    float* part = static_cast<float*>(workspace);
    const int wide = C > H ? C : H;
    float* dgates = part + (size_t)(((4 * H > N ? 4 * H : N) + STRIP - 1) / STRIP) * wide;       // [L][4H], behind the widest partial region
    for (int i = 0; i < 4 * L + 2; ++i) {                   // the kernels use 16-byte loads on the parameter rows
        EGZ_CHECK_ARG(P_ALIGNED(reinterpret_cast<const float* const*>(params)[i]), "egz_lstm_b1_bwd: parameter %d is not 16-byte aligned", i);
    }
    EGZ_CHECK_ARG(P_ALIGNED(xt) && P_ALIGNED(h0) && P_ALIGNED(hn) && P_ALIGNED(dout), "egz_lstm_b1_bwd: xt / h0 / hn / dout must be 16-byte aligned");
    const float* const* P = reinterpret_cast<const float* const*>(params);
    float* const* G = reinterpret_cast<float* const*>(grads);
    // Linear + ReLU head: dW = dpre x h_top, db = dpre, partial columns of W^T dpre
    int strips = (N + STRIP - 1) / STRIP;
    hipLaunchKernelGGL(outer_b1_kernel, dim3(strips), dim3(128), 0, st, P[4 * L], dout, out, hn + (long)(L - 1) * H, nullptr,
                       G[4 * L], nullptr, G[4 * L + 1], nullptr, part, N, H, 0);
    EGZ_CHECK_LAUNCH("egz_lstm_b1_bwd(lin)");
    for (int l = L - 1; l >= 0; --l) {
        float* dg = dgates + (long)l * 4 * H;
        hipLaunchKernelGGL(cell_bwd_b1_kernel, dim3((H + 31) / 32), dim3(256), 0, st, part, strips,
                           dhn ? dhn + (long)l * H : nullptr, dcn ? dcn + (long)l * H : nullptr, acts + (long)l * 4 * H,
                           cn + (long)l * H, c0 + (long)l * H, dg, H);
        EGZ_CHECK_LAUNCH("egz_lstm_b1_bwd(cell)");
        const int cin = l ? H : C;
        const float* vin = l ? hn + (long)(l - 1) * H : xt;
        strips = (4 * H + STRIP - 1) / STRIP;
        hipLaunchKernelGGL(outer_b1_kernel, dim3(strips), dim3(128), 0, st, P[4 * l], dg, nullptr, vin, h0 + (long)l * H,
                           G[4 * l], G[4 * l + 1], G[4 * l + 2], G[4 * l + 3], l ? part : nullptr, 4 * H, cin, H);
        EGZ_CHECK_LAUNCH("egz_lstm_b1_bwd(outer)");
    }
    return 0;
}
