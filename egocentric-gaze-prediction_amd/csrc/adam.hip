// Fused Adam step over one flat fp32 parameter buffer (torch.optim.Adam defaults as the reference configures it:
// SP.py:110-113, AT.py:84, LF.py:77 -- betas (0.9, 0.999), eps 1e-8, no weight decay, no amsgrad).
// HBM-bound: reads p, g, m, v and writes p, m, v once (7 fp32 streams), float4 per lane, grid-stride.
// 1-beta1 / 1-beta2 are formed in fp64 on the host like torch's Python scalars (1-0.999f in fp32 is off by 5e-5).
// Mirrors torch's single-tensor update order:  m.lerp_(g, 1-b1);  v = v*b2 + (1-b2)*g*g;
//   denom = sqrt(v)/sqrt(1-b2^t) + eps;  p -= (lr/(1-b1^t)) * m/denom.
#include "egz_common.h"

namespace {
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, long n, float omb1, float beta2, float omb2, float eps,
                                                   float step_size, float bc2_sqrt, float grad_scale,
                                                   unsigned int* __restrict__ nonfinite) {
#pragma clang fp contract(off)       // the host-counter and device-counter forms must round alike (no fused multiply-add in one of them)
    const long n4 = n >> 2;
    bool bad = false;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        f32x4 pv = reinterpret_cast<f32x4*>(p)[i];
        const f32x4 gv = reinterpret_cast<const f32x4*>(g)[i];
        f32x4 mv = reinterpret_cast<f32x4*>(m)[i];
        f32x4 vv = reinterpret_cast<f32x4*>(v)[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float ge = gv[e] * grad_scale;
            if (!(fabsf(ge) <= 3.402823466e38f)) { bad = true; continue; }       // NaN / inf: this element keeps p, m, v
            mv[e] = mv[e] + omb1 * (ge - mv[e]);
            vv[e] = vv[e] * beta2 + omb2 * ge * ge;
            const float denom = sqrtf(vv[e]) / bc2_sqrt + eps;
            pv[e] = pv[e] - step_size * (mv[e] / denom);
        }
        reinterpret_cast<f32x4*>(p)[i] = pv;
        reinterpret_cast<f32x4*>(m)[i] = mv;
        reinterpret_cast<f32x4*>(v)[i] = vv;
    }
    // tail (n not a multiple of 4)
    const long t = n4 * 4 + blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (t < n) {
        const float ge = g[t] * grad_scale;
        if (!(fabsf(ge) <= 3.402823466e38f)) bad = true;
        else {
            const float mm = m[t] + omb1 * (ge - m[t]);
            const float vq = v[t] * beta2 + omb2 * ge * ge;
            m[t] = mm;
            v[t] = vq;
            p[t] = p[t] - step_size * (mm / (sqrtf(vq) / bc2_sqrt + eps));
        }
    }
    if (bad && nonfinite) atomicOr(nonfinite, 1u);
}
// Step count on the device (hipGraph-captured training steps: a replay cannot receive a new host scalar).  The bias
// corrections are formed in fp64 from step[0] + 1 exactly as the host form does; adam_bump_kernel increments the counter after
// the update (a separate launch: every block of the update reads the same value).  Round 3 tried to let the LAST block to finish
// do the increment (a ticket in step[1]): one device-scope atomic per block on one address costs ~45 ns each -- 2048 blocks
// took the kernel from 23 to 112 us, 512 blocks (23 tickets-us) cost as much as they saved in bandwidth -- so the 4.9 us bump
// launch stays.  step[1] is reserved (0).
__global__ __launch_bounds__(256) void adam_dev_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                       float* __restrict__ v, long n, double lr, double beta1, double beta2,
                                                       float eps, int* __restrict__ step, float grad_scale,
                                                       unsigned int* __restrict__ nonfinite) {
#pragma clang fp contract(off)
    bool bad = false;
    const double t = (double)(*static_cast<volatile int*>(step) + 1);
    const float omb1 = (float)(1.0 - beta1), b2 = (float)beta2, omb2 = (float)(1.0 - beta2);
    const float step_size = (float)(lr / (1.0 - pow(beta1, t)));
    const float bc2_sqrt = (float)sqrt(1.0 - pow(beta2, t));
    const long n4 = n >> 2;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        f32x4 pv = reinterpret_cast<f32x4*>(p)[i];
        const f32x4 gv = reinterpret_cast<const f32x4*>(g)[i];
        f32x4 mv = reinterpret_cast<f32x4*>(m)[i];
        f32x4 vv = reinterpret_cast<f32x4*>(v)[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float ge = gv[e] * grad_scale;
            if (!(fabsf(ge) <= 3.402823466e38f)) { bad = true; continue; }       // NaN / inf: this element keeps p, m, v
            mv[e] = mv[e] + omb1 * (ge - mv[e]);
            vv[e] = vv[e] * b2 + omb2 * ge * ge;
            const float denom = sqrtf(vv[e]) / bc2_sqrt + eps;
            pv[e] = pv[e] - step_size * (mv[e] / denom);
        }
        reinterpret_cast<f32x4*>(p)[i] = pv;
        reinterpret_cast<f32x4*>(m)[i] = mv;
        reinterpret_cast<f32x4*>(v)[i] = vv;
    }
    const long tl = n4 * 4 + blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (tl < n) {
        const float ge = g[tl] * grad_scale;
        if (!(fabsf(ge) <= 3.402823466e38f)) bad = true;
        else {
            const float mm = m[tl] + omb1 * (ge - m[tl]);
            const float vq = v[tl] * b2 + omb2 * ge * ge;
            m[tl] = mm;
            v[tl] = vq;
            p[tl] = p[tl] - step_size * (mm / (sqrtf(vq) / bc2_sqrt + eps));
        }
    }
    if (bad && nonfinite) atomicOr(nonfinite, 1u);
}
__global__ void adam_bump_kernel(int* step) { *step += 1; }
}  // namespace

// egz_adam_step with the (0-based, completed-steps) counter on the device: applies step step[0] + 1 and increments step[0].
// step points at TWO ints: {completed steps, reserved 0}.
EGZ_API int egz_adam_step_dev(float* p, const float* g, float* m, float* v, long n, double lr, double beta1, double beta2,
                              double eps, int* step, double grad_scale, unsigned int* nonfinite, hipStream_t st) {
    EGZ_CHECK_ARG(p && g && m && v && step && n > 0, "egz_adam_step_dev: bad arguments");
    EGZ_CHECK_ARG(((uintptr_t)p % 16 == 0) && ((uintptr_t)g % 16 == 0) && ((uintptr_t)m % 16 == 0) && ((uintptr_t)v % 16 == 0),
                  "egz_adam_step_dev: buffers must be 16-byte aligned");
    long g4 = (n / 4 + 255) / 256;
    if (g4 < 1) g4 = 1;
    const int grid = (int)(g4 > 8192 ? 8192 : g4);
    hipLaunchKernelGGL(adam_dev_kernel, dim3(grid), dim3(256), 0, st, p, g, m, v, n, lr, beta1, beta2, (float)eps, step,
                       (float)grad_scale, nonfinite);
    EGZ_CHECK_LAUNCH("egz_adam_step_dev");
    hipLaunchKernelGGL(adam_bump_kernel, dim3(1), dim3(1), 0, st, step);
    EGZ_CHECK_LAUNCH("egz_adam_step_dev(bump)");
    return 0;
}

// Hyper-parameters are doubles (Python floats) so 1-beta keeps its precision.  step: 1-based step count.  grad_scale multiplies the gradient first (1/world_size after a sum all-reduce).
// An element whose (scaled) gradient is NaN / inf is SKIPPED -- p, m, v keep their values -- and `nonfinite` (device word, may be
// null; the caller zeroes it) gets bit 0 set: a poisoned backward pass (a persistent-LSTM hand-off that timed out, an overflow
// upstream) cannot destroy the weights before the host looks (FusedAdam.check_finite).  Finite gradients: unchanged bits.
EGZ_API int egz_adam_step(float* p, const float* g, float* m, float* v, long n, double lr, double beta1, double beta2,
                          double eps, int step, double grad_scale, unsigned int* nonfinite, hipStream_t st) {
    EGZ_CHECK_ARG(p && g && m && v && n > 0 && step >= 1, "egz_adam_step: bad arguments");
    EGZ_CHECK_ARG(((uintptr_t)p % 16 == 0) && ((uintptr_t)g % 16 == 0) && ((uintptr_t)m % 16 == 0) && ((uintptr_t)v % 16 == 0),
                  "egz_adam_step: buffers must be 16-byte aligned");
    const double bc1 = 1.0 - pow(beta1, (double)step);
    const double bc2 = 1.0 - pow(beta2, (double)step);
    const float step_size = (float)(lr / bc1);
    const float bc2_sqrt = (float)sqrt(bc2);
    long g4 = (n / 4 + 255) / 256;
    if (g4 < 1) g4 = 1;
    const int grid = (int)(g4 > 8192 ? 8192 : g4);
    hipLaunchKernelGGL(adam_kernel, dim3(grid), dim3(256), 0, st, p, g, m, v, n,
                       (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps, step_size, bc2_sqrt, (float)grad_scale, nonfinite);
    EGZ_CHECK_LAUNCH("egz_adam_step");
    return 0;
}
