// Measurement aid of bench.py (not on the reference's path): sustained v_mfma_f32_32x32x16_f16 throughput of THIS chip under
// its power management -- back-to-back MFMAs on register operands, no memory traffic in the loop, eight operand fragments
// cycled so that consecutive instructions see different bits.  With random operand bits MI355X sustains ~1.5 PFLOP/s (2.4
// on all-zero operands): the matrix cores are limited by the power budget, not by issue, and split-half arithmetic pays three
// MFMA MACs per algorithmic MAC -- bench.py reports roofline.frac against the 2.5 PF nominal peak AND against this ceiling.
#include "egz_common.h"

namespace {
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256, 2) void mfma_probe_kernel(const u32x4* __restrict__ frag, float* __restrict__ out, int iters) {
    u32x4 a[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        a[i] = frag[(i * 2 + 0) * 64 + (threadIdx.x & 63)];
        b[i] = frag[(i * 2 + 1) * 64 + (threadIdx.x & 63)];
    }
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
            acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[i]),
                                                                __builtin_bit_cast(f16x8, b[(i + it) & 7]), acc[i & 3], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
}  // namespace

// frag: 16 x 64 x 16 bytes of f16 operand fragments (device); out: blocks x 256 floats.  Executes
// blocks x 4 waves x iters x 8 MFMAs of 32 x 32 x 16 (32768 flop each).
EGZ_API int egz_mfma_probe(const void* frag, float* out, int blocks, int iters, hipStream_t st) {
    EGZ_CHECK_ARG(frag && out && blocks > 0 && iters > 0, "egz_mfma_probe: bad arguments");
    hipLaunchKernelGGL(mfma_probe_kernel, dim3(blocks), dim3(256), 0, st, static_cast<const u32x4*>(frag), out, iters);
    EGZ_CHECK_LAUNCH("egz_mfma_probe");
    return 0;
}
