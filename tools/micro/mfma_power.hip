// Sustained MFMA throughput under the chip's power management: back-to-back MFMAs on register operands (no memory traffic
// in the loop), eight different operand fragments cycled so that consecutive instructions see different bits.
// Shapes: v_mfma_f32_32x32x16_{f16,bf16} and v_mfma_f32_16x16x32_{f16,bf16}; operands: random normal values or zeros.
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_power.hip -o mfma_power ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE, bool BF>
__global__ __launch_bounds__(256, 2) void mfma_loop(const u32x4* __restrict__ frag, float* __restrict__ out, int iters) {
    u32x4 a[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        a[i] = frag[(i * 2 + 0) * 64 + (threadIdx.x & 63)];
        b[i] = frag[(i * 2 + 1) * 64 + (threadIdx.x & 63)];
    }
    if constexpr (SHAPE == 32) {
        f32x16 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if constexpr (BF)
                    acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[i]), __builtin_bit_cast(bf16x8, b[(i + it) & 7]), acc[i & 3], 0, 0, 0);
                else
                    acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[i]), __builtin_bit_cast(f16x8, b[(i + it) & 7]), acc[i & 3], 0, 0, 0);
            }
        }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += acc[i][r];
        out[blockIdx.x * 256 + threadIdx.x] = s;
    } else {
        f32x4 acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if constexpr (BF)
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[i]), __builtin_bit_cast(bf16x8, b[(i + it) & 7]), acc[i], 0, 0, 0);
                else
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a[i]), __builtin_bit_cast(f16x8, b[(i + it) & 7]), acc[i], 0, 0, 0);
            }
        }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
        out[blockIdx.x * 256 + threadIdx.x] = s;
    }
}

static unsigned short f2h(float x, bool bf) {
    if (bf) { unsigned u; memcpy(&u, &x, 4); return (unsigned short)((u + 0x7fff + ((u >> 16) & 1)) >> 16); }
    _Float16 h = (_Float16)x; unsigned short r; memcpy(&r, &h, 2); return r;
}

template <int SHAPE, bool BF>
void run(const char* name, int mode) {
    const int blocks = 2048, iters = 4000;
    std::vector<unsigned short> h(16 * 64 * 8);
    srand(1);
    for (auto& v : h) {
        float x = 0.f;
        if (mode) { float u1 = (rand() + 1.f) / (RAND_MAX + 2.f), u2 = rand() / (float)RAND_MAX; x = sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2); }
        if (mode == 2) x *= 4.8e-4f;          // "lo half" magnitudes: 2^-11 of the hi half
        v = f2h(x, BF);
    }
    u32x4* d; float* o;
    hipMalloc(&d, h.size() * 2); hipMalloc(&o, blocks * 256 * 4);
    hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        for (int k = 0; k < 5; ++k) hipLaunchKernelGGL((mfma_loop<SHAPE, BF>), dim3(blocks), dim3(256), 0, 0, d, o, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double flop = 5.0 * blocks * 4 * (double)iters * 8 * 2.0 * (SHAPE == 32 ? 32.0 * 32 * 16 : 16.0 * 16 * 32);
        if (rep) printf("%-28s %-8s %8.1f TFLOP/s  (%.1f ms)\n", name, mode == 0 ? "zeros" : mode == 1 ? "random" : "random-lo", flop / (ms * 1e-3) / 1e12, ms);
    }
    hipFree(d); hipFree(o);
}

int main() {
    for (int mode = 0; mode < 3; ++mode) {
        run<32, false>("v_mfma_f32_32x32x16_f16", mode);
        run<16, false>("v_mfma_f32_16x16x32_f16", mode);
        run<32, true>("v_mfma_f32_32x32x16_bf16", mode);
        run<16, true>("v_mfma_f32_16x16x32_bf16", mode);
    }
    return 0;
}
