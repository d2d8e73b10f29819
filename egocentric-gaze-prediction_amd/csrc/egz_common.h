// Internal helpers shared by the HIP translation units of libegaze_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>

#define EGZ_API extern "C" __attribute__((visibility("default")))

// Error convention (include/egaze_hip.h): every entry point returns 0 on success or a non-zero
// hipError_t-style code; the message is kept per host thread and read with egz_last_error().
void egz_set_error(const char* fmt, ...);

#define EGZ_CHECK_ARG(cond, ...)                 \
    do {                                         \
        if (!(cond)) {                           \
            egz_set_error(__VA_ARGS__);          \
            return (int)hipErrorInvalidValue;    \
        }                                        \
    } while (0)

#define EGZ_CHECK_LAUNCH(name)                                                        \
    do {                                                                              \
        hipError_t e__ = hipGetLastError();                                           \
        if (e__ != hipSuccess) {                                                      \
            egz_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));     \
            return (int)e__;                                                          \
        }                                                                             \
    } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

static inline int egz_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// 32x32 accumulator element `reg` of lane `lane` sits at row (reg&3)+8*(reg>>2)+4*(lane>>5),
// column lane&31 (cdna_hip_programming.md section 3; dtype-independent on gfx950).
__device__ __forceinline__ int egz_acc_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }// Power-of-two scale that brings a tensor whose max |value| is *absmax (float bit pattern, egz_absmax / the gradient
// producers of bn_pool.hip) into [2^12, 2^13): gradients of 1e-3 .. 1e-9 become f16-representable with 22 significant
// bits in the hi + lo pair.  Multiplying by it and dividing the accumulators by it afterwards is exact.
__device__ __forceinline__ float absmax_scale(const unsigned int* __restrict__ absmax) {
    if (!absmax) return 1.f;
    const float am = __uint_as_float(*absmax);
    if (!(am > 0.f) || !(am < INFINITY)) return 1.f;
    int e;
    frexpf(am, &e);
    int se = 13 - e;
    se = se > 100 ? 100 : (se < -100 ? -100 : se);
    return ldexpf(1.f, se);
}



