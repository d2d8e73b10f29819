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

// Operand-type tag "f16 split halves, TWO products per MAC": a b ~ a_hi b_hi + a_hi b_lo -- the FIRST (A) operand enters with its f16
// hi half only, rounded to nearest (11 significant bits), the second keeps 22.  The split-half kernels (conv3x3_igemm_x3s,
// conv3x3_wgrad) are instantiated with it for the BACKWARD convolutions (dtype | 0x10 / wgrad flag 0x20000): A is the operand that
// goes through LDS as a halo image -- dy in a data gradient, x in a weight gradient -- so one plane is staged instead of two and one
// fragment is read per MFMA pair (with two products per MAC the LDS reads, not the MFMAs, bound the mirror-image form that keeps
// both halves of A and drops the weights' lo half: measured 5 % slower).  Same packings, 2/3 of the MFMA work.
struct egz_f16p2 {};
template <typename T> struct egz_drop_alo { static constexpr bool value = false; };
template <> struct egz_drop_alo<egz_f16p2> { static constexpr bool value = true; };

static inline int egz_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// 32x32 accumulator element `reg` of lane `lane` sits at row (reg&3)+8*(reg>>2)+4*(lane>>5),
// column lane&31 (cdna_hip_programming.md section 3; dtype-independent on gfx950).
__device__ __forceinline__ int egz_acc_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }// Abs-max buffer of a tensor (egz_absmax_elems() uints, ZERO-FILLED by the caller before the producing launch): EGZ_AM_SLOTS slots,
// one per 128-byte line (index slot * EGZ_AM_STRIDE), each holding the bit pattern of a non-negative float.  A producing
// block folds its own maximum into slot (block index % EGZ_AM_SLOTS) with ONE device-scope atomic max -- skipped when the slot
// already holds a larger value, so a launch of thousands of blocks issues a few dozen atomics, spread over 32 lines; bit patterns
// of non-negative floats order like unsigned ints and max is exact and order independent: the result is deterministic.  A
// consumer takes the maximum of the slots (one load per lane + five shuffles) at the top of its kernel -- no fold launch sits
// between producer and consumer (round 3: 84 one-block fold launches per SP step, each waiting ~17 us for a CU slot).
constexpr int EGZ_AM_SLOTS = 32, EGZ_AM_STRIDE = 32;
__device__ __forceinline__ void absmax_commit(unsigned int* __restrict__ absmax, unsigned int idx, float m) {
    unsigned int* p = absmax + (idx & (EGZ_AM_SLOTS - 1)) * EGZ_AM_STRIDE;
    const unsigned int bits = __float_as_uint(m);
    if (bits > __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
        __hip_atomic_fetch_max(p, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// max over the slots; every lane of the calling wave must be active (call it at the top of a kernel)
__device__ __forceinline__ unsigned int absmax_bits(const unsigned int* __restrict__ absmax) {
    unsigned int v = absmax[(threadIdx.x & (EGZ_AM_SLOTS - 1)) * EGZ_AM_STRIDE];
#pragma unroll
    for (int off = EGZ_AM_SLOTS / 2; off > 0; off >>= 1) {
        const unsigned int o = (unsigned int)__shfl_xor((int)v, off);
        v = v > o ? v : o;
    }
    return v;
}
// Power-of-two scale that brings a tensor whose max |value| is in `absmax` (layout above; egz_absmax / the producers of
// bn_pool.hip and the conv epilogues) into [2^12, 2^13): gradients of 1e-3 .. 1e-9 become f16-representable with 22 significant
// bits in the hi + lo pair.  Multiplying by it and dividing the accumulators by it afterwards is exact.
__device__ __forceinline__ float absmax_scale(const unsigned int* __restrict__ absmax) {
    if (!absmax) return 1.f;
    const float am = __uint_as_float(absmax_bits(absmax));
    if (!(am > 0.f) || !(am < INFINITY)) return 1.f;
    int e;
    frexpf(am, &e);
    int se = 13 - e;
    se = se > 100 ? 100 : (se < -100 ? -100 : se);
    return ldexpf(1.f, se);
}
