// Split-half (hi + lo 16-bit) operand helpers shared by the split-half conv kernels (conv3x3_igemm_x3.hip,
// conv3x3_igemm_x3s.hip): packed conversions of 4 floats and the three-product MFMA wrapper.
#pragma once
#include "egz_common.h"
#include <type_traits>

namespace x3 {

constexpr float F16_WSCALE = 1024.f;    // 2^10: f16 weights are pre-scaled so that the lo half stays normal

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <typename T> struct Half;
template <> struct Half<_Float16> {
    static __device__ __forceinline__ void split(float x, unsigned short& h, unsigned short& l) {
        x = __builtin_amdgcn_fmed3f(x, -65504.f, 65504.f);
        const _Float16 hi = (_Float16)x;
        const _Float16 lo = (_Float16)(x - (float)hi);
        h = __builtin_bit_cast(unsigned short, hi);
        l = __builtin_bit_cast(unsigned short, lo);
    }
    // 4 floats -> packed hi / lo halves in 3 VALU per float: hi = v_cvt_pkrtz (round toward zero: the residual is then
    // exact in fp32 and an out-of-range input saturates hi instead of producing inf), lo = RNE of the residual, so the
    // pair carries 22 significant bits with an unbiased error.  |x| > 65504 is outside the f16 x3 domain (lo overflows).
    static __device__ __forceinline__ void split4(const f32x4 v, u32x2& hi, u32x2& lo) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const f16x2 h = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(v[2 * e], v[2 * e + 1]));
            const f16x2 l = __builtin_convertvector(f32x2{v[2 * e] - (float)h[0], v[2 * e + 1] - (float)h[1]}, f16x2);
            hi[e] = __builtin_bit_cast(unsigned, h);
            lo[e] = __builtin_bit_cast(unsigned, l);
        }
    }
    // split4 of v * scale.  (EGZ_TIMING_NOSPLIT: a timing-only build variant for the pre-split-activation go / no-go -- the
    // operand is taken as if it already held [4 hi halves | 4 lo halves]; garbage numerics, two bit-ops per quad.)
    static __device__ __forceinline__ void split4s(const f32x4 v, const float scale, u32x2& hi, u32x2& lo) {
#ifdef EGZ_TIMING_NOSPLIT
        const u32x4 b = __builtin_bit_cast(u32x4, v);
        hi = u32x2{b[0] & 0x7bff7bffu, b[1] & 0x7bff7bffu};
        lo = u32x2{b[2] & 0x7bff7bffu, b[3] & 0x7bff7bffu};
#else
        split4(v * scale, hi, lo);
#endif
    }
    static __device__ __forceinline__ f32x16 mfma(u32x4 a, u32x4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
};
template <> struct Half<__bf16> {
    static __device__ __forceinline__ void split(float x, unsigned short& h, unsigned short& l) {
        const __bf16 hi = (__bf16)x;
        const __bf16 lo = (__bf16)(x - (float)hi);
        h = __builtin_bit_cast(unsigned short, hi);
        l = __builtin_bit_cast(unsigned short, lo);
    }
    // v_cvt_pk_bf16_f32 (RNE), two bit ops to widen the halves back, one packed subtract, v_cvt_pk_bf16_f32: 2.5 VALU / float
    static __device__ __forceinline__ void split4(const f32x4 v, u32x2& hi, u32x2& lo) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const f32x2 x = {v[2 * e], v[2 * e + 1]};
            const unsigned hu = __builtin_bit_cast(unsigned, __builtin_convertvector(x, bf16x2));
            const f32x2 hf = {__builtin_bit_cast(float, hu << 16), __builtin_bit_cast(float, hu & 0xffff0000u)};
            hi[e] = hu;
            lo[e] = __builtin_bit_cast(unsigned, __builtin_convertvector(x - hf, bf16x2));
        }
    }
    static __device__ __forceinline__ void split4s(const f32x4 v, const float scale, u32x2& hi, u32x2& lo) { split4(v * scale, hi, lo); }
    static __device__ __forceinline__ f32x16 mfma(u32x4 a, u32x4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};

template <> struct Half<egz_f16p2> : Half<_Float16> {
    // the hi-only operand, rounded to NEAREST: of 4 scaled floats, and of a stored pair quad (hi + lo in packed f16 arithmetic is the
    // nearest f16 of the value the pair holds; the stored hi half itself is a round-toward-zero image)
    static __device__ __forceinline__ u32x2 rne4s(const f32x4 v, const float scale) {
        const f32x4 w = v * scale;
        return u32x2{__builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{w[0], w[1]}, f16x2)),
                     __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{w[2], w[3]}, f16x2))};
    }
    static __device__ __forceinline__ u32x2 pair_rne(const u32x2 h, const u32x2 l) {
        typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
        return __builtin_bit_cast(u32x2, __builtin_bit_cast(f16x4, h) + __builtin_bit_cast(f16x4, l));
    }
};
template <typename T> constexpr bool IS_F16 = std::is_same<T, _Float16>::value || std::is_same<T, egz_f16p2>::value;

}  // namespace x3
