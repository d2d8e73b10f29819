// Does a power-capped MI355X lose matrix-core throughput when an MFMA-bound kernel is confined to fewer CUs -- and can an
// HBM-bound pass run beside it on the CUs left over?  (Round-5 question behind the SP step's ~4 ms of exposed streaming passes:
// the conv kernels' two resident blocks own their CU's register file, so a streaming pass from another stream only gets tail
// slots; a CU partition would give it CUs of its own -- IF the MFMA kernels are limited by the chip's power budget and not by
// their CU count.)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/cu_mask_probe.hip -o tools/micro/cu_mask_probe && tools/micro/cu_mask_probe
// 1. census: which (XCC, SE, CU) a masked stream's blocks land on, per mask pattern;
// 2. MFMA rate (v_mfma_f32_32x32x16_f16, random bits, register operands) on N CUs;
// 3. the same beside a streaming copy on the complementary CUs: both rates, against each alone.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <set>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e__)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void census(unsigned* __restrict__ ids, int spin) {
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);        // HW_REG_HW_ID
    const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);      // HW_REG_XCC_ID
    long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < spin) {}
    if (threadIdx.x == 0) {
        ids[2 * blockIdx.x] = hw;
        ids[2 * blockIdx.x + 1] = xcc;
    }
}

__global__ __launch_bounds__(256, 2) void mfma_loop(const u32x4* __restrict__ frag, float* __restrict__ out, int iters) {
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
        for (int i = 0; i < 8; ++i)     // B fixed over four MFMAs, A changes every time: the conv kernels' operand pattern
            acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[i]),
                                                                __builtin_bit_cast(f16x8, b[((i >> 2) + it) & 7]), acc[i & 3], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void stream_copy(const f32x4* __restrict__ src, f32x4* __restrict__ dst, long n4) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        f32x4 v = src[i];
        v[0] = fmaxf(v[0] * 1.0001f + 0.5f, 0.f);
        dst[i] = v;
    }
}

static hipStream_t masked_stream(const std::vector<unsigned>& mask) {
    hipStream_t s;
    CK(hipExtStreamCreateWithCUMask(&s, (unsigned)mask.size(), mask.data()));
    return s;
}

// mask over `bits` CU bits: pattern "low": the first n bits; "stride": clear every k-th bit so that n of 256 stay
static std::vector<unsigned> make_mask(int n, const char* kind, bool complement = false) {
    std::vector<unsigned> m(8, 0u);
    std::vector<int> on(256, 0);
    if (!strcmp(kind, "low")) {
        for (int i = 0; i < n; ++i) on[i] = 1;
    } else {                          // spread: bit i is on when floor((i + 1) n / 256) > floor(i n / 256)
        for (int i = 0; i < 256; ++i) on[i] = ((long)(i + 1) * n / 256) > ((long)i * n / 256);
    }
    for (int i = 0; i < 256; ++i)
        if (on[i] != (int)complement) m[i >> 5] |= 1u << (i & 31);
    return m;
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);          // (a first version hung with its output still buffered)
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    printf("device %s, %d CUs\n", p.name, p.multiProcessorCount);
    unsigned* ids;
    CK(hipMalloc(&ids, 2 * 8192 * sizeof(unsigned)));
    std::vector<unsigned> h(2 * 8192);
    // ---- 1. census
    for (const char* kind : {"spread"})          // ("low": a mask that leaves whole XCDs without CUs -- not tried again after a hang)
        for (int n : {256, 192, 128, 64, 32}) {
            auto mk = make_mask(n, kind);
            hipStream_t s = masked_stream(mk);
            hipLaunchKernelGGL(census, dim3(8192), dim3(256), 0, s, ids, 20000);
            CK(hipStreamSynchronize(s));
            CK(hipMemcpy(h.data(), ids, h.size() * 4, hipMemcpyDeviceToHost));
            std::set<unsigned> cus;
            int per_xcc[8] = {0};
            std::set<unsigned> seen[8];
            for (int b = 0; b < 8192; ++b) {
                const unsigned hw = h[2 * b], xcc = h[2 * b + 1] & 15;
                const unsigned key = (xcc << 16) | (hw & 0xff00);          // cu_id [11:8], sh [12], se [15:13]
                cus.insert(key);
                seen[xcc & 7].insert(hw & 0xff00);
            }
            for (int x = 0; x < 8; ++x) per_xcc[x] = (int)seen[x].size();
            printf("census mask %-6s %3d bits: %3zu distinct CUs; per XCC %d %d %d %d %d %d %d %d\n", kind, n, cus.size(), per_xcc[0],
                   per_xcc[1], per_xcc[2], per_xcc[3], per_xcc[4], per_xcc[5], per_xcc[6], per_xcc[7]);
            CK(hipStreamDestroy(s));
        }
    // ---- 2. MFMA rate on N CUs
    const int blocks = 2048, iters = 20000;
    std::vector<unsigned short> hf(16 * 64 * 8);
    srand(3);
    for (auto& v : hf) v = (unsigned short)((rand() & 0x3ff) | ((13 + rand() % 5) << 10) | ((rand() & 1) << 15));
    u32x4* frag;
    float* out;
    CK(hipMalloc(&frag, hf.size() * 2));
    CK(hipMemcpy(frag, hf.data(), hf.size() * 2, hipMemcpyHostToDevice));
    CK(hipMalloc(&out, blocks * 256 * sizeof(float)));
    const long n4 = 64L << 20;        // 1 GiB in, 1 GiB out per pass
    f32x4 *src, *dst;
    CK(hipMalloc(&src, n4 * 16));
    CK(hipMalloc(&dst, n4 * 16));
    CK(hipMemset(src, 0, n4 * 16));
    hipEvent_t e0, e1, c0, c1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&c0)); CK(hipEventCreate(&c1));
    auto mfma_tf = [&](hipStream_t s, int it) {
        hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(256), 0, s, frag, out, it / 10);
        CK(hipEventRecord(e0, s));
        hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(256), 0, s, frag, out, it);
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        return blocks * 4.0 * it * 8 * 32768.0 / (ms * 1e-3) / 1e12;
    };
    for (const char* kind : {"spread"})
        for (int n : {256, 240, 224, 208, 192, 160, 128, 64}) {
            hipStream_t s = masked_stream(make_mask(n, kind));
            const double tf = mfma_tf(s, iters);
            printf("mfma alone   mask %-6s %3d CUs: %7.1f TFLOP/s (%.2f per CU)\n", kind, n, tf, tf / n);
            CK(hipStreamDestroy(s));
        }
    // ---- 3. streaming copy alone on M CUs, then beside the MFMA loop on the complementary mask
    auto copy_tbs = [&](hipStream_t s, int reps) {
        hipLaunchKernelGGL(stream_copy, dim3(4096), dim3(256), 0, s, src, dst, n4);
        CK(hipEventRecord(c0, s));
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(stream_copy, dim3(4096), dim3(256), 0, s, src, dst, n4);
        CK(hipEventRecord(c1, s));
        CK(hipStreamSynchronize(s));
        float ms;
        CK(hipEventElapsedTime(&ms, c0, c1));
        return reps * 2.0 * n4 * 16 / (ms * 1e-3) / 1e12;
    };
    for (int m : {256, 64, 48, 32, 16}) {
        hipStream_t s = masked_stream(make_mask(m, "spread"));
        printf("copy alone   mask spread %3d CUs: %5.2f TB/s\n", m, copy_tbs(s, 4));
        CK(hipStreamDestroy(s));
    }
    for (int m : {64, 48, 32, 16, 0}) {
        // MFMA on 256 - m CUs (m = 0: both streams unmasked, sharing all CUs), copy on the other m
        hipStream_t sm = m ? masked_stream(make_mask(m, "spread", true)) : masked_stream(make_mask(256, "low"));
        hipStream_t sc = m ? masked_stream(make_mask(m, "spread")) : masked_stream(make_mask(256, "low"));
        const int it = iters * 2;
        hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(256), 0, sm, frag, out, 200);
        CK(hipStreamSynchronize(sm));
        CK(hipEventRecord(e0, sm));
        hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(256), 0, sm, frag, out, it);
        CK(hipEventRecord(e1, sm));
        CK(hipEventRecord(c0, sc));
        const int reps = 12;
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(stream_copy, dim3(4096), dim3(256), 0, sc, src, dst, n4);
        CK(hipEventRecord(c1, sc));
        CK(hipStreamSynchronize(sm));
        CK(hipStreamSynchronize(sc));
        float ms_m, ms_c;
        CK(hipEventElapsedTime(&ms_m, e0, e1));
        CK(hipEventElapsedTime(&ms_c, c0, c1));
        printf("together: mfma on %3d CUs %7.1f TFLOP/s over %.1f ms | copy on %3d CUs %5.2f TB/s over %.1f ms\n", 256 - m,
               blocks * 4.0 * it * 8 * 32768.0 / (ms_m * 1e-3) / 1e12, ms_m, m ? m : 256, reps * 2.0 * n4 * 16 / (ms_c * 1e-3) / 1e12, ms_c);
        CK(hipStreamDestroy(sm));
        CK(hipStreamDestroy(sc));
    }
    return 0;
}
