// How many independent non-MFMA instructions hide behind one v_mfma_f32_32x32x16_f16 on gfx950, with one and with two waves per
// SIMD?  (Sizing question behind the Winograd no-go, profiles/r04_winograd_gonogo.txt: a fused F(2x2,3x3) kernel needs 5-7
// VALU per MFMA for its input transform + f16 split, the direct kernel ~0.3.)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/issue_probe.hip -o tools/micro/issue_probe && tools/micro/issue_probe
// Each block runs ITER iterations of [12 MFMAs on 4 accumulators, each followed by N fillers]; one block per CU; the table is
// shader cycles (s_memtime) per MFMA per SIMD.  Operands are random bits (power-limited clock; cycles are what is reported).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e__)); exit(1); } } while (0)

enum { F_FMA = 0, F_CVT = 1, F_MIX = 2, F_LDS = 3, F_PKADD = 4, F_NONE = 5 };

template <int N, int KIND, int NTHR, int AGPR>
__global__ __launch_bounds__(NTHR) void probe(const u32x4* __restrict__ in, float* __restrict__ out, long long* __restrict__ cyc, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[8192];
    const int tid = threadIdx.x;
    for (int i = tid; i < 8192; i += NTHR) lds[i] = (float)i * 1e-3f;
    __syncthreads();
    u32x4 a = in[tid & 63], b = in[64 + (tid & 63)];
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = __builtin_bit_cast(float, a[i & 3]) * 1e-20f + (float)i;
    f32x4 ld = {0.f, 0.f, 0.f, 0.f};
    f32x4 ldn[4] = {ld, ld, ld, ld};
    const float c1 = 1.0000001f, c2 = 1e-9f;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 12; ++m) {
            if (AGPR == 1) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[m & 3]) : "v"(a), "v"(b));
            else if (AGPR == 2) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[m & 3]) : "v"(a), "v"(b));
            else acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc[m & 3], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < N; ++k) {
                const int j = (m * N + k) & 7;
                if (KIND == F_FMA) v[j] = __builtin_fmaf(v[j], c1, c2);
                if (KIND == F_CVT) {
                    unsigned h = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(v[j], v[(j + 1) & 7]));
                    asm volatile("" : "+v"(h));
                    v[j] = __builtin_bit_cast(float, h | 0x3f800000u);
                }
                if (KIND == F_MIX) {
                    float r;
                    asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(__builtin_bit_cast(unsigned, v[(j + 3) & 7])), "v"(v[j]));
                    v[j] = r;
                }
                if (KIND == F_LDS) {
                    ld += ldn[k];                 // the value read one slot ago
                    ldn[k] = *reinterpret_cast<const f32x4*>(&lds[((tid * 4 + (m * N + k) * 256) & 8188)]);
                }
                if (KIND == F_PKADD) {
                    typedef float f32x2 __attribute__((ext_vector_type(2)));
                    f32x2 p = {v[j], v[(j + 4) & 7]};
                    p = p + f32x2{c2, c2};
                    v[j] = p[0];
                    v[(j + 4) & 7] = p[1];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = ld[0] + ld[1] + ld[2] + ld[3];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][7];
    out[blockIdx.x * NTHR + tid] = s;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int N, int KIND, int NTHR, int AGPR>
static void run(const u32x4* din, float* dout, long long* dcyc, const char* name) {
    const int iters = 2000, blocks = 256;
    hipLaunchKernelGGL((probe<N, KIND, NTHR, AGPR>), dim3(blocks), dim3(NTHR), 0, 0, din, dout, dcyc, 10);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((probe<N, KIND, NTHR, AGPR>), dim3(blocks), dim3(NTHR), 0, 0, din, dout, dcyc, iters);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<long long> h(blocks);
    CK(hipMemcpy(h.data(), dcyc, blocks * sizeof(long long), hipMemcpyDeviceToHost));
    double avg = 0;
    for (auto c : h) avg += (double)c;
    avg /= blocks;
    const int wps = NTHR / 256;                              // waves per SIMD
    const double mfma_per_simd = 12.0 * iters * wps;
    printf("%-8s acc=%s N=%2d  waves/SIMD %d  cycles(s_memtime) per MFMA per SIMD %7.1f   wall ns per MFMA per SIMD %6.2f\n", name, AGPR == 1 ? "AGPR" : AGPR == 2 ? "VGPR" : "auto", N, wps,
           avg / mfma_per_simd, ms * 1e6 / mfma_per_simd);
    fflush(stdout);
}

int main() {
    std::vector<unsigned> hin(128 * 4);
    unsigned s = 12345;
    for (auto& x : hin) { s = s * 1664525u + 1013904223u; x = (s & 0x7fff7fffu) | 0x30003000u; }     // finite random halves
    u32x4* din; float* dout; long long* dcyc;
    CK(hipMalloc(&din, hin.size() * 4)); CK(hipMalloc(&dout, 256 * 512 * 4)); CK(hipMalloc(&dcyc, 256 * 8));
    CK(hipMemcpy(din, hin.data(), hin.size() * 4, hipMemcpyHostToDevice));
#define ROW(N, K, NAME) run<N, K, 256, 1>(din, dout, dcyc, NAME); run<N, K, 256, 2>(din, dout, dcyc, NAME); run<N, K, 512, 1>(din, dout, dcyc, NAME); run<N, K, 512, 2>(din, dout, dcyc, NAME);
    ROW(0, F_NONE, "none")
    ROW(2, F_FMA, "v_fma") ROW(4, F_FMA, "v_fma") ROW(6, F_FMA, "v_fma") ROW(8, F_FMA, "v_fma") ROW(12, F_FMA, "v_fma")
    ROW(4, F_CVT, "cvt_pk") ROW(8, F_CVT, "cvt_pk")
    ROW(4, F_MIX, "fma_mix") ROW(8, F_MIX, "fma_mix")
    ROW(2, F_PKADD, "pk_add") ROW(4, F_PKADD, "pk_add")
    ROW(1, F_LDS, "ds_b128") ROW(2, F_LDS, "ds_b128")
    return 0;
}
