// What an in-kernel fold of the weight-gradient split partials would cost against the separate reduce launch.
// The split-half weight-gradient kernel leaves S partial tiles part[s][tap][c][k] per 64x64 (c, k) tile; today ONE launch on every CU
// sums them and transposes into dw[k][c][tap] (wgrad_reduce_tile_kernel).  Folding inside the producing launch means: the block that
// draws the last ticket of a (c, k) tile reads the S - 1 other partials (written through by blocks on other XCDs) plus its own,
// sums in split order and writes the transposed tile -- on ONE CU per tile, while the launch's other blocks have already left.
// This probe times exactly that tail (one block per (c, k) tile, sc1 loads of freshly written partials) beside the all-CU reduce:
//   hipcc --offload-arch=gfx950 -O3 tools/micro/wgrad_fold_probe.hip -o /tmp/wgrad_fold_probe && /tmp/wgrad_fold_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// the producer stand-in: every partial written write-through, like a hand-off inside a launch
__global__ void fill_kernel(float* p, long n, unsigned seed) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u + seed;
        h ^= h >> 15;
        __hip_atomic_store(p + i, (float)(h & 0xffff) * 1e-4f - 3.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
// the product's final pass (csrc/conv3x3_wgrad.hip, wgrad_reduce_tile_kernel): 8 (c) x 32 (k) x 9 taps per block, every CU busy
__global__ __launch_bounds__(256) void reduce_tile_kernel(const float* __restrict__ part, float* __restrict__ dw, int C, int K, int S) {
    __shared__ float tile[32 * 73];
    const long ck = (long)C * K, n = 9 * ck;
    const int c0 = blockIdx.x * 8, k0 = blockIdx.y * 32;
    const int kl = threadIdx.x & 31, cs = threadIdx.x >> 5;
    float acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = 0.f;
    const float* p = part + (long)(c0 + cs) * K + k0 + kl;
    for (int sp = 0; sp < S; ++sp, p += n)
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[t] += p[t * ck];
#pragma unroll
    for (int t = 0; t < 9; ++t) tile[kl * 73 + cs * 9 + t] = acc[t];
    __syncthreads();
    for (int e = threadIdx.x; e < 32 * 72; e += 256) {
        const int k = e / 72, r = e - k * 72;
        dw[((long)(k0 + k) * C + c0) * 9 + r] = tile[k * 73 + r];
    }
}
// the tail an in-kernel fold adds: ONE block per 64x64 (c, k) tile sums its S partial tiles (16-byte sc1 loads, 8 in flight per
// lane) in split order and writes dw[k][c][tap] through LDS, 8 channels at a time
__global__ __launch_bounds__(256) void last_arriver_kernel(const float* part, float* __restrict__ dw, int C, int K, int S) {
    __shared__ float tile[64 * 73];                    // [k][c8 * 9 + tap]
    const long ck = (long)C * K, n = 9 * ck;
    const int c0 = blockIdx.x * 64, k0 = blockIdx.y * 64;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(part), 0, 0x7fffffff, 0x00020000);
    const int kq = threadIdx.x & 15, cl = threadIdx.x >> 4;           // 16 lanes x 16 bytes along k, 16 channel rows per pass
    for (int cg = 0; cg < 64; cg += 8) {                              // 8 channels x 9 taps x 64 k per LDS tile
        const int c = c0 + cg + (cl & 7);
        const int thalf = cl >> 3;                                    // taps 0..4 / 5..8 on the two halves of the block
        f32x4 acc[5];
#pragma unroll
        for (int t = 0; t < 5; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int sp = 0; sp < S; ++sp)
#pragma unroll
            for (int t = 0; t < 5; ++t) {
                const int tap = thalf * 5 + t;
                if (tap < 9) {
                    const long off = ((long)sp * n + (long)tap * ck + (long)c * K + k0 + 4 * kq) * 4;
                    acc[t] += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)off, 0, 16));
                }
            }
#pragma unroll
        for (int t = 0; t < 5; ++t) {
            const int tap = thalf * 5 + t;
            if (tap < 9)
#pragma unroll
                for (int e = 0; e < 4; ++e) tile[(4 * kq + e) * 73 + (cl & 7) * 9 + tap] = acc[t][e];
        }
        __syncthreads();
        for (int e = threadIdx.x; e < 64 * 72; e += 256) {
            const int k = e / 72, r = e - k * 72;
            dw[((long)(k0 + k) * C + c0 + cg) * 9 + r] = tile[k * 73 + r];
        }
        __syncthreads();
    }
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int shapes[][3] = {{512, 512, 8}, {256, 256, 32}, {512, 256, 16}};
    for (auto& sh : shapes) {
        const int C = sh[0], K = sh[1], S = sh[2];
        const long n = 9L * C * K;
        if ((long)S * n * 4 > 0x7fffffffL) { printf("skip\n"); continue; }
        float *part, *dwa, *dwb;
        CK(hipMalloc(&part, S * n * 4)); CK(hipMalloc(&dwa, n * 4)); CK(hipMalloc(&dwb, n * 4));
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        float ta = 0.f, tb = 0.f;
        const int REP = 20;
        for (int rep = 0; rep < REP + 2; ++rep) {
            for (int which = 0; which < 2; ++which) {
                hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, 0, part, S * n, (unsigned)rep);      // fresh partials
                CK(hipEventRecord(e0, 0));
                if (which == 0) hipLaunchKernelGGL(reduce_tile_kernel, dim3(C / 8, K / 32), dim3(256), 0, 0, part, dwa, C, K, S);
                else            hipLaunchKernelGGL(last_arriver_kernel, dim3(C / 64, K / 64), dim3(256), 0, 0, part, dwb, C, K, S);
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep >= 2) (which ? tb : ta) += ms;
            }
        }
        std::vector<float> ha(n), hb(n);
        CK(hipMemcpy(ha.data(), dwa, n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hb.data(), dwb, n * 4, hipMemcpyDeviceToHost));
        long bad = 0;
        for (long i = 0; i < n; ++i) bad += ha[i] != hb[i];
        printf("C=%d K=%d S=%d: reduce launch on every CU %6.1f us (+ ~1.5 us boundary)   last-arriver tail (%d blocks, %.1f MB each) %6.1f us   "
               "results %s\n", C, K, S, ta / REP * 1e3, (C / 64) * (K / 64), S * 9 * 64 * 64 * 4 / 1e6, tb / REP * 1e3,
               bad ? "DIFFER" : "bit-identical");
        CK(hipFree(part)); CK(hipFree(dwa)); CK(hipFree(dwb));
    }
    return 0;
}
