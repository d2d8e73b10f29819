// Operand / result layout of v_mfma_f32_4x4x1_16B_f32 on gfx950, read off the instruction itself: every lane feeds a = 1 + lane
// and b = 100 * (1 + lane); D[e] of lane l then names its (A lane, B lane) pair:  D = a_src * b_src.
// hipcc --offload-arch=gfx950 -O2 tools/micro/mfma4x4_probe.hip -o /tmp/mfma4x4_probe && /tmp/mfma4x4_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(float* out) {
    const int l = threadIdx.x;
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(1.f + l, 100.f * (1 + l), c, 0, 0, 0);
    for (int e = 0; e < 4; ++e) out[l * 4 + e] = c[e];
}
int main() {
    float* d;
    hipMalloc(&d, 256 * 4);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    float h[256];
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int e = 0; e < 4; ++e) {
            const long v = (long)(h[l * 4 + e] + 0.5f);
            // expected if D_blk[i = e][j = l % 4] = A_blk[i] * B_blk[j] with blk = l / 4:  a lane = 4 * (l / 4) + e, b lane = l
            const long want = (long)(1 + 4 * (l / 4) + e) * 100 * (1 + l);
            if (v != want) ++bad;
            if (l < 8 || v != want) printf("lane %2d e %d  D = %ld  (expected %ld)\n", l, e, v, want);
        }
    printf("4x4x1 layout assumption (A row = lane %% 4 of block lane / 4, D[e] = row e, column lane %% 4): %s\n", bad ? "WRONG" : "confirmed");
    return 0;
}
