// Measurement aid (round 5): take N CUs away from every other kernel for a while -- each block of the blocker allocates the whole
// LDS of a CU (160 KB), so no block that needs LDS can share its CU, and sleeps (s_sleep: no issue slots, no power to speak of)
// until `cycles` ticks of the 100 MHz reference clock have passed.  With it tools/micro/cu_share_probe.py answers: what does a power-capped conv kernel
// lose when it runs on 224 / 208 / 192 of the 256 CUs?  (hipExtStreamCreateWithCUMask is ignored on this stack,
// tools/micro/cu_mask_probe.hip.)   hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/micro/cu_blocker.hip -o tools/micro/libcu_blocker.so
#include <hip/hip_runtime.h>

__global__ __launch_bounds__(64) void cu_blocker_kernel(long long cycles, unsigned* __restrict__ where) {
    extern __shared__ char lds[];
    lds[threadIdx.x] = 0;                                        // (keeps the allocation)
    if (threadIdx.x == 0) {
        where[2 * blockIdx.x] = __builtin_amdgcn_s_getreg((31 << 11) | 4);       // HW_ID
        where[2 * blockIdx.x + 1] = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // XCC_ID
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();          // constant 100 MHz reference clock
    while ((long long)(__builtin_amdgcn_s_memrealtime() - t0) < cycles) __builtin_amdgcn_s_sleep(127);
}

extern "C" __attribute__((visibility("default"))) int cu_blocker_launch(int nblocks, long long cycles, unsigned* where, hipStream_t st) {
    static bool once = false;
    const int lds = 160 * 1024;
    if (!once) {
        if (hipFuncSetAttribute((const void*)cu_blocker_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return 1;
        once = true;
    }
    hipLaunchKernelGGL(cu_blocker_kernel, dim3(nblocks), dim3(64), lds, st, cycles, where);
    return (int)hipGetLastError();
}
