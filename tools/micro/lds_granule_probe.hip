// How many 64-thread workgroups fit on a CU as a function of their dynamic LDS size (gfx950): each workgroup just waits a fixed time, so the
// launch time of 256 x 48 workgroups steps exactly where the resident count per CU changes.   hipcc --offload-arch=gfx950 -O2 lds_granule_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(64) void wait_kernel(unsigned long long ticks, int* sink) {
    extern __shared__ int lds[];
    lds[threadIdx.x] = threadIdx.x;
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (lds[(threadIdx.x + 1) & 63] == -1) sink[0] = 1;
}
int main() {
    int* sink; hipMalloc(&sink, 4);
    hipFuncSetAttribute((const void*)wait_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const unsigned long long ticks = 2000;      // 100 MHz wall clock: 20 us per workgroup
    const int per_cu = 48, grid = 256 * per_cu;
    int last = -1;
    for (int bytes = 8192; bytes <= 24576; bytes += 128) {
        hipLaunchKernelGGL(wait_kernel, dim3(grid), dim3(64), bytes, 0, ticks, sink);
        hipEventRecord(a);
        hipLaunchKernelGGL(wait_kernel, dim3(grid), dim3(64), bytes, 0, ticks, sink);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        const double rounds = ms * 1e3 / 20.0;            // 48 / resident
        const int resident = (int)(per_cu / rounds + 0.5);
        if (resident != last) { printf("dyn LDS %6d B: %.3f ms -> ~%d workgroups per CU\n", bytes, ms, resident); last = resident; }
    }
    return 0;
}
