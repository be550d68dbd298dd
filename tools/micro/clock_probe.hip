// clock_probe.hip -- what the shader clock runs at under a VALU-only load, what __builtin_readcyclecounter counts, and the
// issue interval of dependent / independent fp32 FMAs for one wavefront per SIMD (the regime kp_step_kernel lives in).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int DEP>
__global__ void k_fma(float* out, unsigned long long* cyc, unsigned long long* rt, int iters) {
    float a[8];
    for (int i = 0; i < 8; i++) a[i] = threadIdx.x * 0.001f + i;
    const float b = 1.0001f, c = 0.5f;
    unsigned long long t0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    for (int it = 0; it < iters; it++) {
        if (DEP) {
#pragma unroll
            for (int k = 0; k < 8; k++) a[0] = __builtin_fmaf(a[0], b, c);      // one dependent chain
        } else {
#pragma unroll
            for (int k = 0; k < 8; k++) a[k] = __builtin_fmaf(a[k], b, c);      // eight independent chains
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    float s = 0; for (int i = 0; i < 8; i++) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { cyc[blockIdx.x] = t1 - t0; rt[blockIdx.x] = r1 - r0; }
}

int main() {
    const int iters = 2000000;
    for (int cfg = 0; cfg < 6; cfg++) {
        const int dep = cfg & 1, wpc = cfg < 2 ? 1 : (cfg < 4 ? 4 : 8);     // waves per CU (256 CUs)
        const int blocks = 256 * wpc;
        float* out; unsigned long long *cyc, *rt;
        hipMalloc(&out, blocks * 64 * 4); hipMalloc(&cyc, blocks * 8); hipMalloc(&rt, blocks * 8);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            if (dep) hipLaunchKernelGGL(k_fma<1>, dim3(blocks), dim3(64), 0, 0, out, cyc, rt, iters);
            else hipLaunchKernelGGL(k_fma<0>, dim3(blocks), dim3(64), 0, 0, out, cyc, rt, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(blocks), r(blocks);
        hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost); hipMemcpy(r.data(), rt, blocks * 8, hipMemcpyDeviceToHost);
        double c = 0, w = 0; for (int i = 0; i < blocks; i++) { c += h[i]; w += r[i]; } c /= blocks; w /= blocks;
        printf("%s waves/CU=%d: %.3f ms  readcyclecounter=%.0f (%.1f MHz)  wall_clock64=%.0f (%.1f MHz)  cyc/FMA=%.3f  ns/FMA=%.3f\n",
               dep ? "dependent  " : "independent", wpc, ms, c, c / ms / 1e3, w, w / ms / 1e3, c / (8.0 * iters), ms * 1e6 / (8.0 * iters));
    }
    return 0;
}
