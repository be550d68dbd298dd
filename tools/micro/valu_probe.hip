// valu_probe.hip -- how many cycles one wave64 fp32 VALU instruction occupies a SIMD on gfx950, measured at 1 .. 8 waves per SIMD with
// independent FMA chains (the microarchitecture guide's "2 cycles" vs the 4 cycles the VALU-busy estimate used).
// SIMD-side cost = elapsed shader cycles / (wave instructions issued on that SIMD).   hipcc --offload-arch=gfx950 -O3 valu_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int PACKED>
__global__ void k_fma(float* out, unsigned long long* cyc, int iters) {
    float a[16];
    for (int i = 0; i < 16; i++) a[i] = threadIdx.x * 0.001f + i;
    const float b = 1.0001f, c = 0.5f;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int k = 0; k < 16; k++) {                                       // sixteen independent chains per lane
            if (PACKED) a[k] = __builtin_fmaf(a[k], b, c);                   // the compiler pairs these into 8 v_pk_fma_f32
            else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));   // 16 plain v_fma_f32
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0; for (int i = 0; i < 16; i++) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

int main() {
    const int iters = 400000;
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    printf("%s: %d CUs, clockRate %d kHz\n", prop.gcnArchName, cus, prop.clockRate);
    for (int packed = 0; packed < 2; packed++)
    for (int wps = 1; wps <= 8; wps *= 2) {                 // waves per SIMD; a workgroup of 4 * wps waves fills one CU's four SIMDs
        const int threads = 64 * 4 * wps > 1024 ? 1024 : 64 * 4 * wps, per_cu = (64 * 4 * wps) / threads, blocks = cus * per_cu;
        const int nw = blocks * threads / 64;
        float* out; unsigned long long* cyc;
        hipMalloc(&out, (size_t)blocks * threads * 4); hipMalloc(&cyc, nw * 8);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        float ms = 0;
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            if (packed) hipLaunchKernelGGL(k_fma<1>, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
            else hipLaunchKernelGGL(k_fma<0>, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        std::vector<unsigned long long> h(nw);
        hipMemcpy(h.data(), cyc, nw * 8, hipMemcpyDeviceToHost);
        double c = 0; for (int i = 0; i < nw; i++) c += h[i]; c /= nw;
        const double wave_insts_per_simd = (double)wps * (packed ? 8.0 : 16.0) * iters;
        printf("%s waves/SIMD=%d: %.3f ms, %.0f cycles/wave (%.0f MHz), SIMD cycles per wave64 VALU instruction = %.3f, chip fp32 rate = %.1f TFLOP/s\n",
               packed ? "v_pk_fma_f32" : "v_fma_f32   ", wps, ms, c, c / ms / 1e3, c / wave_insts_per_simd, 2.0 * 64.0 * 16.0 * iters * nw / (ms * 1e-3) / 1e12);
        hipFree(out); hipFree(cyc);
    }
    return 0;
}
