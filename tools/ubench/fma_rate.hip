// Micro-benchmark: FP64 FMA issue rate of ONE wave per SIMD versus TWO (and 4) on gfx950.
// Build: hipcc --offload-arch=gfx950 -O3 -o fma_rate fma_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int NACC>
__global__ __launch_bounds__(64) void fma_kernel(double* out, int iters, double a, double b) {
    double acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = threadIdx.x * 1e-3 + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_fma(acc[i], a, b);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(64) void fma32_kernel(float* out, int iters, float a, float b) {
    float acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = threadIdx.x * 1e-3f + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_fmaf(acc[i], a, b);
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
int main() {
    double* d; hipMalloc(&d, 8 * 64 * 8192);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int waves : {256, 512, 1024, 2048, 4096, 8192}) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(fma_kernel<48>, dim3(waves), dim3(64), 0, 0, d, iters, 1.0000001, 1e-9);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("f64 waves=%5d  %.3f ms  %.2f TFLOP/s\n", waves, ms, 2.0 * 48 * iters * 64.0 * waves / ms / 1e9);
        }
    }
    for (int waves : {1024, 2048, 4096}) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(fma32_kernel<48>, dim3(waves), dim3(64), 0, 0, (float*)d, iters, 1.0000001f, 1e-9f);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("f32 waves=%5d  %.3f ms  %.2f TFLOP/s\n", waves, ms, 2.0 * 48 * iters * 64.0 * waves / ms / 1e9);
        }
    }
    return 0;
}
