// Micro-benchmark: issue rate of v_pk_fma_f32 against v_fma_f64 in the shape the lag loops use
// (acc += x * row, x a VGPR broadcast, row wave-uniform kernel arguments = SGPRs), with one,
// two and four waves per SIMD on gfx950.  Compiler-generated code only.
// Build: hipcc --offload-arch=gfx950 -O3 -o pkfma_rate pkfma_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float2_t __attribute__((ext_vector_type(2)));
struct Row { float r[8]; };
struct RowD { double r[4]; };
template <int NDAY>
__global__ __launch_bounds__(64) void pk_kernel(float* out, int iters, Row row, const float* xs) {
    float2_t acc[4][NDAY];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int t = 0; t < NDAY; ++t) acc[c][t] = float2_t{threadIdx.x * 1e-3f, (float)c};
    float x[NDAY];
#pragma unroll
    for (int t = 0; t < NDAY; ++t) x[t] = xs[t * 64 + threadIdx.x];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < NDAY; ++t)
#pragma unroll
            for (int c = 0; c < 4; ++c)
                acc[c][t] = __builtin_elementwise_fma(float2_t{x[t], x[t]}, float2_t{row.r[2 * c], row.r[2 * c + 1]}, acc[c][t]);
    }
    float s = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int t = 0; t < NDAY; ++t) s += acc[c][t].x + acc[c][t].y;
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <int NDAY>
__global__ __launch_bounds__(64) void f64_kernel(double* out, int iters, RowD row, const double* xs) {
    double acc[4][NDAY];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int t = 0; t < NDAY; ++t) acc[c][t] = threadIdx.x * 1e-3 + c;
    double x[NDAY];
#pragma unroll
    for (int t = 0; t < NDAY; ++t) x[t] = xs[t * 64 + threadIdx.x];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < NDAY; ++t)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c][t] = __builtin_fma(x[t], row.r[c], acc[c][t]);
    }
    double s = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int t = 0; t < NDAY; ++t) s += acc[c][t];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
int main() {
    void* d; hipMalloc(&d, 8 * 64 * 8192);
    void* xs; hipMalloc(&xs, 8 * 64 * 16); hipMemset(xs, 0, 8 * 64 * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    Row row; for (int i = 0; i < 8; ++i) row.r[i] = 1e-3f * (i + 1);
    RowD rowd; for (int i = 0; i < 4; ++i) rowd.r[i] = 1e-3 * (i + 1);
    constexpr int ND = 8;
    for (int waves : {1024, 2048, 4096}) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(pk_kernel<ND>, dim3(waves), dim3(64), 0, 0, (float*)d, iters, row, (const float*)xs);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double instr = 4.0 * ND * iters;              // per wave
            if (rep) printf("pk_f32 waves=%5d  %.3f ms  %.2f ns/instr/wave  %.2f TFLOP/s\n", waves, ms, ms * 1e6 / instr / ((waves + 1023) / 1024),
                            4.0 * instr * 64.0 * waves / ms / 1e9);
        }
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(f64_kernel<ND>, dim3(waves), dim3(64), 0, 0, (double*)d, iters, rowd, (const double*)xs);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double instr = 4.0 * ND * iters;
            if (rep) printf("f64    waves=%5d  %.3f ms  %.2f ns/instr/wave  %.2f TFLOP/s\n", waves, ms, ms * 1e6 / instr / ((waves + 1023) / 1024),
                            2.0 * instr * 64.0 * waves / ms / 1e9);
        }
    }
    return 0;
}
