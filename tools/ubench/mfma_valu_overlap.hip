// Do v_mfma_f64_16x16x4_f64 and v_fma_f64 run concurrently from ONE wave per SIMD on gfx950?
// Times: 12 MFMA tiles alone, NV VALU FMAs alone, both interleaved in one loop body.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int NM, int NV>
__global__ __launch_bounds__(64) void k(double* out, int iters, double a0, double b0) {
    double4_t acc[NM > 0 ? NM : 1];
    double v[NV > 0 ? NV : 1];
    for (int t = 0; t < NM; ++t) acc[t] = {0.0, 0.0, 0.0, 0.0};
    for (int i = 0; i < NV; ++i) v[i] = i * 1e-3;
    double a = a0 + threadIdx.x * 1e-9, b = b0 + threadIdx.x * 1e-9;
    for (int it = 0; it < iters; ++it) {
        // interleave: after each MFMA, NV/NM VALU FMAs
#pragma unroll
        for (int t = 0; t < (NM > 0 ? NM : 1); ++t) {
            if (NM > 0) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < NV / (NM > 0 ? NM : 1); ++i) {
                const int j = t * (NV / (NM > 0 ? NM : 1)) + i;
                v[j] = __builtin_fma(v[j], a, b);
            }
        }
    }
    double s = 0;
    for (int t = 0; t < NM; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    for (int i = 0; i < NV; ++i) s += v[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int NM, int NV>
int run(double* d, const char* name) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 4000, waves = 1024;
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k<NM, NV>), dim3(waves), dim3(64), 0, 0, d, iters, 1.0000001, 0.9999999);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
    }
    const double fl = 2.0 * (1024.0 * NM + 64.0 * NV) * iters * waves;
    printf("%-28s %.3f ms  total %.2f TFLOP/s (mfma part %.2f, valu part %.2f)\n", name, ms, fl / ms / 1e9,
           2.0 * 1024 * NM * iters * waves / ms / 1e9, 2.0 * 64 * NV * iters * waves / ms / 1e9);
    return 0;
}
int main() {
    double* d; CK(hipMalloc(&d, 8 * 64 * 4096));
    run<12, 0>(d, "12 mfma");
    run<0, 96>(d, "96 valu fma");
    run<0, 192>(d, "192 valu fma");
    run<12, 96>(d, "12 mfma + 96 valu");
    run<12, 192>(d, "12 mfma + 192 valu");
    run<12, 288>(d, "12 mfma + 288 valu");
    run<12, 384>(d, "12 mfma + 384 valu");
    return 0;
}
