// Micro-benchmark: the lag loop of the wave mapping's phase A in isolation.  NACC accumulators, each receiving a
// dependent chain of v_fmac_f64 (plain, or with a DPP row_newbcast source), NACC independent chains interleaved --
// how many independent chains does a lone wave need to issue FP64 multiply-adds at its full rate?
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int NACC, bool DPP>
__global__ __launch_bounds__(64) void k(double* out, int iters, double a) {
    double acc[NACC];
    double t[NACC];
    double h = a + threadIdx.x * 1e-12;
#pragma unroll
    for (int i = 0; i < NACC; ++i) { acc[i] = i; t[i] = 1.0 + ((threadIdx.x + i) & 15) * 1e-9; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int lag = 0; lag < 16; ++lag) {
#pragma unroll
            for (int i = 0; i < NACC; ++i) {
                if constexpr (DPP) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(acc[i]) : "v"(t[i]), "v"(h));
                else asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(acc[i]) : "v"(t[i]), "v"(h));
            }
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int NACC, bool DPP>
int run(double* d, const char* name) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 4000, waves = 1024;
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k<NACC, DPP>), dim3(waves), dim3(64), 0, 0, d, iters, 1.0000001);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
    }
    const double fm = 16.0 * NACC * iters;
    printf("%s chains=%2d  %.3f ms  %.2f ns per multiply-add per wave (%.1f cycles at 2.4 GHz)\n", name, NACC, ms, ms * 1e6 / fm, ms * 1e6 / fm * 2.4);
    return 0;
}
int main() {
    double* d; CK(hipMalloc(&d, 8 * 64 * 1024));
    run<7, false>(d, "v_fmac_f64    "); run<7, true>(d, "v_fmac_f64_dpp");
    run<14, false>(d, "v_fmac_f64    "); run<14, true>(d, "v_fmac_f64_dpp");
    run<28, false>(d, "v_fmac_f64    "); run<28, true>(d, "v_fmac_f64_dpp");
    run<3, false>(d, "v_fmac_f64    "); run<3, true>(d, "v_fmac_f64_dpp");
    run<1, false>(d, "v_fmac_f64    "); run<1, true>(d, "v_fmac_f64_dpp");
    return 0;
}
