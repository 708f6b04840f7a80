// Micro-benchmark: throughput of v_fmac_f64 with a DPP row_newbcast source against the plain
// instruction, one wave per SIMD (1024 waves) and two (2048), plus a value check.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int NACC, bool DPP>
__global__ __launch_bounds__(64) void k(double* out, int iters, double a) {
    double acc[NACC];
    double row = 1.0 + (threadIdx.x & 15) * 1e-9;      // lane c of each 16-lane row holds "R_c"
    double nk = a + threadIdx.x * 1e-12;
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            if constexpr (DPP) {
                if (i % 7 == 0) asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:0 row_mask:0xf bank_mask:0xf" : "+v"(acc[i]) : "v"(row), "v"(nk));
                else if (i % 7 == 1) asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:1 row_mask:0xf bank_mask:0xf" : "+v"(acc[i]) : "v"(row), "v"(nk));
                else if (i % 7 == 2) asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:2 row_mask:0xf bank_mask:0xf" : "+v"(acc[i]) : "v"(row), "v"(nk));
                else if (i % 7 == 3) asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(acc[i]) : "v"(row), "v"(nk));
                else if (i % 7 == 4) asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:4 row_mask:0xf bank_mask:0xf" : "+v"(acc[i]) : "v"(row), "v"(nk));
                else if (i % 7 == 5) asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(acc[i]) : "v"(row), "v"(nk));
                else asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:6 row_mask:0xf bank_mask:0xf" : "+v"(acc[i]) : "v"(row), "v"(nk));
            } else {
                acc[i] = __builtin_fma(row, nk, acc[i]);
            }
        }
        nk *= 0.999999;
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

// value check: lane l, accumulator c: fma(R[(l&~15)+c], n[l], acc) against the host
__global__ __launch_bounds__(64) void check(const double* R, const double* n, double* out) {
    const double row = R[threadIdx.x], nk = n[threadIdx.x];
    double a0 = 0.5, a3 = 0.25, a6 = 0.125;
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:0 row_mask:0xf bank_mask:0xf" : "+v"(a0) : "v"(row), "v"(nk));
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a3) : "v"(row), "v"(nk));
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:6 row_mask:0xf bank_mask:0xf" : "+v"(a6) : "v"(row), "v"(nk));
    out[threadIdx.x] = a0; out[64 + threadIdx.x] = a3; out[128 + threadIdx.x] = a6;
}

int main() {
    double* d; CK(hipMalloc(&d, 8 * 64 * 4096));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 20000;
    for (int dpp = 0; dpp < 2; ++dpp)
        for (int waves : {1024, 2048}) {
            float ms = 0;
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipEventRecord(e0));
                if (dpp) hipLaunchKernelGGL((k<49, true>), dim3(waves), dim3(64), 0, 0, d, iters, 1.0000001);
                else hipLaunchKernelGGL((k<49, false>), dim3(waves), dim3(64), 0, 0, d, iters, 1.0000001);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms, e0, e1));
            }
            printf("%s waves=%d  %.3f ms  %.2f TFLOP/s\n", dpp ? "fmac_f64_dpp" : "fma_f64     ", waves, ms, 2.0 * 49 * iters * 64.0 * waves / ms / 1e9);
        }
    std::vector<double> R(64), n(64), o(192);
    for (int i = 0; i < 64; ++i) { R[i] = 1.0 / (3 + i); n[i] = 7.0 / (11 + i); }
    double *dR, *dn, *dout;
    CK(hipMalloc(&dR, 512)); CK(hipMalloc(&dn, 512)); CK(hipMalloc(&dout, 192 * 8));
    CK(hipMemcpy(dR, R.data(), 512, hipMemcpyHostToDevice)); CK(hipMemcpy(dn, n.data(), 512, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(check, dim3(1), dim3(64), 0, 0, dR, dn, dout);
    CK(hipMemcpy(o.data(), dout, 192 * 8, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        const int base = l & ~15;
        if (o[l] != __builtin_fma(R[base + 0], n[l], 0.5)) ++bad;
        if (o[64 + l] != __builtin_fma(R[base + 3], n[l], 0.25)) ++bad;
        if (o[128 + l] != __builtin_fma(R[base + 6], n[l], 0.125)) ++bad;
    }
    printf("value check: %d mismatches of 192\n", bad);
    return bad != 0;
}
