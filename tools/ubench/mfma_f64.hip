// v_mfma_f64_16x16x4_f64 on gfx950: (1) operand/result lane maps, (2) is D bit-for-bit the
// k-ordered FMA chain fma(a3,b3,fma(a2,b2,fma(a1,b1,fma(a0,b0,C)))) ?, (3) issue rate of one
// wave per SIMD with 28 independent accumulator tiles.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_f64 mfma_f64.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <cmath>
#include <vector>
typedef double double4_t __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// A[16][4], B[4][16], C[16][16] row-major in memory; lane l: a = A[l&15][l>>4], b = B[l>>4][l&15],
// c/d reg i = C[(l>>4) + 4*i][l&15]
__global__ __launch_bounds__(64) void one(const double* A, const double* B, const double* C, double* D) {
    const int l = threadIdx.x;
    const double a = A[(l & 15) * 4 + (l >> 4)], b = B[(l >> 4) * 16 + (l & 15)];
    double4_t c;
    for (int i = 0; i < 4; ++i) c[i] = C[((l >> 4) + 4 * i) * 16 + (l & 15)];
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int i = 0; i < 4; ++i) D[((l >> 4) + 4 * i) * 16 + (l & 15)] = c[i];
}

template <int NT>
__global__ __launch_bounds__(64) void rate(double* out, int iters, double a0, double b0) {
    double4_t acc[NT];
    for (int t = 0; t < NT; ++t) acc[t] = {0.0, 0.0, 0.0, 0.0};
    double a = a0 + threadIdx.x * 1e-9, b = b0 + threadIdx.x * 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
    }
    double s = 0;
    for (int t = 0; t < NT; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

static uint64_t rs = 88172645463325252ULL;
static double rnd() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return (double)(rs >> 11) / 9007199254740992.0; }

int main() {
    double *dA, *dB, *dC, *dD;
    CK(hipMalloc(&dA, 64 * 8)); CK(hipMalloc(&dB, 64 * 8)); CK(hipMalloc(&dC, 256 * 8)); CK(hipMalloc(&dD, 256 * 8));
    int bad_chain = 0, bad_rev = 0, trials = 200;
    for (int tr = 0; tr < trials; ++tr) {
        std::vector<double> A(64), B(64), C(256), D(256);
        for (auto& v : A) v = (rnd() - 0.3) * std::pow(10.0, (int)(rnd() * 12) - 6);
        for (auto& v : B) v = (rnd() - 0.5) * std::pow(10.0, (int)(rnd() * 8) - 4);
        for (auto& v : C) v = (rnd() - 0.5) * std::pow(10.0, (int)(rnd() * 10) - 5);
        if (tr == 0) for (auto& v : C) v = 0.0;
        if (tr == 1) { for (int i = 0; i < 64; i += 3) B[i] = 0.0; }            // zero products
        CK(hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice));
        CK(hipMemcpy(dC, C.data(), 2048, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(one, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
        CK(hipMemcpy(D.data(), dD, 2048, hipMemcpyDeviceToHost));
        for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) {
            double f = C[m * 16 + n], r = C[m * 16 + n];
            for (int k = 0; k < 4; ++k) f = std::fma(A[m * 4 + k], B[k * 16 + n], f);
            for (int k = 3; k >= 0; --k) r = std::fma(A[m * 4 + k], B[k * 16 + n], r);
            if (memcmp(&f, &D[m * 16 + n], 8)) ++bad_chain;
            if (memcmp(&r, &D[m * 16 + n], 8)) ++bad_rev;
        }
    }
    printf("ascending-k fma chain: %d mismatches of %d; descending-k chain: %d mismatches\n", bad_chain, trials * 256, bad_rev);
    double* d; CK(hipMalloc(&d, 8 * 64 * 4096));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 4000;
    for (int waves : {1024, 2048}) {
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(rate<28>, dim3(waves), dim3(64), 0, 0, d, iters, 1.0000001, 0.9999999);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
        }
        printf("mfma_f64_16x16x4 x28 tiles, waves=%d: %.3f ms  %.2f TFLOP/s\n", waves, ms, 2.0 * 1024 * 28 * iters * (double)waves / ms / 1e9);
    }
    for (int waves : {1024}) {
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(rate<4>, dim3(waves), dim3(64), 0, 0, d, iters, 1.0000001, 0.9999999);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
        }
        printf("mfma_f64_16x16x4 x4 tiles, waves=%d: %.3f ms  %.2f TFLOP/s\n", waves, ms, 2.0 * 1024 * 4 * iters * (double)waves / ms / 1e9);
    }
    // dependent chains: how long one wave waits for its own previous result (the small-ensemble kernel's phase A)
    for (int waves : {1024}) {
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(rate<1>, dim3(waves), dim3(64), 0, 0, d, iters, 1.0000001, 0.9999999);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
        }
        printf("mfma_f64_16x16x4 x1 tile (dependent chain), waves=%d: %.3f ms  %.2f TFLOP/s  %.1f ns per instruction\n", waves, ms,
               2.0 * 1024 * 1 * iters * (double)waves / ms / 1e9, ms * 1e6 / iters);
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(rate<2>, dim3(waves), dim3(64), 0, 0, d, iters, 1.0000001, 0.9999999);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
        }
        printf("mfma_f64_16x16x4 x2 tiles, waves=%d: %.3f ms  %.2f TFLOP/s  %.1f ns per instruction\n", waves, ms,
               2.0 * 1024 * 2 * iters * (double)waves / ms / 1e9, ms * 1e6 / iters / 2);
    }
    return bad_chain != 0;
}
