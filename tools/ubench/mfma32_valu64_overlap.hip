// Do v_mfma_f32_32x32x2_f32 (the fast mode's stream) and FP64 vector FMAs (the fast mode's day loop)
// run concurrently from ONE wave per SIMD on gfx950?  Times NM MFMAs alone, NV FP64 FMAs alone
// (independent, and as CH dependent chains like the day loop's feedback), and both interleaved.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float16v_t __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// NM MFMAs on 4 accumulators (as the stream has), NV FMAs spread over CH dependent chains
template <int NM, int NV, int CH>
__global__ __launch_bounds__(64) void k(double* out, int iters, float a0, float b0, double c0) {
    float16v_t acc[4];
    double v[CH > 0 ? CH : 1];
    for (int t = 0; t < 4; ++t)
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.0f;
    for (int i = 0; i < CH; ++i) v[i] = i * 1e-3;
    const float a = a0 + threadIdx.x * 1e-7f, b = b0 + threadIdx.x * 1e-7f;
    const double c = c0 + threadIdx.x * 1e-9;
    constexpr int PER = NM > 0 ? NV / NM : NV;       // FMAs after each MFMA
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < (NM > 0 ? NM : 1); ++t) {
            if (NM > 0) acc[t & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t & 3], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const int j = (t * PER + i) % (CH > 0 ? CH : 1);
                if (CH > 0) v[j] = __builtin_fma(v[j], c, c);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    double s = 0;
    for (int t = 0; t < 4; ++t)
        for (int i = 0; i < 16; ++i) s += acc[t][i];
    for (int i = 0; i < CH; ++i) s += v[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

// Two waves per SIMD (one 512-thread workgroup per CU: wave w sits on SIMD w % 4): waves 0..3 run MFMAs only,
// waves 4..7 FP64 FMAs only; `roles` bit 0 enables the MFMA waves, bit 1 the FMA waves.
__global__ __launch_bounds__(512) void k2(double* out, int iters, float a0, float b0, double c0, int roles) {
    const int w = threadIdx.x / 64;
    double s = 0;
    if (w < 4) {
        if (!(roles & 1)) return;
        float16v_t acc[4];
        for (int t = 0; t < 4; ++t)
            for (int i = 0; i < 16; ++i) acc[t][i] = 0.0f;
        const float a = a0 + threadIdx.x * 1e-7f, b = b0 + threadIdx.x * 1e-7f;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int t = 0; t < 16; ++t) acc[t & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t & 3], 0, 0, 0);
        for (int t = 0; t < 4; ++t)
            for (int i = 0; i < 16; ++i) s += acc[t][i];
    } else {
        if (!(roles & 2)) return;
        double v[16];
        for (int i = 0; i < 16; ++i) v[i] = i * 1e-3;
        const double c = c0 + threadIdx.x * 1e-9;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 192; ++i) v[i & 15] = __builtin_fma(v[i & 15], c, c);
        for (int i = 0; i < 16; ++i) s += v[i];
    }
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
int run2(double* d, int roles, const char* name) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 4000;
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k2, dim3(256), dim3(512), 0, 0, d, iters, 1.0000001f, 0.9999999f, 0.999999, roles);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
    }
    printf("%-44s %8.3f ms  %7.0f cycles/iter @2.4GHz\n", name, ms, ms * 1e-3 * 2.4e9 / iters);
    return 0;
}

template <int NM, int NV, int CH>
int run(double* d, const char* name) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 4000, waves = 1024;
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k<NM, NV, CH>), dim3(waves), dim3(64), 0, 0, d, iters, 1.0000001f, 0.9999999f, 0.999999);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
    }
    // cycles per loop iteration of one wave at a nominal 2.4 GHz
    printf("%-44s %8.3f ms  %7.0f cycles/iter @2.4GHz  (mfma alone would be %d, fma issue alone %d)\n", name, ms,
           ms * 1e-3 * 2.4e9 / iters, NM * 64, NV * 4);
    return 0;
}
int main() {
    double* d; CK(hipMalloc(&d, 8 * 64 * 4096));   // >= 256 x 512 results
    run<16, 0, 0>(d, "16 mfma");
    run<0, 96, 16>(d, "96 f64 fma, 16 chains");
    run<0, 96, 4>(d, "96 f64 fma, 4 chains");
    run<0, 96, 1>(d, "96 f64 fma, 1 chain");
    run<16, 96, 16>(d, "16 mfma + 96 fma (16 chains)");
    run<16, 96, 4>(d, "16 mfma + 96 fma (4 chains)");
    run<16, 96, 1>(d, "16 mfma + 96 fma (1 chain)");
    run<16, 192, 16>(d, "16 mfma + 192 fma (16 chains)");
    run<16, 192, 2>(d, "16 mfma + 192 fma (2 chains)");
    run<16, 48, 1>(d, "16 mfma + 48 fma (1 chain)");
    run2(d, 1, "two waves/SIMD: 16 mfma in wave A only");
    run2(d, 2, "two waves/SIMD: 192 f64 fma in wave B only");
    run2(d, 3, "two waves/SIMD: A mfma || B fma");
    return 0;
}
