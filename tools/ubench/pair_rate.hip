// Micro-benchmark for the "helper wave" question: the phase-A lag loop of eval_model (one history load and
// one scalar table-row load per lag step, T x NF multiply-adds with the row in SGPRs) run
//   (1) as in the product kernel: one wave per SIMD, NF = 7 families, 1024 tiles;
//   (2) with TWO waves per SIMD on ONE tile, the seven families split 3 + 4 between them, both streaming
//       the same history rows (the second read of a row hits the CU's L1 / the XCD's L2);
//   (3) two waves per SIMD, each with all 7 families on a tile of its own (2048 tiles: doubled streams).
// Same number of multiply-adds in (1) and (2).  Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off
// -I../../include -I../../mcmc_amd/csrc -o pair_rate pair_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "mcmc_kernels.hip.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int T, int F0, int NF>
__device__ __forceinline__ void chunk_nf(double (&acc)[NF][T], double (&W)[T], const double (&N)[T], double8_t& r,
                                         const double* Rpos, int Lc, const double* hn_next_u, unsigned lane) {
    const double* rowBase = Rpos + (long long)(Lc - T) * 8;
    double8_t r1 = smem_row8(rowBase, T - 1);
#pragma unroll
    for (int u = 0; u < T; ++u) {
        const double8_t rnext = r1;
        r1 = smem_row8(rowBase, T - 2 - u);
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const int q = t + u;
            const double nk = (q < T) ? W[q < T ? q : 0] : N[q < T ? 0 : q - T];
#pragma unroll
            for (int f = 0; f < NF; ++f) acc[f][t] = __builtin_fma(nk, r[F0 + f], acc[f][t]);
        }
        W[u] = ld_row(hn_next_u, u, lane);
        r = rnext;
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int T, int F0, int NF>
__device__ __forceinline__ double sweep(const double* hist_u, const double* Rpos, int nBlocks, unsigned lane) {
    double total = 0;
    const double* hbase = hist_u;
#pragma unroll 1
    for (int b = 1; b <= nBlocks; ++b) {
        double acc[NF][T];
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int t = 0; t < T; ++t) acc[f][t] = 0.0;
        double WA[T], WB[T];
#pragma unroll
        for (int q = 0; q < T; ++q) { WA[q] = hbase[q * 64 + lane]; WB[q] = hbase[(q + T) * 64 + lane]; }
        int Lc = b * T + T - 1;
        double8_t r = smem_row8(Rpos, Lc);
        const double* hnx = hbase + 2 * T * 64;
        unsigned laneL = lane;
        asm volatile("" : "+v"(laneL));
        int nch = b;
#pragma unroll 1
        for (; nch >= 2; nch -= 2) {
            chunk_nf<T, F0, NF>(acc, WA, WB, r, Rpos, Lc, hnx, laneL);
            chunk_nf<T, F0, NF>(acc, WB, WA, r, Rpos, Lc - T, hnx + T * 64, laneL);
            hnx += 2 * T * 64;
            Lc -= 2 * T;
        }
        if (nch > 0) chunk_nf<T, F0, NF>(acc, WA, WB, r, Rpos, Lc, hnx, laneL);
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int t = 0; t < T; ++t) total += acc[f][t];
    }
    return total;
}

// mode 1 / 3: every wave all 7 families on its own tile
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(1, 2)))
void full_kernel(const double* hist, const double* Rpos, double* out, int rows, int nBlocks) {
    const unsigned lane = threadIdx.x & 63;
    const int tile = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64));
    const double v = sweep<8, 0, 7>(hist + (size_t)tile * rows * 64, Rpos, nBlocks, lane);
    out[(size_t)tile * 64 + lane] = v;
}
// mode 2: waves w and w+4 of a 512-thread workgroup share a SIMD and a tile; 3 + 4 families
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
void split_kernel(const double* hist, const double* Rpos, double* out, int rows, int nBlocks) {
    const unsigned lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / 64));
    const int tile = (int)blockIdx.x * 4 + (wave & 3);
    const double* h = hist + (size_t)tile * rows * 64;
    double v;
    if (wave < 4) v = sweep<8, 0, 3>(h, Rpos, nBlocks, lane);
    else v = sweep<8, 3, 4>(h, Rpos, nBlocks, lane);
    out[((size_t)tile * 2 + (wave >> 2)) * 64 + lane] = v;
}

int main() {
    const int rows = 512, nBlocks = 50, tiles1 = 1024;
    const size_t histN = (size_t)2 * tiles1 * rows * 64;
    double *hist, *tab, *out;
    CK(hipMalloc(&hist, histN * 8)); CK(hipMalloc(&tab, 600 * 8 * 8)); CK(hipMalloc(&out, (size_t)4 * tiles1 * 64 * 8));
    std::vector<double> h(histN), t(600 * 8);
    for (size_t i = 0; i < histN; ++i) h[i] = 1.0 + 1e-6 * (double)(i % 977);
    for (size_t i = 0; i < t.size(); ++i) t[i] = 1e-3 * (double)(1 + i % 13);
    CK(hipMemcpy(hist, h.data(), histN * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(tab, t.data(), t.size() * 8, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double fma1 = 7.0 * 64 * 64 * (double)tiles1 * (nBlocks * (nBlocks + 1) / 2);   // multiply-adds of modes 1 and 2
    for (int mode = 1; mode <= 3; ++mode) {
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipEventRecord(e0));
            if (mode == 1) hipLaunchKernelGGL(full_kernel, dim3(tiles1 / 4), dim3(256), 0, 0, hist, tab, out, rows, nBlocks);
            else if (mode == 2) hipLaunchKernelGGL(split_kernel, dim3(tiles1 / 4), dim3(512), 0, 0, hist, tab, out, rows, nBlocks);
            else hipLaunchKernelGGL(full_kernel, dim3(tiles1 / 4), dim3(512), 0, 0, hist, tab, out, rows, nBlocks);
            CK(hipGetLastError());
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep && ms < best) best = ms;
        }
        const double fma = mode == 3 ? 2 * fma1 : fma1;
        printf("mode %d (%s): %.3f ms, %.2f TFLOP/s issued\n", mode,
               mode == 1 ? "1 wave/SIMD, 7 families" : mode == 2 ? "2 waves/SIMD on one tile, 3+4 families" : "2 waves/SIMD, 7 families each, own tiles",
               best, 2 * fma / best / 1e9);
    }
    return 0;
}
