// Prototype of the fast mode's phase A on the FP32 matrix cores: for one
// tile of 64 chains and one 8-day block starting at day j0,
//     acc[chain][t][c] = sum over source days k < K of hist[k][chain] * table[j0 + t - k][c]
// with v_mfma_f32_32x32x2_f32: M = 32 chains, N = 32 = 4 days x 8 curves, K = 2 source days.
// (1) checks the operand / result lane maps against a host sum, (2) times the loop with one wave
// per SIMD, every wave streaming a history slab of its own.  Compiler builtins only.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_conv_proto mfma_conv_proto.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef float float16_t __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr int PAD = 64;          // rows of zeros behind every history slab: the prefetch ring reads ahead
constexpr int RING = 16;         // source-day pairs in flight (8: 86 ns per source day, latency-bound)

// hist: [tiles][K + PAD][64], table: [rowsT][8], out: [tiles][64 chains][64 = 8 days x 8 curves]
template <bool TRANSPOSE>
__global__ __launch_bounds__(64) void conv_mfma(const float* __restrict__ hist, const float* __restrict__ table,
                                                float* __restrict__ out, int K, int rowsT, int j0, int reps) {
    extern __shared__ float lds[];
    float* tr = lds + rowsT * 8;                                 // TRANSPOSE: [64 chains][65] floats behind the table
    float keep = 0.0f;
    for (int i = threadIdx.x; i < rowsT * 8; i += 64) lds[i] = table[i];
    __syncthreads();
    const int l = threadIdx.x, half = l >> 5, n = l & 31;
    const float* h = hist + (size_t)blockIdx.x * (K + PAD) * 64;
    float16_t acc[2][2];
    for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 2; ++b)
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.0f;
    for (int rep = 0; rep < reps; ++rep) {
        float ra[RING][2];
#pragma unroll
        for (int u = 0; u < RING; ++u) {
            ra[u][0] = h[(2 * u + half) * 64 + n];
            ra[u][1] = h[(2 * u + half) * 64 + n + 32];
        }
        /* B operands one pair ahead of their MFMAs (LDS latency off the critical path); one running LDS
         * index and one running history pointer, no clamps (the shape check guarantees j0 >= K + 2) */
        int li = (j0 - half) * 8 + n;
        float b0 = lds[li], b1 = lds[li + 32];
        const float* hp = h + (2 * RING + half) * 64 + n;       // refill source of pair 0
        for (int k0 = 0; k0 < K; k0 += 2 * RING) {
#pragma unroll
            for (int u = 0; u < RING; ++u) {
                li -= 16;                                        // table row of the NEXT pair: two days on
                const float nb0 = lds[li];                       // days j0 .. j0+3
                const float nb1 = lds[li + 32];                  // days j0+4 .. j0+7
                /* the MFMAs block the wave's issue for 64 cycles each: the reads for the next pair have
                 * to be in the queue BEFORE them, or their latency is exposed after them */
                __builtin_amdgcn_sched_barrier(0);
                const float a0 = ra[u][0], a1 = ra[u][1];
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
                ra[u][0] = hp[u * 128];                          // pair k + 2*RING: rows <= K + 2*RING - 1 < K + PAD
                ra[u][1] = hp[u * 128 + 32];
                b0 = nb0;
                b1 = nb1;
                __builtin_amdgcn_sched_barrier(0);               // keep every pair's loads in its step
            }
            hp += 2 * RING * 64;
        }
        if constexpr (TRANSPOSE) {
            /* what the end of every 8-day block would do: the 64 sums of a chain from the MFMA layout
             * (a lane holds one column of 16 chains per tile) to chain-per-lane, through LDS rows of 65 */
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int v = 0; v < 16; ++v)
                        tr[(32 * mt + 8 * (v / 4) + 4 * half + (v % 4)) * 65 + 32 * nt + n] = acc[mt][nt][v];
            __syncthreads();
            float vals[64];
#pragma unroll
            for (int c = 0; c < 64; ++c) vals[c] = tr[l * 65 + c];
            __syncthreads();
            if (rep == reps - 1) {
                float* o = out + (size_t)blockIdx.x * 64 * 64;
#pragma unroll
                for (int c = 0; c < 64; ++c) o[l * 64 + c] = vals[c];
            } else {
#pragma unroll
                for (int c = 0; c < 64; ++c) keep += vals[c];
            }
        }
    }
    if constexpr (TRANSPOSE) {
        if (keep == 123.456f) out[0] = keep;                     // keeps the read-back alive
        return;
    }
    float* o = out + (size_t)blockIdx.x * 64 * 64;
    for (int mt = 0; mt < 2; ++mt)
        for (int nt = 0; nt < 2; ++nt)
            for (int v = 0; v < 16; ++v) {
                const int chain = 32 * mt + 8 * (v / 4) + 4 * half + (v % 4);
                o[chain * 64 + 32 * nt + n] = acc[mt][nt][v];
            }
}

int main() {
    const int K = 384, j0 = 400, rowsT = 416, tilesCheck = 2, tiles = 1024;
    if (K % (2 * RING) != 0 || j0 < K + 2 || j0 + 7 + 4 >= rowsT || 2 * RING > PAD) { printf("bad shape\n"); return 1; }
    std::vector<float> hist((size_t)tiles * (K + PAD) * 64, 0.0f), table((size_t)rowsT * 8, 0.0f);
    srand(7);
    for (int t = 0; t < tiles; ++t)
        for (int k = 0; k < K; ++k)
            for (int c = 0; c < 64; ++c) hist[((size_t)t * (K + PAD) + k) * 64 + c] = (float)(rand() % 2000) / 1000.0f;
    for (int r = 0; r < rowsT; ++r)
        for (int c = 0; c < 7; ++c) table[r * 8 + c] = (float)(rand() % 1000) / 1000.0f;
    float *dh, *dt, *dout;
    CK(hipMalloc(&dh, hist.size() * 4)); CK(hipMalloc(&dt, table.size() * 4)); CK(hipMalloc(&dout, (size_t)tiles * 4096 * 4));
    CK(hipMemcpy(dh, hist.data(), hist.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dt, table.data(), table.size() * 4, hipMemcpyHostToDevice));
    const size_t shmem = (size_t)rowsT * 8 * 4 + 64 * 65 * 4;
  for (int variant = 0; variant < 2; ++variant) {
    if (variant == 0) hipLaunchKernelGGL(conv_mfma<false>, dim3(tilesCheck), dim3(64), shmem, 0, dh, dt, dout, K, rowsT, j0, 1);
    else hipLaunchKernelGGL(conv_mfma<true>, dim3(tilesCheck), dim3(64), shmem, 0, dh, dt, dout, K, rowsT, j0, 1);
    CK(hipGetLastError()); CK(hipDeviceSynchronize());
    std::vector<float> out((size_t)tilesCheck * 4096);
    CK(hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0;
    for (int t = 0; t < tilesCheck; ++t)
        for (int ch = 0; ch < 64; ++ch)
            for (int d = 0; d < 8; ++d)
                for (int c = 0; c < 8; ++c) {
                    double s = 0;
                    for (int k = 0; k < K; ++k) s += (double)hist[((size_t)t * (K + PAD) + k) * 64 + ch] * table[(j0 + d - k) * 8 + c];
                    const double got = out[(size_t)t * 4096 + ch * 64 + d * 8 + c];
                    const double err = fabs(got - s) / (fabs(s) + 1e-3);
                    if (err > worst) worst = err;
                }
    printf("%s: worst relative error against the host sum %.3g (%s)\n", variant ? "after the LDS transpose" : "lane maps", worst, worst < 1e-4 ? "ok" : "WRONG");
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 20;
    float ms = 0;
    for (int r = 0; r < 2; ++r) {
        CK(hipEventRecord(e0));
        if (variant == 0) hipLaunchKernelGGL(conv_mfma<false>, dim3(tiles), dim3(64), shmem, 0, dh, dt, dout, K, rowsT, j0, reps);
        else hipLaunchKernelGGL(conv_mfma<true>, dim3(tiles), dim3(64), shmem, 0, dh, dt, dout, K, rowsT, j0, reps);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
    }
    const double pairs = (double)tiles * reps * K * 64 * 8;               // (chain, day, source day) triples
    printf("%s1024 waves, %d source days x %d reps: %.3f ms  %.1f TFLOP/s issued (8 columns), %.1f useful (7 curves); "
           "%.0f ns per source day per wave\n", variant ? "+ transpose per rep: " : "", K, reps, ms, pairs * 8 * 2 / ms / 1e9, pairs * 7 * 2 / ms / 1e9,
           ms * 1e6 / ((double)reps * K));
  }
    return 0;
}
