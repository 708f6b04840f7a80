// Prototype of DESIGN.md section 8, 2(d): the fast mode's phase A on the BF16 matrix cores with FP32 recovered
// from three-way BF16 splits.  For one tile of 64 chains and one 9-day block starting at day j0,
//     acc[chain][t][c] = sum over source days k < K of hist[k][chain] * R_c[j0 + t - k]      (7 curves c)
// with v_mfma_f32_32x32x16_bf16: M = 32 chains, N = 32 of the 63 columns (7 t + c), K = 16 source days, and
// the six products a1 b1, a1 b2, a2 b1, a2 b2, a1 b3, a3 b1 of x = x1 + x2 + x3 (each a BF16) per FP32 product.
//   * history: [comp][group of 8 days][chain][8] BF16 -- one 16-byte load per lane, comp and chain tile
//   * table:   per comp and PARITY copy, [curve][reversed lag] BF16 in LDS: the 8 consecutive lags a lane needs
//              are 4 aligned 4-byte reads (copy p holds the row shifted by p elements)
// (1) accuracy against a double sum, beside the FP32-MFMA form's; (2) time per 16 source days, one wave per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_bf16x3_proto mfma_bf16x3_proto.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr int T = 9, NC = 7;
constexpr int NCOMP = 3;

// hs: [tiles][NCOMP][groups][64][8] u16;  tb: [NCOMP][2][NC][rowLen] u16 (rowLen even);  out: [tiles][64][64] floats
// xmax: reversed-lag index of lag 0.  NPROD: 6 (full) or 3 (a1b1, a1b2, a2b1: ~2^-16) for comparison.
template <int NPROD>
__global__ __launch_bounds__(64) void conv_bf16(const uint16_t* __restrict__ hs, const uint16_t* __restrict__ tb,
                                                float* __restrict__ out, int K, int groups, int rowLen, int xmax,
                                                int j0, int reps) {
    extern __shared__ uint32_t lds[];
    const int tabWords = NCOMP * 2 * NC * rowLen / 2;
    for (int i = threadIdx.x; i < tabWords; i += 64) lds[i] = ((const uint32_t*)tb)[i];
    __syncthreads();
    const int l = threadIdx.x, h = l >> 5, r = l & 31;
    f32x16 acc[2][2];
    for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 2; ++b)
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.0f;
    // this lane's two columns: n = r and 32 + r -> (t, c); column 63 does not exist: it reads curve 0, day 9 (unused)
    int wordBase[2];                                     // LDS word index of element x0 (parity folded in), comp 0
    for (int nt = 0; nt < 2; ++nt) {
        const int n = 32 * nt + r, t = n / NC, c = n % NC;
        const int L0 = j0 + t - 8 * h;                   // lag of element j = 0 at k0 = 0
        const int x0 = xmax - L0, p = x0 & 1;
        wordBase[nt] = ((p * NC + c) * rowLen + (x0 - p)) / 2;
    }
    const int compWords = 2 * NC * rowLen / 2;
    const uint4* hp = (const uint4*)hs + ((size_t)blockIdx.x * NCOMP * groups + h) * 64 + r;   // comp 0, group h, chain r
    const size_t compStride = (size_t)groups * 64;       // uint4s per comp
    auto loadA = [&](bf16x8 (&A)[2][NCOMP], int k0) {
        const int g = k0 / 8;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int q = 0; q < NCOMP; ++q)
                A[mt][q] = __builtin_bit_cast(bf16x8, hp[q * compStride + (size_t)g * 64 + 32 * mt]);
    };
    auto loadB = [&](bf16x8 (&B)[2][NCOMP], int k0) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int q = 0; q < NCOMP; ++q) {
                const uint32_t* w = lds + q * compWords + wordBase[nt] + k0 / 2;
                const uint4 v = {w[0], w[1], w[2], w[3]};
                B[nt][q] = __builtin_bit_cast(bf16x8, v);
            }
    };
    for (int rep = 0; rep < reps; ++rep) {
        bf16x8 A[2][NCOMP], B[2][NCOMP];
        loadA(A, 0);
        loadB(B, 0);
#pragma unroll 1
        for (int k0 = 0; k0 < K; k0 += 16) {
            bf16x8 An[2][NCOMP], Bn[2][NCOMP];
            const int k1 = min(k0 + 16, K - 16);          // operands of the next step, requested ahead of this step's MFMAs
            loadA(An, k1);
            loadB(Bn, k1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    f32x16 c = acc[mt][nt];
                    if (NPROD >= 6) {                     // the small terms first
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[mt][2], B[nt][0], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[mt][0], B[nt][2], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[mt][1], B[nt][1], c, 0, 0, 0);
                    }
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[mt][1], B[nt][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[mt][0], B[nt][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[mt][0], B[nt][0], c, 0, 0, 0);
                    acc[mt][nt] = c;
                }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int q = 0; q < NCOMP; ++q) { A[mt][q] = An[mt][q]; B[mt][q] = Bn[mt][q]; }
        }
    }
    float* o = out + (size_t)blockIdx.x * 64 * 64;
    for (int mt = 0; mt < 2; ++mt)
        for (int nt = 0; nt < 2; ++nt)
            for (int v = 0; v < 16; ++v) {
                const int chain = 32 * mt + (v & 3) + 8 * (v >> 2) + 4 * h;
                o[chain * 64 + 32 * nt + r] = acc[mt][nt][v];
            }
}

static uint16_t f2bf(float x) {                           // round to nearest even
    uint32_t u; memcpy(&u, &x, 4);
    u += 0x7FFF + ((u >> 16) & 1);
    return (uint16_t)(u >> 16);
}
static float bf2f(uint16_t b) { uint32_t u = (uint32_t)b << 16; float x; memcpy(&x, &u, 4); return x; }
static void split3(float x, uint16_t s[3]) {
    s[0] = f2bf(x); float r1 = x - bf2f(s[0]);
    s[1] = f2bf(r1); float r2 = r1 - bf2f(s[1]);
    s[2] = f2bf(r2);
}

int main() {
    const int K = 384, j0 = 396, tilesCheck = 2, tiles = 1024, groups = K / 8;
    const int maxLag = j0 + T + 8, xmax = maxLag + 1, rowLen = (xmax + 8 + K + 16 + 1) & ~1;
    if (K % 16 || j0 < K) { printf("bad shape\n"); return 1; }
    std::vector<float> hist((size_t)tiles * K * 64), table((size_t)(maxLag + 2) * NC);
    srand(7);
    // new infections per day span orders of magnitude; table values in [0, 1)
    for (size_t i = 0; i < hist.size(); ++i) hist[i] = (float)((rand() % 200000) / 100.0 * pow(10.0, (rand() % 4) - 1));
    for (size_t i = 0; i < table.size(); ++i) table[i] = (float)(rand() % 100000) / 100000.0f;
    std::vector<uint16_t> hs((size_t)tiles * NCOMP * groups * 64 * 8), tb((size_t)NCOMP * 2 * NC * rowLen, 0);
    for (int t = 0; t < tiles; ++t)
        for (int k = 0; k < K; ++k)
            for (int ch = 0; ch < 64; ++ch) {
                uint16_t s[3]; split3(hist[((size_t)t * K + k) * 64 + ch], s);
                for (int q = 0; q < NCOMP; ++q) hs[((((size_t)t * NCOMP + q) * groups + k / 8) * 64 + ch) * 8 + k % 8] = s[q];
            }
    for (int c = 0; c < NC; ++c)
        for (int x = 0; x < rowLen; ++x) {
            const int L = xmax - x;                        // Rrev_c[x] = R_c[xmax - x], zero outside the table
            uint16_t s[3] = {0, 0, 0};
            if (L >= 0 && L <= maxLag) split3(table[(size_t)L * NC + c], s);
            for (int q = 0; q < NCOMP; ++q)
                for (int p = 0; p < 2; ++p)
                    if (x - p >= 0) tb[((size_t)(q * 2 + p) * NC + c) * rowLen + (x - p)] = s[q];   // copy p: shifted by p
        }
    uint16_t *dh, *dt; float* dout;
    CK(hipMalloc(&dh, hs.size() * 2)); CK(hipMalloc(&dt, tb.size() * 2)); CK(hipMalloc(&dout, (size_t)tiles * 4096 * 4));
    CK(hipMemcpy(dh, hs.data(), hs.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dt, tb.data(), tb.size() * 2, hipMemcpyHostToDevice));
    const size_t shmem = tb.size() * 2;
    CK(hipFuncSetAttribute((const void*)conv_bf16<6>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    CK(hipFuncSetAttribute((const void*)conv_bf16<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    printf("LDS table image: %zu bytes (3 comps x 2 parity copies x 7 curves x %d lags)\n", shmem, rowLen);
    // reference sums: double, and what an FP32 multiply-add chain gives (the FP32-MFMA form's accuracy class)
    for (int variant = 0; variant < 2; ++variant) {
        if (variant == 0) hipLaunchKernelGGL(conv_bf16<6>, dim3(tilesCheck), dim3(64), shmem, 0, dh, dt, dout, K, groups, rowLen, xmax, j0, 1);
        else hipLaunchKernelGGL(conv_bf16<3>, dim3(tilesCheck), dim3(64), shmem, 0, dh, dt, dout, K, groups, rowLen, xmax, j0, 1);
        CK(hipGetLastError()); CK(hipDeviceSynchronize());
        std::vector<float> out((size_t)tilesCheck * 4096);
        CK(hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost));
        double worst = 0, worstF = 0, sumsq = 0, sumsqF = 0; size_t cnt = 0;
        for (int t = 0; t < tilesCheck; ++t)
            for (int ch = 0; ch < 64; ++ch)
                for (int d = 0; d < T; ++d)
                    for (int c = 0; c < NC; ++c) {
                        double s = 0; float sf = 0.0f;
                        for (int k = 0; k < K; ++k) {
                            const float a = hist[((size_t)t * K + k) * 64 + ch], b = table[(size_t)(j0 + d - k) * NC + c];
                            s += (double)a * b;
                            sf = fmaf(a, b, sf);
                        }
                        const double got = out[(size_t)t * 4096 + ch * 64 + d * NC + c];
                        const double e = fabs(got - s) / fabs(s), ef = fabs((double)sf - s) / fabs(s);
                        worst = fmax(worst, e); worstF = fmax(worstF, ef); sumsq += e * e; sumsqF += ef * ef; ++cnt;
                    }
        printf("%d products: relative error against the double sum: worst %.3g, rms %.3g   (an FP32 fmaf chain: worst %.3g, rms %.3g)\n",
               variant ? 3 : 6, worst, sqrt(sumsq / cnt), worstF, sqrt(sumsqF / cnt));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        const int reps = 40;
        float ms = 0;
        for (int rr = 0; rr < 2; ++rr) {
            CK(hipEventRecord(e0));
            if (variant == 0) hipLaunchKernelGGL(conv_bf16<6>, dim3(tiles), dim3(64), shmem, 0, dh, dt, dout, K, groups, rowLen, xmax, j0, reps);
            else hipLaunchKernelGGL(conv_bf16<3>, dim3(tiles), dim3(64), shmem, 0, dh, dt, dout, K, groups, rowLen, xmax, j0, reps);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
        }
        printf("  1024 waves, %d source days x %d reps: %.3f ms; %.1f ns per source day per wave (the FP32-MFMA form: 69 in its prototype, "
               "~60 in the kernel); %.0f cycles per 16 source days at 2.4 GHz (pure MFMA time %d)\n",
               K, reps, ms, ms * 1e6 / ((double)reps * K), ms * 1e-3 * 2.4e9 / ((double)reps * K / 16), (variant ? 3 : 6) * 4 * 32);
    }
    return 0;
}
