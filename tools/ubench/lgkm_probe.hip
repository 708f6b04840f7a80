// Probe: does `s_waitcnt lgkmcnt(0)` stall on an outstanding global_load on gfx950?
// Each wave issues one cold global_load, then times (s_memtime) three waits in turn.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(64) void probe(const double* __restrict__ src, unsigned long long* out, double* sink, size_t stride, int mode) {
    extern __shared__ double lds[];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    const double* p = src + (size_t)blockIdx.x * stride + threadIdx.x + (size_t)mode * 4096;
    double v, l = 0;
    unsigned long long t0, t1, t2, t3;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0));
    asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    if (mode & 1) {
        // an LDS read issued after the global load, then wait for the LDS read only
        asm volatile("ds_read_b64 %0, %1" : "=v"(l) : "v"((unsigned)(threadIdx.x * 8)) : "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1));
    asm volatile("s_nop 0\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t2));
    asm volatile("s_waitcnt vmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t3) :: "memory");
    asm volatile("" : "+v"(v), "+v"(l));
    if (threadIdx.x == 0) {
        out[blockIdx.x * 3 + 0] = t1 - t0;
        out[blockIdx.x * 3 + 1] = t2 - t1;
        out[blockIdx.x * 3 + 2] = t3 - t2;
    }
    sink[blockIdx.x * 64 + threadIdx.x] = v + l;
}

int main() {
    const int nb = 256;
    const size_t stride = 1 << 20;   // 8 MB apart: cold lines
    double* src; unsigned long long* out; double* sink;
    CK(hipMalloc(&src, nb * stride * sizeof(double)));
    CK(hipMemset(src, 0, nb * stride * sizeof(double)));
    CK(hipMalloc(&out, nb * 3 * sizeof(unsigned long long)));
    CK(hipMalloc(&sink, nb * 64 * sizeof(double)));
    for (int mode = 0; mode < 4; ++mode) {
        hipLaunchKernelGGL(probe, dim3(nb), dim3(64), 1024, 0, src, out, sink, stride, mode);
        CK(hipDeviceSynchronize());
        std::vector<unsigned long long> h(nb * 3);
        CK(hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost));
        double a = 0, b = 0, c = 0;
        for (int i = 0; i < nb; ++i) { a += h[3 * i]; b += h[3 * i + 1]; c += h[3 * i + 2]; }
        printf("mode %d (%s): issue+lgkmcnt(0) wait %.0f cyc | idle %.0f cyc | vmcnt(0) wait %.0f cyc\n", mode,
               (mode & 1) ? "global_load then ds_read" : "global_load only", a / nb, b / nb, c / nb);
    }
    return 0;
}
