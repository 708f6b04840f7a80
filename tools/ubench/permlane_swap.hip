// Prints what v_permlane16_swap / v_permlane32_swap do to two registers (a = lane, b = 100 + lane), so that the
// wave mapping's row transposition (mcmc_wave_kernel.hip.h) rests on observed, not assumed, semantics.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
    const unsigned lane = threadIdx.x;
    unsigned a = lane, b = 100 + lane;
    auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    out[lane] = r[0]; out[64 + lane] = r[1];
    auto s = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    out[128 + lane] = s[0]; out[192 + lane] = s[1];
}
int main() {
    unsigned* d; unsigned h[256];
    if (hipMalloc(&d, sizeof h) != hipSuccess) return 1;
    k<<<1, 64>>>(d);
    if (hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    const char* names[4] = {"p16 vdst", "p16 src ", "p32 vdst", "p32 src "};
    for (int r = 0; r < 4; ++r) {
        printf("%s:", names[r]);
        for (int i = 0; i < 64; ++i) printf(" %u", h[r * 64 + i]);
        printf("\n");
    }
    (void)hipFree(d);
    return 0;
}
