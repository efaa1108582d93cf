// Which XCD does workgroup b of a large grid run on?  Every workgroup records HW_REG_XCC_ID; the host reports how often
// xcc == (b + c) % 8 holds for the best c, for grids that fit at once and for grids many times the chip (workgroups that start
// when earlier ones retire), with uniform and with uneven work per workgroup.   hipcc --offload-arch=gfx950 -O2 -o tools/bin/ubench_xcc_map tools/ubench_xcc_map.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256) void k(unsigned* out, unsigned spin, unsigned uneven, unsigned long long* sink) {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    if (threadIdx.x == 0) out[blockIdx.x] = x & 15u;
    unsigned n = spin * (uneven ? 1u + (blockIdx.x * 2654435761u >> 29) : 1u);
    unsigned long long a = threadIdx.x;
    for (unsigned i = 0; i < n; ++i) a = a * 6364136223846793005ull + 1442695040888963407ull;
    if (a == 42) *sink = a;
}
int main() {
    unsigned* d; unsigned long long* s;
    const unsigned grids[] = {256, 2048, 8192, 40000, 140000};
    hipMalloc(&d, 140000 * 4); hipMalloc(&s, 8);
    for (unsigned uneven = 0; uneven < 2; ++uneven)
        for (unsigned g : grids) {
            hipLaunchKernelGGL(k, dim3(g), dim3(256), 0, 0, d, 20000u, uneven, s);
            hipDeviceSynchronize();
            std::vector<unsigned> h(g);
            hipMemcpy(h.data(), d, g * 4, hipMemcpyDeviceToHost);
            unsigned best = 0, bc = 0;
            for (unsigned c = 0; c < 8; ++c) { unsigned ok = 0; for (unsigned b = 0; b < g; ++b) ok += h[b] == (b + c) % 8; if (ok > best) { best = ok; bc = c; } }
            unsigned cnt[16] = {0}; for (unsigned b = 0; b < g; ++b) cnt[h[b]]++;
            printf("grid %6u %s work: xcc == (b + %u) %% 8 for %u of %u workgroups (%.1f %%); per XCD:", g, uneven ? "uneven" : "equal ", bc, best, g, 100.0 * best / g);
            for (unsigned x = 0; x < 8; ++x) printf(" %u", cnt[x]);
            printf("\n");
        }
    return 0;
}
