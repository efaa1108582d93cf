// Microbenchmark: how many returning atomicAdds per second does ONE global cursor sustain on MI355X when every wave of
// a resident grid bumps it (output reservation of a single-pass materialising join)?  And with the cursor sharded?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void bump(unsigned long long* cur, unsigned nshards, unsigned iters, unsigned long long* sink) {
    const unsigned lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    unsigned long long acc = 0;
    for (unsigned i = 0; i < iters; ++i) {
        unsigned long long b = 0;
        if (lane == 0) b = atomicAdd(&cur[(wave % nshards) * 16], 37ull);       // shards 128 B apart
        b = __shfl(b, 0, 64);
        acc += b;
    }
    if (lane == 0 && acc == 1) sink[0] = acc;
}
int main() {
    unsigned long long *cur, *sink;
    CK(hipMalloc(&cur, 1 << 20)); CK(hipMalloc(&sink, 8));
    CK(hipMemset(cur, 0, 1 << 20));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("waves,shards,iters,ms,Matomics_per_s\n");
    for (unsigned grid : {256u, 512u}) for (unsigned nt : {256u, 1024u}) for (unsigned shards : {1u, 8u, 64u, 4096u}) {
        const unsigned iters = 2000;
        float best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(bump, dim3(grid), dim3(nt), 0, 0, cur, shards, iters, sink);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        const double n = (double)grid * (nt / 64) * iters;
        printf("%u,%u,%u,%.3f,%.1f\n", grid * (nt / 64), shards, iters, best, n / best / 1e3);
    }
    return 0;
}
