// Does the 256 MiB Infinity Cache keep freshly WRITTEN data for a following reader?  (Design question: could the second
// partition pass read the first pass's output from the cache if the probe side were processed in batches?)
// write X MB (streaming 16-B stores), optionally stream Y MB of unrelated reads+writes, then read the X MB back; report the
// read-back bandwidth.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint64_t u64;
struct __attribute__((aligned(16))) u64x2 { u64 x, y; };
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ __launch_bounds__(256) void wr(u64x2* out, u64 n16, u64 v) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (u64)gridDim.x * blockDim.x) { u64x2 q; q.x = v + i; q.y = i; out[i] = q; }
}
__global__ __launch_bounds__(256) void rd(const u64x2* in, u64 n16, u64* sink) {
    u64 acc = 0;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (u64)gridDim.x * blockDim.x) { u64x2 q = in[i]; acc += q.x ^ q.y; }
    if (acc == 0x123456789abcull) *sink = acc;
}
__global__ __launch_bounds__(256) void cp(const u64x2* in, u64x2* out, u64 n16) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (u64)gridDim.x * blockDim.x) out[i] = in[i];
}
int main() {
    const u64 big = 4ull << 30;
    u64x2 *buf, *a, *b; u64* sink;
    CK(hipMalloc(&buf, 1ull << 30)); CK(hipMalloc(&a, big)); CK(hipMalloc(&b, big)); CK(hipMalloc(&sink, 8));
    CK(hipMemset(a, 1, big)); CK(hipMemset(b, 0, big));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("written_MB,between_MB,readback_ms,readback_GBps\n");
    for (u64 mb : {16ull, 32ull, 64ull, 96ull, 128ull, 192ull, 256ull, 512ull, 1024ull}) {
        for (u64 between : {0ull, 64ull, 128ull, 256ull}) {
            float best = 1e9;
            for (int rep = 0; rep < 5; ++rep) {
                const u64 n16 = (mb << 20) / 16;
                hipLaunchKernelGGL(cp, dim3(2048), dim3(256), 0, 0, a, b, big / 16);          // flush the cache with 8 GB of traffic
                hipLaunchKernelGGL(wr, dim3(2048), dim3(256), 0, 0, buf, n16, (u64)rep);
                if (between) hipLaunchKernelGGL(cp, dim3(2048), dim3(256), 0, 0, a, b, (between << 20) / 32);   // `between` MB of reads+writes in total
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(rd, dim3(2048), dim3(256), 0, 0, buf, n16, sink);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
            }
            printf("%llu,%llu,%.4f,%.0f\n", (unsigned long long)mb, (unsigned long long)between, best, (double)(mb << 20) / best / 1e6);
        }
    }
    return 0;
}
