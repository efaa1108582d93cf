// Does a non-temporal hint on the stores (and loads) of a streaming copy raise the copy ceiling on MI355X?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint64_t u64;
typedef u64 u64x2v __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <int U, int MODE>   // MODE 0 plain, 1 nt store, 2 nt load + nt store
__global__ __launch_bounds__(256) void copy_kernel(const u64x2v* __restrict__ in, u64x2v* __restrict__ out, u64 n16) {
    const u64 stride = (u64)gridDim.x * blockDim.x * U;
    for (u64 base = (u64)blockIdx.x * blockDim.x * U + threadIdx.x; base < n16; base += stride) {
        u64x2v v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { const u64 i = base + (u64)u * blockDim.x; if (i < n16) v[u] = MODE == 2 ? __builtin_nontemporal_load(in + i) : in[i]; }
#pragma unroll
        for (int u = 0; u < U; ++u) { const u64 i = base + (u64)u * blockDim.x; if (i < n16) { if (MODE >= 1) __builtin_nontemporal_store(v[u], out + i); else out[i] = v[u]; } }
    }
}
template <int U, int MODE> float run(const u64x2v* in, u64x2v* out, u64 n16, int g, hipEvent_t e0, hipEvent_t e1) {
    float best = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((copy_kernel<U, MODE>), dim3(g), dim3(256), 0, 0, in, out, n16);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    return best;
}
int main() {
    const u64 bytes = 8ull << 30, n16 = bytes / 16;
    u64x2v *in, *out; CK(hipMalloc(&in, bytes)); CK(hipMalloc(&out, bytes)); CK(hipMemset(in, 1, bytes)); CK(hipMemset(out, 0, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("mode,unroll,grid,ms,GBps\n");
    for (int g : {1024, 2048, 16384}) {
        float t;
        t = run<1, 0>(in, out, n16, g, e0, e1); printf("plain,1,%d,%.3f,%.0f\n", g, t, 2.0 * bytes / t / 1e6);
        t = run<1, 1>(in, out, n16, g, e0, e1); printf("nt_store,1,%d,%.3f,%.0f\n", g, t, 2.0 * bytes / t / 1e6);
        t = run<1, 2>(in, out, n16, g, e0, e1); printf("nt_both,1,%d,%.3f,%.0f\n", g, t, 2.0 * bytes / t / 1e6);
        t = run<4, 0>(in, out, n16, g, e0, e1); printf("plain,4,%d,%.3f,%.0f\n", g, t, 2.0 * bytes / t / 1e6);
        t = run<4, 1>(in, out, n16, g, e0, e1); printf("nt_store,4,%d,%.3f,%.0f\n", g, t, 2.0 * bytes / t / 1e6);
        t = run<4, 2>(in, out, n16, g, e0, e1); printf("nt_both,4,%d,%.3f,%.0f\n", g, t, 2.0 * bytes / t / 1e6);
    }
    return 0;
}
