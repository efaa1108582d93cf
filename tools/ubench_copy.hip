// What is the plain copy ceiling on this box?  (guide: 6.29 TB/s for a float4 copy)  Variants: loads in flight per lane.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint64_t u64;
struct __attribute__((aligned(16))) u64x2 { u64 x, y; };
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <int U>
__global__ __launch_bounds__(256) void copy_kernel(const u64x2* __restrict__ in, u64x2* __restrict__ out, u64 n16) {
    const u64 stride = (u64)gridDim.x * blockDim.x * U;
    for (u64 base = (u64)blockIdx.x * blockDim.x * U + threadIdx.x; base < n16; base += stride) {
        u64x2 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) if (base + (u64)u * blockDim.x < n16) v[u] = in[base + (u64)u * blockDim.x];
#pragma unroll
        for (int u = 0; u < U; ++u) if (base + (u64)u * blockDim.x < n16) out[base + (u64)u * blockDim.x] = v[u];
    }
}
template <int U>
__global__ __launch_bounds__(256) void read_kernel(const u64x2* __restrict__ in, u64* __restrict__ out, u64 n16) {
    const u64 stride = (u64)gridDim.x * blockDim.x * U;
    u64 acc = 0;
    for (u64 base = (u64)blockIdx.x * blockDim.x * U + threadIdx.x; base < n16; base += stride) {
#pragma unroll
        for (int u = 0; u < U; ++u) if (base + (u64)u * blockDim.x < n16) { u64x2 v = in[base + (u64)u * blockDim.x]; acc += v.x ^ v.y; }
    }
    if (acc == 0x1234567) out[0] = acc;
}
int main() {
    const u64 bytes = 8ull << 30, n16 = bytes / 16;
    u64x2 *in, *out; CK(hipMalloc(&in, bytes)); CK(hipMalloc(&out, bytes)); CK(hipMemset(in, 1, bytes)); CK(hipMemset(out, 0, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("kind,unroll,grid,ms,GBps\n");
    for (int g : {1024, 2048, 4096, 16384}) {
        for (int U : {1, 2, 4, 8}) {
            for (int kind = 0; kind < 2; ++kind) {
                float best = 1e9;
                for (int rep = 0; rep < 3; ++rep) {
                    CK(hipEventRecord(e0));
                    if (kind == 0) { if (U==1) hipLaunchKernelGGL(copy_kernel<1>, dim3(g), dim3(256), 0, 0, in, out, n16); if (U==2) hipLaunchKernelGGL(copy_kernel<2>, dim3(g), dim3(256), 0, 0, in, out, n16); if (U==4) hipLaunchKernelGGL(copy_kernel<4>, dim3(g), dim3(256), 0, 0, in, out, n16); if (U==8) hipLaunchKernelGGL(copy_kernel<8>, dim3(g), dim3(256), 0, 0, in, out, n16); }
                    else { if (U==1) hipLaunchKernelGGL(read_kernel<1>, dim3(g), dim3(256), 0, 0, in, (u64*)out, n16); if (U==2) hipLaunchKernelGGL(read_kernel<2>, dim3(g), dim3(256), 0, 0, in, (u64*)out, n16); if (U==4) hipLaunchKernelGGL(read_kernel<4>, dim3(g), dim3(256), 0, 0, in, (u64*)out, n16); if (U==8) hipLaunchKernelGGL(read_kernel<8>, dim3(g), dim3(256), 0, 0, in, (u64*)out, n16); }
                    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
                }
                printf("%s,%d,%d,%.3f,%.0f\n", kind == 0 ? "copy" : "read", U, g, best, (kind == 0 ? 2.0 : 1.0) * bytes / best / 1e6);
            }
        }
    }
    return 0;
}
