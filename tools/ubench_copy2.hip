// Round 6: the plain-copy ceiling of these boxes, measured wider than round 1's 4 x 4 sweep.  The guide quotes
// 6.29 TB/s for a float4 copy; profiles/r01_ubench_copy.csv's best point was 5.62 TB/s at its SMALLEST grid.  A radix pass streams
// 8 B in and 8 B out per key: a copy is its ceiling.  Variants:
//   stride  : grid-stride loop, 256-thread workgroups, U 16-byte loads in flight per lane, grids 256 .. 16384
//   slice   : persistent workgroups (256 x 1024 threads: the pass kernel's shape, and 512 x 512, 1024 x 256), each owning one
//             contiguous slice of the buffer, U loads in flight
//   nt      : the same with nontemporal loads and stores
//   memcpy  : hipMemcpyDtoDAsync
// sizes 8 / 16 / 32 GiB per buffer.  Output: CSV, best of 3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint64_t u64;
typedef __attribute__((ext_vector_type(4))) unsigned int v4;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int U, bool NT>
__global__ void copy_stride(const v4* __restrict__ in, v4* __restrict__ out, u64 n16) {
    const u64 stride = (u64)gridDim.x * blockDim.x * U;
    for (u64 base = (u64)blockIdx.x * blockDim.x * U + threadIdx.x; base < n16; base += stride) {
        v4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) if (base + (u64)u * blockDim.x < n16) v[u] = NT ? __builtin_nontemporal_load(in + base + (u64)u * blockDim.x) : in[base + (u64)u * blockDim.x];
#pragma unroll
        for (int u = 0; u < U; ++u) if (base + (u64)u * blockDim.x < n16) { if (NT) __builtin_nontemporal_store(v[u], out + base + (u64)u * blockDim.x); else out[base + (u64)u * blockDim.x] = v[u]; }
    }
}
// every workgroup copies ONE contiguous slice, U loads in flight per lane
template <int U, bool NT>
__global__ void copy_slice(const v4* __restrict__ in, v4* __restrict__ out, u64 n16) {
    const u64 per = (n16 + gridDim.x - 1) / gridDim.x, lo = per * blockIdx.x, hi = lo + per < n16 ? lo + per : n16;
    for (u64 base = lo + threadIdx.x; base < hi; base += (u64)blockDim.x * U) {
        v4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) if (base + (u64)u * blockDim.x < hi) v[u] = NT ? __builtin_nontemporal_load(in + base + (u64)u * blockDim.x) : in[base + (u64)u * blockDim.x];
#pragma unroll
        for (int u = 0; u < U; ++u) if (base + (u64)u * blockDim.x < hi) { if (NT) __builtin_nontemporal_store(v[u], out + base + (u64)u * blockDim.x); else out[base + (u64)u * blockDim.x] = v[u]; }
    }
}
template <int U, bool NT> static void launch(int kind, int grid, int block, const v4* in, v4* out, u64 n16) {
    if (kind == 0) hipLaunchKernelGGL((copy_stride<U, NT>), dim3(grid), dim3(block), 0, 0, in, out, n16);
    else hipLaunchKernelGGL((copy_slice<U, NT>), dim3(grid), dim3(block), 0, 0, in, out, n16);
}
static void launch_u(int U, bool nt, int kind, int grid, int block, const v4* in, v4* out, u64 n16) {
#define L(u) if (U == u) { if (nt) launch<u, true>(kind, grid, block, in, out, n16); else launch<u, false>(kind, grid, block, in, out, n16); }
    L(1) L(2) L(4) L(8)
#undef L
}
int main() {
    const u64 maxb = 32ull << 30;
    v4 *in, *out; CK(hipMalloc(&in, maxb)); CK(hipMalloc(&out, maxb)); CK(hipMemset(in, 1, maxb)); CK(hipMemset(out, 0, maxb));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("kind,nt,unroll,grid,block,GiB,ms,GBps\n");
    auto timeit = [&](auto&& fn, float* best) -> int {
        *best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0)); fn(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < *best) *best = ms;
        }
        return 0;
    };
    for (u64 gib : {8ull, 16ull, 32ull}) {
        const u64 bytes = gib << 30, n16 = bytes / 16;
        float best;
        if (timeit([&] { (void)hipMemcpyDtoDAsync((hipDeviceptr_t)out, (hipDeviceptr_t)in, bytes, 0); }, &best)) return 1;
        printf("memcpy,0,0,0,0,%llu,%.3f,%.0f\n", (unsigned long long)gib, best, 2.0 * bytes / best / 1e6);
        for (int nt = 0; nt < 2; ++nt) {
            for (int U : {1, 2, 4, 8}) {
                if (gib != 16 && (U == 1 || nt)) continue;                     // the full sweep at 16 GiB, the main points at 8 and 32
                for (int g : {256, 512, 1024, 2048, 4096, 8192, 16384}) {
                    if (timeit([&] { launch_u(U, nt, 0, g, 256, in, out, n16); }, &best)) return 1;
                    printf("stride,%d,%d,%d,256,%llu,%.3f,%.0f\n", nt, U, g, (unsigned long long)gib, best, 2.0 * bytes / best / 1e6);
                }
                for (int shape = 0; shape < 4; ++shape) {
                    const int g = shape == 0 ? 256 : shape == 1 ? 512 : shape == 2 ? 1024 : 2048, b = shape == 0 ? 1024 : shape == 1 ? 512 : 256;
                    if (timeit([&] { launch_u(U, nt, 1, g, b, in, out, n16); }, &best)) return 1;
                    printf("slice,%d,%d,%d,%d,%llu,%.3f,%.0f\n", nt, U, g, b, (unsigned long long)gib, best, 2.0 * bytes / best / 1e6);
                }
            }
        }
    }
    return 0;
}
