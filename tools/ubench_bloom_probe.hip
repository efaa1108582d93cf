// Microbenchmark: how fast can MI355X test streamed keys against a membership filter, by where the filter lives?
//   mode 0: stream only (16 B per lane, hash, count)                          -> the read ceiling of the loop
//   mode 1: + one random 8-B load per key from a table of T bytes PER XCD (workgroup's XCD = blockIdx.x % 8)
//   mode 2: + one random 4-B load per key, same placement
//   mode 3: + one random 8-B load per key from ONE table of T bytes shared by all XCDs
//   mode 4: + one ds_read_b32 per key from a filter held in LDS (T <= 128 KiB, every workgroup loads its copy first)
// Decides the placement of the bloom precheck of the partitioned join: per-bucket filter in L2 vs LDS.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint64_t u64; typedef uint32_t u32;
struct __attribute__((aligned(16))) u64x2 { u64 x, y; };
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ u32 fmix32(u32 x) { x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16; return x; }
__device__ __forceinline__ u32 hw(u64 k) { return fmix32((u32)k * 0x9E3779B1u ^ (u32)(k >> 32) * 0x85EBCA77u); }

__global__ void fill(u64* k, u64 n) {
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) k[i] = (i + 1) * 0x9E3779B97F4A7C15ull;
}

template <int MODE, int NT>
__global__ __launch_bounds__(NT) void probe(const u64x2* __restrict__ keys, u64 n16, const u64* __restrict__ table, u32 tmask8,
                                            unsigned long long* __restrict__ out) {
    extern __shared__ u32 lds[];
    const u32 tid = threadIdx.x;
    const u64* tb = table;
    if (MODE == 1 || MODE == 2) tb = table + (u64)(blockIdx.x & 7) * ((u64)tmask8 + 1);
    if (MODE == 4) {
        for (u32 i = tid; i < (tmask8 + 1) * 2; i += NT) lds[i] = ((const u32*)table)[i];
        __syncthreads();
    }
    u32 cnt = 0;
    const u64 per = (n16 + gridDim.x - 1) / gridDim.x;
    const u64 lo = per * blockIdx.x, hi = lo + per < n16 ? lo + per : n16;
    for (u64 i = lo + tid; i < hi; i += 4 * NT) {
        u64x2 q[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const u64 j = i + (u64)u * NT; q[u] = keys[j < hi ? j : hi - 1]; }
        u64 w[8];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const u32 h0 = hw(q[u].x), h1 = hw(q[u].y);
            if (MODE == 0) { w[2 * u] = h0; w[2 * u + 1] = h1; }
            else if (MODE == 2) { w[2 * u] = ((const u32*)tb)[h0 & (2 * tmask8 + 1)]; w[2 * u + 1] = ((const u32*)tb)[h1 & (2 * tmask8 + 1)]; }
            else if (MODE == 4) { w[2 * u] = lds[h0 & (2 * tmask8 + 1)]; w[2 * u + 1] = lds[h1 & (2 * tmask8 + 1)]; }
            else { w[2 * u] = tb[h0 & tmask8]; w[2 * u + 1] = tb[h1 & tmask8]; }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const u32 h0 = hw(q[u].x) >> 20, h1 = hw(q[u].y) >> 20;
            cnt += (u32)((w[2 * u] >> (h0 & 31)) & 1) + (u32)((w[2 * u + 1] >> (h1 & 31)) & 1);
        }
    }
    for (int d = 32; d >= 1; d >>= 1) cnt += __shfl_xor(cnt, d, 64);
    if ((tid & 63) == 0) atomicAdd(out, (unsigned long long)cnt);
}

template <int MODE>
int run(const u64x2* keys, u64 n16, const u64* table, u64 tbytes, unsigned long long* out, int grid) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const u32 tmask8 = (u32)(tbytes / 8 - 1);
    const u32 ldsb = MODE == 4 ? (u32)tbytes : 0;
    auto k = probe<MODE, 512>;
    if (ldsb) CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
    float best = 1e9;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k, dim3(grid), dim3(512), ldsb, 0, keys, n16, table, tmask8, out);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    printf("%d,%llu,%d,%.3f,%.1f,%.0f\n", MODE, (unsigned long long)tbytes, grid, best, 2.0 * n16 / best / 1e6, 16.0 * n16 / best / 1e6);
    fflush(stdout);
    return 0;
}

int main() {
    const u64 n = 1ull << 29;                               // 512M keys = 4 GiB
    u64* keys; u64* table; unsigned long long* out;
    CK(hipMalloc(&keys, n * 8)); CK(hipMalloc(&table, 8ull * (256ull << 20))); CK(hipMalloc(&out, 8));
    hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, keys, n);
    CK(hipMemset(table, 0x5A, 8ull * (256ull << 20)));
    CK(hipDeviceSynchronize());
    printf("mode,table_bytes,grid,ms,Gkeys_per_s,stream_GBps\n");
    for (int grid : {512, 1024}) {
        if (run<0>((const u64x2*)keys, n / 2, table, 8, out, grid)) return 1;
        for (u64 tb : {64ull << 10, 256ull << 10, 1ull << 20, 2ull << 20, 4ull << 20, 16ull << 20, 256ull << 20}) {
            if (run<1>((const u64x2*)keys, n / 2, table, tb, out, grid)) return 1;
            if (run<2>((const u64x2*)keys, n / 2, table, tb, out, grid)) return 1;
            if (run<3>((const u64x2*)keys, n / 2, table, tb, out, grid)) return 1;
        }
        for (u64 tb : {32ull << 10, 64ull << 10}) if (run<4>((const u64x2*)keys, n / 2, table, tb, out, grid)) return 1;
    }
    return 0;
}
