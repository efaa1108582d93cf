// Microbenchmark: what does a wave-wide LDS atomic cost on MI355X?  One workgroup per CU (512 threads), every lane hits a
// pseudo-random slot of an 8192-entry table (a cuckoo insert's access pattern); reports shader-clock ticks (s_memtime) per wave-instruction; every
// iteration depends on the previous one (the returned value is consumed), so this is LATENCY with 8 waves sharing the LDS.
//   xchg64  : ds_wrxchg_rtn_b64 (the cuckoo insert's eviction step)        or32 : ds_or_rtn_b32 on a 1-KiB bitmap + ds_write_b64
//   cas64   : ds_cmpst_rtn_b64                                              add32: ds_add_rtn_u32 (the tagged table's slot claim)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint64_t u64; typedef uint32_t u32;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr u32 S = 8192, NT = 512, REPS = 64;

template <int MODE>
__global__ __launch_bounds__(NT) void k(u64* out, unsigned long long* cyc) {
    __shared__ u64 tab[S];
    __shared__ u32 bits[S / 32 + 2048];
    const u32 tid = threadIdx.x;
    for (u32 i = tid; i < S; i += NT) tab[i] = ~0ull;
    for (u32 i = tid; i < S / 32 + 2048; i += NT) bits[i] = 0;
    __syncthreads();
    u32 x = tid * 2654435761u + blockIdx.x * 40503u + 12345u;
    u64 acc = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (u32 r = 0; r < REPS; ++r) {
        x = x * 1664525u + 1013904223u;
        const u32 slot = (x >> 10) & (S - 1);
        const u64 key = ((u64)x << 32) | r | 1;
        if (MODE == 0) acc += atomicExch((unsigned long long*)&tab[slot], (unsigned long long)key);
        else if (MODE == 1) acc += atomicCAS((unsigned long long*)&tab[slot], ~0ull, (unsigned long long)key);
        else if (MODE == 2) { const u32 old = atomicOr(&bits[slot >> 5], 1u << (slot & 31)); if (!((old >> (slot & 31)) & 1u)) tab[slot] = key; acc += old; }
        else if (MODE == 3) acc += atomicAdd(&bits[slot >> 2], 1u << ((slot & 3) * 8));
        else if (MODE == 4) { acc += tab[slot]; }                                   // plain 64-bit read, for scale
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * NT + tid] = acc;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    u64* out; unsigned long long* cyc;
    const int G = 256;
    CK(hipMalloc(&out, G * NT * 8)); CK(hipMalloc(&cyc, G * 8));
    unsigned long long h[G];
    const char* names[] = {"xchg64", "cas64", "or32+write64", "add32", "read64"};
    printf("op,ticks_per_dependent_iteration_of_one_wave\n");
    for (int m = 0; m < 5; ++m) {
        for (int rep = 0; rep < 2; ++rep) {
            switch (m) {
                case 0: hipLaunchKernelGGL(k<0>, dim3(G), dim3(NT), 0, 0, out, cyc); break;
                case 1: hipLaunchKernelGGL(k<1>, dim3(G), dim3(NT), 0, 0, out, cyc); break;
                case 2: hipLaunchKernelGGL(k<2>, dim3(G), dim3(NT), 0, 0, out, cyc); break;
                case 3: hipLaunchKernelGGL(k<3>, dim3(G), dim3(NT), 0, 0, out, cyc); break;
                default: hipLaunchKernelGGL(k<4>, dim3(G), dim3(NT), 0, 0, out, cyc); break;
            }
            CK(hipDeviceSynchronize());
        }
        CK(hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost));
        double s = 0; for (int i = 0; i < G; ++i) s += (double)h[i];
        printf("%s,%.1f\n", names[m], s / G / REPS);                // s_memtime counts shader clocks
    }
    return 0;
}
