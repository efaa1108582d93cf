// Microbenchmark: what does a wave-wide LDS atomic cost on MI355X?  One workgroup per CU (512 or 1024 threads); every lane
// hits a pseudo-random word (a cuckoo insert's / a radix rank's access pattern).  Two measurements per operation, in shader
// clocks (s_memtime) per wave-instruction:
//   latency    - every iteration consumes the previous result (one operation in flight per wave);
//   throughput - 8 independent operations per iteration, results only summed at the end: the CU's LDS pipe is the limit,
//                reported as clocks of CU time per wave-instruction (all waves of the workgroup issuing).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint64_t u64; typedef uint32_t u32;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr u32 S = 8192, REPS = 64;

template <int MODE, int NT, bool DEP>
__global__ __launch_bounds__(NT) void k(u64* out, unsigned long long* cyc, u32 range_mask) {
    __shared__ u64 tab[S];
    __shared__ u32 words[S];
    const u32 tid = threadIdx.x;
    for (u32 i = tid; i < S; i += NT) { tab[i] = ~0ull; words[i] = 0; }
    __syncthreads();
    u32 x = tid * 2654435761u + blockIdx.x * 40503u + 12345u;
    u64 acc = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (u32 r = 0; r < REPS; ++r) {
        u64 part[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            x = x * 1664525u + 1013904223u + (DEP ? (u32)acc : 0u);
            const u32 slot = (x >> 10) & range_mask;
            const u64 key = ((u64)x << 32) | r | 1;
            if (MODE == 0) part[u] = atomicExch((unsigned long long*)&tab[slot], (unsigned long long)key);
            else if (MODE == 1) part[u] = atomicCAS((unsigned long long*)&tab[slot], ~0ull, (unsigned long long)key);
            else if (MODE == 2) part[u] = atomicOr(&words[slot >> 5], 1u << (slot & 31));
            else if (MODE == 3) part[u] = atomicAdd(&words[slot], 1u);
            else if (MODE == 4) { atomicAdd(&words[slot], 1u); part[u] = 0; }          // no return value
            else if (MODE == 5) part[u] = tab[slot];                                   // plain 64-bit read, for scale
            else { tab[slot] = key; part[u] = 0; }                                     // plain 64-bit write
            if (DEP) acc += part[u];
        }
        if (!DEP) {
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += part[u];
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * NT + tid] = acc;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE, int NT>
static int run(const char* name, u32 mask, u64* out, unsigned long long* cyc) {
    const int G = 256;
    unsigned long long h[G];
    double res[2];
    for (int dep = 0; dep < 2; ++dep) {
        for (int rep = 0; rep < 2; ++rep) {
            if (dep) hipLaunchKernelGGL((k<MODE, NT, true>), dim3(G), dim3(NT), 0, 0, out, cyc, mask);
            else hipLaunchKernelGGL((k<MODE, NT, false>), dim3(G), dim3(NT), 0, 0, out, cyc, mask);
            CK(hipDeviceSynchronize());
        }
        CK(hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost));
        double s = 0; for (int i = 0; i < G; ++i) s += (double)h[i];
        res[dep] = s / G;
    }
    // latency: clocks per dependent operation of one wave; throughput: CU clocks per wave-instruction with NT/64 waves issuing
    printf("%s,%d,%u,%.1f,%.1f\n", name, NT, mask + 1, res[1] / (REPS * 8.0), res[0] / (REPS * 8.0 * (NT / 64)));
    return 0;
}

int main() {
    u64* out; unsigned long long* cyc;
    CK(hipMalloc(&out, 256 * 1024 * 8)); CK(hipMalloc(&cyc, 256 * 8));
    printf("op,threads,distinct_words,latency_clocks_per_dependent_op,cu_clocks_per_wave_instruction\n");
    run<0, 512>("xchg_rtn_b64", S - 1, out, cyc);   run<1, 512>("cmpst_rtn_b64", S - 1, out, cyc);
    run<2, 512>("or_rtn_b32 (bitmap of 8192 bits)", S - 1, out, cyc);
    run<3, 512>("add_rtn_u32", S - 1, out, cyc);    run<3, 512>("add_rtn_u32", 255, out, cyc);
    run<4, 512>("add_u32 (no return)", 255, out, cyc);
    run<5, 512>("read_b64", S - 1, out, cyc);        run<6, 512>("write_b64", S - 1, out, cyc);
    run<0, 1024>("xchg_rtn_b64", S - 1, out, cyc);  run<3, 1024>("add_rtn_u32", 255, out, cyc);
    return 0;
}
