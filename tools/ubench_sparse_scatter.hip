// Microbenchmark: stream keys in at full rate and write a SPARSE subset (the survivors of a filter) out -
//   mode 0: count only;  mode 1: survivors compacted per wave, 512 B contiguous per store (what fj_bloom_filter_kernel does);
//   mode 2: every survivor stored on its own (8 B) into one of 64 wave-private output streams chosen by a radix digit -
//           a sub-partitioning filter stage WITHOUT LDS write-combining: does L2 / Infinity Cache merge the partial lines?
// 256 workgroups x 1024 threads (16 waves per CU, as the filter kernel), 4 x 16 B per thread in flight.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint64_t u64; typedef uint32_t u32;
struct __attribute__((aligned(16))) u64x2 { u64 x, y; };
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__device__ __forceinline__ u32 mix(u64 k) { u32 x = (u32)k ^ (u32)(k >> 32); x *= 0x9E3779B1u; x ^= x >> 15; x *= 0x85EBCA77u; x ^= x >> 13; return x; }

template <int MODE>
__global__ __launch_bounds__(1024) void k(const u64x2* __restrict__ in, u64* __restrict__ out, u64 n16, u32 keep_per_1024, u32 stream_keys_log,
                                          unsigned long long* __restrict__ total) {
    __shared__ u32 cnt[16][64];
    __shared__ u64 stg[16][128];
    const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const u64 gwave = (u64)blockIdx.x * 16 + wave;
    cnt[wave][lane] = 0;
    u32 ns = 0; u64 dense_pos = 0, found = 0;
    const u64 per_wg = n16 / gridDim.x, base = (u64)blockIdx.x * per_wg;
    const u64 smask = (1ull << stream_keys_log) - 1;
    for (u64 i = tid; i < per_wg; i += 4096) {
        u64x2 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const u64 j = i + (u64)u * 1024; v[u] = in[base + (j < per_wg ? j : per_wg - 1)]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool live = i + (u64)u * 1024 < per_wg;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const u64 key = h ? v[u].y : v[u].x;
                const u32 x = mix(key);
                const bool keep = live && (x & 1023u) < keep_per_1024;
                if (MODE == 0) { found += keep; continue; }
                if (MODE == 2) {
                    if (keep) {
                        const u32 d = (x >> 10) & 63u;
                        const u32 pos = atomicAdd(&cnt[wave][d], 1u);
                        out[((gwave * 64 + d) << stream_keys_log) + (pos & smask)] = key;
                    }
                    continue;
                }
                const u64 m = __ballot(keep);
                if (m) {
                    if (keep) stg[wave][ns + __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u))] = key;
                    ns += (u32)__popcll(m);
                    if (ns >= 64) {
                        out[(gwave << (stream_keys_log + 6)) + ((dense_pos + lane) & ((smask << 6) | 63))] = stg[wave][lane];
                        dense_pos += 64; ns -= 64;
                        if (lane < ns) { const u64 t = stg[wave][64 + lane]; stg[wave][lane] = t; }
                    }
                }
            }
        }
    }
    if (MODE == 0 && found) atomicAdd(total, found);
}

int main() {
    const u64 bytes = 8ull << 30, n16 = bytes / 16;
    const u32 stream_keys_log = 10;                         // 8 KiB per (wave, digit) stream: 4096 waves x 64 x 8 KiB = 2 GiB
    u64x2* in; u64* out; unsigned long long* tot;
    CK(hipMalloc(&in, bytes)); CK(hipMalloc(&out, (4096ull * 64) << (stream_keys_log + 3))); CK(hipMalloc(&tot, 8));
    // keys: a counter hash so that `keep` and the digits look random
    { u64* h = (u64*)malloc(1 << 26); for (u64 i = 0; i < (1 << 23); ++i) h[i] = (i + 1) * 0x9E3779B97F4A7C15ull; for (u64 o = 0; o < bytes; o += 1 << 26) CK(hipMemcpy((char*)in + o, h, 1 << 26, hipMemcpyHostToDevice)); free(h); }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("mode,keep_percent,ms,read_GBps\n");
    for (u32 keep : {51u, 133u, 307u, 1024u}) {             // 5 %, 13 %, 30 %, 100 % of the keys survive
        for (int mode = 0; mode < 3; ++mode) {
            float best = 1e9;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipEventRecord(e0));
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(1024), 0, 0, in, out, n16, keep, stream_keys_log, tot);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(1024), 0, 0, in, out, n16, keep, stream_keys_log, tot);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(1024), 0, 0, in, out, n16, keep, stream_keys_log, tot);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
            }
            printf("%d,%.1f,%.3f,%.0f\n", mode, keep / 10.24, best, bytes / best / 1e6);
        }
    }
    return 0;
}
