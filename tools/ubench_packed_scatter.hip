// Microbenchmark for an open lead: would a radix pass that stores keys WITHOUT the digits their
// bucket implies - 7 or 6 bytes per key, i.e. 112-B or 96-B lines in 1792-B / 1536-B chunks - write as fast per byte as the
// pass does today with 128-B lines in 2-KiB chunks?  Packed lines straddle 128-B cache lines, so neighbouring lines of a chunk,
// written a few tiles apart, meet in the caches as partial lines.
//
// The write pattern of fj_partition_kernel is emulated without its sort: one 1024-thread workgroup per CU streams 8 B per key
// (flat input, 16 B per lane and load) and, per 8192-key tile, writes 512 lines of 16 keys; every line goes to a pseudo-random
// bucket of F, behind that bucket's fill position in its open chunk; a full chunk is replaced by the next id of the
// workgroup's dense region (the allocation order of the flat pass).  Line payloads are whatever was loaded.
//   usage: ./ubench_packed_scatter           -> csv: keys per line, line_bytes, F, ms, GB/s (bytes read + bytes written)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint64_t u64; typedef uint32_t u32;
struct __attribute__((aligned(16))) u64x2 { u64 x, y; };
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr u32 NT = 1024, T = 8192;                      // threads, keys per tile

// KPL = keys per line (16 or 32), LB = bytes per line.  KPL 16: LB / 16 lanes of every 8-lane group store 16 B each, lines
// contiguous (128 / 112 / 96 B).  KPL 32: 16-lane groups; LB 256 = contiguous 8-byte keys, LB 192 = a 6-byte key as a 4-byte and
// a 2-byte plane: the chunk is a 1-KiB plane of eight 128-B line pieces followed by a 512-B plane of eight 64-B pieces - every
// piece aligned to its own size, nothing straddles a cache line.
template <u32 LB, u32 KPL>
__global__ __launch_bounds__(NT) void packed_scatter(const u64x2* __restrict__ in, unsigned char* __restrict__ out, u64 nkeys, u32 F,
                                                     u64 region_bytes) {
    __shared__ u32 fill[512];          // lines written into the bucket's open chunk
    __shared__ u32 chunk[512];         // the bucket's open chunk (index inside the workgroup's region)
    __shared__ u32 prev[512];          // ... at the start of the tile
    __shared__ u32 next_chunk;
    constexpr u32 LINES = T / KPL, LPC = 256 / KPL, GL = KPL / 2;      // lines per tile, lines per chunk, lanes per line
    __shared__ u32 line_dst[LINES];    // bucket | slot of each line of the tile
    const u32 tid = threadIdx.x;
    constexpr u32 CHUNK_B = LB * LPC;
    unsigned char* region = out + (u64)blockIdx.x * region_bytes;
    const u64 tiles = nkeys / T, t0 = tiles * blockIdx.x / gridDim.x, t1 = tiles * (blockIdx.x + 1) / gridDim.x;
    if (tid < F) { fill[tid] = 0; chunk[tid] = tid; }
    if (tid == 0) next_chunk = F;
    __syncthreads();
    // two key buffers that rotate by NAME (the loop body is written twice): a register copy of a buffer whose load is still in
    // flight would make the wave wait for it, i.e. no prefetch at all
    u64x2 ka[4], kb[4];
    auto load = [&](u64 t, u64x2 (&kk)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) kk[i] = in[t * (T / 2) + (u64)i * NT + tid];
    };
    u32 rng = blockIdx.x * 2654435761u + 12345u;
    auto tile = [&](u64 t, const u64x2 (&k)[4], u64x2 (&kn)[4]) {
        if (t + 1 < t1) load(t + 1, kn);
        // line l of the tile -> bucket (uniform pseudo-random, the same sequence for every variant) -> destination
        if (tid < LINES) {
            const u32 h = (rng + tid * 0x9E3779B1u) * 0x85EBCA77u;
            const u32 b = (h >> 7) & (F - 1);
            const u32 f = atomicAdd(&fill[b], 1u);                    // (slots LPC.. belong to the bucket's next chunk)
            line_dst[tid] = b | (f << 16);
        }
        __syncthreads();
        if (tid < F) {                                               // a bucket whose chunk filled up in this tile moves to a fresh chunk
            const u32 f = fill[tid];                                 // (< 2 * LPC: at most one overflow per bucket and tile at these fan-outs)
            prev[tid] = chunk[tid];
            if (f >= LPC) { chunk[tid] = atomicAdd(&next_chunk, 1u); fill[tid] = f - LPC; }
        }
        __syncthreads();
        // ---- write: GL lanes per line ----
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const u32 e = (u32)i * NT + tid, l = e / GL, q = e % GL;
            const u32 d = line_dst[l], b = d & 0xFFFFu, f = d >> 16;
            const u32 c = f < LPC ? prev[b] : chunk[b], slot = f % LPC;
            unsigned char* cb = region + (u64)c * CHUNK_B;
            if (KPL == 16) {
                if (q < LB / 16) *reinterpret_cast<u64x2*>(cb + slot * LB + q * 16) = k[i];
            } else if (LB == 256) {
                *reinterpret_cast<u64x2*>(cb + slot * 256 + q * 16) = k[i];
            } else {                                                  // planes: lanes 0-7 the 128-B piece, lanes 8-11 the 64-B piece
                if (q < 8) *reinterpret_cast<u64x2*>(cb + slot * 128 + q * 16) = k[i];
                else if (q < 12) *reinterpret_cast<u64x2*>(cb + LPC * 128 + slot * 64 + (q - 8) * 16) = k[i];
            }
        }
        rng = rng * 1664525u + 1013904223u;
        __syncthreads();
    };
    if (t0 < t1) load(t0, ka);
    for (u64 t = t0; t < t1; t += 2) {
        tile(t, ka, kb);
        if (t + 1 < t1) tile(t + 1, kb, ka);
    }
}

template <u32 LB, u32 KPL>
int run(const u64x2* in, unsigned char* out, u64 nkeys, u32 F, u64 region_bytes, hipEvent_t e0, hipEvent_t e1) {
    float best = 1e9;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((packed_scatter<LB, KPL>), dim3(256), dim3(NT), 0, 0, in, out, nkeys, F, region_bytes);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    const double bytes = (double)nkeys * 8 + (double)nkeys / KPL * LB;
    printf("%u,%u,%u,%.3f,%.0f\n", KPL, LB, F, best, bytes / best / 1e6);
    return 0;
}

int main() {
    const u64 nkeys = 1ull << 30;                            // 1Gi keys: 8 GiB in
    u64x2* in; unsigned char* out;
    const u64 region_bytes = ((nkeys / 256) * 8 / 2048 + 1024) * 2048;      // every workgroup's dense output region (sized for 128-B lines)
    CK(hipMalloc(&in, nkeys * 8)); CK(hipMalloc(&out, region_bytes * 256));
    CK(hipMemset(in, 1, nkeys * 8)); CK(hipMemset(out, 0, region_bytes * 256));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("keys_per_line,line_bytes,F,ms,GBps(read+write)\n");
    for (u32 F : {128u, 256u, 512u}) {
        if (run<128, 16>(in, out, nkeys, F, region_bytes, e0, e1)) return 1;
        if (run<112, 16>(in, out, nkeys, F, region_bytes, e0, e1)) return 1;
        if (run<96, 16>(in, out, nkeys, F, region_bytes, e0, e1)) return 1;
        if (run<256, 32>(in, out, nkeys, F, region_bytes, e0, e1)) return 1;
        if (run<192, 32>(in, out, nkeys, F, region_bytes, e0, e1)) return 1;
    }
    return 0;
}
