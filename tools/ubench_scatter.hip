// Microbenchmark: what HBM rate does MI355X sustain for "stream in, scatter out in pieces of X bytes"?
// (read side coalesced 16 B/lane; write side: contiguous pieces of X bytes at pseudo-random piece positions)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef uint64_t u64; typedef uint32_t u32;
struct __attribute__((aligned(16))) u64x2 { u64 x, y; };
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// n16 = number of 16-B elements; piece16 = 16-B elements per piece; in_piece16 = read-side piece size (0 = linear)
__global__ __launch_bounds__(256) void scatter_kernel(const u64x2* __restrict__ in, u64x2* __restrict__ out, u64 n16, u32 piece_log,
                                                      u64 npieces_mask, u64 mul, u32 in_piece_log, u64 mul_in) {
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
        u64 src = i;                                        // n16 is a power of two: index math is shifts and masks only
        if (in_piece_log) { const u64 p = i >> in_piece_log, o = i & ((1ull << in_piece_log) - 1); src = ((((p * mul_in) >> 7) & ((n16 >> in_piece_log) - 1)) << in_piece_log) + o; }
        const u64x2 v = in[src];
        const u64 p = i >> piece_log, o = i & ((1ull << piece_log) - 1);
        const u64 q = (p * mul) & ((n16 >> piece_log) - 1);  // odd multiplier mod 2^k: a bijection of the piece index
        out[(q << piece_log) + o] = v;
    }
}

int main(int argc, char** argv) {
    const bool by_grid = argc > 1 && argv[1][0] == 'g';       // `ubench_scatter grid`: 128-B / 2-KiB / 1-MiB pieces over many grid sizes
    const u64 bytes = 8ull << 30;                           // 8 GiB in, 8 GiB out
    const u64 n16 = bytes / 16;
    u64x2 *in, *out;
    CK(hipMalloc(&in, bytes)); CK(hipMalloc(&out, bytes));
    CK(hipMemset(in, 1, bytes)); CK(hipMemset(out, 0, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<int> grids = {2048, 16384};
    if (by_grid) grids = {512, 1024, 2048, 4096, 8192, 16384, 65536, 262144};
    printf("piece_bytes,in_piece_bytes,grid,ms,GBps(read+write)\n");
    for (int g : grids)
    for (u32 in_log : {0u, 7u}) {                            // read side linear, or random 2-KiB chunks
        for (u32 piece_log = 0; piece_log <= 16; ++piece_log) {   // 16 B .. 16 KiB pieces (`grid` mode: 128 B, 2 KiB, 1 MiB)
            if (by_grid ? (piece_log != 3 && piece_log != 7 && piece_log != 16) : piece_log > 10) continue;
            float best = 1e9;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(scatter_kernel, dim3(g), dim3(256), 0, 0, in, out, n16, piece_log, ~0ull, 0x9E3779B97F4A7C15ull | 1, in_log, 0xD6E8FEB86659FD93ull | 1);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
            }
            printf("%u,%u,%d,%.3f,%.0f\n", 16u << piece_log, in_log ? (16u << in_log) : 0u, g, best, 2.0 * bytes / best / 1e6);
        }
    }
    return 0;
}
