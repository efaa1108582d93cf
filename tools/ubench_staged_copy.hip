// Can a two-step data movement keep its intermediate in the 256 MiB Infinity Cache?
//   in --(step A)--> staging (batch-sized, reused every batch) --(step B)--> out
// One persistent kernel, 512 workgroups, grid barrier + agent-scope fence between the steps of every batch; step B of a
// workgroup reads what ANOTHER workgroup (another XCD) wrote in step A.  Compared with two plain copies through a full-size
// intermediate (the shape of the two partition passes today).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint64_t u64; typedef uint32_t u32;
struct __attribute__((aligned(16))) u64x2 { u64 x, y; };
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ void grid_barrier(u32* counter, u32 nwg, u32& epoch) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();                                        // release: this workgroup's stores reach memory / other XCDs
        ++epoch;
        atomicAdd(counter, 1u);
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch * nwg) __builtin_amdgcn_s_sleep(2);
        __threadfence();                                        // acquire
    }
    __syncthreads();
}

// barrier without cache maintenance: the staging bytes themselves travel with agent scope (sc1: past the XCD's L2)
__device__ __forceinline__ void grid_barrier_nofence(u32* counter, u32 nwg, u32& epoch) {
    __builtin_amdgcn_s_waitcnt(0);                              // this thread's stores have been acknowledged
    __syncthreads();
    if (threadIdx.x == 0) {
        ++epoch;
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch * nwg) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
}

__global__ __launch_bounds__(512) void staged_sc1(const u64* __restrict__ in, u64* stg, u64* __restrict__ out, u64 n8, u64 batch8, u32* counter) {
    const u32 nwg = gridDim.x, w = blockIdx.x;
    const u64 share = batch8 / nwg;                             // 8-B elements per workgroup and batch
    u32 epoch = 0;
    for (u64 b0 = 0; b0 < n8; b0 += batch8) {
        for (u64 i = threadIdx.x; i < share; i += blockDim.x)
            __hip_atomic_store(&stg[(u64)w * share + i], in[b0 + (u64)w * share + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        grid_barrier_nofence(counter, nwg, epoch);
        const u32 src = (w + 37u) % nwg;
        for (u64 i = threadIdx.x; i < share; i += blockDim.x)
            out[b0 + (u64)src * share + i] = __hip_atomic_load(&stg[(u64)src * share + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        grid_barrier_nofence(counter, nwg, epoch);
    }
}

// per batch: every workgroup copies its share of the batch in -> staging, barrier, then the share of ANOTHER workgroup staging -> out
__global__ __launch_bounds__(512) void staged(const u64x2* __restrict__ in, u64x2* __restrict__ stg, u64x2* __restrict__ out,
                                              u64 n16, u64 batch16, u32* counter) {
    const u32 nwg = gridDim.x, w = blockIdx.x;
    const u64 share = batch16 / nwg;                            // 16-B elements per workgroup and batch
    u32 epoch = 0;
    for (u64 b0 = 0; b0 < n16; b0 += batch16) {
        for (u64 i = threadIdx.x; i < share; i += blockDim.x) stg[(u64)w * share + i] = in[b0 + (u64)w * share + i];
        grid_barrier(counter, nwg, epoch);
        const u32 src = (w + 37u) % nwg;                        // 37 is odd: a different CU, mostly a different XCD
        for (u64 i = threadIdx.x; i < share; i += blockDim.x) {
            const u64x2 v = stg[(u64)src * share + i];
            out[b0 + (u64)src * share + i] = v;
        }
        grid_barrier(counter, nwg, epoch);                      // staging is reused by the next batch
    }
}
__global__ __launch_bounds__(256) void cp(const u64x2* __restrict__ in, u64x2* __restrict__ out, u64 n16) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (u64)gridDim.x * blockDim.x) out[i] = in[i];
}
int main() {
    const u64 bytes = 8ull << 30, n16 = bytes / 16;
    u64x2 *in, *mid, *out, *stg; u32* counter;
    CK(hipMalloc(&in, bytes)); CK(hipMalloc(&mid, bytes)); CK(hipMalloc(&out, bytes)); CK(hipMalloc(&stg, 512ull << 20)); CK(hipMalloc(&counter, 4));
    CK(hipMemset(in, 1, bytes)); CK(hipMemset(out, 0, bytes)); CK(hipMemset(mid, 0, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("variant,batch_MB,ms,GBps_in_to_out\n");
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(cp, dim3(2048), dim3(256), 0, 0, in, mid, n16);
        hipLaunchKernelGGL(cp, dim3(2048), dim3(256), 0, 0, mid, out, n16);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    printf("two_plain_copies,-,%.3f,%.0f\n", best, (double)bytes / best / 1e6);
    for (u64 mb : {16ull, 32ull, 64ull, 128ull, 256ull}) {
        const u64 batch16 = (mb << 20) / 16;
        best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemset(counter, 0, 4));
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(staged, dim3(512), dim3(512), 0, 0, in, stg, out, n16, batch16, counter);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        printf("staged_persistent,%llu,%.3f,%.0f\n", (unsigned long long)mb, best, (double)bytes / best / 1e6);
    }
    for (u64 mb : {16ull, 32ull, 64ull, 128ull, 256ull}) {
        const u64 batch8 = (mb << 20) / 8;
        CK(hipMemset(out, 0, bytes));
        best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemset(counter, 0, 4));
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(staged_sc1, dim3(512), dim3(512), 0, 0, (const u64*)in, (u64*)stg, (u64*)out, n16 * 2, batch8, counter);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        printf("staged_sc1_8B,%llu,%.3f,%.0f\n", (unsigned long long)mb, best, (double)bytes / best / 1e6);
    }
    // correctness of the staged path: out == in
    u64x2 h; CK(hipMemcpy(&h, out + (n16 - 12345), 16, hipMemcpyDeviceToHost));
    printf("check,%s\n", (h.x == 0x0101010101010101ull && h.y == 0x0101010101010101ull) ? "ok" : "MISMATCH");
    return 0;
}
