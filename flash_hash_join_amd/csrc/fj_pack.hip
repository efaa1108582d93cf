// fj_pack.hip -- owner shuffle, sender side: a level-1 chunk set -> what goes on the xGMI links.
//
// No reference counterpart (the reference is one process, hash_join.cpp:318); what is exploited is that radix partitions are
// independent join units (hash_join.cpp:340-356, :515-525): the first radix pass of the plan for the TOTAL build side is the
// owner split, bucket b of its F buckets belongs to rank (b * nranks) >> log2(F).  The exchange is bound by the bytes it puts
// on ONE link (xGMI is a point-to-point mesh), so after that pass - the ordinary fj_partition_kernel over the rank's local
// rows - the rows are rewritten for the wire:
//   * dense: every bucket's keys are packed into full 256-key chunks (one partial chunk per bucket and piece instead of one
//     per (workgroup, bucket): 3-10 % fewer chunks), bucket after bucket, so that an owner's share is ONE contiguous range;
//   * narrow: chunk pools hold MIXED keys (a bijection of the key, fj_common.h) and a chunk of bucket b need not carry the
//     top bits b implies - 7 bytes per key (FJ_WIRE7_*) when the first pass has >= 256 buckets;
//   * one directory word (bucket << 9 | keys) per chunk, as the partition pass writes them.
// Four launches: fj_pack_count (keys per bucket), fj_pack_offsets (output chunk ranges per bucket and owner), fj_pack_scan (one
// descriptor per OUTPUT chunk: the input chunk its first key sits in), fj_pack_squeeze (the copy: 8 B read + 7 B written per key).
// The receiver's second radix pass reads the wire format directly (fj_partition_kernel<..., PK7>).
#include "fj_internal.h"

namespace {

constexpr u32 PK_NT = 256;             // threads of the scan and squeeze workgroups = keys per chunk
constexpr u32 PK_CPW = 4;              // output chunks a squeeze workgroup handles per step (independent load chains in flight)

// keys per first-pass bucket: one workgroup per bucket sums its chunk list's counts
__global__ __launch_bounds__(PK_NT) void fj_pack_count(FjPackArgs a) {
    __shared__ u32 wsum[PK_NT / 64];
    const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.x;
    const u32 l0 = a.boff[b], n = a.boff[b + 1] - l0;
    u32 sum = 0;
    for (u32 j = tid; j < n; j += PK_NT) sum += FJ_LIST_CNT(a.list[l0 + j]);
    sum = fj_wave_sum(sum);
    if (lane == 0) wsum[wave] = sum;
    __syncthreads();
    if (tid == 0) { u32 t = 0; for (u32 w = 0; w < PK_NT / 64; ++w) t += wsum[w]; a.bkeys[b] = t; }
}

// one workgroup: output chunks before every bucket (bucket order = owner order), chunks per owner
__global__ __launch_bounds__(512) void fj_pack_offsets(FjPackArgs a) {
    __shared__ u32 s_ob[(1u << FJ_MAX_FAN_LOG) + 1];
    __shared__ u32 wsum[8];
    const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, F = a.nb;
    const u32 x = tid < F ? (a.bkeys[tid] + FJ_CHUNK - 1) >> FJ_CHUNK_LOG : 0u;
    u32 inc = x;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const u32 y = __shfl_up(inc, d, 64); if ((int)lane >= d) inc += y; }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    u32 pre = inc - x, tot = 0;
    for (u32 w = 0; w < 8; ++w) { if (w < wave) pre += wsum[w]; tot += wsum[w]; }
    if (tid < F) { s_ob[tid] = pre; a.obase[tid] = pre; }
    if (tid == 0) { s_ob[F] = tot; a.obase[F] = tot; }
    __syncthreads();
    if (tid < a.nranks) {
        const u32 lo = (tid * F + a.nranks - 1) / a.nranks, hi = ((tid + 1) * F + a.nranks - 1) / a.nranks;    // first bucket b with (b * nranks) >> log2(F) == tid
        a.used[tid] = s_ob[hi] - s_ob[lo];
        a.obase[F + 1 + tid] = s_ob[lo];                     // ... and the first output chunk of every owner
    }
}

// one workgroup per bucket: prefix-sum the chunk list's key counts and write one DESCRIPTOR per output chunk, at its global
// output index: the input chunk that holds key 256 * c of the bucket's stream starts output chunk c (a chunk has <= 256 keys: at
// most one such c per input chunk, exactly one input chunk per c).  The descriptor carries that chunk's list entry and the next
// one, so that the copy reaches its keys with ONE dependent load (full chunks make a third input chunk rare).
__global__ __launch_bounds__(PK_NT) void fj_pack_scan(FjPackArgs a) {
    __shared__ u32 wsum[PK_NT / 64];
    const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.x;
    const u32 l0 = a.boff[b], n = a.boff[b + 1] - l0, ob = a.obase[b], nk = a.bkeys[b];
    u32 run = 0;
    for (u32 c0 = 0; c0 < n; c0 += PK_NT) {
        const u32 j = c0 + tid;
        const u32 e = j < n ? a.list[l0 + j] : 0u, e1 = j + 1 < n ? a.list[l0 + j + 1] : e;
        const u32 cnt = j < n ? FJ_LIST_CNT(e) : 0u;
        u32 inc = cnt;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const u32 y = __shfl_up(inc, d, 64); if ((int)lane >= d) inc += y; }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        u32 pre = run + inc - cnt, tot = 0;
        for (u32 w = 0; w < PK_NT / 64; ++w) { if (w < wave) pre += wsum[w]; tot += wsum[w]; }
        const u32 c = (pre + FJ_CHUNK - 1) >> FJ_CHUNK_LOG;
        if (cnt && (c << FJ_CHUNK_LOG) < pre + cnt) {
            const u32 left = nk - (c << FJ_CHUNK_LOG);
            a.fi[ob + c] = make_uint4(e, e1, (c << FJ_CHUNK_LOG) - pre, l0 + j);          // .z: position of the output chunk's key 0 inside input chunk e
            a.fb[ob + c] = (b << FJ_DIR_CNT_BITS) | (left < FJ_CHUNK ? left : FJ_CHUNK);   // = the chunk's directory word
        }
        run += tot;
        __syncthreads();
    }
}

// the copy.  A workgroup takes PK_CPW consecutive output chunks per step, thread t owns key t of each; every wave is on its own
// (no LDS, no barrier): the chunk's descriptor and directory word come through the scalar cache (wave-uniform loads), one
// dependent vector load reaches the key, and the three planes of the wire format leave straight from registers - every lane
// stores its low word, even lanes the pair of 16-bit fields they collect from their neighbour, every fourth lane the four bytes
// of its group: a wave's stores cover 256 / 128 / 64 contiguous bytes.  (Rounds of this kernel with an LDS transpose and with a
// per-chunk binary search over the bucket offsets ran at 2.7-3.0 TB/s.)
template <bool W7, bool VALS>
__global__ __launch_bounds__(PK_NT) void fj_pack_squeeze(FjPackArgs a, const uint4* __restrict__ fi, const u32* __restrict__ fb, const u32* __restrict__ list,
                                                         const u32* __restrict__ obase, const u64* __restrict__ keys, const u64* __restrict__ vals) {
    const u32 tid = threadIdx.x, F = a.nb, N = a.nranks;
    const u32 total = obase[F];
    for (u32 g0 = blockIdx.x * PK_CPW; g0 < total; g0 += gridDim.x * PK_CPW) {
        uint4 f[PK_CPW]; u32 dw[PK_CPW];
#pragma unroll
        for (u32 u = 0; u < PK_CPW; ++u) {
            const u32 g = g0 + u < total ? g0 + u : total - 1;            // (clamped: the loads stay unconditional; g is wave-uniform)
            f[u] = fi[g]; dw[u] = fb[g];
        }
        u64 key[PK_CPW], val[PK_CPW];
        bool ok[PK_CPW];
#pragma unroll
        for (u32 u = 0; u < PK_CPW; ++u) {
            ok[u] = g0 + u < total && tid < (dw[u] & FJ_DIR_CNT_MASK);
            u32 rel = f[u].z + tid, e = f[u].x;
            if (ok[u] && rel >= FJ_LIST_CNT(e)) {
                rel -= FJ_LIST_CNT(e); e = f[u].y;
                u32 j = f[u].w + 1;
                while (rel >= FJ_LIST_CNT(e)) { rel -= FJ_LIST_CNT(e); ++j; e = list[j]; }      // (the key exists: the walk ends inside the bucket's list)
            }
            const u64 src = (u64)FJ_LIST_ID(e) * FJ_CHUNK + (ok[u] ? rel : 0u);
            key[u] = keys[src];
            if (VALS) val[u] = vals[src];
        }
#pragma unroll
        for (u32 u = 0; u < PK_CPW; ++u) {
            if (g0 + u >= total) continue;                                 // (workgroup-uniform)
            const u32 b = dw[u] >> FJ_DIR_CNT_BITS, r = (b * N) >> a.fan_log, idx = g0 + u - obase[F + 1 + r];
            const u64 k = ok[u] ? key[u] : 0ull;
            if (W7) {
                unsigned char* d = a.dst_k[r] + (u64)idx * FJ_WIRE7_BYTES;
                const u32 md = (u32)(k >> 32) & 0xFFFFu, hb = (u32)(k >> 48) & 0xFFu;
                const u32 md2 = md | (__shfl_down(md, 1, 64) << 16);
                u32 hb4 = hb | (__shfl_down(hb, 1, 64) << 8);
                hb4 |= __shfl_down(hb4, 2, 64) << 16;
                reinterpret_cast<u32*>(d)[tid] = (u32)k;
                if (!(tid & 1u)) reinterpret_cast<u32*>(d + FJ_WIRE7_MID)[tid >> 1] = md2;
                if (!(tid & 3u)) reinterpret_cast<u32*>(d + FJ_WIRE7_HI)[tid >> 2] = hb4;
            } else {
                reinterpret_cast<u64*>(a.dst_k[r] + (u64)idx * (FJ_CHUNK * 8u))[tid] = k;
            }
            if (VALS) a.dst_v[r][(u64)idx * FJ_CHUNK + tid] = ok[u] ? val[u] : 0ull;
            if (tid == 0) a.dst_d[r][idx] = dw[u];
        }
    }
}

}  // namespace

hipError_t fj_launch_pack_plan(const FjPackArgs& a, hipStream_t s) {
    if (a.nb > (1u << FJ_MAX_FAN_LOG) || a.nb != (1u << a.fan_log) || a.nranks < 1 || a.nranks > 64 || a.nranks > a.nb) return hipErrorInvalidValue;
    hipLaunchKernelGGL(fj_pack_count, dim3(a.nb), dim3(PK_NT), 0, s, a);
    hipLaunchKernelGGL(fj_pack_offsets, dim3(1), dim3(512), 0, s, a);
    hipLaunchKernelGGL(fj_pack_scan, dim3(a.nb), dim3(PK_NT), 0, s, a);
    return hipGetLastError();
}

hipError_t fj_launch_pack_squeeze(const FjPackArgs& a, u32 grid, hipStream_t s) {
    if (a.wire7 && a.fan_log < 8) return hipErrorInvalidValue;
    auto kern = a.wire7 ? (a.vals ? fj_pack_squeeze<true, true> : fj_pack_squeeze<true, false>) : (a.vals ? fj_pack_squeeze<false, true> : fj_pack_squeeze<false, false>);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(PK_NT), 0, s, a, a.fi, a.fb, a.list, a.obase, a.keys, a.vals);
    return hipGetLastError();
}
