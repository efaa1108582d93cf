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
// Three launches: fj_pack_scan (per bucket: key counts, and for every OUTPUT chunk the input chunk its first key sits in),
// fj_pack_offsets (output chunk ranges per bucket and owner), fj_pack_squeeze (the copy: 8 B read + 7 B written per key).
// The receiver's second radix pass reads the wire format directly (fj_partition_kernel<..., PK7>).
#include "fj_internal.h"

namespace {

constexpr u32 PK_NT = 256;             // threads of the scan and squeeze workgroups = keys per chunk
constexpr u32 PK_CPW = 8;              // output chunks a squeeze workgroup handles per step (independent load chains in flight)

// one workgroup per first-pass bucket: walk its chunk list, prefix-sum the chunks' key counts
__global__ __launch_bounds__(PK_NT) void fj_pack_scan(FjPackArgs a) {
    __shared__ u32 wsum[PK_NT / 64];
    const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.x;
    const u32 l0 = a.boff[b], n = a.boff[b + 1] - l0;
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
        // the input chunk that holds key 256 * c of the bucket's stream starts output chunk c (a chunk has <= 256 keys: at most
        // one such c per input chunk, exactly one input chunk per c).  The descriptor carries that chunk's list entry and the
        // next one, so that the copy reaches its keys with ONE dependent load (full chunks make a third input chunk rare).
        const u32 c = (pre + FJ_CHUNK - 1) >> FJ_CHUNK_LOG;
        if (cnt && (c << FJ_CHUNK_LOG) < pre + cnt) a.fi[l0 + c] = make_uint4(e, e1, pre, j);
        run += tot;
        __syncthreads();
    }
    if (tid == 0) a.bkeys[b] = run;
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
    }
}

// the copy.  A resident workgroup takes PK_CPW consecutive output chunks per step; thread t owns key t of each.  It is written
// against latency: the per-bucket tables live in LDS, an output chunk's descriptor (one 16-byte load) names the two input chunks
// its keys can sit in (a third one is rare: only after a run of tiny partial chunks), the descriptors of the NEXT step are
// requested before this step's keys, and a chunk's 1792 (2048) bytes leave as whole 16-byte pieces.
template <bool W7, bool VALS>
__global__ __launch_bounds__(PK_NT) void fj_pack_squeeze(FjPackArgs a) {
    constexpr u32 NB1 = (1u << FJ_MAX_FAN_LOG) + 1;
    __shared__ u32 s_ob[NB1], s_bo[NB1], s_nk[NB1];
    __shared__ __attribute__((aligned(16))) unsigned char s_pk[PK_CPW][W7 ? FJ_WIRE7_BYTES : 16];
    __shared__ unsigned char* s_dst[PK_CPW];          // W7: where chunk u of the step goes (null: past the end)
    const u32 tid = threadIdx.x, F = a.nb, N = a.nranks;
    for (u32 i = tid; i <= F; i += PK_NT) { s_ob[i] = a.obase[i]; s_bo[i] = a.boff[i]; s_nk[i] = i < F ? a.bkeys[i] : 0u; }
    __syncthreads();
    const u32 total = s_ob[F];
    u32 bb[PK_CPW], cc[PK_CPW];
    uint4 f[PK_CPW];
    auto describe = [&](u32 g0, u32 (&b_)[PK_CPW], u32 (&c_)[PK_CPW], uint4 (&f_)[PK_CPW]) {
#pragma unroll
        for (u32 u = 0; u < PK_CPW; ++u) {
            const u32 g = g0 + u < total ? g0 + u : total - 1;            // (clamped: the loads stay unconditional)
            u32 lo = 0, hi = F;                                            // last b with s_ob[b] <= g (buckets without keys repeat an offset)
            while (hi - lo > 1) { const u32 mid = (lo + hi) >> 1; if (s_ob[mid] <= g) lo = mid; else hi = mid; }
            b_[u] = lo; c_[u] = g - s_ob[lo];
            f_[u] = a.fi[s_bo[lo] + c_[u]];
        }
    };
    u32 g0 = blockIdx.x * PK_CPW;
    if (g0 < total) describe(g0, bb, cc, f);
    for (; g0 < total; g0 += gridDim.x * PK_CPW) {
        u32 bn[PK_CPW], cn[PK_CPW];
        uint4 fn[PK_CPW];
        const u32 gn = g0 + gridDim.x * PK_CPW;
        if (gn < total) describe(gn, bn, cn, fn);                          // next step's descriptors fly under this step's keys
        u64 key[PK_CPW], val[PK_CPW];
        bool ok[PK_CPW];
#pragma unroll
        for (u32 u = 0; u < PK_CPW; ++u) {
            const u32 pos = (cc[u] << FJ_CHUNK_LOG) + tid, nk = s_nk[bb[u]];
            ok[u] = g0 + u < total && pos < nk;
            u32 rel = pos - f[u].z, e = f[u].x;
            if (ok[u] && rel >= FJ_LIST_CNT(e)) {
                rel -= FJ_LIST_CNT(e); e = f[u].y;
                u32 j = f[u].w + 1;
                while (rel >= FJ_LIST_CNT(e)) { rel -= FJ_LIST_CNT(e); ++j; e = a.list[s_bo[bb[u]] + j]; }      // (pos < nk: the walk ends inside the list)
            }
            const u64 src = (u64)FJ_LIST_ID(e) * FJ_CHUNK + (ok[u] ? rel : 0u);
            key[u] = a.keys[src];
            if (VALS) val[u] = a.vals[src];
        }
#pragma unroll
        for (u32 u = 0; u < PK_CPW; ++u) {
            if (g0 + u >= total) { if (W7 && tid == 0) s_dst[u] = nullptr; continue; }     // (workgroup-uniform)
            const u32 b = bb[u], r = (b * N) >> a.fan_log, idx = g0 + u - s_ob[(r * F + N - 1) / N];
            const u64 k = ok[u] ? key[u] : 0ull;
            if (W7) {
                *reinterpret_cast<u32*>(&s_pk[u][tid * 4u]) = (u32)k;
                *reinterpret_cast<u16*>(&s_pk[u][FJ_WIRE7_MID + tid * 2u]) = (u16)(k >> 32);
                s_pk[u][FJ_WIRE7_HI + tid] = (unsigned char)(k >> 48);
            } else {
                reinterpret_cast<u64*>(a.dst_k[r] + (u64)idx * (FJ_CHUNK * 8u))[tid] = k;
            }
            if (VALS) a.dst_v[r][(u64)idx * FJ_CHUNK + tid] = ok[u] ? val[u] : 0ull;
            if (tid == 0) {
                const u32 left = s_nk[b] - (cc[u] << FJ_CHUNK_LOG);
                a.dst_d[r][idx] = (b << FJ_DIR_CNT_BITS) | (left < FJ_CHUNK ? left : FJ_CHUNK);
                if (W7) s_dst[u] = a.dst_k[r] + (u64)idx * FJ_WIRE7_BYTES;
            }
        }
        if (W7) {
            __syncthreads();
            for (u32 i = tid; i < PK_CPW * (FJ_WIRE7_BYTES / 16u); i += PK_NT) {
                const u32 u = i / (FJ_WIRE7_BYTES / 16u), q = i % (FJ_WIRE7_BYTES / 16u);
                unsigned char* d = s_dst[u];
                if (d) reinterpret_cast<uint4*>(d)[q] = reinterpret_cast<const uint4*>(&s_pk[u][0])[q];
            }
            __syncthreads();
        }
#pragma unroll
        for (u32 u = 0; u < PK_CPW; ++u) { bb[u] = bn[u]; cc[u] = cn[u]; f[u] = fn[u]; }
    }
}

}  // namespace

hipError_t fj_launch_pack_plan(const FjPackArgs& a, hipStream_t s) {
    if (a.nb > (1u << FJ_MAX_FAN_LOG) || a.nb != (1u << a.fan_log) || a.nranks < 1 || a.nranks > 64 || a.nranks > a.nb) return hipErrorInvalidValue;
    hipLaunchKernelGGL(fj_pack_scan, dim3(a.nb), dim3(PK_NT), 0, s, a);
    hipLaunchKernelGGL(fj_pack_offsets, dim3(1), dim3(512), 0, s, a);
    return hipGetLastError();
}

hipError_t fj_launch_pack_squeeze(const FjPackArgs& a, u32 grid, hipStream_t s) {
    if (a.wire7 && a.fan_log < 8) return hipErrorInvalidValue;
    if (a.wire7) {
        if (a.vals) hipLaunchKernelGGL((fj_pack_squeeze<true, true>), dim3(grid), dim3(PK_NT), 0, s, a);
        else hipLaunchKernelGGL((fj_pack_squeeze<true, false>), dim3(grid), dim3(PK_NT), 0, s, a);
    } else {
        if (a.vals) hipLaunchKernelGGL((fj_pack_squeeze<false, true>), dim3(grid), dim3(PK_NT), 0, s, a);
        else hipLaunchKernelGGL((fj_pack_squeeze<false, false>), dim3(grid), dim3(PK_NT), 0, s, a);
    }
    return hipGetLastError();
}
