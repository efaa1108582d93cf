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
constexpr u32 PK_CPW = 4;              // output chunks a squeeze workgroup handles per step (two per half workgroup: independent load chains in flight)

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

// the copy.  A workgroup takes PK_CPW consecutive output chunks per step - its lower 128 threads the even ones, its upper 128 the
// odd ones - and a thread owns keys 2q, 2q + 1 of its chunk; every wave is on its own (no LDS, no barrier): the chunk's
// descriptor and directory word come through the scalar cache (wave-uniform loads), one dependent 16-byte load reaches the two
// keys (8-byte aligned: after a partial input chunk the stream's positions are odd as often as even), and the three planes of the
// wire format leave straight from registers - 8 bytes of low words and 4 bytes of 16-bit fields per lane, even lanes the four
// bytes of their group: a wave's stores cover 512 / 256 / 128 contiguous bytes.  (Rounds of this kernel: LDS transpose + a
// per-chunk binary search over the bucket offsets 3.0 TB/s; one key per lane 4.4 TB/s.)
// The launch gives a workgroup one or two steps (512 workgroups per CU in the grid): with 16 per CU and a static share of ~40 steps
// each the kernel waited for its slowest CUs - 1.05 ms per 312M keys instead of 0.84 ms = 5.6 TB/s, the pool's plain-copy rate.
struct __attribute__((packed, aligned(8))) u64x2u { u64 x, y; };
template <bool W7, bool VALS>
__global__ __launch_bounds__(PK_NT) void fj_pack_squeeze(FjPackArgs a, const uint4* __restrict__ fi, const u32* __restrict__ fb, const u32* __restrict__ list,
                                                         const u32* __restrict__ obase, const u64* __restrict__ keys, const u64* __restrict__ vals) {
    constexpr u32 SLOTS = PK_CPW / 2;
    const u32 tid = threadIdx.x, F = a.nb, N = a.nranks;
    const u32 half = (u32)__builtin_amdgcn_readfirstlane((int)(tid >> 7)), q = tid & 127u;
    const u32 total = obase[F];
    // the key at position `rel` of the stream that starts in list entry e (index j of the chunk list): walk on while it lies beyond
    auto resolve = [&](u32 rel, u32 e, u32 e1, u32 j) -> u64 {
        if (rel >= FJ_LIST_CNT(e)) {
            rel -= FJ_LIST_CNT(e); e = e1; ++j;
            while (rel >= FJ_LIST_CNT(e)) { rel -= FJ_LIST_CNT(e); ++j; e = list[j]; }      // (the key exists: the walk ends inside the bucket's list)
        }
        return (u64)FJ_LIST_ID(e) * FJ_CHUNK + rel;
    };
    for (u32 g0 = blockIdx.x * PK_CPW; g0 < total; g0 += gridDim.x * PK_CPW) {
        uint4 f[SLOTS]; u32 dw[SLOTS];
#pragma unroll
        for (u32 u = 0; u < SLOTS; ++u) {
            const u32 gg = g0 + 2 * u + half, g = gg < total ? gg : total - 1;     // (clamped: the loads stay unconditional; g is wave-uniform)
            f[u] = fi[g]; dw[u] = fb[g];
        }
        u64 k0[SLOTS], k1[SLOTS], v0[SLOTS], v1[SLOTS];
#pragma unroll
        for (u32 u = 0; u < SLOTS; ++u) {
            const u32 cnt = g0 + 2 * u + half < total ? (dw[u] & FJ_DIR_CNT_MASK) : 0u;
            const bool ok0 = 2 * q < cnt, ok1 = 2 * q + 1 < cnt;
            const u32 rel = f[u].z + 2 * q;
            k0[u] = 0; k1[u] = 0; v0[u] = 0; v1[u] = 0;
            if (ok1 && rel + 1 < FJ_LIST_CNT(f[u].x)) {                           // both keys in the first input chunk: one 16-byte load
                const u64 src = (u64)FJ_LIST_ID(f[u].x) * FJ_CHUNK + rel;
                const u64x2u kk = *reinterpret_cast<const u64x2u*>(keys + src);
                k0[u] = kk.x; k1[u] = kk.y;
                if (VALS) { const u64x2u vv = *reinterpret_cast<const u64x2u*>(vals + src); v0[u] = vv.x; v1[u] = vv.y; }
            } else {
                if (ok0) { const u64 s0 = resolve(rel, f[u].x, f[u].y, f[u].w); k0[u] = keys[s0]; if (VALS) v0[u] = vals[s0]; }
                if (ok1) { const u64 s1 = resolve(rel + 1, f[u].x, f[u].y, f[u].w); k1[u] = keys[s1]; if (VALS) v1[u] = vals[s1]; }
            }
        }
#pragma unroll
        for (u32 u = 0; u < SLOTS; ++u) {
            const u32 g = g0 + 2 * u + half;
            if (g >= total) continue;                                      // (wave-uniform)
            const u32 b = dw[u] >> FJ_DIR_CNT_BITS, r = (b * N) >> a.fan_log, idx = g - obase[F + 1 + r];
            if (W7) {
                unsigned char* d = a.dst_k[r] + (u64)idx * FJ_WIRE7_BYTES;
                const u32 md = ((u32)(k0[u] >> 32) & 0xFFFFu) | ((u32)(k1[u] >> 32) << 16);
                const u32 hb2 = ((u32)(k0[u] >> 48) & 0xFFu) | (((u32)(k1[u] >> 48) & 0xFFu) << 8);
                const u32 hb4 = hb2 | (__shfl_down(hb2, 1, 64) << 16);
                reinterpret_cast<uint2*>(d)[q] = make_uint2((u32)k0[u], (u32)k1[u]);
                reinterpret_cast<u32*>(d + FJ_WIRE7_MID)[q] = md;
                if (!(q & 1u)) reinterpret_cast<u32*>(d + FJ_WIRE7_HI)[q >> 1] = hb4;
            } else {
                u64x2 kk; kk.x = k0[u]; kk.y = k1[u];
                reinterpret_cast<u64x2*>(a.dst_k[r] + (u64)idx * (FJ_CHUNK * 8u))[q] = kk;
            }
            if (VALS) { u64x2 vv; vv.x = v0[u]; vv.y = v1[u]; reinterpret_cast<u64x2*>(a.dst_v[r] + (u64)idx * FJ_CHUNK)[q] = vv; }
            if (q == 0) a.dst_d[r][idx] = dw[u];
        }
    }
}

// ---- sender-side precheck in chunk form: one small Bloom filter per FINAL partition of the global plan ---------------------------
// An owner writes, for every final partition it owns, a blocked Bloom filter of the build keys it received (FJ_PFILT_BYTES = 4 KiB
// for ~3-4K keys: ~8 bits per key, four bits inside one 64-bit block), all owners' filters are all-gathered (1 byte per build key
// in all), and a sender drops the probe keys that no filter admits BEFORE the copy into the wire format - at 50 % hits ~45 % of
// what would travel, at 5 % hits ~90 %.  The filters are far too many for LDS (config 5: 1 GB), but a sender works through its
// level-1 chunk lists bucket after bucket, and the filters of ONE level-1 bucket's partitions are a contiguous 2 MiB: they stay in
// the XCDs' L2s while that bucket's keys stream by (round 2's microbenchmark: 205-270 G keys/s against an L2-resident filter,
// profiles/r02_ubench_bloom_probe.csv).  Role of the reference's bloom directory (hash_join.cpp:60-74, :122, :183-189): insert
// and test use the same bits, so no key of the build side is ever dropped.
__device__ __forceinline__ void pf_bits(u64 h, u32& block, u64& mask) {
    const u32 w = FJ_HW2(h);
    block = __umulhi(w, FJ_PFILT_BYTES / 8u);                                       // top bits of hash word 2
    mask = (1ull << (w & 63u)) | (1ull << ((w >> 6) & 63u)) | (1ull << ((w >> 12) & 63u)) | (1ull << ((w >> 18) & 63u));
}

// owner: one 256-thread workgroup per final partition (grid-stride): filter in LDS, written out whole
__global__ __launch_bounds__(256) void fj_part_filter_export(FjChunkSet build, u64* __restrict__ out) {
    __shared__ unsigned long long filt[FJ_PFILT_BYTES / 8];
    const u32 tid = threadIdx.x;
    for (u32 p = blockIdx.x; p < build.nb; p += gridDim.x) {
        for (u32 i = tid; i < FJ_PFILT_BYTES / 8; i += 256) filt[i] = 0;
        __syncthreads();
        const u32 l0 = build.boff[p], n = build.boff[p + 1] - l0;
        for (u32 c = 0; c < n; ++c) {
            const u32 e = build.list[l0 + c];
            if (tid < FJ_LIST_CNT(e)) {
                u32 b; u64 m;
                pf_bits(build.keys[(u64)FJ_LIST_ID(e) * FJ_CHUNK + tid], b, m);
                atomicOr(&filt[b], (unsigned long long)m);
            }
        }
        __syncthreads();
        for (u32 i = tid; i < FJ_PFILT_BYTES / 8; i += 256) out[(u64)p * (FJ_PFILT_BYTES / 8) + i] = filt[i];
        __syncthreads();
    }
}

// sender: the chunks of a level-1 chunk set are compacted IN PLACE to the keys some filter admits, and a chunk's list entry gets its
// new count.  A WAVE owns a chunk - four keys per lane, two 16-byte loads - and is on its own while it works on it: four independent
// filter words in flight per lane, survivor positions from ballots (their order inside the chunk is free), no LDS traffic; every key
// of the chunk is in a register before the first survivor is written back.  A chunk without survivors keeps its first key (a key that
// matches nothing is harmless on the wire; a list entry cannot say "empty").  part_shift: H >> part_shift = final partition.
// Bound by the texture path: one divergent filter-word load per key (tools/ubench_bloom_probe.hip: ~205 G/s against 1-2 MiB in L2),
// beside the key streaming; the survivors' stores are free - provided the filters in use stay in L2, which is what the order the
// chunks are taken in is about (below).  (Rounds of this kernel per 312M keys at the 1-rank / 8-rank plan: a workgroup per chunk,
// one key per thread, two barriers per chunk 3.9 ms / -; a wave per chunk, a static strided share per wave 2.3-3.7 / 4.7; XCD x
// takes buckets x, x + 8, ... 1.7-2.0 / 4.4; chunks handed out in order 1.5-2.1 / 2.3.)
template <u32 B>
__global__ __launch_bounds__(256) void fj_part_filter_inplace(u64* keys, u32* list, const u32* __restrict__ boff, u32 nb,
                                                              const u64* __restrict__ filters, u32 part_shift, unsigned long long* __restrict__ kept,
                                                              u32* __restrict__ next_of_xcd) {
    __shared__ u32 s_next;
    const u32 lane = threadIdx.x & 63, wave = (u32)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // XCD x (workgroup b runs on XCD b % 8: tools/ubench_xcc_map.hip) takes buckets x, x + 8, ... in order, and its workgroups take
    // that list's chunks B at a time in the order they become free (one counter per XCD; an agent-scope atomic on a hot address
    // takes ~100 ns: one per chunk made the kernel 15 ms): the chunks in flight on an XCD stay within one or two buckets, whose
    // filters (2 MiB per bucket at 512 partitions) stay in that XCD's L2.  A static share per wave - strided over the whole list -
    // had every wave in another bucket after each step: 4.4-4.7 ms per 312M keys at the 8-rank plan; 8 workgroups per CU taking
    // batches of 32 from one counter per XCD 3.6 ms; 3 per CU and batches of 16 (at most 1536 chunks in flight per XCD, 2384 per
    // bucket) 2.5 ms; 4 per CU and batches of 4 from four interleaved counters per XCD 2.3 ms (a plateau: 2.3-2.5 ms for 2-6
    // workgroups per CU and batches of 4-16).
    const u32 xcd = blockIdx.x & 7u;
    const u32 nbx = nb > xcd ? (nb - xcd + 7u) >> 3 : 0u;                        // <= 64 buckets per XCD (FJ_MAX_FAN_LOG = 9): one per lane
    const u32 bstart = lane < nbx ? boff[xcd + 8u * lane] : 0u;
    const u32 bcnt = lane < nbx ? boff[xcd + 8u * lane + 1u] - bstart : 0u;
    u32 inc = bcnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const u32 y = __shfl_up(inc, d, 64); if ((int)lane >= d) inc += y; }
    const u32 pre = inc - bcnt, total = __shfl(inc, 63, 64);                     // this XCD's chunks, in bucket order: ordinal o -> list index
    // (K counters per XCD, a workgroup uses one of them: batch t of counter k covers ordinals (t * K + k) * B ..., so the counters
    //  advance together and an agent-scope atomic on a hot address - ~100 ns - is paid once per K * B chunks of the XCD)
    constexpr u32 K = FJ_PF_COUNTERS;
    const u32 kk = (blockIdx.x >> 3) % K;
    u32* ctr = next_of_xcd + xcd * K + kk;
    auto locate = [&](u32 o) -> u32 {
        const u32 j = (u32)__popcll(__ballot(lane < nbx && pre <= o)) - 1u;      // (o < total: the last bucket that starts at or before o holds it)
        return (u32)__shfl((int)bstart, (int)j, 64) + (o - (u32)__shfl((int)pre, (int)j, 64));
    };
    const u64 lt = (1ull << lane) - 1ull;
    unsigned long long mine = 0;
    if (threadIdx.x == 0) s_next = (atomicAdd(ctr, 1u) * K + kk) * B;
    __syncthreads();
    u32 o0 = s_next;
    while (o0 < total) {
        __syncthreads();                                                         // (everyone has read s_next)
        u32 nxt = 0;
        if (threadIdx.x == 0) nxt = (atomicAdd(ctr, 1u) * K + kk) * B;           // the next batch's ordinal is on its way while this one is worked on
        // this wave's chunks of the batch: o0 + wave, + 4, ...; software pipeline: the NEXT chunk's keys are requested before this
        // chunk's filter words are waited for (an ordinal beyond the end is clamped: loaded, never used)
        const u32 oend = o0 + B < total ? o0 + B : total;
        u32 o = o0 + wave;
        if (o < oend) {
            u32 li = locate(o);
            u32 e = list[li];
            const uint4* c4 = reinterpret_cast<const uint4*>(keys + (u64)FJ_LIST_ID(e) * FJ_CHUNK);
            uint4 a = c4[2 * lane], c = c4[2 * lane + 1];                        // keys 4 * lane .. 4 * lane + 3 (slack beyond the count: never counted)
            for (; o < oend; o += 4) {
                const u32 li1 = locate(o + 4 < oend ? o + 4 : o);
                const u32 e1 = list[li1];
                const uint4* n4 = reinterpret_cast<const uint4*>(keys + (u64)FJ_LIST_ID(e1) * FJ_CHUNK);
                const uint4 an = n4[2 * lane], cn = n4[2 * lane + 1];
                const u32 cnt = FJ_LIST_CNT(e), id = FJ_LIST_ID(e);
                const u64 h[4] = {(u64)a.y << 32 | a.x, (u64)a.w << 32 | a.z, (u64)c.y << 32 | c.x, (u64)c.w << 32 | c.z};
                u64 word[4], m[4];
#pragma unroll
                for (u32 j = 0; j < 4; ++j) {
                    u32 b;
                    pf_bits(h[j], b, m[j]);
                    const u64 part = 4 * lane + j < cnt ? h[j] >> part_shift : 0ull;     // (slack lanes read partition 0's filter: unconditional loads)
                    word[j] = filters[part * (FJ_PFILT_BYTES / 8) + b];
                }
                u64* ck = keys + (u64)id * FJ_CHUNK;
                u32 base = 0;
#pragma unroll
                for (u32 j = 0; j < 4; ++j) {
                    const bool pass = (word[j] & m[j]) == m[j] && 4 * lane + j < cnt;
                    const u64 bal = __ballot(pass);
                    if (pass) ck[base + (u32)__popcll(bal & lt)] = h[j];
                    base += (u32)__popcll(bal);
                }
                if (lane == 0) list[li] = ((base ? base - 1u : 0u) << 24) | id;   // (no survivor: key 0 stays - its slot was not written)
                mine += base;
                e = e1; li = li1; a = an; c = cn;
            }
        }
        if (threadIdx.x == 0) s_next = nxt;
        __syncthreads();
        o0 = s_next;
    }
    if (lane == 0 && mine) atomicAdd(kept, mine);
}

// how many of n raw probe keys (a strided sample) would pass: the "auto" decision
__global__ __launch_bounds__(256) void fj_part_filter_sample(const u64* __restrict__ raw, u64 n, u64 stride, const u64* __restrict__ filters, u32 part_shift,
                                                             unsigned long long* __restrict__ kept) {
    const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    bool pass = false;
    if (i < n) {
        const u64 h = fj_key_mix(raw[i * stride]);
        u32 b; u64 m;
        pf_bits(h, b, m);
        pass = (filters[(h >> part_shift) * (FJ_PFILT_BYTES / 8) + b] & m) == m;
    }
    const u64 bal = __ballot(pass);
    if ((threadIdx.x & 63) == 0 && bal) atomicAdd(kept, (unsigned long long)__popcll(bal));
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

hipError_t fj_launch_part_filter_export(const FjChunkSet& build, u64* out, u32 grid, hipStream_t s) {
    if (!build.list || build.nb == 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(fj_part_filter_export, dim3(grid < build.nb ? grid : build.nb), dim3(256), 0, s, build, out);
    return hipGetLastError();
}

hipError_t fj_launch_part_filter_inplace(const FjChunkSet& cs, const u64* filters, u32 part_shift, unsigned long long* kept, u32* next_of_xcd, u32 grid, hipStream_t s) {
    grid = (grid + 7u) & ~7u;                                  // (the same number of workgroups on every XCD)
    hipLaunchKernelGGL(fj_part_filter_inplace<4>, dim3(grid), dim3(256), 0, s, cs.keys, cs.list, cs.boff, cs.nb, filters, part_shift, kept, next_of_xcd);
    return hipGetLastError();
}

hipError_t fj_launch_part_filter_sample(const u64* raw, u64 n, u64 stride, const u64* filters, u32 part_shift, unsigned long long* kept, hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(fj_part_filter_sample, dim3((u32)((n + 255) / 256)), dim3(256), 0, s, raw, n, stride, filters, part_shift, kept);
    return hipGetLastError();
}
