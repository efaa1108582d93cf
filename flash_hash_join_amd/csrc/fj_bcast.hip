// fj_bcast.hip -- the multi-GPU counting join in its BUILD-BROADCAST form: what one rank does to its own rows.
//
// No reference counterpart: the reference is one process (hash_join.cpp:318).  What is exploited is that radix partitions are
// independent join units (hash_join.cpp:340-356, :515-525) and that a final partition's build keys share their top radix bits.
// Every rank plans for the TOTAL build side (the same 2^bits final partitions everywhere), runs BOTH passes of that plan over
// its own build rows and writes them out densely, partition after partition, in a wire format that drops the bits the
// partition implies (fj_bcast_pack):
//     region = offset table u32[nparts + 1] (keys before partition p) | low words u32[n] | the low (32 - bits) bits of the high
//              words as u16[n] (bits >= 16) or u32[n]
// - 6 bytes per key for build sides of 134M rows and more, exact (no chunk padding).  The regions of all ranks are exchanged
// (every rank sends its region to every peer: csrc/fj_dist.hip, in pieces of consecutive partitions), the probe rows NEVER
// move: every rank partitions its own probe rows with the same plan (fj_bcast_probe) and joins them against the runs of all
// ranks, partition range by partition range as the ranges land (fj_bcast_join -> fj_count_join_wide<DENSE>, which reads the
// runs where they lie: the receiver neither re-partitions nor copies anything).
// Against the owner shuffle (7 bytes per probe AND build key across the links, a pack copy and a pass over wire chunks on the
// receiving side) a rank puts 6 * nb_local bytes on every link however many probe rows there are: at BASELINE configs[4]
// (125M x 1.25B rows per GPU, 8 GPUs) 0.75 GB per link instead of 1.2 GB, and no kernel beyond the plain join's but the pack.
#include "fj_host.h"
#include <type_traits>
using namespace fjh;

namespace {

constexpr u32 DP_NT = 256;

// keys per final partition, from the final level's chunk lists (entry = (count - 1) << 24 | id)
__global__ __launch_bounds__(DP_NT) void fj_dense_count(const u32* __restrict__ boff, const u32* __restrict__ list, u32 nparts, u32* __restrict__ cnt) {
    const u32 p = blockIdx.x * DP_NT + threadIdx.x;
    if (p >= nparts) return;
    u32 n = 0;
    for (u32 i = boff[p]; i < boff[p + 1]; ++i) n += FJ_LIST_CNT(list[i]);
    cnt[p] = n;
}

// in-place exclusive scan of cnt[0 .. nparts) (one 1024-thread workgroup; nparts <= 2^22), cnt[nparts] = total; the key index at
// which piece q of `pieces` pieces of consecutive partitions starts goes to bounds[q] (bounds[pieces] = total).  16384 entries per
// sweep (four 16-byte loads per thread in flight: 262144 entries in 16 sweeps; 4096 per sweep took 150 us)
__global__ __launch_bounds__(1024) void fj_dense_scan(u32* __restrict__ cnt, u32 nparts, u32 pieces, u32* __restrict__ bounds) {
    __shared__ u32 wsum[4][16];
    const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    u32 carry = 0;
    for (u32 base = 0; base < nparts; base += 16384) {
        uint4 x[4]; u32 sum[4], inc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const u32 e0 = base + 4u * ((u32)j * 1024u + tid);
            x[j] = make_uint4(0, 0, 0, 0);
            if (e0 + 3 < nparts) x[j] = *reinterpret_cast<const uint4*>(cnt + e0);
            else { if (e0 < nparts) x[j].x = cnt[e0]; if (e0 + 1 < nparts) x[j].y = cnt[e0 + 1]; if (e0 + 2 < nparts) x[j].z = cnt[e0 + 2]; }
            sum[j] = x[j].x + x[j].y + x[j].z + x[j].w;
            u32 v = sum[j];
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const u32 y = __shfl_up(v, d, 64); if ((int)lane >= d) v += y; }
            inc[j] = v;
            if (lane == 63) wsum[j][wave] = v;
        }
        __syncthreads();
        u32 total = carry;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            u32 mine = 0, all = 0;
            for (u32 v = 0; v < 16; ++v) { const u32 c = wsum[j][v]; all += c; if (v < wave) mine += c; }
            u32 run = total + mine + inc[j] - sum[j];
            const u32 e0 = base + 4u * ((u32)j * 1024u + tid);
            if (e0 < nparts) cnt[e0] = run; run += x[j].x;
            if (e0 + 1 < nparts) cnt[e0 + 1] = run; run += x[j].y;
            if (e0 + 2 < nparts) cnt[e0 + 2] = run; run += x[j].z;
            if (e0 + 3 < nparts) cnt[e0 + 3] = run;
            total += all;
        }
        carry = total;
        __syncthreads();
    }
    if (tid == 0) cnt[nparts] = carry;
    __threadfence_block();
    __syncthreads();
    if (tid <= pieces) bounds[tid] = tid == pieces ? carry : cnt[(u32)(((u64)nparts * tid) / pieces)];
}

// the same scan for 8192 .. 2^19 partitions (a power of two), out of place, one workgroup per 4096 counts: each sums the counts in
// front of its piece itself (<= 2 MiB, in L2) and scans its piece - 262144 partitions in ~10 us instead of 150 on one CU
__global__ __launch_bounds__(1024) void fj_dense_scan_wide(const u32* __restrict__ cnt, u32* __restrict__ offs, u32 nparts, u32 pieces, u32* __restrict__ bounds) {
    __shared__ u32 wtot[2][16];
    const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = blockIdx.x;
    const u32 e0 = g * 4096u + 4u * tid;
    const uint4 x = *reinterpret_cast<const uint4*>(cnt + e0);
    u32 pre = 0;
#pragma unroll 4
    for (u32 j = 0; j < g; ++j) { const uint4 y = *reinterpret_cast<const uint4*>(cnt + 4u * (j * 1024u + tid)); pre += y.x + y.y + y.z + y.w; }
    const u32 own = x.x + x.y + x.z + x.w;
    u32 inc = own;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const u32 y = __shfl_up(inc, d, 64), z = __shfl_up(pre, d, 64);
        if ((int)lane >= d) { inc += y; pre += z; }
    }
    if (lane == 63) { wtot[0][wave] = inc; wtot[1][wave] = pre; }
    __syncthreads();
    u32 run = inc - own;
    for (u32 w = 0; w < 16; ++w) { run += wtot[1][w]; if (w < wave) run += wtot[0][w]; }
    const uint4 o = make_uint4(run, run + x.x, run + x.x + x.y, run + x.x + x.y + x.z);
    *reinterpret_cast<uint4*>(offs + e0) = o;
    const u32 ov[4] = {o.x, o.y, o.z, o.w};
    for (u32 q = 0; q < pieces; ++q) {
        const u32 idx = (u32)(((u64)nparts * q) / pieces);
        if (idx - e0 < 4u) bounds[q] = ov[idx - e0];
    }
    if (e0 + 4u == nparts) { offs[nparts] = run + own; bounds[pieces] = run + own; }
}

// one wave per final partition: its chunks' keys -> the two planes at the partition's offset.  The keys are staged in a per-wave
// LDS tile whose index mirrors the alignment of the output (tile_base is a multiple of 8 keys) and leave as whole 16-byte words
// wherever a word belongs to this partition alone; the partition's first and last few keys - words shared with the neighbouring
// partitions, other waves' business - go out one by one.  (One 4-byte and one 2-byte store per KEY made this copy 0.41 ms per 125M
// keys: the memory pipeline charges per store instruction.)  LDS operations of one wave execute in order: no barrier.
constexpr u32 DC_T = 512;                      // keys per tile (a multiple of 8)
template <int MIDB>
__global__ __launch_bounds__(DP_NT) void fj_dense_copy(const u64* __restrict__ keys, const u32* __restrict__ boff, const u32* __restrict__ list,
                                                      u32 nparts, const u32* __restrict__ offs, u32 midmask, u32* __restrict__ lo, void* __restrict__ mid) {
    typedef typename std::conditional<MIDB == 2, u16, u32>::type mid_t;
    __shared__ __attribute__((aligned(16))) u32 s_lo[DP_NT / 64][DC_T];
    __shared__ __attribute__((aligned(16))) mid_t s_mid[DP_NT / 64][DC_T];
    const u32 wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const u32 p = blockIdx.x * (DP_NT / 64) + wv;
    if (p >= nparts) return;
    u32* tl = s_lo[wv]; mid_t* tm = s_mid[wv];
    mid_t* gm = reinterpret_cast<mid_t*>(mid);
    const u32 o = offs[p], n = offs[p + 1] - o, end = o + n;
    if (n == 0) return;
    constexpr u32 MG = 16 / sizeof(mid_t);                           // high-word elements per 16-byte word
    // the tile [tile_base, tile_base + DC_T) of the output leaves the LDS stage: whole words inside [o, end), single elements at its edges
    auto flush = [&](u32 tile_base) {
        const u32 vlo = o > tile_base ? o : tile_base, vhi = end < tile_base + DC_T ? end : tile_base + DC_T;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        for (u32 g = (vlo - tile_base) / 4 + lane; g * 4 < vhi - tile_base; g += 64) {
            const u32 g0 = tile_base + 4 * g;
            const uint4 q = *reinterpret_cast<const uint4*>(tl + 4 * g);
            if (g0 >= vlo && g0 + 4 <= vhi) *reinterpret_cast<uint4*>(lo + g0) = q;
            else {
                const u32 w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                for (u32 e = 0; e < 4; ++e) if (g0 + e >= vlo && g0 + e < vhi) lo[g0 + e] = w[e];
            }
        }
        for (u32 g = (vlo - tile_base) / MG + lane; g * MG < vhi - tile_base; g += 64) {
            const u32 g0 = tile_base + MG * g;
            const uint4 q = *reinterpret_cast<const uint4*>(tm + MG * g);
            if (g0 >= vlo && g0 + MG <= vhi) *reinterpret_cast<uint4*>(gm + g0) = q;
            else {
#pragma unroll
                for (u32 e = 0; e < MG; ++e) if (g0 + e >= vlo && g0 + e < vhi) gm[g0 + e] = tm[MG * g + e];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    };
    u32 tile_base = o & ~7u, pos = o;                                // pos: output index of the next chunk's first key
    for (u32 i = boff[p]; i < boff[p + 1]; ++i) {
        const u32 e = list[i], cnt = FJ_LIST_CNT(e);
        const u64* ck = keys + (u64)FJ_LIST_ID(e) * FJ_CHUNK;
        u64x2 q[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {                                // (only the lines that hold keys: the chunks of a final partition are ~2/3 full)
            q[t].x = 0; q[t].y = 0;
            if (2 * lane + 128 * t < cnt) q[t] = *reinterpret_cast<const u64x2*>(ck + 2 * lane + 128 * t);     // (chunks are allocated whole: the pair is readable)
        }
        const u64 h[4] = {q[0].x, q[0].y, q[1].x, q[1].y};
        auto stage = [&]() {                                         // this chunk's keys that fall into the current tile
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const u32 k = 2 * lane + 128 * (t >> 1) + (t & 1), at = pos + k - tile_base;      // (pos + k < tile_base wraps to a huge index)
                if (k < cnt && at < DC_T) { tl[at] = FJ_HW2(h[t]); tm[at] = (mid_t)(FJ_HW1(h[t]) & midmask); }
            }
        };
        stage();
        if (pos + cnt >= tile_base + DC_T) {                         // the tile is full (cnt <= 256 < DC_T: at most once per chunk)
            flush(tile_base);
            tile_base += DC_T;
            stage();
        }
        pos += cnt;
    }
    if (pos > tile_base) flush(tile_base);
}


// materialising joins: one wave per final partition copies its chunks' VALUES to the partition's place in the values plane (same
// order as fj_dense_copy wrote the keys: chunk after chunk, key after key)
__global__ __launch_bounds__(DP_NT) void fj_dense_copy_vals(const u64* __restrict__ vals, const u32* __restrict__ boff, const u32* __restrict__ list,
                                                           u32 nparts, const u32* __restrict__ offs, u64* __restrict__ out) {
    const u32 wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const u32 p = blockIdx.x * (DP_NT / 64) + wv;
    if (p >= nparts) return;
    u32 pos = offs[p];
    for (u32 i = boff[p]; i < boff[p + 1]; ++i) {
        const u32 e = list[i], cnt = FJ_LIST_CNT(e);
        const u64* cv = vals + (u64)FJ_LIST_ID(e) * FJ_CHUNK;
        for (u32 k = lane; k < cnt; k += 64) out[pos + k] = cv[k];
        pos += cnt;
    }
}

// ---- the pair writer of a MATERIALISING build-broadcast step (insert_local + probe_vectorized of one radix partition with the build
// rows' values, hash_join.cpp:112-128, :153-182, inside _hash_join_radix_materialize, :315-381).  The step is COUNTED by the counting
// step's kernel (fj_count_join_wide<DENSE>: a probe row counts once however many copies of its key the build side holds - the
// reference's rule, duplicates dropped at insert, :125 - and its per-item counts are what the pairs' offsets are scanned from); that
// kernel is a deep software pipeline without room for values, so the pairs are written by this plain one behind fj_emit_pairs: one
// 1024-thread workgroup per CU walks a run of items; per partition the runs of all sources go into an 8192-slot bucketed table
// (2048 buckets of 4 slots: keys 64 KiB + values 64 KiB of LDS; one returning atomic add on the bucket's fill count per key, a full
// bucket sends the key to the next one), then the item's probe chunks are looked up and every hit writes (probe key un-mixed, build
// value) at out_off[item] + a cursor.  A duplicated build key occupies several slots and a probe row takes the first it meets: ONE
// pair per matching probe row, carrying the value of one of the copies - across GPUs there is no "first" occurrence to prefer.
// The counting launch refuses partitions of more than 24 load units (FjWideArgs::max_units: ~6000 keys), so that what it accepts
// fits this table; should a partition overflow it all the same, the launch says so (FJ_STAT_RETRY) and fj_emit_pairs fails.
constexpr u32 DM_NT = 1024, DM_SLOTS = 8192, DM_BS = 4, DM_NBK = DM_SLOTS / DM_BS, DM_MAXWALK = 256;
struct DenseMatArgs {
    FjChunkSet probe; const uint4* items; const u32* toff; u32 part_lo, part_hi;
    const unsigned char* base; u32 nsrc, bits, mid_bytes;
    u64 offs_off[FJ_WIDE_MAXSRC], lo_off[FJ_WIDE_MAXSRC], mid_off[FJ_WIDE_MAXSRC], val_off[FJ_WIDE_MAXSRC];
    u32* part_count; unsigned long long* total; u32* err;
    const u64* out_off; u64* out_keys; u64* out_vals;
};
__global__ __launch_bounds__(DM_NT) void fj_dense_mat_join(DenseMatArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dm_smem[];
    u64* tkeys = reinterpret_cast<u64*>(dm_smem);
    u64* tvals = tkeys + DM_SLOTS;
    u32* fill = reinterpret_cast<u32*>(tvals + DM_SLOTS);
    u32* sh = fill + DM_NBK;                                       // [1] table overflow (kept while the partition stays), [3] pairs written by the item
    const u32 tid = threadIdx.x, lane = tid & 63u;
    const u32 wave = (u32)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const u32 item_lo = a.toff[a.part_lo], item_hi = a.toff[a.part_hi];
    // a workgroup takes a RUN of consecutive items: the items of one partition (few ranks: several per partition) share one table build
    const u32 nit = item_hi - item_lo;
    const u32 my_lo = item_lo + (u32)((u64)nit * blockIdx.x / gridDim.x), my_hi = item_lo + (u32)((u64)nit * (blockIdx.x + 1u) / gridDim.x);
    if (my_lo >= my_hi) return;
    // every source's run is loaded by its own waves (16 / nsrc rounded up to a power of two: the source, its bounds and its planes are
    // wave-uniform - scalar loads, no per-key search), four keys per lane and round
    u32 wps = DM_NT / 64u;                                         // waves per source
    while (wps > 1u && (DM_NT / 64u) / wps < a.nsrc) wps >>= 1;
    const u32 src = wave / wps, stride = wps * 64u, wofs = (wave % wps) * 64u + lane;
    const bool has_src = src < a.nsrc;
    const u32* offs = reinterpret_cast<const u32*>(a.base + a.offs_off[has_src ? src : 0]);
    const u32* lo = reinterpret_cast<const u32*>(a.base + a.lo_off[has_src ? src : 0]);
    const unsigned char* midp = a.base + a.mid_off[has_src ? src : 0];
    const u64* vp = reinterpret_cast<const u64*>(a.base + a.val_off[has_src ? src : 0]);

    // The item loop is a software pipeline over three levels of dependent loads (one 1024-thread workgroup per CU - the table and the
    // value array take 136 KiB - so nothing else hides them): item descriptors two items ahead; the next item's run bounds and probe
    // list entries requested before this item's table is built; the next item's first round of build rows and its probe keys requested
    // between this item's build and its lookups.  (Plain form, everything loaded where it was needed: 16.3 ms as one rank of 8.)
    auto desc = [&](u32 it) -> uint4 { return it < my_hi ? a.items[it] : make_uint4(0, 0, 0xFFFFFFFFu, 0); };   // {probe list pos, probe chunks, partition, -}
    auto bounds = [&](u32 part, u32& b, u32& e) { b = 0; e = 0; if (has_src && part != 0xFFFFFFFFu) { b = offs[part]; e = offs[part + 1]; } };
    auto build_loads = [&](u32 k0, u32 b, u32 e, u32 (&lw)[4], u32 (&mw)[4], u64 (&vw)[4], u32& okm) {
        okm = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const u32 k = k0 + (u32)j * stride;
            const bool ok = k < e;
            const u32 kk = ok ? k : (b < e ? b : 0u);                // (a readable key of the run; an empty run reads key 0 of the plane - the region is never empty: it has its offset table)
            lw[j] = lo[kk];
            mw[j] = a.mid_bytes == 2 ? (u32)reinterpret_cast<const u16*>(midp)[kk] : reinterpret_cast<const u32*>(midp)[kk];
            vw[j] = vp[kk];
            okm |= (ok ? 1u : 0u) << j;
        }
    };
    auto insert4 = [&](u32 top, const u32 (&lw)[4], const u32 (&mw)[4], const u64 (&vw)[4], u32 okm) {
        u64 key[4]; u32 bk[4], o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {                              // the four returning adds in flight together
            key[j] = ((u64)(top | mw[j]) << 32) | lw[j];
            bk[j] = FJ_HW2(key[j]) & (DM_NBK - 1u);
            o[j] = ((okm >> j) & 1u) ? atomicAdd(&fill[bk[j]], 1u) : 0u;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (!((okm >> j) & 1u)) continue;
            u32 slot = o[j] < DM_BS ? bk[j] * DM_BS + o[j] : 0xFFFFFFFFu;
            if (slot == 0xFFFFFFFFu) {                             // the home bucket was full: on to the next ones
                u32 b2 = bk[j];
                for (u32 step = 1; step < DM_MAXWALK; ++step) {
                    b2 = (b2 + 1u) & (DM_NBK - 1u);
                    const u32 o2 = atomicAdd(&fill[b2], 1u);
                    if (o2 < DM_BS) { slot = b2 * DM_BS + o2; break; }
                }
            }
            if (slot == 0xFFFFFFFFu) sh[1] = 1;
            else { tkeys[slot] = key[j]; tvals[slot] = vw[j]; }
        }
    };
    auto list_loads = [&](const uint4& d, u32 idx0, u32 (&le)[8]) {    // the list entries of the lane's eight key slots idx0 + j * 1024
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const u32 idx = idx0 + (u32)j * DM_NT;
            le[j] = (d.y && d.z != 0xFFFFFFFFu) ? a.probe.list[d.x + (idx < d.y * FJ_CHUNK ? (idx >> FJ_CHUNK_LOG) : 0u)] : 0u;
        }
    };
    auto key_loads = [&](const uint4& d, u32 idx0, const u32 (&le)[8], u64 (&pk)[8], u32& okm) {
        okm = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const u32 idx = idx0 + (u32)j * DM_NT, k = idx & (FJ_CHUNK - 1u);
            const bool ok = d.z != 0xFFFFFFFFu && idx < d.y * FJ_CHUNK && k < FJ_LIST_CNT(le[j]);
            pk[j] = a.probe.keys[(u64)FJ_LIST_ID(le[j]) * FJ_CHUNK + (ok ? k : 0u)];       // (chunk 0 stands in where there is no item)
            okm |= (ok ? 1u : 0u) << j;
        }
    };
    // the lookups of a lane's eight probe keys, four at a time: the home bucket whole (two 16-byte reads) and its fill count, all reads
    // of the four in flight, then the compares; only a key that is not in an OVERFLOWED home bucket walks on (rare).  (First form: per
    // key a loop over the bucket's slots with an early exit - nested divergent loops of dependent LDS reads: 9.7 of the kernel's 16.4
    // ms as one rank of 8, by ablation.)  obase: the item's first output position - read once per item (behind every pair store the
    // compiler would have to read out_off[it] again: the output arrays might alias it)
    auto probe8 = [&](u64 obase, const u64 (&pk)[8], u32 okm) {
#pragma unroll
        for (int h0 = 0; h0 < 8; h0 += 4) {
            uint4 q0[4], q1[4]; u32 fc[4], bkt[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                bkt[i] = FJ_HW2(pk[h0 + i]) & (DM_NBK - 1u);
                const uint4* bp = reinterpret_cast<const uint4*>(tkeys + bkt[i] * DM_BS);
                q0[i] = bp[0]; q1[i] = bp[1];
                fc[i] = fill[bkt[i]];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const u64 key = pk[h0 + i];
                const bool ok = (okm >> (h0 + i)) & 1u;
                const u32 n = fc[i] < DM_BS ? fc[i] : DM_BS;
                const u64 e0 = ((u64)q0[i].y << 32) | q0[i].x, e1 = ((u64)q0[i].w << 32) | q0[i].z, e2 = ((u64)q1[i].y << 32) | q1[i].x, e3 = ((u64)q1[i].w << 32) | q1[i].z;
                const bool m0 = (e0 == key) & (n > 0u), m1 = (e1 == key) & (n > 1u), m2 = (e2 == key) & (n > 2u), m3 = (e3 == key) & (n > 3u);
                u32 hit = m0 ? 0u : m1 ? 1u : m2 ? 2u : m3 ? 3u : 0xFFFFFFFFu;
                if (hit != 0xFFFFFFFFu) hit += bkt[i] * DM_BS;
                if (!ok) hit = 0xFFFFFFFFu;
                bool walk = ok & (hit == 0xFFFFFFFFu) & (fc[i] > DM_BS);      // somebody was sent on from this bucket
                if (__ballot(walk)) {
                    u32 bk = bkt[i];
                    for (u32 step = 1; step < DM_MAXWALK && __ballot(walk); ++step) {
                        if (walk) {
                            bk = (bk + 1u) & (DM_NBK - 1u);
                            const u32 f = fill[bk], nn = f < DM_BS ? f : DM_BS;
                            for (u32 t = 0; t < nn; ++t) if (tkeys[bk * DM_BS + t] == key) { hit = bk * DM_BS + t; break; }
                            walk = hit == 0xFFFFFFFFu && f > DM_BS;
                        }
                    }
                }
                // the item's cursor is bumped once per wave and key slot; the lanes rank themselves inside the ballot
                const bool h = hit != 0xFFFFFFFFu;
                const u64 m = __ballot(h);
                if (m) {
                    u32 wb = 0;
                    if (lane == (u32)__builtin_ctzll(m)) wb = atomicAdd(&sh[3], (u32)__popcll(m));
                    wb = (u32)__builtin_amdgcn_readlane((int)wb, __builtin_ctzll(m));
#ifndef FJ_DM_ABLATE      // (timing-only variants: 1 no pair stores | 2 no lookups | 4 no inserts)
#define FJ_DM_ABLATE 0
#endif
                    if (h && !(FJ_DM_ABLATE & 1)) {
                        const u64 o = obase + wb + (u32)__popcll(m & ((1ull << lane) - 1ull));
                        a.out_keys[o] = fj_key_unmix(key); a.out_vals[o] = tvals[hit];
                    }
                }
            }
        }
    };

    // ---- prologue: item my_lo's loads, level by level ----
    uint4 d = desc(my_lo), dn = desc(my_lo + 1);
    u32 b = 0, e = 0;
    bounds(d.z, b, e);
    u32 le[8];
    list_loads(d, tid, le);
    u32 lw[4], mw[4], bokm = 0; u64 vw[4];
    build_loads(b + wofs, b, e, lw, mw, vw, bokm);
    u64 pk[8]; u32 pokm = 0;
    key_loads(d, tid, le, pk, pokm);
    u32 cur_part = 0xFFFFFFFFu;
    if (tid < 4) sh[tid] = 0;

    for (u32 it = my_lo; it < my_hi; ++it) {
        const u32 part = d.z;
        const bool newpart = part != cur_part;
        // level 1 of the next item: its descriptor is here; bounds of its partition (if another one), its probe list entries; the
        // descriptor after it
        const uint4 dnn = desc(it + 2);
        u32 nb_ = 0, ne_ = 0, len[8];
        if (dn.z != part) bounds(dn.z, nb_, ne_);
        list_loads(dn, tid, len);
        if (tid == 3) sh[3] = 0;
        if (newpart) {
            for (u32 i = tid; i < DM_NBK; i += DM_NT) fill[i] = 0;
            if (tid == 1) sh[1] = 0;
        }
        __syncthreads();
        if (newpart) {                                             // (items of one partition that follow each other in a workgroup share the table)
            const u32 top = a.bits ? part << (32u - a.bits) : 0u;
            if (!(FJ_DM_ABLATE & 4)) insert4(top, lw, mw, vw, bokm);
            for (u32 k0 = b + wofs + 4u * stride; k0 - lane < e && has_src; k0 += 4u * stride) {      // (runs of more than 4 keys per lane: skew; loaded on the spot)
                build_loads(k0, b, e, lw, mw, vw, bokm);
                insert4(top, lw, mw, vw, bokm);
            }
            cur_part = part;
        }
        __syncthreads();
        // level 2 of the next item: the first round of its partition's build rows (if another partition), its probe keys
        u64 pkn[8]; u32 pokn = 0;
        if (dn.z != part) build_loads(nb_ + wofs, nb_, ne_, lw, mw, vw, bokm);
        key_loads(dn, tid, len, pkn, pokn);
        // ---- this item's lookups and pairs ----
        const bool bad = sh[1] != 0;
        const u64 obase = a.out_off[it];
        if (!bad && !(FJ_DM_ABLATE & 2)) {
            probe8(obase, pk, pokm);
            for (u32 idx0 = tid + 8u * DM_NT; idx0 - tid < d.y * FJ_CHUNK; idx0 += 8u * DM_NT) {     // (items of more than 32 chunks: none from fj_bcast_probe; loaded on the spot)
                u32 le2[8]; u64 pk2[8]; u32 ok2;
                list_loads(d, idx0, le2);
                key_loads(d, idx0, le2, pk2, ok2);
                probe8(obase, pk2, ok2);
            }
        }
        if (bad && tid == 0) atomicOr(a.err, FJ_STAT_RETRY);       // (the counting launch accepted a partition this table cannot hold: reported, fj_emit_pairs fails)
        __syncthreads();
        // ---- rotate ----
        if (dn.z != part) { b = nb_; e = ne_; }
        d = dn; dn = dnn;
#pragma unroll
        for (int j = 0; j < 8; ++j) pk[j] = pkn[j];
        pokm = pokn;
    }
}

struct Layout { u32 bits, nparts, mid_bytes; size_t lo_off, mid_off, val_off, bytes; };
// with_vals: the region of a MATERIALISING join carries a fourth part, the build values u64[n] (14 bytes per build row in all)
int layout_of(size_t nb_total, size_t nkeys, Layout* L, bool with_vals = false) {
    const Plan p = make_plan(nb_total, 64);
    if (p.npass < 1 || p.bits < 5) return set_err("build broadcast: a build side of %zu rows in all needs no partitioning (use the owner-scatter form)", nb_total);
    if (nkeys >= (1ull << 32)) return set_err("build broadcast: %zu build rows on one rank (the offset tables are 32-bit)", nkeys);
    L->bits = (u32)p.bits; L->nparts = 1u << p.bits; L->mid_bytes = 32 - p.bits <= 16 ? 2u : 4u;
    L->lo_off = (((size_t)L->nparts + 1) * 4 + 15) & ~(size_t)15;
    L->mid_off = L->lo_off + ((nkeys * 4 + 15) & ~(size_t)15) + 16;
    L->val_off = L->mid_off + ((nkeys * L->mid_bytes + 15) & ~(size_t)15) + 16;
    L->bytes = with_vals ? L->val_off + ((nkeys * 8 + 15) & ~(size_t)15) + 16 : L->val_off;
    return 0;
}

}  // namespace

extern "C" {

int fj_bcast_plan(size_t nb_total, int* bits, uint32_t* nparts, int* mid_bytes) {
    Layout L;
    if (layout_of(nb_total, 0, &L)) return 1;
    if (bits) *bits = (int)L.bits;
    if (nparts) *nparts = L.nparts;
    if (mid_bytes) *mid_bytes = (int)L.mid_bytes;
    return 0;
}

size_t fj_bcast_region_bytes(size_t nb_total, size_t nkeys, int with_vals) {
    Layout L;
    return layout_of(nb_total, nkeys, &L, with_vals != 0) ? 0 : L.bytes;
}

// where piece q's bytes of a region of `nkeys` keys lie: part 0 = the offset table (travels with piece 0), part 1 = low words,
// part 2 = high-word plane, part 3 = values (regions of materialising joins), of the keys [k_lo, k_hi)
int fj_bcast_piece_span(size_t nb_total, size_t nkeys, size_t k_lo, size_t k_hi, int part, size_t* offset, size_t* bytes) {
    Layout L;
    if (layout_of(nb_total, nkeys, &L, true)) return 1;
    if (k_lo > k_hi || k_hi > nkeys || part < 0 || part > 3) return set_err("fj_bcast_piece_span: bad range");
    if (part == 0) { *offset = 0; *bytes = ((size_t)L.nparts + 1) * 4; }
    else if (part == 1) { *offset = L.lo_off + k_lo * 4; *bytes = (k_hi - k_lo) * 4; }
    else if (part == 2) { *offset = L.mid_off + k_lo * L.mid_bytes; *bytes = (k_hi - k_lo) * L.mid_bytes; }
    else { *offset = L.val_off + k_lo * 8; *bytes = (k_hi - k_lo) * 8; }
    return 0;
}

// This rank's build rows -> its region (asynchronous on `stream`; starts the step: the plan's scalars are cleared here).
// pieces: the region will travel in that many pieces of consecutive partitions; fj_bcast_pack_bounds blocks until their key
// boundaries are known.
int fj_bcast_pack(fj_ctx* c, const uint64_t* d_keys, const uint64_t* d_vals, size_t nb, size_t nb_total, void* d_region, int pieces, int with_vals_in, void* stream) {
    if (!c) return set_err("fj_bcast_pack: null context");
    if ((nb && !d_keys) || !d_region || (((uintptr_t)d_keys | (uintptr_t)d_vals | (uintptr_t)d_region) & 15)) return set_err("fj_bcast_pack: null or misaligned pointer");
    if (pieces < 1 || pieces > 16) return set_err("fj_bcast_pack: pieces must be 1..16");
    if (c->st.active) return set_err("fj_bcast_pack: a stream join is open on this context");
    const bool with_vals = with_vals_in != 0;                  // a materialising join: the values travel as a fourth part of the region
    if (with_vals && nb && !d_vals) return set_err("fj_bcast_pack: a materialising join needs the build values");
    Layout L;
    if (layout_of(nb_total, nb, &L, with_vals)) return 1;
    FJ_ENTER(c);
    hipStream_t s = (hipStream_t)stream;
    c->pend.valid = false;
    BcastState& bc = c->bc;
    bc = BcastState();
    bc.nb_total = nb_total; bc.pieces = pieces; bc.plan = make_plan(nb_total, 64); bc.with_vals = with_vals;
    begin_plan(c);
    HIPCHK(hipEventRecord(c->ev[E_START], s));
    if (clear_plan_scalars(c, s)) return 1;
    unsigned char* reg = (unsigned char*)d_region;
    u32* offs = (u32*)reg;
    if (nb == 0) {
        HIPCHK(hipMemsetAsync(offs, 0, ((size_t)L.nparts + 1) * 4, s));
        HIPCHK(hipMemsetAsync(c->d_sc->bc_bounds, 0, sizeof(c->d_sc->bc_bounds), s));
    } else {
        PassIter it;
        pass_init(it, 0, with_vals, nb, bc.plan, 64);
        FjChunkSet cs{};
        if (run_passes(c, it, (const u64*)d_keys, with_vals ? (const u64*)d_vals : nullptr, s, &cs, nullptr)) return 1;
        if (L.nparts >= 8192u && L.nparts <= (1u << 19)) {
            void* tmp = nullptr;                             // (a slot of the owner shuffle's packing pass: idle in this form)
            if (get_buf(c, W_PK_BKEYS, (size_t)L.nparts * 4, &tmp)) return 1;
            hipLaunchKernelGGL(fj_dense_count, dim3((L.nparts + DP_NT - 1) / DP_NT), dim3(DP_NT), 0, s, cs.boff, cs.list, L.nparts, (u32*)tmp);
            hipLaunchKernelGGL(fj_dense_scan_wide, dim3(L.nparts / 4096u), dim3(1024), 0, s, (const u32*)tmp, offs, L.nparts, (u32)pieces, c->d_sc->bc_bounds);
        } else {
            hipLaunchKernelGGL(fj_dense_count, dim3((L.nparts + DP_NT - 1) / DP_NT), dim3(DP_NT), 0, s, cs.boff, cs.list, L.nparts, offs);
            hipLaunchKernelGGL(fj_dense_scan, dim3(1), dim3(1024), 0, s, offs, L.nparts, (u32)pieces, c->d_sc->bc_bounds);
        }
        const u32 midmask = L.bits ? (L.bits >= 32 ? 0u : (0xFFFFFFFFu >> L.bits)) : 0xFFFFFFFFu;
        const u32 grid = (L.nparts + DP_NT / 64 - 1) / (DP_NT / 64);
        if (L.mid_bytes == 2) hipLaunchKernelGGL(fj_dense_copy<2>, dim3(grid), dim3(DP_NT), 0, s, cs.keys, cs.boff, cs.list, L.nparts, offs, midmask, (u32*)(reg + L.lo_off), (void*)(reg + L.mid_off));
        else hipLaunchKernelGGL(fj_dense_copy<4>, dim3(grid), dim3(DP_NT), 0, s, cs.keys, cs.boff, cs.list, L.nparts, offs, midmask, (u32*)(reg + L.lo_off), (void*)(reg + L.mid_off));
        if (with_vals) hipLaunchKernelGGL(fj_dense_copy_vals, dim3(grid), dim3(DP_NT), 0, s, cs.vals, cs.boff, cs.list, L.nparts, offs, (u64*)(reg + L.val_off));
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipMemcpyAsync(c->pk_h, c->d_sc->bc_bounds, (size_t)(pieces + 1) * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(c->pk_h + 32, &c->d_sc->err, 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipEventRecord(c->pk_ev, s));
    HIPCHK(hipEventRecord(c->ev[E_BUILD], s));
    bc.packed = true; bc.nb = nb;
    return 0;
}

// blocks until the pack has run: h_bounds[q] = index of the first key of piece q in this rank's region, h_bounds[pieces] = nb
int fj_bcast_pack_bounds(fj_ctx* c, uint64_t* h_bounds) {
    if (!c || !h_bounds) return set_err("fj_bcast_pack_bounds: null argument");
    if (!c->bc.packed) return set_err("fj_bcast_pack_bounds: no pack in flight");
    FJ_ENTER(c);
    HIPCHK(hipEventSynchronize(c->pk_ev));
    const u32* b = reinterpret_cast<const u32*>(c->pk_h);
    const u32 err = *reinterpret_cast<const u32*>(c->pk_h + 32);
    if (err & FJ_ERR_POOL) return set_err("internal error: chunk pool exhausted while partitioning the build side for the broadcast");
    for (int q = 0; q <= c->bc.pieces; ++q) h_bounds[q] = b[q];
    if (h_bounds[c->bc.pieces] != c->bc.nb) return set_err("internal error: the broadcast region holds %llu of %zu build rows", (unsigned long long)h_bounds[c->bc.pieces], c->bc.nb);
    return 0;
}

// This rank's probe rows through both passes of the global plan (asynchronous); they stay where they are.
int fj_bcast_probe(fj_ctx* c, const uint64_t* d_pk, size_t np, size_t nb_total, void* stream) {
    if (!c) return set_err("fj_bcast_probe: null context");
    if ((np && !d_pk) || ((uintptr_t)d_pk & 15)) return set_err("fj_bcast_probe: null or misaligned pointer");
    BcastState& bc = c->bc;
    if (!bc.packed || bc.nb_total != nb_total) return set_err("fj_bcast_probe: fj_bcast_pack of the same step comes first");
    FJ_ENTER(c);
    hipStream_t s = (hipStream_t)stream;
    bc.np = np;
    if (np) {
        pass_init(bc.pit, 1, false, np, bc.plan, 64);
        bc.pit.want_items = true; bc.pit.item_tc_max = 32;
        bc.ja = FjLdsJoinArgs();
        if (run_passes(c, bc.pit, (const u64*)d_pk, nullptr, s, &bc.ja.probe, &bc.evc)) return 1;
        bc.ja.items = bc.pit.tiles; bc.ja.nitems_dev = bc.pit.ntiles; bc.ja.items_cap = bc.pit.items_cap; bc.ja.part_count = bc.pit.part_count;
        bc.ja.nparts = bc.ja.probe.nb; bc.ja.nsplit = 1;
        bc.ja.total = &c->d_sc->total; bc.ja.err = &c->d_sc->err;
    }
    HIPCHK(hipEventRecord(c->ev[E_PPART], s));
    bc.probed = true;
    return 0;
}

// join the probe rows of partitions [part_lo, part_hi) against the runs of nsrc sources: source i's region (of nkeys[i] keys,
// laid out by fj_bcast_pack on its rank) starts region_off[i] bytes into d_base
static int dense_mat_args(fj_ctx* c, const void* d_base, int nsrc, const uint64_t* region_off, const uint64_t* nkeys, DenseMatArgs* a) {
    BcastState& bc = c->bc;
    Layout L0;
    if (layout_of(bc.nb_total, 0, &L0)) return 1;
    *a = DenseMatArgs();
    a->probe = bc.ja.probe; a->items = bc.ja.items; a->toff = bc.pit.toff;
    a->base = (const unsigned char*)d_base; a->nsrc = (u32)nsrc; a->bits = L0.bits; a->mid_bytes = L0.mid_bytes;
    for (int i = 0; i < nsrc; ++i) {
        Layout L;
        if (layout_of(bc.nb_total, (size_t)nkeys[i], &L, true)) return 1;
        if (region_off[i] & 15) return set_err("fj_bcast_join: region offsets must be multiples of 16");
        a->offs_off[i] = region_off[i]; a->lo_off[i] = region_off[i] + L.lo_off; a->mid_off[i] = region_off[i] + L.mid_off; a->val_off[i] = region_off[i] + L.val_off;
    }
    a->part_count = bc.ja.part_count; a->total = &c->d_sc->total; a->err = &c->d_sc->err;
    return 0;
}
static u32 dense_mat_lds() { return DM_SLOTS * 8 * 2 + DM_NBK * 4 + 16; }

int fj_bcast_join(fj_ctx* c, const void* d_base, int nsrc, const uint64_t* region_off, const uint64_t* nkeys, uint32_t part_lo, uint32_t part_hi, void* stream) {
    if (!c) return set_err("fj_bcast_join: null context");
    BcastState& bc = c->bc;
    if (!bc.probed) return set_err("fj_bcast_join: fj_bcast_probe of the same step comes first");
    if (!d_base || nsrc < 1 || nsrc > (int)FJ_WIDE_MAXSRC || !region_off || !nkeys) return set_err("fj_bcast_join: 1..%u sources", FJ_WIDE_MAXSRC);
    Layout L0;
    if (layout_of(bc.nb_total, 0, &L0)) return 1;
    if (part_lo > part_hi || part_hi > L0.nparts) return set_err("fj_bcast_join: partitions [%u, %u) of %u", part_lo, part_hi, L0.nparts);
    if (bc.np == 0 || part_lo == part_hi) return 0;
    FJ_ENTER(c);
    const u32 cus = c->reserve_cus < c->num_cus ? c->num_cus - c->reserve_cus : 1u;
    if (bc.with_vals) {      // a materialising step is counted like a counting one; fj_emit_pairs needs to find the regions again (they stay where they are)
        bc.mat_base = d_base; bc.mat_nsrc = nsrc;
        for (int i = 0; i < nsrc; ++i) { bc.mat_off[i] = region_off[i]; bc.mat_nk[i] = nkeys[i]; }
    }
    FjWideArgs w{};
    w.toff = bc.pit.toff; w.part_lo = part_lo; w.part_hi = part_hi;
    w.base = (const unsigned char*)d_base; w.nsrc = (u32)nsrc; w.bits = L0.bits; w.mid_bytes = L0.mid_bytes; w.pmask = fj_wide_pmask((int)L0.bits, 64);
    w.max_units = bc.with_vals ? 24u : 0u;                   // (what is counted for a materialising step must fit the pair writer's 8192-slot table)
    // few ranks = few, fat partitions: > ~28 probe chunks each means two or three items per partition (items hold <= 32 probe chunks),
    // dealt in runs of 8 so that a partition's items find their table built (2 and 4 ranks of config 5: 3 and 2 items per partition)
    w.group_log = bc.np / L0.nparts > 7000 ? 3u : 0u;
    for (int i = 0; i < nsrc; ++i) {
        Layout L;
        if (layout_of(bc.nb_total, (size_t)nkeys[i], &L, bc.with_vals)) return 1;
        if (region_off[i] & 15) return set_err("fj_bcast_join: region offsets must be multiples of 16");
        w.offs_off[i] = region_off[i]; w.lo_off[i] = region_off[i] + L.lo_off; w.mid_off[i] = region_off[i] + L.mid_off;
    }
    HIPCHK(fj_launch_count_join_wide(bc.ja, w, true, cus, (hipStream_t)stream));
    return 0;
}

// the pairs of the materialising step that fj_bcast_finish just counted: (probe key, build value) of this rank's probe rows, out_capacity >= the count
int fj_bcast_emit(fj_ctx* c, uint64_t* d_out_keys, uint64_t* d_out_vals, size_t out_capacity, void* stream) {
    if (!c) return set_err("fj_bcast_emit: null context");
    BcastState& bc = c->bc;
    if (!bc.mat_ready) return set_err("fj_emit_pairs: no materialising build-broadcast step is pending on this context");
    if (bc.mat_count > out_capacity) return set_err("fj_emit_pairs: output capacity %zu < %llu pairs", out_capacity, (unsigned long long)bc.mat_count);
    FJ_ENTER(c);
    hipStream_t s = (hipStream_t)stream;
    bc.mat_ready = false;
    if (bc.mat_count == 0) return 0;
    if (((uintptr_t)d_out_keys | (uintptr_t)d_out_vals) & 7) return set_err("output buffers must be 8-byte aligned");
    void* p;
    if (get_buf(c, W_OUT_OFF, ((size_t)bc.mat_items + 1) * 8, &p)) return 1;
    HIPCHK(fj_launch_scan_u32_to_u64(bc.ja.part_count, (u64*)p, bc.mat_items, s));
    DenseMatArgs a;
    if (dense_mat_args(c, bc.mat_base, bc.mat_nsrc, bc.mat_off, bc.mat_nk, &a)) return 1;
    Layout L0;
    if (layout_of(bc.nb_total, 0, &L0)) return 1;
    a.part_lo = 0; a.part_hi = L0.nparts;
    a.out_off = (const u64*)p; a.out_keys = d_out_keys; a.out_vals = d_out_vals;
    HIPCHK(hipMemsetAsync(&c->d_sc->err, 0, 4, s));
    HIPCHK(fj_set_max_lds_once(reinterpret_cast<const void*>(fj_dense_mat_join), dense_mat_lds()));
    hipLaunchKernelGGL(fj_dense_mat_join, dim3(c->num_cus), dim3(DM_NT), dense_mat_lds(), s, a);
    HIPCHK(hipGetLastError());
    if (read_scalars(c, s)) return 1;
    if (c->h_sc->err & FJ_STAT_RETRY) return set_err("internal error: a partition the counting kernel accepted does not fit the pair writer's table (fj_bcast_emit)");
    return 0;
}

int fj_bcast_finish(fj_ctx* c, void* stream, uint64_t* out_count, fj_timings* timings) {
    if (!c) return set_err("fj_bcast_finish: null context");
    BcastState& bc = c->bc;
    if (!bc.probed) return set_err("fj_bcast_finish: no broadcast join in flight");
    FJ_ENTER(c);
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(hipEventRecord(c->ev[E_JOIN], s));
    if (read_scalars(c, s)) return 1;
    const u32 err = c->h_sc->err;
    const int npass = bc.plan.npass, bits = bc.plan.bits, evc = bc.evc;
    const bool mat = bc.with_vals;
    if (!mat || (err & (FJ_ERR_POOL | FJ_STAT_RETRY | FJ_ERR_LDS_FULL))) bc = BcastState();
    else { bc.packed = bc.probed = false; bc.mat_ready = true; bc.mat_count = c->h_sc->total; bc.mat_items = bc.pit.items_cap; }      // (the emit reads the regions and the probe partitions where they lie)
    if (err & FJ_ERR_POOL) return set_err("internal error: chunk pool exhausted during a partition pass");
    end_plan(c);
    // a final partition beyond the LDS table (> ~8000 build keys in all: build-side skew): the caller takes another form
    if (err & (FJ_STAT_RETRY | FJ_ERR_LDS_FULL)) return set_err("build broadcast: a final partition does not fit the LDS table (skewed build keys)");
    if (out_count) *out_count = c->h_sc->total;
    fj_timings t; memset(&t, 0, sizeof t);
    t.path = 0; t.passes = npass; t.radix_bits = bits; t.partitions = 1ull << bits; t.sampled_hit_bp = -1;
    for (int i = 0; i < evc && i < 4; ++i) t.probe_part_kernel_ms[i] = ev_ms(c, E_PK0 + 2 * i, E_PK0 + 2 * i + 1);
    t.build_phase_ms = ev_ms(c, E_START, E_BUILD); t.join_ms = ev_ms(c, E_PPART, E_JOIN);
    t.probe_phase_ms = ev_ms(c, E_BUILD, E_JOIN); t.total_ms = ev_ms(c, E_START, E_JOIN);
    if (timings) *timings = t;
    last_timings() = t;
    return 0;
}

void fj_bcast_abort(fj_ctx* c) { if (c) { c->bc = BcastState(); } }

}  // extern "C"
