// fj_join.hip -- build + probe kernels for MI355X (gfx950).
//
// Per final radix partition -- work item = a tile of the partition's probe chunk list (several slices for few partitions
// or a partition swollen by a hot key, each slice rebuilding the small table) --, mirroring insert_local +
// probe_vectorized of the reference (hash_join.cpp:112-128, :153-182) inside _hash_join_radix_{count,materialize}
// (:315-381, :498-534):
//  * fj_count_join_kernel      : counting joins.  Cuckoo table of bare keys in LDS: a lookup is two independent 8-byte
//    reads and two v_cmp_eq_u64, hit masks and the count live in SGPRs.  One workgroup per item.
//  * fj_count_join_persistent  : the same table and lookups for plans with many items: two resident workgroups per CU
//    take items from a global counter and prefetch the next item's chunk lists and build keys.
//  * fj_lds_join_kernel<MAT>   : materialising joins (a value must be fetched).  Open addressing with two candidate
//    4-slot groups per key, one-byte tags, slot claims by a 32-bit LDS atomic, linear probing as the overflow path.
//    Duplicate build keys: optional row-index dedup so the FIRST occurrence's value is emitted (hash_join.cpp:125).
// Both were shaped by measurements (in-kernel stamps, PMC, ablations): plain linear probing with
// 64-bit ds_cmpst was issue-, LDS-bank-conflict- and atomic-bound in turn.
//  * fj_gt_*                   : non-partitioned table in HBM / Infinity Cache (hash_join.cpp:130-151 insert_concurrent,
//    :383-496 / :536-567 scalar drivers), 8-key (64-B) groups, linear probing over groups, optional bloom word per
//    group (role of the reference's bloom directory, :183-189).
//  * fj_owner_*                : multi-GPU owner split.      * fj_gen_* : synthetic relations.
//
// Materialisation is two-pass like the reference's small-table strategy (hash_join.cpp:394-444): count per work item
// -> exclusive scan -> re-probe and write at exact offsets, so the output arrays have exactly `count` rows and no
// global cursor is contended.
#include "fj_internal.h"

namespace {

constexpr u32 S = FJ_LDS_SLOTS;

struct JoinHdr {              // small scalars at the front of the dynamic LDS block
    u32 cnt, has_empty, cursor, claimed, full, ovf;
    u32 pad[2];
    u64 empty_val;
    u64 pad2;
};

// ---- LDS table: open addressing, two candidate 4-slot groups per key ---------------------------
// With plain linear probing at load ~0.4 a few percent of the lanes must walk past their home
// group, which means almost every 64-lane wave pays the divergent walk (measured: ~160 issued
// instructions per 64 lookups).  So a key may live in either of two groups g1(k), g2(k) (it is
// inserted into the emptier one); a lookup always reads exactly those 8 slots, branch-free.  Only if
// both groups of some key were full at insert time does the table fall back to linear probing from
// g1 for that key, and it raises a per-table flag so lookups know they may have to walk (rare: the
// host's plan keeps the load at ~0.4, where a double-full pair is a ~1e-4 event per key).
constexpr u32 NGRP = S / FJ_LDS_GROUP;
__device__ __forceinline__ void lds_groups(u64 key, u32& g1, u32& g2) {
    const u32 w = FJ_HW2(key);
    g1 = (w & (NGRP - 1)) * FJ_LDS_GROUP;
    g2 = ((w >> 11) & (NGRP - 1)) * FJ_LDS_GROUP;
}

__device__ __forceinline__ u32 lds_tag(u32 w2) { const u32 t = w2 >> 24; return t ? t : 1u; }
__device__ __forceinline__ u32 tag_zero_bytes(u32 t) {                   // bit 7 of every zero byte, exact
    const u32 y = (t & 0x7F7F7F7Fu) + 0x7F7F7F7Fu;
    return ~(y | t | 0x7F7F7F7Fu);
}
__device__ __forceinline__ u32 tag_matches(u32 tags, u32 pattern) {      // bit 7 of every byte that equals the tag, EXACT:
    return tag_zero_bytes(tags ^ pattern);                                  // an empty slot (tag 0, stale key bytes) must never be a candidate
}


constexpr u32 LDS_MAX_WALK = 128;                 // groups (512 slots) an insert's linear-probing overflow walk may visit
template <bool MAT>
__device__ __forceinline__ bool lds_insert(u64* __restrict__ tkeys, u64* __restrict__ tvals, u32* __restrict__ ttags,
                                        u32* __restrict__ gcnt, JoinHdr* hdr, u64 key, u64 val, bool dedup) {
    // Tag-guided insert.  A slot is claimed with ONE 32-bit returning LDS atomic on the group's fill
    // counter (two 16-bit counters per word) -- 64-bit ds_cmpst was measured to dominate the build
    // phase -- then key, value and tag are plain stores.  Duplicates already visible are dropped via
    // the tags (hash_join.cpp:125); two racing copies of one key may both be stored, which no lookup
    // can observe (a lookup only asks whether ANY slot matches).
    unsigned char* ttag8 = reinterpret_cast<unsigned char*>(ttags);
    const u32 w = FJ_HW2(key);
    const u32 g1 = w & (NGRP - 1), g2 = (w >> 11) & (NGRP - 1);            // group indices
    const u32 tag = lds_tag(w), pat = tag * 0x01010101u;
    const u32 t1 = ttags[g1], t2 = ttags[g2];
    u32 m1 = tag_matches(t1, pat), m2 = tag_matches(t2, pat);
    // (dedup: the "value" is the original row index and the smallest one must win -> ds_min_u64 on the existing copy)
    while (m1) { const u32 c = g1 * FJ_LDS_GROUP + ((u32)__builtin_ctz(m1) >> 3);
                 if (tkeys[c] == key) { if (MAT && dedup) atomicMin((unsigned long long*)&tvals[c], (unsigned long long)val); return false; } m1 &= m1 - 1; }
    while (m2) { const u32 c = g2 * FJ_LDS_GROUP + ((u32)__builtin_ctz(m2) >> 3);
                 if (tkeys[c] == key) { if (MAT && dedup) atomicMin((unsigned long long*)&tvals[c], (unsigned long long)val); return false; } m2 &= m2 - 1; }
    // emptier group first, then the other one, then walk forward from g1 (linear-probing overflow)
    const u32 e1 = (u32)__popc(tag_zero_bytes(t1)), e2 = (u32)__popc(tag_zero_bytes(t2));
    u32 g = e1 >= e2 ? g1 : g2;
    const u32 galt = e1 >= e2 ? g2 : g1;
    // The overflow walk gives up after LDS_MAX_WALK groups: a table that full (load > ~0.95) is reported like an overflowing
    // one - the caller redoes the partition another way - instead of being squeezed in by walks over the whole table (a
    // partition of 30K keys spent 0.6 ms per item filling the table to the last slot before it failed).
#pragma unroll 1
    for (u32 step = 0; step < LDS_MAX_WALK + 2; ++step) {
        const u32 sh = (g & 1u) * 16u;
        const u32 idx = (atomicAdd(&gcnt[g >> 1], 1u << sh) >> sh) & 0xFFFFu;
        if (idx < FJ_LDS_GROUP) {
            const u32 slot = g * FJ_LDS_GROUP + idx;
            tkeys[slot] = key;
            if (MAT) { if (dedup) atomicMin((unsigned long long*)&tvals[slot], (unsigned long long)val); else tvals[slot] = val; }
            ttag8[slot] = (unsigned char)tag;
            return true;
        }
        if (step == 0) { g = galt; continue; }
        // Both candidate groups are full.  Before the key walks on: is a copy of it in there by now?  Hundreds of copies of ONE key
        // inserted in the same instant all miss each other's tags at the top (a claim's tag is written a few instructions after its
        // add), fill both groups with copies and would then walk the table until it counts as full - a build side of 756 rows, all
        // the same key, failed every materialising join that way (tools/r6_api_fuzz.py).  The copies that took the groups' slots
        // have their tags in place by the time a loser gets here (two LDS round trips later), and a loser that still sees none
        // looks again at every step of its walk.
        {
            u32 n1 = tag_matches(ttags[g1], pat), n2 = tag_matches(ttags[g2], pat);
            while (n1) { const u32 c = g1 * FJ_LDS_GROUP + ((u32)__builtin_ctz(n1) >> 3);
                         if (tkeys[c] == key) { if (MAT && dedup) atomicMin((unsigned long long*)&tvals[c], (unsigned long long)val); return false; } n1 &= n1 - 1; }
            while (n2) { const u32 c = g2 * FJ_LDS_GROUP + ((u32)__builtin_ctz(n2) >> 3);
                         if (tkeys[c] == key) { if (MAT && dedup) atomicMin((unsigned long long*)&tvals[c], (unsigned long long)val); return false; } n2 &= n2 - 1; }
        }
        if (step == 1) { hdr->ovf = 1; g = g1; }           // both candidate groups full: walk from g1
        // (somebody else already walked the whole table in vain: an oversized partition - do not repeat the 2048-step walk per key)
        if ((step & 31u) == 31u && *reinterpret_cast<volatile u32*>(&hdr->full)) return false;
        g = (g + 1) & (NGRP - 1);
    }
    hdr->full = 1;
    return false;
}



// Lookups go through one-byte tags (one u32 = the 4 tags of a group): the kernel is LDS-bound
// when a lookup reads all 8 candidate keys (random ds_read_b128: ~70 % of the LDS cycles are bank
// conflicts, profiles/r01_v5_pmc), so a lookup reads the two tag words (8 B), finds the slots whose
// tag equals the key's tag with a SWAR zero-byte test, and reads only that key (8 B).  Tag 0 = empty.
//   okm[i]  : lanes whose key i is a real key          he : all-ones if the build side held FJ_EMPTY_KEY
//   hitm[i] : lanes whose key i matched                where[i] (MAT only): matching slot
template <bool MAT, int NK>
__device__ __forceinline__ void lds_probe(const u64* __restrict__ tkeys, const u32* __restrict__ ttags, const u64 (&k)[NK],
                                          const u64 (&okm)[NK], u64 he, bool ovf, u32 lane, u64 (&hitm)[NK], u32 (&where)[NK]) {
    u32 g1[NK], g2[NK], z1[NK], z2[NK];
#pragma unroll
    for (int i = 0; i < NK; ++i) {
        const u32 w = FJ_HW2(k[i]);
        g1[i] = (w & (NGRP - 1)) * FJ_LDS_GROUP;
        g2[i] = ((w >> 11) & (NGRP - 1)) * FJ_LDS_GROUP;
        const u32 pat = lds_tag(w) * 0x01010101u;
        z1[i] = tag_matches(ttags[g1[i] / FJ_LDS_GROUP], pat);
        z2[i] = tag_matches(ttags[g2[i] / FJ_LDS_GROUP], pat);
    }
    u32 cand[NK];
    u64 kc[NK];
#pragma unroll
    for (int i = 0; i < NK; ++i) {                           // stage B: the first candidate's key (unconditional read keeps the
        const bool in1 = z1[i] != 0;                         // NK reads independent and in flight together)
        const u32 z = in1 ? z1[i] : z2[i];
        cand[i] = (in1 ? g1[i] : g2[i]) + ((u32)__builtin_ctz(z | 0x80000000u) >> 3);
        kc[i] = tkeys[cand[i]];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NK; ++i) {
        const bool in1 = z1[i] != 0;
        const bool first = ((z1[i] | z2[i]) != 0) & (kc[i] == k[i]);
        const u64 hit1 = __ballot(first);
        const u64 ise = __ballot(k[i] == FJ_EMPTY_KEY);      // the empty marker is never stored in the table
        if (MAT) where[i] = cand[i];
        // more candidates left?  (false-positive tags: rare)
        const u32 r1 = in1 ? (z1[i] & (z1[i] - 1)) : 0u, r2 = in1 ? z2[i] : (z2[i] & (z2[i] - 1));
        const u64 more = __ballot(!first && (r1 | r2) != 0) & okm[i] & ~ise;
        u64 hit = hit1;
        if (more) {
            bool found = false;
            if ((more >> lane) & 1ull) {
                u32 ra = r1, rb = r2;
                while (ra | rb) {
                    const bool ina = ra != 0;
                    const u32 zz = ina ? ra : rb;
                    const u32 c = (ina ? g1[i] : g2[i]) + ((u32)__builtin_ctz(zz) >> 3);
                    if (tkeys[c] == k[i]) { found = true; if (MAT) where[i] = c; break; }
                    if (ina) ra &= ra - 1; else rb &= rb - 1;
                }
            }
            hit |= __ballot(found);
        }
        if (ovf) {                                           // this table holds linearly probed overflow keys
            const u64 undec = okm[i] & ~ise & ~hit;
            bool found = false;
            if ((undec >> lane) & 1ull) {
                u32 g = g1[i] / FJ_LDS_GROUP;
                for (u32 step = 0; step < NGRP; ++step) {    // an overflow key sits in the first non-full group after g1
                    g = (g + 1) & (NGRP - 1);
                    const u32 t = ttags[g];
                    u32 m = tag_matches(t, lds_tag(FJ_HW2(k[i])) * 0x01010101u);
                    while (m) {
                        const u32 c = g * FJ_LDS_GROUP + ((u32)__builtin_ctz(m) >> 3);
                        if (tkeys[c] == k[i]) { found = true; if (MAT) where[i] = c; break; }
                        m &= m - 1;
                    }
                    if (found || tag_zero_bytes(t)) break;
                }
            }
            hit |= __ballot(found);
        }
        hitm[i] = okm[i] & ((hit & ~ise) | (ise & he));
    }
}

// encoded chunk-list entry ((count-1) << 24 | id) of chunk `idx`; a flat array is read as a virtual chunk list
__device__ __forceinline__ u32 chunk_entry(const FjChunkSet& cs, u32 idx) {
    if (cs.list) return cs.list[idx];
    const u64 rem = cs.n_flat - (u64)idx * FJ_CHUNK;
    const u32 cnt = rem >= FJ_CHUNK ? FJ_CHUNK : (u32)rem;
    return ((cnt - 1u) << 24) | idx;
}

constexpr u32 JB_META = 128;    // build-side chunk-list entries staged in LDS per batch
constexpr u32 JP_META = 512;    // probe-side chunk-list entries staged in LDS per batch

// grid = nparts * nsplit work items: item = (partition p, slice of p's probe chunks).  Every item
// rebuilds p's table in LDS (cheap: the build side of a partition is <= a few thousand rows and
// is L2 / Infinity-Cache resident), so small-build joins still fill the chip.
//
// A work item lives ~25 us, so it is written against HBM *latency*: the chunk-list entries of both
// sides are fetched together, all build keys of the partition are requested in one shot (not chunk
// by chunk), and probe keys are prefetched two rounds (2 x 8 keys per lane) ahead; the first two
// rounds are requested before the table is even initialised.
template <bool MAT, int NT, bool LIST>
__global__ __launch_bounds__(NT, 4) void fj_lds_join_kernel(FjLdsJoinArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    JoinHdr* hdr = reinterpret_cast<JoinHdr*>(smem);
    u64* tkeys = reinterpret_cast<u64*>(smem + sizeof(JoinHdr));
    u64* tvals = tkeys + S;        // only present when MAT
    u32* ttags = reinterpret_cast<u32*>(tkeys + (MAT ? 2 * S : S));     // S one-byte tags
    u32* gcnt = ttags + S / 4;                    // group fill counters, 16 bits each
    u32* pm = gcnt + NGRP / 2;
    u32* bm = pm + JP_META;
    const u32 tid = threadIdx.x, lane = tid & 63;
#define FJ_STAMP(i) do { if (a.dbg && tid == 0 && blockIdx.x < 4096) a.dbg[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
    FJ_STAMP(0);
    // work item -> (partition p, probe chunk-list range [p0 + s_lo, p0 + s_hi))
    const u32 item = blockIdx.x;
    u32 p, b0 = 0, nbc, p0 = 0, s_lo, s_hi;
    if (a.items) {                                   // chunk-list probe side: the item table (skew-proof slices)
        if (item >= *a.nitems_dev) return;
        const uint4 it = a.items[item];
        p = it.z; s_lo = it.x; s_hi = it.x + it.y;
    } else {                                         // flat probe side: equal slices
        const u32 slice = item % a.nsplit;
        p = item / a.nsplit;
        const u32 npc = (u32)((a.probe.n_flat + FJ_CHUNK - 1) >> FJ_CHUNK_LOG);
        s_lo = (u32)(((u64)slice * npc) / a.nsplit); s_hi = (u32)(((u64)(slice + 1) * npc) / a.nsplit);
    }
    if (a.build.list) { b0 = a.build.boff[p]; nbc = a.build.boff[p + 1] - b0; }
    else nbc = (u32)((a.build.n_flat + FJ_CHUNK - 1) >> FJ_CHUNK_LOG);
    if (nbc == 0 || s_lo >= s_hi) {      // an empty side is skipped (hash_join.cpp:343, :518)
        if (!MAT && tid == 0) a.part_count[item] = 0;
        return;
    }
    if (MAT && a.part_count[item] == 0) return;
    if (a.retry_only && a.part_count[item] != FJ_ITEM_RETRY) return;           // second chance for the cuckoo kernels' overflows only

    constexpr u32 CPL = NT / (FJ_CHUNK / 2);          // chunks covered by one 16-B load per lane
    constexpr u32 CPR = 4 * CPL;                      // chunks per round (4 loads per lane = 8 keys)
    constexpr u32 BKPT = 4096 / NT;                   // build keys per lane and build batch (16 chunks)
    // request one round of probe keys of the current metadata batch
    // LIST: both sides are chunk lists (every partitioned plan).  Their chunks are whole 2-KiB pool blocks, so the loads
    // are issued unconditionally (validity is a mask) -- straight-line loads let the waits be counted.
    auto load_round = [&](u32 r, u32 nbatch, u64 (&kk)[8], u32& okm) {
        okm = 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const u32 c = r * CPR + u * CPL + tid / (FJ_CHUNK / 2), off = (tid % (FJ_CHUNK / 2)) * 2;
            if (LIST) {
                const u32 e = pm[c < nbatch ? c : nbatch - 1], cnt = c < nbatch ? FJ_LIST_CNT(e) : 0;
                const u64x2 q = *reinterpret_cast<const u64x2*>(a.probe.keys + (u64)FJ_LIST_ID(e) * FJ_CHUNK + off);
                kk[2 * u] = q.x; kk[2 * u + 1] = q.y;
                okm |= ((off < cnt ? 1u : 0u) | (off + 1 < cnt ? 2u : 0u)) << (2 * u);
                continue;
            }
            kk[2 * u] = 0; kk[2 * u + 1] = 0;
            if (c < nbatch) {
                const u32 e = pm[c], cnt = FJ_LIST_CNT(e);
                const u64 base = (u64)FJ_LIST_ID(e) * FJ_CHUNK + off;
                if (off + 1 < cnt) {
                    const u64x2 q = *reinterpret_cast<const u64x2*>(a.probe.keys + base);
                    kk[2 * u] = a.probe.list ? q.x : fj_key_mix(q.x); kk[2 * u + 1] = a.probe.list ? q.y : fj_key_mix(q.y); okm |= 3u << (2 * u);     // flat arrays hold raw keys
                } else if (off < cnt) {
                    kk[2 * u] = a.probe.list ? a.probe.keys[base] : fj_key_mix(a.probe.keys[base]); okm |= 1u << (2 * u);
                }
            }
        }
    };
    u64 bk[BKPT], bv[BKPT];
    u32 bok = 0;
    auto load_build = [&](u32 c0, u32 nbb) {                // 16 chunks = 4096 rows requested at once
        bok = 0;
#pragma unroll
        for (u32 j = 0; j < BKPT; ++j) {
            const u32 kidx = j * NT + tid, c = c0 + (kidx >> FJ_CHUNK_LOG), off = kidx & (FJ_CHUNK - 1);
            if (LIST) {
                const u32 e = bm[c < nbb ? c : nbb - 1];
                const u64 src = (u64)FJ_LIST_ID(e) * FJ_CHUNK + off;
                bk[j] = a.build.keys[src];
                bv[j] = MAT ? a.build.vals[src] : 0;
                bok |= (c < nbb && off < FJ_LIST_CNT(e) ? 1u : 0u) << j;
                continue;
            }
            bk[j] = 0; bv[j] = 0;
            if (c < nbb) {
                const u32 e = bm[c];
                if (off < FJ_LIST_CNT(e)) {
                    const u64 src = (u64)FJ_LIST_ID(e) * FJ_CHUNK + off;
                    bk[j] = a.build.list ? a.build.keys[src] : fj_key_mix(a.build.keys[src]);
                    if (MAT) bv[j] = a.build.vals[src];
                    bok |= 1u << j;
                }
            }
        }
    };

    // ---- chunk-list entries of both sides, table init; then the first two probe rounds go in flight
    u32 nbatch = (s_hi - s_lo) < JP_META ? (s_hi - s_lo) : JP_META;
    for (u32 i = tid; i < nbatch; i += NT) pm[i] = chunk_entry(a.probe, p0 + s_lo + i);
    u32 nbb = nbc < JB_META ? nbc : JB_META;
    if (tid < nbb) bm[tid] = chunk_entry(a.build, b0 + tid);
    for (u32 i = tid; i < S / 4 + NGRP / 2; i += NT) ttags[i] = 0;       // tags and group counters
    const bool dedup = MAT && a.dedup != 0;
    if (dedup) for (u32 i = tid; i < S; i += NT) tvals[i] = ~0ull;        // row indices: the minimum wins
    if (tid == 0) { hdr->cnt = 0; hdr->has_empty = 0; hdr->cursor = 0; hdr->claimed = 0; hdr->full = 0; hdr->ovf = 0; hdr->empty_val = (MAT && a.dedup) ? ~0ull : 0ull; }
    __syncthreads();
    FJ_STAMP(1);
    u64 ka[8], kb[8];
    u32 oka = 0, okb = 0;
    u32 nrounds = (nbatch + CPR - 1) / CPR;
    load_build(0, nbb);                              // build rows first: their inserts start while the probe keys fly
    load_round(0, nbatch, ka, oka);
    if (nrounds > 1) load_round(1, nbatch, kb, okb);

    // ---- build: first claim of a key wins, later duplicates are dropped (hash_join.cpp:125) ----
    u32 claimed = 0;
    for (u32 bb = 0; bb < nbc; bb += JB_META) {
        if (bb) {                                   // only partitions with > 128 build chunks get here
            nbb = (nbc - bb) < JB_META ? (nbc - bb) : JB_META;
            __syncthreads();
            if (tid < nbb) bm[tid] = chunk_entry(a.build, b0 + bb + tid);
            __syncthreads();
        }
        for (u32 c0 = 0; c0 < nbb; c0 += 16) {
            if (*reinterpret_cast<volatile u32*>(&hdr->full)) break;       // the table is full: the item is reported below, no point in more inserts
            if (bb | c0) load_build(c0, nbb);
#pragma unroll
            for (u32 j = 0; j < BKPT; ++j) {
                if (bok & (1u << j)) {
                    const u64 key = bk[j];
                    if (key == FJ_EMPTY_KEY) {
                        hdr->has_empty = 1;
                        if (MAT) hdr->empty_val = bv[j];
                    } else if (!(a.dbg_flags & 2u)) {
                        claimed += lds_insert<MAT>(tkeys, tvals, ttags, gcnt, hdr, key, bv[j], dedup) ? 1u : 0u;
                    }
                }
            }
        }
    }
    FJ_STAMP(2);
    claimed = fj_wave_sum(claimed);
    if (lane == 0 && claimed) atomicAdd(&hdr->claimed, claimed);
    __syncthreads();
    FJ_STAMP(3);
    if (hdr->full || hdr->claimed > S - 64) {       // table (nearly) full: host falls back to the global-table path
        if (tid == 0) {
            if (!MAT && a.mark_toobig) { atomicOr(a.err, FJ_STAT_TOOBIG); a.part_count[item] = FJ_ITEM_TOOBIG; }   // the host re-partitions this partition alone
            else { atomicOr(a.err, FJ_ERR_LDS_FULL); if (!MAT) a.part_count[item] = 0; }
        }
        return;
    }
    // counting pass of a materialising join, second-chance table: racing copies of a duplicated key can sit in the table
    // unnoticed, and this table has no sweep for them - report "duplicates possible", which only selects the (always
    // correct) first-occurrence emit path
    if (!MAT && a.want_dups && tid == 0) atomicOr(a.err, FJ_STAT_DUPS);
    if (dedup) {
        // Two racing copies of one key may both have been stored, each with the smallest row index IT saw: give every
        // copy the minimum over all copies (every copy lives in g1, g2 or, for overflow keys, the walk from g1), then
        // turn the winning row index into its value with one gather from the caller's build_values.
        const bool ovf0 = hdr->ovf != 0;
        for (u32 sl = tid; sl < S; sl += NT) {
            if (reinterpret_cast<const unsigned char*>(ttags)[sl] == 0) continue;
            const u64 key = tkeys[sl];
            const u32 w = FJ_HW2(key), pat = lds_tag(w) * 0x01010101u;
            u32 g = w & (NGRP - 1);
            const u32 g2 = (w >> 11) & (NGRP - 1);
            u64 best = tvals[sl];
            for (u32 step = 0; step < NGRP + 2; ++step) {
                const u32 t = ttags[g];
                u32 m = tag_matches(t, pat);
                while (m) { const u32 c = g * FJ_LDS_GROUP + ((u32)__builtin_ctz(m) >> 3); if (c != sl && tkeys[c] == key) { const u64 o = tvals[c]; best = o < best ? o : best; } m &= m - 1; }
                if (step == 0) { g = g2; continue; }                     // second candidate group
                if (!ovf0) break;
                if (step == 1) g = w & (NGRP - 1);                      // then the overflow walk from g1 + 1
                else if (tag_zero_bytes(t)) break;                       // a non-full group ends the walk
                g = (g + 1) & (NGRP - 1);
            }
            atomicMin((unsigned long long*)&tvals[sl], (unsigned long long)best);
        }
        __syncthreads();
        for (u32 sl = tid; sl < S; sl += NT)
            if (reinterpret_cast<const unsigned char*>(ttags)[sl] != 0) tvals[sl] = a.orig_vals[tvals[sl]];
        if (tid == 0 && hdr->has_empty) hdr->empty_val = a.orig_vals[hdr->empty_val];
        __syncthreads();
    }
    const bool has_empty = hdr->has_empty != 0;
    const u64 obase = MAT ? a.out_off[item] : 0;

    // ---- probe ------------------------------------------------------------------------------------
    u32 wave_hits = 0;                                        // wave-uniform
    const u64 he = has_empty ? ~0ull : 0ull;
    const bool ovf = hdr->ovf != 0;                          // wave-uniform: does this table need the walking lookup?
    for (u32 pb = s_lo; pb < s_hi; pb += JP_META) {
        if (pb != s_lo) {                           // later batches (only very large partitions get here)
            nbatch = (s_hi - pb) < JP_META ? (s_hi - pb) : JP_META;
            __syncthreads();
            for (u32 i = tid; i < nbatch; i += NT) pm[i] = chunk_entry(a.probe, p0 + pb + i);
            __syncthreads();
            nrounds = (nbatch + CPR - 1) / CPR;
            load_round(0, nbatch, ka, oka);
            if (nrounds > 1) load_round(1, nbatch, kb, okb);
        }
        for (u32 r = 0; r < nrounds; ++r) {
            u64 k[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) { k[i] = ka[i]; ka[i] = kb[i]; }
            const u32 okm = oka;
            oka = okb;
            if (r + 2 < nrounds) load_round(r + 2, nbatch, kb, okb);
#pragma unroll
            for (int hgrp = 0; hgrp < 2; ++hgrp) {
                const u64 k2[4] = {k[4 * hgrp], k[4 * hgrp + 1], k[4 * hgrp + 2], k[4 * hgrp + 3]};
                u64 ok2[4], hitm[4];
                u32 where[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) ok2[i] = __ballot((okm >> (4 * hgrp + i)) & 1u);
                if (a.dbg_flags & 1u) { hitm[0] = __ballot((k2[0] ^ k2[1]) & 1ull); hitm[1] = hitm[2] = hitm[3] = 0; where[0] = where[1] = where[2] = where[3] = 0; }
                else lds_probe<MAT, 4>(tkeys, ttags, k2, ok2, he, ovf, lane, hitm, where);
                if (MAT) {      // ONE LDS cursor bump per wave for the four key slots; lanes ranked inside the ballots
                    const u32 n0 = (u32)__popcll(hitm[0]), n1 = (u32)__popcll(hitm[1]), n2 = (u32)__popcll(hitm[2]), n3 = (u32)__popcll(hitm[3]);
                    if (n0 + n1 + n2 + n3) {
                        u32 wb = 0;
                        if (lane == 0) wb = atomicAdd(&hdr->cursor, n0 + n1 + n2 + n3);
                        wb = (u32)__builtin_amdgcn_readfirstlane((int)wb);
                        const u32 off[4] = {0u, n0, n0 + n1, n0 + n1 + n2};
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const u64 m = hitm[i];
                            if ((m >> lane) & 1ull) {
                                const u64 o = obase + wb + off[i] + (u32)__popcll(m & ((1ull << lane) - 1ull));
                                if (a.dbg_flags & 4u) continue;             // (ablation: no output stores)
                                a.out_keys[o] = fj_key_unmix(k2[i]);          // (tables and chunk pools hold mixed keys)
                                a.out_vals[o] = k2[i] == FJ_EMPTY_KEY ? hdr->empty_val : tvals[where[i]];
                            }
                        }
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) wave_hits += (u32)__popcll(hitm[i]);   // scalar: the count never touches the VALU
                }
            }
        }
    }
    FJ_STAMP(4);
    if (!MAT) {
        if (lane == 0 && wave_hits) atomicAdd(&hdr->cnt, wave_hits);
        __syncthreads();
        FJ_STAMP(5);
        if (tid == 0) {
            a.part_count[item] = hdr->cnt;
            if (hdr->cnt) atomicAdd(a.total, (unsigned long long)hdr->cnt);
        }
    } else if (a.dbg) {
        __syncthreads();                                 // diagnostic only: stamp 5 = the slowest wave is done
        FJ_STAMP(5);
    }
}

// ---- count-only join: cuckoo table ---------------------------------------------------------------
// For counting, a lookup only has to answer "is the key there".  Two single-slot candidate locations
// (cuckoo hashing with evictions at build time) make that 2 independent 8-byte LDS reads and two
// v_cmp_eq_u64 -- about a third of the issued instructions of the tagged 2x4-slot table above, which
// stays in use where a value must be fetched (materialising joins).  Load is ~0.37 by plan; a key
// whose eviction chain does not terminate goes to a 32-entry stash that lookups scan only when it is
// non-empty (a wave-uniform branch); a full stash raises FJ_ERR_LDS_FULL like a full table.
constexpr u32 CK_STASH = 32, CK_MAXIT = 48;
struct CkHdr { u32 cnt, has_empty, nstash, full, dups, empties, novf, pad1; u64 pad[1]; u64 stash[CK_STASH]; };

template <typename Hdr>
__device__ __forceinline__ void cuckoo_insert(u64* __restrict__ tkeys, Hdr* hdr, u64 key) {
    u32 w = FJ_HW2(key);
    u32 l1 = w & (S - 1), l2 = (w >> 13) & (S - 1);
    if (tkeys[l1] == key || tkeys[l2] == key) { hdr->dups = 1; return; }   // duplicate build key already stored (hash_join.cpp:125)
    u32 loc = l1;
#pragma unroll 1
    for (u32 it = 0; it < CK_MAXIT; ++it) {
        const u64 old = atomicExch((unsigned long long*)&tkeys[loc], (unsigned long long)key);
        if (old == key) hdr->dups = 1;
        if (old == FJ_EMPTY_KEY || old == key) return;            // free slot, or displaced a copy of the same key
        key = old;                                                // carry the evicted key to its other location
        w = FJ_HW2(key);
        l1 = w & (S - 1); l2 = (w >> 13) & (S - 1);
        loc = loc == l1 ? l2 : l1;
    }
    const u32 sidx = atomicAdd(&hdr->nstash, 1u);
    if (sidx < CK_STASH) hdr->stash[sidx] = key; else hdr->full = 1;
}

// Two-phase build.  An evicting insert is a chain of dependent LDS operations (~100 clocks each, tools/ubench_lds_atomics)
// and a wave stays in its loop until its unluckiest lane is done, while at load 0.37 nine in ten inserts only need an EMPTY
// candidate slot: phase 1 claims one with a 32-bit atomic OR on a slot bitmap and stores the key with a plain write (no
// loop: a thread's keys overlap); a key that finds both candidates taken goes to a dense overflow list (unless a copy of it
// is already visible: a duplicate, dropped as hash_join.cpp:125 does).  After a barrier - every phase-1 store has landed -
// phase 2 runs the evicting insert over the list only: full waves, a tenth of the keys.  tools/ubench: 3052 keys 8.0 -> 4.8 us per table, 3950 keys 16 -> 8.6 us.  A duplicated key may
// end up stored twice (both candidates) as before; a list overflow (heavy duplication racing the stores) reports a full table.
constexpr u32 CK_OVF = 1024;
template <typename Hdr>
__device__ __forceinline__ void cuckoo_claim(u64* __restrict__ tkeys, u32* __restrict__ bits, u64* __restrict__ ovf, Hdr* hdr, u64 key) {
    const u32 w = FJ_HW2(key), l1 = w & (S - 1), l2 = (w >> 13) & (S - 1);
    const u32 o1 = atomicOr(&bits[l1 >> 5], 1u << (l1 & 31));
    if (!((o1 >> (l1 & 31)) & 1u)) { tkeys[l1] = key; return; }
    const u32 o2 = atomicOr(&bits[l2 >> 5], 1u << (l2 & 31));
    if (!((o2 >> (l2 & 31)) & 1u)) { tkeys[l2] = key; return; }
    if (tkeys[l1] == key || tkeys[l2] == key) { hdr->dups = 1; return; }
    const u32 i = atomicAdd(&hdr->novf, 1u);
    if (i < CK_OVF) ovf[i] = key; else hdr->full = 1;
}
template <int NT, typename Hdr>
__device__ __forceinline__ void cuckoo_finish(u64* __restrict__ tkeys, const u64* __restrict__ ovf, Hdr* hdr, u32 tid) {   // after the barrier
    const u32 n = hdr->novf < CK_OVF ? hdr->novf : CK_OVF;
    for (u32 i = tid; i < n; i += NT) cuckoo_insert(tkeys, hdr, ovf[i]);
}

template <int NT, bool LIST>
__global__ __launch_bounds__(NT, 4) void fj_count_join_kernel(FjLdsJoinArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    CkHdr* hdr = reinterpret_cast<CkHdr*>(smem);
    u64* tkeys = reinterpret_cast<u64*>(smem + sizeof(CkHdr));
    u32* pm = reinterpret_cast<u32*>(tkeys + S);
    u32* bm = pm + JP_META;
    u32* bits = bm + JB_META;                                // slot bitmap of the two-phase build
    u64* ovf = reinterpret_cast<u64*>(bits + S / 32);        // its overflow list
    const u32 tid = threadIdx.x, lane = tid & 63;
    FJ_STAMP(0);
    // work item -> (partition p, probe chunk-list range [p0 + s_lo, p0 + s_hi))
    const u32 item = blockIdx.x;
    u32 p, b0 = 0, nbc, p0 = 0, s_lo, s_hi;
    if (a.items) {                                   // chunk-list probe side: the item table (skew-proof slices)
        if (item >= *a.nitems_dev) return;
        const uint4 it = a.items[item];
        p = it.z; s_lo = it.x; s_hi = it.x + it.y;
    } else {                                         // flat probe side: equal slices
        const u32 slice = item % a.nsplit;
        p = item / a.nsplit;
        const u32 npc = (u32)((a.probe.n_flat + FJ_CHUNK - 1) >> FJ_CHUNK_LOG);
        s_lo = (u32)(((u64)slice * npc) / a.nsplit); s_hi = (u32)(((u64)(slice + 1) * npc) / a.nsplit);
    }
    if (a.build.list) { b0 = a.build.boff[p]; nbc = a.build.boff[p + 1] - b0; }
    else nbc = (u32)((a.build.n_flat + FJ_CHUNK - 1) >> FJ_CHUNK_LOG);
    if (nbc == 0 || s_lo >= s_hi) {      // an empty side is skipped (hash_join.cpp:343, :518)
        if (tid == 0) a.part_count[item] = 0;
        return;
    }
    constexpr u32 CPL = NT / (FJ_CHUNK / 2), CPR = 4 * CPL, BKPT = 4096 / NT;
    // Chunk-list inputs live in whole 2-KiB pool chunks, so their loads are issued unconditionally (validity is a mask):
    // straight-line loads let the compiler count outstanding requests instead of draining them all at the first use.
    constexpr bool plist = LIST, blist = LIST;       // LIST: both sides are chunk lists (every partitioned plan)
    auto load_round = [&](u32 r, u32 nbatch, u64 (&kk)[8], u32& okm) {
        okm = 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const u32 c = r * CPR + u * CPL + tid / (FJ_CHUNK / 2), off = (tid % (FJ_CHUNK / 2)) * 2;
            if (plist) {
                const u32 e = pm[c < nbatch ? c : nbatch - 1], cnt = c < nbatch ? FJ_LIST_CNT(e) : 0;
                const u64x2 q = *reinterpret_cast<const u64x2*>(a.probe.keys + (u64)FJ_LIST_ID(e) * FJ_CHUNK + off);
                kk[2 * u] = q.x; kk[2 * u + 1] = q.y;
                okm |= ((off < cnt ? 1u : 0u) | (off + 1 < cnt ? 2u : 0u)) << (2 * u);
                continue;
            }
            kk[2 * u] = 0; kk[2 * u + 1] = 0;
            if (c < nbatch) {
                const u32 e = pm[c], cnt = FJ_LIST_CNT(e);
                const u64 base = (u64)FJ_LIST_ID(e) * FJ_CHUNK + off;
                if (off + 1 < cnt) {
                    const u64x2 q = *reinterpret_cast<const u64x2*>(a.probe.keys + base);
                    kk[2 * u] = a.probe.list ? q.x : fj_key_mix(q.x); kk[2 * u + 1] = a.probe.list ? q.y : fj_key_mix(q.y); okm |= 3u << (2 * u);     // flat arrays hold raw keys
                } else if (off < cnt) {
                    kk[2 * u] = a.probe.list ? a.probe.keys[base] : fj_key_mix(a.probe.keys[base]); okm |= 1u << (2 * u);
                }
            }
        }
    };
    u64 bk[BKPT];
    u32 bok = 0;
    auto load_build = [&](u32 c0, u32 nbb) {                // 16 chunks = 4096 keys requested at once
        bok = 0;
#pragma unroll
        for (u32 j = 0; j < BKPT; ++j) {
            const u32 kidx = j * NT + tid, c = c0 + (kidx >> FJ_CHUNK_LOG), off = kidx & (FJ_CHUNK - 1);
            if (blist) {
                const u32 e = bm[c < nbb ? c : nbb - 1];
                bk[j] = a.build.keys[(u64)FJ_LIST_ID(e) * FJ_CHUNK + off];
                bok |= (c < nbb && off < FJ_LIST_CNT(e) ? 1u : 0u) << j;
                continue;
            }
            bk[j] = 0;
            if (c < nbb) {
                const u32 e = bm[c];
                if (off < FJ_LIST_CNT(e)) { const u64 raw = a.build.keys[(u64)FJ_LIST_ID(e) * FJ_CHUNK + off]; bk[j] = a.build.list ? raw : fj_key_mix(raw); bok |= 1u << j; }
            }
        }
    };

    u32 nbatch = (s_hi - s_lo) < JP_META ? (s_hi - s_lo) : JP_META;
    for (u32 i = tid; i < nbatch; i += NT) pm[i] = chunk_entry(a.probe, p0 + s_lo + i);
    u32 nbb = nbc < JB_META ? nbc : JB_META;
    if (tid < nbb) bm[tid] = chunk_entry(a.build, b0 + tid);
    for (u32 i = tid; i < S; i += NT) tkeys[i] = FJ_EMPTY_KEY;
    if (tid < S / 32) bits[tid] = 0;
    if (tid == 0) { hdr->cnt = 0; hdr->has_empty = 0; hdr->nstash = 0; hdr->full = 0; hdr->dups = 0; hdr->empties = 0; hdr->novf = 0; }
    __syncthreads();
    FJ_STAMP(1);
    u64 ka[8], kb[8];
    u32 oka = 0, okb = 0;
    u32 nrounds = (nbatch + CPR - 1) / CPR;
    load_build(0, nbb);                              // build keys first: their inserts start while the probe keys fly
    load_round(0, nbatch, ka, oka);
    if (nrounds > 1) load_round(1, nbatch, kb, okb);

    // ---- build ------------------------------------------------------------------------------------
    for (u32 bb = 0; bb < nbc; bb += JB_META) {
        if (bb) {
            nbb = (nbc - bb) < JB_META ? (nbc - bb) : JB_META;
            __syncthreads();
            if (tid < nbb) bm[tid] = chunk_entry(a.build, b0 + bb + tid);
            __syncthreads();
        }
        for (u32 c0 = 0; c0 < nbb; c0 += 16) {
            if (bb | c0) load_build(c0, nbb);
#pragma unroll
            for (u32 j = 0; j < BKPT; ++j) {
                if (bok & (1u << j)) {
                    if (bk[j] == FJ_EMPTY_KEY) { hdr->has_empty = 1; if (a.want_dups && atomicAdd(&hdr->empties, 1u) > 0) hdr->dups = 1; }
                    else if (!(a.dbg_flags & 2u)) cuckoo_claim(tkeys, bits, ovf, hdr, bk[j]);
                }
            }
        }
    }
    FJ_STAMP(2);
    __syncthreads();                                 // phase-1 stores have landed
    cuckoo_finish<NT>(tkeys, ovf, hdr, tid);
    __syncthreads();
    FJ_STAMP(3);
    if (hdr->full) {                                 // stash overflow: the item is redone with the tagged table (fj_launch_lds_join_retry)
        if (tid == 0) { atomicOr(a.err, FJ_STAT_RETRY); a.part_count[item] = FJ_ITEM_RETRY; }
        return;
    }
    const u64 he = hdr->has_empty ? ~0ull : 0ull;
    const u32 nstash = hdr->nstash < CK_STASH ? hdr->nstash : CK_STASH;
    if (a.want_dups) {
        // exact duplicate report for the materialising pass that follows: a second copy of a key was either noticed
        // by its insert (dups flag) or it raced past the check and the key now sits in both of its slots / the stash
        for (u32 sl = tid; sl < S; sl += NT) {
            const u64 key = tkeys[sl];
            if (key == FJ_EMPTY_KEY) continue;
            const u32 w = FJ_HW2(key), l1 = w & (S - 1), l2 = (w >> 13) & (S - 1);
            const u32 other = sl == l1 ? l2 : l1;
            if (other != sl && tkeys[other] == key) hdr->dups = 1;
            for (u32 si = 0; si < nstash; ++si) if (hdr->stash[si] == key) hdr->dups = 1;
        }
        for (u32 si = tid; si < nstash; si += NT)
            for (u32 sj = si + 1; sj < nstash; ++sj) if (hdr->stash[si] == hdr->stash[sj]) hdr->dups = 1;
        __syncthreads();
        if (tid == 0 && hdr->dups) atomicOr(a.err, FJ_STAT_DUPS);
    }

    // ---- probe ------------------------------------------------------------------------------------
    u32 wave_hits = 0;                                        // wave-uniform, accumulated on the scalar unit
    for (u32 pb = s_lo; pb < s_hi; pb += JP_META) {
        if (pb != s_lo) {
            nbatch = (s_hi - pb) < JP_META ? (s_hi - pb) : JP_META;
            __syncthreads();
            for (u32 i = tid; i < nbatch; i += NT) pm[i] = chunk_entry(a.probe, p0 + pb + i);
            __syncthreads();
            nrounds = (nbatch + CPR - 1) / CPR;
            load_round(0, nbatch, ka, oka);
            if (nrounds > 1) load_round(1, nbatch, kb, okb);
        }
        for (u32 r = 0; r < nrounds; ++r) {
            u64 k[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) { k[i] = ka[i]; ka[i] = kb[i]; }
            const u32 okm = oka;
            oka = okb;
            if (r + 2 < nrounds) load_round(r + 2, nbatch, kb, okb);
            if (a.dbg_flags & 1u) { wave_hits += (u32)__popcll(__ballot((k[0] ^ k[7]) & 1ull)); continue; }
            u64 c1[8], c2[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {                    // 16 independent LDS reads in flight
                const u32 w = FJ_HW2(k[i]);
                c1[i] = tkeys[w & (S - 1)];
                c2[i] = tkeys[(w >> 13) & (S - 1)];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                u64 hit = __ballot(c1[i] == k[i]) | __ballot(c2[i] == k[i]);
                if (nstash) {                                 // rare tables only
                    bool f = false;
                    for (u32 si = 0; si < nstash; ++si) f |= hdr->stash[si] == k[i];
                    hit |= __ballot(f);
                }
                const u64 ise = __ballot(k[i] == FJ_EMPTY_KEY);  // the empty marker is never stored in the table
                const u64 ok = __ballot((okm >> i) & 1u);
                wave_hits += (u32)__popcll(ok & ((hit & ~ise) | (ise & he)));
            }
        }
    }
    FJ_STAMP(4);
    if (lane == 0 && wave_hits) atomicAdd(&hdr->cnt, wave_hits);
    __syncthreads();
    FJ_STAMP(5);
    if (tid == 0) {
        a.part_count[item] = hdr->cnt;
        if (hdr->cnt) atomicAdd(a.total, (unsigned long long)hdr->cnt);
    }
}

// ---- count-only join over chunk lists, persistent form ----------------------------------------------
// Same table and lookups as fj_count_join_kernel, but a workgroup stays resident and takes items from a global
// counter, so that the dependent-load latencies of an item (list entries -> build keys) are paid while the PREVIOUS
// item is being probed: after the build barrier of item i the list entries of item i+1 are requested, after the
// first probe round they are parked in the second half of the LDS staging arrays, and right after the last probe
// round the build keys of item i+1 are requested into registers (holding them across the probe loop would spill).
// Item i+1 then starts with a table memset, its first probe rounds' loads, and inserts.
// Measured at c3 (32768 items of ~3052 build / ~30518 probe keys): see DESIGN.md section 5.
template <int NT>
__global__ __launch_bounds__(NT, 4) void fj_count_join_persistent(FjLdsJoinArgs a, u32* __restrict__ next_item) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    CkHdr* hdr = reinterpret_cast<CkHdr*>(smem);
    u64* tkeys = reinterpret_cast<u64*>(smem + sizeof(CkHdr));
    u32* pm0 = reinterpret_cast<u32*>(tkeys + S);          // [2][JP_META] probe-side list entries
    u32* bm0 = pm0 + 2 * JP_META;                            // [2][JB_META] build-side list entries
    u32* s_next = bm0 + 2 * JB_META;                         // [2] item ids handed out by the global counter
    u32* bits = s_next + 4;                                  // slot bitmap of the two-phase build
    u64* ovf = reinterpret_cast<u64*>(bits + S / 32);        // its overflow list
    const u32 tid = threadIdx.x, lane = tid & 63;
    const u32 nitems = *a.nitems_dev;                       // chunk-list inputs only: items come from the item table
    constexpr u32 CPL = NT / (FJ_CHUNK / 2), CPR = 4 * CPL, BKPT = 4096 / NT;
    static_assert(JP_META <= NT && JB_META <= NT, "one staged list entry per thread");

    struct Desc { u32 item, b0, nbc, p0, s_lo, s_hi; };
    auto describe = [&](u32 item) -> Desc {                  // wave-uniform scalar loads
        Desc d; d.item = item; d.b0 = 0; d.nbc = 0; d.p0 = 0; d.s_lo = 0; d.s_hi = 0;
        if (item >= nitems) return d;
        const uint4 it = a.items[item];
        d.b0 = a.build.boff[it.z]; d.nbc = a.build.boff[it.z + 1] - d.b0;
        d.p0 = 0; d.s_lo = it.x; d.s_hi = it.x + it.y;
        return d;
    };
    auto live = [&](const Desc& d) { return d.item < nitems && d.nbc > 0 && d.s_lo < d.s_hi; };

    u64 bk[BKPT];
    u32 bok = 0;
    auto load_build = [&](const u32* bm, u32 c0, u32 nbb) {  // 16 chunks = 4096 keys requested at once, unconditionally
        bok = 0;
#pragma unroll
        for (u32 j = 0; j < BKPT; ++j) {
            const u32 kidx = j * NT + tid, c = c0 + (kidx >> FJ_CHUNK_LOG), off = kidx & (FJ_CHUNK - 1);
            const u32 e = bm[c < nbb ? c : nbb - 1];
            bk[j] = a.build.keys[(u64)FJ_LIST_ID(e) * FJ_CHUNK + off];
            bok |= (c < nbb && off < FJ_LIST_CNT(e) ? 1u : 0u) << j;
        }
    };
    auto load_round = [&](const u32* pm, u32 r, u32 nbatch, u64 (&kk)[8], u32& okm) {
        okm = 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const u32 c = r * CPR + u * CPL + tid / (FJ_CHUNK / 2), off = (tid % (FJ_CHUNK / 2)) * 2;
            const u32 e = pm[c < nbatch ? c : nbatch - 1], cnt = c < nbatch ? FJ_LIST_CNT(e) : 0;
            const u64x2 q = *reinterpret_cast<const u64x2*>(a.probe.keys + (u64)FJ_LIST_ID(e) * FJ_CHUNK + off);
            kk[2 * u] = q.x; kk[2 * u + 1] = q.y;
            okm |= ((off < cnt ? 1u : 0u) | (off + 1 < cnt ? 2u : 0u)) << (2 * u);
        }
    };
    auto reset_table = [&]() {
        for (u32 i = tid; i < S; i += NT) tkeys[i] = FJ_EMPTY_KEY;
        if (tid < S / 32) bits[tid] = 0;
        if (tid == 0) { hdr->cnt = 0; hdr->has_empty = 0; hdr->nstash = 0; hdr->full = 0; hdr->dups = 0; hdr->empties = 0; hdr->novf = 0; }
    };

    // ---- prologue: first item (static), its list entries, its first build batch ----
    Desc d = describe(blockIdx.x);
    if (d.item >= nitems) return;
    u32 buf = 0;
    {
        const bool lv = live(d);
        const u32 nb0 = (d.s_hi - d.s_lo) < JP_META ? (d.s_hi - d.s_lo) : JP_META;
        if (lv && tid < nb0) pm0[tid] = a.probe.list[d.p0 + d.s_lo + tid];
        const u32 nbb0 = d.nbc < JB_META ? d.nbc : JB_META;
        if (lv && tid < nbb0) bm0[tid] = a.build.list[d.b0 + tid];
        reset_table();
        if (tid == 0) s_next[0] = atomicAdd(next_item, 1u) + gridDim.x;
        __syncthreads();
        if (lv) load_build(bm0, 0, nbb0);
    }

    for (;;) {
        // invariant: table empty, hdr reset, pm/bm[buf] hold d's first batches, bk[] = d's first build batch (in flight),
        // s_next[buf] = id of the item after d
        u32* pm = pm0 + buf * JP_META; u32* bm = bm0 + buf * JB_META;
        u32* pmn = pm0 + (buf ^ 1) * JP_META; u32* bmn = bm0 + (buf ^ 1) * JB_META;
        const bool lv = live(d);
        u64 ka[8], kb[8];
        u32 oka = 0, okb = 0;
        u32 nbatch = (d.s_hi - d.s_lo) < JP_META ? (d.s_hi - d.s_lo) : JP_META;
        u32 nrounds = (nbatch + CPR - 1) / CPR;
        if (lv) {
            load_round(pm, 0, nbatch, ka, oka);
            if (nrounds > 1) load_round(pm, 1, nbatch, kb, okb);
            // ---- build ----
            u32 nbb = d.nbc < JB_META ? d.nbc : JB_META;
            for (u32 bb = 0; bb < d.nbc; bb += JB_META) {
                if (bb) {
                    nbb = (d.nbc - bb) < JB_META ? (d.nbc - bb) : JB_META;
                    __syncthreads();
                    if (tid < nbb) bm[tid] = a.build.list[d.b0 + bb + tid];
                    __syncthreads();
                }
                for (u32 c0 = 0; c0 < nbb; c0 += 16) {
                    if (bb | c0) load_build(bm, c0, nbb);
#pragma unroll
                    for (u32 j = 0; j < BKPT; ++j) {
                        if (bok & (1u << j)) {
                            if (bk[j] == FJ_EMPTY_KEY) { hdr->has_empty = 1; if (a.want_dups && atomicAdd(&hdr->empties, 1u) > 0) hdr->dups = 1; }
                            else cuckoo_claim(tkeys, bits, ovf, hdr, bk[j]);
                        }
                    }
                }
            }
        }
        __syncthreads();                                     // phase-1 stores have landed
        if (lv) cuckoo_finish<NT>(tkeys, ovf, hdr, tid);
        __syncthreads();                                     // table complete; s_next[buf] visible
        // ---- P1: request the next item's list entries ----
        const Desc dn = describe(__builtin_amdgcn_readfirstlane(s_next[buf]));   // uniform: keeps the descriptor in SGPRs
        const bool lvn = live(dn);
        const u32 nbn = (dn.s_hi - dn.s_lo) < JP_META ? (dn.s_hi - dn.s_lo) : JP_META;
        const u32 nbbn = dn.nbc < JB_META ? dn.nbc : JB_META;
        u32 mp = 0, mb = 0;
        if (lvn && tid < nbn) mp = a.probe.list[dn.p0 + dn.s_lo + tid];
        if (lvn && tid < nbbn) mb = a.build.list[dn.b0 + tid];
        bool parked = false;                                  // next item's list entries still in registers?
        auto park = [&]() {
            if (tid < nbn) pmn[tid] = mp;
            if (tid < nbbn) bmn[tid] = mb;
            if (tid == 0) s_next[buf ^ 1] = atomicAdd(next_item, 1u) + gridDim.x;
            parked = true;
        };

        u32 wave_hits = 0;
        bool full = false;
        if (lv) {
            full = hdr->full != 0;
            const u64 he = hdr->has_empty ? ~0ull : 0ull;
            const u32 nstash = hdr->nstash < CK_STASH ? hdr->nstash : CK_STASH;
            if (a.want_dups && !full) {
                for (u32 sl = tid; sl < S; sl += NT) {
                    const u64 key = tkeys[sl];
                    if (key == FJ_EMPTY_KEY) continue;
                    const u32 w = FJ_HW2(key), l1 = w & (S - 1), l2 = (w >> 13) & (S - 1);
                    const u32 other = sl == l1 ? l2 : l1;
                    if (other != sl && tkeys[other] == key) hdr->dups = 1;
                    for (u32 si = 0; si < nstash; ++si) if (hdr->stash[si] == key) hdr->dups = 1;
                }
                for (u32 si = tid; si < nstash; si += NT)
                    for (u32 sj = si + 1; sj < nstash; ++sj) if (hdr->stash[si] == hdr->stash[sj]) hdr->dups = 1;
                __syncthreads();
                if (tid == 0 && hdr->dups) atomicOr(a.err, FJ_STAT_DUPS);
            }
            // ---- probe ----
            if (!full) for (u32 pb = d.s_lo; pb < d.s_hi; pb += JP_META) {
                if (pb != d.s_lo) {
                    nbatch = (d.s_hi - pb) < JP_META ? (d.s_hi - pb) : JP_META;
                    __syncthreads();
                    if (tid < nbatch) pm[tid] = a.probe.list[d.p0 + pb + tid];
                    __syncthreads();
                    nrounds = (nbatch + CPR - 1) / CPR;
                    load_round(pm, 0, nbatch, ka, oka);
                    if (nrounds > 1) load_round(pm, 1, nbatch, kb, okb);
                }
                for (u32 r = 0; r < nrounds; ++r) {
                    u64 k[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) { k[i] = ka[i]; ka[i] = kb[i]; }
                    const u32 okm = oka;
                    oka = okb;
                    if (r + 2 < nrounds) load_round(pm, r + 2, nbatch, kb, okb);
#pragma unroll
                    for (int h = 0; h < 8; h += 4) {              // two halves of 4 keys: 8 LDS reads in flight per lane
                        u64 c1[4], c2[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const u32 w = FJ_HW2(k[h + i]);
                            c1[i] = tkeys[w & (S - 1)];
                            c2[i] = tkeys[(w >> 13) & (S - 1)];
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            u64 hit = __ballot(c1[i] == k[h + i]) | __ballot(c2[i] == k[h + i]);
                            if (nstash) {
                                bool f = false;
                                for (u32 si = 0; si < nstash; ++si) f |= hdr->stash[si] == k[h + i];
                                hit |= __ballot(f);
                            }
                            const u64 ise = __ballot(k[h + i] == FJ_EMPTY_KEY);
                            const u64 ok = __ballot((okm >> (h + i)) & 1u);
                            wave_hits += (u32)__popcll(ok & ((hit & ~ise) | (ise & he)));
                        }
                    }
                    if (!parked) park();                      // after the first round: the entries have long arrived
                }
            }
        }
        if (!parked) park();                                  // skipped probe loop
        // ---- finish item d; request the next item's build keys as soon as its entries are visible ----
        if (lane == 0 && wave_hits) atomicAdd(&hdr->cnt, wave_hits);
        __syncthreads();
        if (lvn) load_build(bmn, 0, nbbn); else bok = 0;
        if (tid == 0) {
            const u32 cnt = (lv && !full) ? hdr->cnt : 0u;
            if (full) atomicOr(a.err, FJ_STAT_RETRY);       // stash overflow: the item is redone with the tagged table
            a.part_count[d.item] = full ? FJ_ITEM_RETRY : cnt;
            if (cnt) atomicAdd(a.total, (unsigned long long)cnt);
        }
        if (dn.item >= nitems) break;
        __syncthreads();                                     // hdr->cnt read before the reset
        reset_table();
        __syncthreads();
        d = dn; buf ^= 1;
    }
}

// ---- materialising join over chunk lists: the emitting pass on the cuckoo table, persistent form ----------------------
// The tagged table of fj_lds_join_kernel<true> costs about three times the issued instructions of a cuckoo lookup, and its
// 128 KiB (keys + values) leave one workgroup per CU: at c3 the emitting pass took 4.3 ms where the counting pass (cuckoo,
// keys only) takes 1.8, and dropping the output stores changed it by 0.4 ms only (ablation FJ_JOIN_ABLATE=4) - the table is
// what costs.  Here the emitting pass keeps the cuckoo KEY table and adds a value array indexed by the same slot:
//   1. build keys go in exactly as in fj_count_join_persistent (evictions move bare keys with one 64-bit exchange each);
//   2. after the barrier every build row looks its key up again (two reads) and drops its value into the slot the key ended
//      up in (or the stash's value row) - plain stores, no races for unique keys;
//   3. a probe key reads its two candidate slots; a hit fetches the value of the matching slot.
// A table whose stash overflows marks its item (part_count = FJ_ITEM_RETRY, FJ_STAT_EMIT_RETRY) and
// the host runs the tagged kernel over the marked items.  Resident workgroups with next-item prefetch as in
// fj_count_join_persistent; output positions = scanned per-item offsets + an LDS cursor bumped once per wave and 4 key slots.
struct EkHdr { u32 has_empty, nstash, full, dups, cursor, novf, cur2[2]; u64 empty_val; u64 gb2[2]; u64 pad; u64 stash[CK_STASH]; u64 stash_val[CK_STASH]; };

// DEDUP (the counting pass saw duplicate build keys): the build "values" are original row indices, every copy of a key
// lowers its slot's index with an LDS atomic minimum, and the winners are turned into values with one gather from the
// caller's build_values - the reference's first-occurrence rule (hash_join.cpp:125 on a stable partition).
// SINGLE: the single-pass materialising join - no counting pass ran and nobody knows the items' match counts.  Every probe
// round (8 keys per thread) counts first: the waves add their hit counts to an LDS cursor (which hands each its offset inside
// the round), after ONE barrier thread 0 reserves the round's range on the GLOBAL cursor a.out_cursor (one returning atomic
// per workgroup and round: ~130K per 1B probe rows, far below the 83 M/s one cursor sustains, tools/ubench_atomic_cursor.hip)
// and does not wait for it: the round's keys, hit bits and 16-bit slot codes stay in registers as the PENDING round while the
// next round is loaded and probed; the range is published behind the next round's barrier and the pending pairs are written
// then (values fetched from the table by slot code).  An item's last round is written behind the table reset for the next
// item (tvals, stash_val and empty_val survive that reset).  So a round costs one barrier and the atomic's ~1-2 us round trip
// hides under a whole round of LDS lookups.  Pair order is unspecified (as everywhere); the cursor ends as the match count.
// Duplicate build keys are reported exactly (FJ_STAT_DUPS, the sweep of the counting kernels) and the host then discards the
// output and runs the two-pass first-occurrence path.
template <int NT, bool DEDUP, bool SINGLE = false>
__global__ __launch_bounds__(NT, 4) void fj_emit_join_persistent(FjLdsJoinArgs a, u32* __restrict__ next_item) {
    static_assert(!(SINGLE && DEDUP), "the single-pass form serves unique build keys");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    EkHdr* hdr = reinterpret_cast<EkHdr*>(smem);
    u64* tkeys = reinterpret_cast<u64*>(smem + sizeof(EkHdr));
    u64* tvals = tkeys + S;
    u32* pm0 = reinterpret_cast<u32*>(tvals + S);           // [2][JP_META] probe-side list entries
    u32* bm0 = pm0 + 2 * JP_META;                            // [2][JB_META] build-side list entries
    u32* s_next = bm0 + 2 * JB_META;                         // [2] item ids handed out by the global counter
    u32* bits = s_next + 4;                                  // slot bitmap of the two-phase build
    u64* ovf = reinterpret_cast<u64*>(bits + S / 32);        // its overflow list
    // tid is NOT const: behind an item's probe rounds it is made afresh from the wave number (a scalar) and the lane id (fresh_tid).
    // With 1024 threads the kernel lives at the 128-register cap; the compiler kept the thread id and nine addresses derived from it
    // alive across the rounds by SPILLING them, and their reloads behind the loop came with s_waitcnt vmcnt(0): once per item every
    // wave waited for all its loads and pair stores in flight.  Recomputed (3 instructions), nothing needs to survive the loop.
    u32 tid = threadIdx.x;
    const u32 lane = tid & 63, wave_s = (u32)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    auto fresh_tid = [&]() -> u32 {
        u32 t;                                               // (the lane id too inside the asm: computed outside, it was hoisted and spilled)
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0\n\tv_lshl_add_u32 %0, %1, 6, %0" : "=&v"(t) : "s"(wave_s));
        return t;
    };
    const u32 nitems = *a.nitems_dev;
    constexpr u32 CPL = NT / (FJ_CHUNK / 2), CPR = 4 * CPL, BKPT = 4096 / NT;
    static_assert(JP_META <= NT && JB_META <= NT, "one staged list entry per thread");

    struct Desc { u32 item, b0, nbc, s_lo, s_hi, cnt; };
    auto describe = [&](u32 item) -> Desc {                  // wave-uniform scalar loads
        Desc d; d.item = item; d.b0 = 0; d.nbc = 0; d.s_lo = 0; d.s_hi = 0; d.cnt = 0;
        if (item >= nitems) return d;
        const uint4 it = a.items[item];
        d.b0 = a.build.boff[it.z]; d.nbc = a.build.boff[it.z + 1] - d.b0;
        d.s_lo = it.x; d.s_hi = it.x + it.y;
        d.cnt = SINGLE ? 1u : a.part_count[item];            // the counting pass: items without a match emit nothing
        return d;
    };
    auto live = [&](const Desc& d) { return d.item < nitems && d.nbc > 0 && d.s_lo < d.s_hi && d.cnt != 0; };

    u64 bk[BKPT], bv[BKPT];
    u32 bok = 0;
    auto load_build = [&](const u32* bm, u32 c0, u32 nbb) {  // 16 chunks = 4096 rows requested at once, unconditionally
        bok = 0;
#pragma unroll
        for (u32 j = 0; j < BKPT; ++j) {
            const u32 kidx = j * NT + tid, c = c0 + (kidx >> FJ_CHUNK_LOG), off = kidx & (FJ_CHUNK - 1);
            const u32 e = bm[c < nbb ? c : nbb - 1];
            const u64 src = (u64)FJ_LIST_ID(e) * FJ_CHUNK + off;
            bk[j] = a.build.keys[src];
            bv[j] = a.build.vals[src];
            bok |= (c < nbb && off < FJ_LIST_CNT(e) ? 1u : 0u) << j;
        }
    };
    auto load_round = [&](const u32* pm, u32 r, u32 nbatch, u64 (&kk)[8], u32& okm) {
        okm = 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const u32 c = r * CPR + u * CPL + tid / (FJ_CHUNK / 2), off = (tid % (FJ_CHUNK / 2)) * 2;
            const u32 e = pm[c < nbatch ? c : nbatch - 1], cnt = c < nbatch ? FJ_LIST_CNT(e) : 0;
            const u64x2 q = *reinterpret_cast<const u64x2*>(a.probe.keys + (u64)FJ_LIST_ID(e) * FJ_CHUNK + off);
            kk[2 * u] = q.x; kk[2 * u + 1] = q.y;
            okm |= ((off < cnt ? 1u : 0u) | (off + 1 < cnt ? 2u : 0u)) << (2 * u);
        }
    };
    auto reset_table = [&]() {
        u32 m1;                                              // all ones, made on the spot (the hoisted 64-bit constant was spilled as well)
        asm volatile("v_mov_b32 %0, -1" : "=v"(m1));
        const u64 empty = ((u64)m1 << 32) | m1;
        static_assert(FJ_EMPTY_KEY == ~0ull, "the empty marker is all ones");
        for (u32 i = tid; i < S; i += NT) { tkeys[i] = empty; if (DEDUP) tvals[i] = ~0ull; }
        if (DEDUP && tid < CK_STASH) hdr->stash_val[tid] = ~0ull;
        if (tid < S / 32) bits[tid] = 0;
        if (tid == 0) {
            hdr->has_empty = 0; hdr->nstash = 0; hdr->full = 0; hdr->dups = 0; hdr->cursor = 0; hdr->novf = 0; hdr->cur2[0] = 0; hdr->cur2[1] = 0;
            if (!SINGLE) hdr->empty_val = DEDUP ? ~0ull : 0ull;    // (SINGLE: a pending round may still need the old one; read only under has_empty)
        }
    };
    // step 2: the value of build row (key, val) goes where the key lives now (table before stash, first location before the
    // second: the order the probe uses).  DEDUP: val is the row index, the smallest one stays.
    auto put = [&](u64* where, u64 val) {
        if (DEDUP) atomicMin((unsigned long long*)where, (unsigned long long)val); else *where = val;
    };
    auto place = [&](u64 key, u64 val) {
        if (key == FJ_EMPTY_KEY) { put(&hdr->empty_val, val); return; }
        const u32 w = FJ_HW2(key), l1 = w & (S - 1), l2 = (w >> 13) & (S - 1);
        if (tkeys[l1] == key) { put(&tvals[l1], val); return; }
        if (tkeys[l2] == key) { put(&tvals[l2], val); return; }
        const u32 ns = hdr->nstash < CK_STASH ? hdr->nstash : CK_STASH;
        for (u32 si = 0; si < ns; ++si) if (hdr->stash[si] == key) { put(&hdr->stash_val[si], val); return; }
    };

    // ---- prologue: first item (static), its list entries, its first build batch ----
    Desc d = describe(blockIdx.x);
    if (d.item >= nitems) return;
    u32 buf = 0;
    {
        const bool lv = live(d);
        const u32 nb0 = (d.s_hi - d.s_lo) < JP_META ? (d.s_hi - d.s_lo) : JP_META;
        if (lv && tid < nb0) pm0[tid] = a.probe.list[d.s_lo + tid];
        const u32 nbb0 = d.nbc < JB_META ? d.nbc : JB_META;
        if (lv && tid < nbb0) bm0[tid] = a.build.list[d.b0 + tid];
        reset_table();
        if (tid == 0) s_next[0] = atomicAdd(next_item, 1u) + gridDim.x;
        __syncthreads();
        if (lv) load_build(bm0, 0, nbb0);
    }

    for (;;) {
        // invariant: table empty, hdr reset, pm/bm[buf] hold d's first batches, bk/bv[] = d's first build batch (in flight),
        // s_next[buf] = id of the item after d
        u32* pm = pm0 + buf * JP_META; u32* bm = bm0 + buf * JB_META;
        u32* pmn = pm0 + (buf ^ 1) * JP_META; u32* bmn = bm0 + (buf ^ 1) * JB_META;
        const bool lv = live(d);
        const bool one_batch = d.nbc <= 16;                  // all build rows of the partition are in bk/bv: step 2 needs no reload
        u64 ka[8], kb[8];
        u32 oka = 0, okb = 0;
        u32 nbatch = (d.s_hi - d.s_lo) < JP_META ? (d.s_hi - d.s_lo) : JP_META;
        u32 nrounds = (nbatch + CPR - 1) / CPR;
        if (lv) {
            load_round(pm, 0, nbatch, ka, oka);
            if (!SINGLE && nrounds > 1) load_round(pm, 1, nbatch, kb, okb);
            // ---- build, step 1: keys ----
            u32 nbb = d.nbc < JB_META ? d.nbc : JB_META;
            for (u32 bb = 0; bb < d.nbc; bb += JB_META) {
                if (bb) {
                    nbb = (d.nbc - bb) < JB_META ? (d.nbc - bb) : JB_META;
                    __syncthreads();
                    if (tid < nbb) bm[tid] = a.build.list[d.b0 + bb + tid];
                    __syncthreads();
                }
                for (u32 c0 = 0; c0 < nbb; c0 += 16) {
                    if (bb | c0) load_build(bm, c0, nbb);
#pragma unroll
                    for (u32 j = 0; j < BKPT; ++j) {
                        if (bok & (1u << j)) {
                            if (bk[j] == FJ_EMPTY_KEY) hdr->has_empty = 1;
                            else cuckoo_claim(tkeys, bits, ovf, hdr, bk[j]);
                        }
                    }
                }
            }
        }
        __syncthreads();                                     // phase-1 stores have landed
        if (lv) cuckoo_finish<NT>(tkeys, ovf, hdr, tid);
        __syncthreads();                                     // keys are final; s_next[buf] visible
        // ---- P1: request the next item's list entries ----
        const Desc dn = describe(__builtin_amdgcn_readfirstlane(s_next[buf]));   // uniform: keeps the descriptor in SGPRs
        const bool lvn = live(dn);
        const u32 nbn = (dn.s_hi - dn.s_lo) < JP_META ? (dn.s_hi - dn.s_lo) : JP_META;
        const u32 nbbn = dn.nbc < JB_META ? dn.nbc : JB_META;
        u32 mp = 0, mb = 0;
        if (lvn && tid < nbn) mp = a.probe.list[dn.s_lo + tid];
        if (lvn && tid < nbbn) mb = a.build.list[dn.b0 + tid];
        bool parked = false;                                  // next item's list entries still in registers?
        auto park = [&]() {
            if (tid < nbn) pmn[tid] = mp;
            if (tid < nbbn) bmn[tid] = mb;
            if (tid == 0) s_next[buf ^ 1] = atomicAdd(next_item, 1u) + gridDim.x;
            parked = true;
        };

        // SINGLE: the pending round (counted, range requested, pairs not written yet)
        u64 kp[8];
        u32 scp[4] = {0u, 0u, 0u, 0u}, hbp = 0, wbp = 0, par = 0;
        bool pend = false;
        u64 gbv = 0; u32 totp = 0, icnt = 0;                 // thread 0: the range request in flight, its size, the item's matches so far
        auto settle = [&]() {                                 // thread 0 publishes the pending round's range (waits for the atomic here)
            if (tid == 0 && pend) {
                u64 gb = gbv;
                if (totp && gb + totp > a.out_capacity) { atomicOr(a.err, FJ_ERR_OUTCAP); gb = ~0ull; }
                hdr->gb2[par ^ 1] = gb;
            }
        };
        auto flush = [&]() {                                  // behind the barrier that follows settle(): write the pending pairs
            const u64 gb = hdr->gb2[par ^ 1];
            u32 pre = 0;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const bool bit = (hbp >> i) & 1u;
                const u64 m = __ballot(bit);
                if (bit && gb != ~0ull) {
                    const u32 code = (scp[i >> 1] >> ((i & 1) * 16)) & 0xFFFFu;
                    u64 val;
                    if (code < S) val = tvals[code];
                    else if (code == 0xFFFFu) val = hdr->empty_val;
                    else val = hdr->stash_val[code - S];
                    const u64 o = gb + wbp + pre + (u32)__popcll(m & ((1ull << lane) - 1ull));
                    a.out_keys[o] = fj_key_unmix(kp[i]);
                    a.out_vals[o] = val;
                }
                pre += (u32)__popcll(m);
            }
        };

        bool full = false;
        if (lv) {
            full = hdr->full != 0 || ((a.dbg_flags & 8u) && d.item % 7u == 3u);   // (8: test hook - every 7th item takes the retry path)
            // ---- build, step 2: values ----
            if (!full) {
                if (one_batch) {
#pragma unroll
                    for (u32 j = 0; j < BKPT; ++j) if (bok & (1u << j)) place(bk[j], bv[j]);
                } else {
                    u32 nbb = d.nbc < JB_META ? d.nbc : JB_META;
                    for (u32 bb = 0; bb < d.nbc; bb += JB_META) {
                        nbb = (d.nbc - bb) < JB_META ? (d.nbc - bb) : JB_META;
                        if (d.nbc > JB_META) {               // (the staged entries were overwritten by later batches)
                            __syncthreads();
                            if (tid < nbb) bm[tid] = a.build.list[d.b0 + bb + tid];
                            __syncthreads();
                        }
                        for (u32 c0 = 0; c0 < nbb; c0 += 16) {
                            load_build(bm, c0, nbb);
#pragma unroll
                            for (u32 j = 0; j < BKPT; ++j) if (bok & (1u << j)) place(bk[j], bv[j]);
                        }
                    }
                }
            }
        }
        __syncthreads();                                     // values are in place
        if (DEDUP && lv && !full) {                          // winning row indices -> values (one gather from the caller's array)
            {   // all gathers of a thread in flight together (unconditional loads: row 0 stands in where there is nothing to fetch)
                u64 v[S / NT]; u32 okm = 0;
#pragma unroll
                for (u32 j = 0; j < S / NT; ++j) {
                    const u32 i = tid + j * NT;
                    const u64 r = tvals[i];
                    const bool ok = tkeys[i] != FJ_EMPTY_KEY && r != ~0ull;
                    v[j] = a.orig_vals[ok ? r : 0ull];
                    okm |= (ok ? 1u : 0u) << j;
                }
#pragma unroll
                for (u32 j = 0; j < S / NT; ++j) if (okm & (1u << j)) tvals[tid + j * NT] = v[j];
            }
            if (tid < CK_STASH && tid < hdr->nstash) { const u64 r = hdr->stash_val[tid]; if (r != ~0ull) hdr->stash_val[tid] = a.orig_vals[r]; }
            if (tid == 0 && hdr->has_empty && hdr->empty_val != ~0ull) hdr->empty_val = a.orig_vals[hdr->empty_val];
            __syncthreads();
        }
        if (SINGLE && lv && !full) {
            // exact duplicate report (as the counting kernels do it): a second copy of a key was noticed by its insert, or it
            // raced past the check and the key now sits in both of its slots / the stash
            const u32 nst = hdr->nstash < CK_STASH ? hdr->nstash : CK_STASH;
            for (u32 sl = tid; sl < S; sl += NT) {
                const u64 key = tkeys[sl];
                if (key == FJ_EMPTY_KEY) continue;
                const u32 w = FJ_HW2(key), l1 = w & (S - 1), l2 = (w >> 13) & (S - 1);
                const u32 other = sl == l1 ? l2 : l1;
                if (other != sl && tkeys[other] == key) hdr->dups = 1;
                for (u32 si = 0; si < nst; ++si) if (hdr->stash[si] == key) hdr->dups = 1;
            }
            for (u32 si = tid; si < nst; si += NT)
                for (u32 sj = si + 1; sj < nst; ++sj) if (hdr->stash[si] == hdr->stash[sj]) hdr->dups = 1;
            __syncthreads();
            if (tid == 0 && hdr->dups) atomicOr(a.err, FJ_STAT_DUPS);
        }
        if (lv && full) {                                    // stash overflow: the host redoes this item on the tagged table
            if (tid == 0) { atomicOr(a.err, FJ_STAT_EMIT_RETRY); a.part_count[d.item] = FJ_ITEM_RETRY; }
        } else if (lv) {
            const u64 he = hdr->has_empty ? ~0ull : 0ull;
            const u32 nstash = hdr->nstash < CK_STASH ? hdr->nstash : CK_STASH;
            const u64 obase = SINGLE ? 0ull : a.out_off[d.item];
            if constexpr (SINGLE) {
                // ---- probe; count; request the range; write the round before ----
                for (u32 pb = d.s_lo; pb < d.s_hi; pb += JP_META) {
                    if (pb != d.s_lo) {
                        nbatch = (d.s_hi - pb) < JP_META ? (d.s_hi - pb) : JP_META;
                        __syncthreads();
                        if (tid < nbatch) pm[tid] = a.probe.list[pb + tid];
                        __syncthreads();
                        nrounds = (nbatch + CPR - 1) / CPR;
                        load_round(pm, 0, nbatch, ka, oka);
                    }
                    for (u32 r = 0; r < nrounds; ++r) {
                        u64 k[8];
#pragma unroll
                        for (int i = 0; i < 8; ++i) k[i] = ka[i];
                        const u32 okm = oka;
                        if (r + 1 < nrounds) load_round(pm, r + 1, nbatch, ka, oka);   // a whole round ahead of its use
                        u32 hb = 0, sc[4] = {0u, 0u, 0u, 0u}, nw = 0;
#pragma unroll
                        for (int h = 0; h < 8; h += 4) {          // two halves of 4 keys: 8 LDS reads in flight per lane
                            u64 c1[4], c2[4];
                            u32 l1[4], l2[4];
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const u32 w = FJ_HW2(k[h + i]);
                                l1[i] = w & (S - 1); l2[i] = (w >> 13) & (S - 1);
                                c1[i] = tkeys[l1[i]];
                                c2[i] = tkeys[l2[i]];
                            }
                            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const u64 key = k[h + i];
                                const bool h1 = c1[i] == key, h2 = c2[i] == key;
                                bool hit = h1 | h2;
                                u32 code = h1 ? l1[i] : l2[i];
                                if (nstash) {                            // (table before stash: the order step 2 used)
                                    for (u32 si = 0; si < nstash; ++si) if (!hit && hdr->stash[si] == key) { hit = true; code = S + si; }
                                }
                                if (key == FJ_EMPTY_KEY) { hit = he != 0; code = 0xFFFFu; }   // the empty marker is never stored in the table
                                hit = hit && ((okm >> (h + i)) & 1u);
                                nw += (u32)__popcll(__ballot(hit));
                                hb |= (hit ? 1u : 0u) << (h + i);
                                sc[(h + i) >> 1] |= code << (((h + i) & 1) * 16);
                            }
                        }
                        u32 wb = 0;
                        if (lane == 0 && nw) wb = atomicAdd(&hdr->cur2[par], nw);
                        wb = (u32)__builtin_amdgcn_readfirstlane((int)wb);
                        settle();                                 // the round before: its range must have arrived by now
                        __syncthreads();                          // every wave has counted this round; the round before has its range
                        if (tid == 0) {
                            totp = hdr->cur2[par]; hdr->cur2[par] = 0;      // (next used two rounds on, behind another barrier)
                            icnt += totp;
                            gbv = totp ? atomicAdd(a.out_cursor, (unsigned long long)totp) : 0ull;   // not waited for here
                        }
                        if (pend) flush();
#pragma unroll
                        for (int i = 0; i < 8; ++i) kp[i] = k[i];
#pragma unroll
                        for (int i = 0; i < 4; ++i) scp[i] = sc[i];
                        hbp = hb; wbp = wb; pend = true; par ^= 1u;
                        if (!parked) park();                      // after the first round: the entries have long arrived
                    }
                }
            } else {
                // ---- probe + emit ----
                for (u32 pb = d.s_lo; pb < d.s_hi; pb += JP_META) {
                    if (pb != d.s_lo) {
                        nbatch = (d.s_hi - pb) < JP_META ? (d.s_hi - pb) : JP_META;
                        __syncthreads();
                        if (tid < nbatch) pm[tid] = a.probe.list[pb + tid];
                        __syncthreads();
                        nrounds = (nbatch + CPR - 1) / CPR;
                        load_round(pm, 0, nbatch, ka, oka);
                        if (nrounds > 1) load_round(pm, 1, nbatch, kb, okb);
                    }
                    for (u32 r = 0; r < nrounds; ++r) {
                        u64 k[8];
    #pragma unroll
                        for (int i = 0; i < 8; ++i) { k[i] = ka[i]; ka[i] = kb[i]; }
                        const u32 okm = oka;
                        oka = okb;
                        if (r + 2 < nrounds) load_round(pm, r + 2, nbatch, kb, okb);
    #pragma unroll
                        for (int h = 0; h < 8; h += 4) {              // two halves of 4 keys: 8 LDS reads in flight per lane
                            u64 c1[4], c2[4];
                            u32 l1[4], l2[4];
    #pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const u32 w = FJ_HW2(k[h + i]);
                                l1[i] = w & (S - 1); l2[i] = (w >> 13) & (S - 1);
                                c1[i] = tkeys[l1[i]];
                                c2[i] = tkeys[l2[i]];
                            }
                            __builtin_amdgcn_sched_barrier(0);
                            u64 hitm[4];
                            u64 val[4];
    #pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const u64 key = k[h + i];
                                const bool h1 = c1[i] == key, h2 = c2[i] == key;
                                bool hit = h1 | h2;
                                val[i] = tvals[h1 ? l1[i] : l2[i]];          // (unconditional read: independent of the compare, harmless on a miss)
                                if (nstash) {                                // (table before stash: the order step 2 used)
                                    for (u32 si = 0; si < nstash; ++si) if (!hit && hdr->stash[si] == key) { hit = true; val[i] = hdr->stash_val[si]; }
                                }
                                const bool ise = key == FJ_EMPTY_KEY;        // the empty marker is never stored in the table
                                if (ise) { hit = he != 0; val[i] = hdr->empty_val; }
                                hitm[i] = __ballot(hit && ((okm >> (h + i)) & 1u));
                            }
                            // ONE LDS cursor bump per wave for the four key slots; lanes ranked inside the ballots
                            const u32 n0 = (u32)__popcll(hitm[0]), n1 = (u32)__popcll(hitm[1]), n2 = (u32)__popcll(hitm[2]), n3 = (u32)__popcll(hitm[3]);
                            if (n0 + n1 + n2 + n3) {
                                u32 wb = 0;
                                if (lane == 0) wb = atomicAdd(&hdr->cursor, n0 + n1 + n2 + n3);
                                wb = (u32)__builtin_amdgcn_readfirstlane((int)wb);
                                const u32 off[4] = {0u, n0, n0 + n1, n0 + n1 + n2};
    #pragma unroll
                                for (int i = 0; i < 4; ++i) {
                                    const u64 m = hitm[i];
                                    if ((m >> lane) & 1ull) {
                                        const u64 o = obase + wb + off[i] + (u32)__popcll(m & ((1ull << lane) - 1ull));
                                        a.out_keys[o] = fj_key_unmix(k[h + i]);
                                        a.out_vals[o] = val[i];
                                    }
                                }
                            }
                        }
                        if (!parked) park();                      // after the first round: the entries have long arrived
                    }
                }
            }
            }
        tid = fresh_tid();                                    // (see the declaration of tid)
        if (!parked) park();                                  // skipped probe loop
        __syncthreads();                                      // every wave is done with the table; the parked entries are visible
        if (SINGLE && tid == 0 && d.item < nitems && !(lv && full)) a.part_count[d.item] = lv ? icnt : 0u;
        if (lvn) load_build(bmn, 0, nbbn); else bok = 0;      // the next item's build rows fly during the reset
        const bool last = dn.item >= nitems;
        if (!SINGLE && last) break;
        if (!last) reset_table();
        if (SINGLE) settle();                                 // the item's last round: its range arrives under the reset
        __syncthreads();
        if (SINGLE && pend) flush();
        if (last) break;
        d = dn; buf ^= 1;
    }
}

// ---- hit-rate sample for the adaptive joins' bloom decision ----------------------------------------------------------------
__global__ __launch_bounds__(256) void fj_sample_hits_kernel(FjChunkSet build, const u64* __restrict__ pk, u64 np, u32 nsamples,
                                                             u32 shift32, u32 pmask, unsigned long long* __restrict__ hits) {
    const u32 lane = threadIdx.x & 63, widx = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (widx >= nsamples) return;
    const u64 key = fj_key_mix(pk[(u64)widx * (np / nsamples)]);   // wave-uniform; the build side's chunks hold mixed keys
    const u32 p = (FJ_HW1(key) >> shift32) & pmask;
    const u32 b0 = build.boff[p], nbc = build.boff[p + 1] - b0;
    bool found = false;
    for (u32 c = 0; c < nbc; ++c) {
        const u32 e = build.list[b0 + c], cnt = FJ_LIST_CNT(e);
        const u64* ck = build.keys + (u64)FJ_LIST_ID(e) * FJ_CHUNK;
        for (u32 o = lane; o < cnt; o += 64) found |= ck[o] == key;
    }
    if (__ballot(found) && lane == 0) atomicAdd(hits, 1ull);
}

// =============================== global (non-partitioned) table ===============================
__device__ __forceinline__ u32 gt_bloom_mask(u64 h) {      // 3 bits of a 32-bit word per 8-slot group
    return (1u << ((h >> 40) & 31)) | (1u << ((h >> 45) & 31)) | (1u << ((h >> 50) & 31));
}

__global__ __launch_bounds__(256) void fj_gt_build_kernel(FjGtArgs a) {
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < a.nb; i += stride) {
        const u64 key = a.bk[i], val = a.bv[i];
        if (key == FJ_EMPTY_KEY) { a.flags[0] = 1; *a.empty_val = val; continue; }
        const u64 h = fj_hash64(key);
        const u64 home = (h & a.cap_mask) & ~(u64)(FJ_GT_GROUP - 1);
        u64 pos = home;
        for (u64 step = 0; step <= a.cap_mask; ++step) {
            const u64 old = atomicCAS((unsigned long long*)&a.tkeys[pos], (unsigned long long)FJ_EMPTY_KEY, (unsigned long long)key);
            if (old == FJ_EMPTY_KEY) {
                a.tvals[pos] = val;
                if (a.bloom) atomicOr(&a.bloom[home >> 3], gt_bloom_mask(h));
                break;
            }
            if (old == key) break;
            pos = (pos + 1) & a.cap_mask;
        }
    }
}

__device__ __forceinline__ bool gt_lookup(const u64* __restrict__ tkeys, u64 cap_mask, u64 key, u64 h, u64& where) {
    u64 pos = (h & cap_mask) & ~(u64)(FJ_GT_GROUP - 1);
    const u64 ngroups = (cap_mask + 1) / FJ_GT_GROUP;
    for (u64 step = 0; step < ngroups; ++step) {
        const u64x2* g = reinterpret_cast<const u64x2*>(tkeys + pos);
        const u64x2 q0 = g[0], q1 = g[1], q2 = g[2], q3 = g[3];
        if (q0.x == key) { where = pos; return true; }
        if (q0.y == key) { where = pos + 1; return true; }
        if (q1.x == key) { where = pos + 2; return true; }
        if (q1.y == key) { where = pos + 3; return true; }
        if (q2.x == key) { where = pos + 4; return true; }
        if (q2.y == key) { where = pos + 5; return true; }
        if (q3.x == key) { where = pos + 6; return true; }
        if (q3.y == key) { where = pos + 7; return true; }
        const bool e = q0.x == FJ_EMPTY_KEY || q0.y == FJ_EMPTY_KEY || q1.x == FJ_EMPTY_KEY || q1.y == FJ_EMPTY_KEY ||
                       q2.x == FJ_EMPTY_KEY || q2.y == FJ_EMPTY_KEY || q3.x == FJ_EMPTY_KEY || q3.y == FJ_EMPTY_KEY;
        if (e) return false;
        pos = (pos + FJ_GT_GROUP) & cap_mask;
    }
    return false;
}

template <bool MAT, bool BLOOM>
__global__ __launch_bounds__(256) void fj_gt_probe_kernel(FjGtArgs a) {
    __shared__ u32 s_cnt, s_cursor;
    const u32 tid = threadIdx.x, lane = tid & 63, g = blockIdx.x, G = gridDim.x;
    if (tid == 0) { s_cnt = 0; s_cursor = 0; }
    __syncthreads();
    const u64 npairs = (a.np + 1) / 2;
    const u64 per = npairs / G, rem = npairs % G;
    const u64 lo = (u64)g * per + (g < rem ? g : rem), hi = lo + per + (g < rem ? 1 : 0);
    const bool has_empty = a.flags[0] != 0;
    const u64 obase = MAT ? a.out_off[g] : 0;
    u32 local = 0;
    for (u64 base = lo; base < hi; base += 256) {
        const u64 pi = base + tid;
        u64 k[2] = {0, 0};
        bool ok[2] = {false, false};
        if (pi < hi) {
            if (2 * pi + 1 < a.np) {
                const u64x2 kk = *reinterpret_cast<const u64x2*>(a.pk + 2 * pi);
                k[0] = kk.x; k[1] = kk.y; ok[0] = ok[1] = true;
            } else { k[0] = a.pk[2 * pi]; ok[0] = true; }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            bool hit = false;
            u64 val = 0;
            if (ok[i]) {
                if (k[i] == FJ_EMPTY_KEY) {
                    hit = has_empty;
                    if (MAT) val = *a.empty_val;
                } else {
                    const u64 h = fj_hash64(k[i]);
                    bool pass = true;
                    if (BLOOM) {       // precheck on the home group's bloom word (role of hash_join.cpp:185-189)
                        const u32 m = gt_bloom_mask(h);
                        pass = (a.bloom[(h & a.cap_mask) >> 3] & m) == m;
                    }
                    if (pass) {
                        u64 where = 0;
                        hit = gt_lookup(a.tkeys, a.cap_mask, k[i], h, where);
                        if (MAT && hit) val = a.tvals[where];
                    }
                }
            }
            if (MAT) {
                const u64 m = __ballot(hit);
                if (m) {
                    u32 wb = 0;
                    if (lane == 0) wb = atomicAdd(&s_cursor, (u32)__popcll(m));
                    wb = __shfl(wb, 0, 64);
                    if (hit) {
                        const u64 o = obase + wb + (u32)__popcll(m & ((1ull << lane) - 1ull));
                        a.out_keys[o] = k[i];
                        a.out_vals[o] = val;
                    }
                }
            } else {
                local += hit ? 1u : 0u;
            }
        }
    }
    if (!MAT) {
        local = fj_wave_sum(local);
        if (lane == 0 && local) atomicAdd(&s_cnt, local);
        __syncthreads();
        if (tid == 0) {
            a.wg_count[g] = s_cnt;
            if (s_cnt) atomicAdd(a.total, (unsigned long long)s_cnt);
        }
    }
}

// =============================== multi-GPU owner split ========================================
// lanes of this wave whose value d (< 2^(nbits-1), or exactly 2^(nbits-1) for "no key") equals mine: nbits ballots
__device__ __forceinline__ u64 fj_match_any(u32 d, u32 nbits) {
    u64 m = ~0ull;
    for (u32 b = 0; b < nbits; ++b) {
        const u64 bal = __ballot((d >> b) & 1u);
        m &= ((d >> b) & 1u) ? bal : ~bal;
    }
    return m;
}

__global__ __launch_bounds__(256) void fj_owner_hist_kernel(const u64* __restrict__ keys, u64 n, u32 nranks,
                                                            unsigned long long* __restrict__ counts) {
    __shared__ u32 h[64];
    const u32 tid = threadIdx.x;
    if (tid < 64) h[tid] = 0;
    __syncthreads();
    // counts are aggregated per wave with ballots: 64 lanes hitting <= nranks LDS counters would serialise
    const u32 lane = tid & 63;
    u32 nbits = 1; while ((1u << (nbits - 1)) < nranks) ++nbits;      // owner ids + one marker bit
    const u32 none = 1u << (nbits - 1);
    const u64 stride = (u64)gridDim.x * blockDim.x, npairs = n / 2;
    const u64 iters = (npairs + stride - 1) / stride;
    for (u64 it = 0; it < iters; ++it) {                  // wave-uniform trip count: every lane reaches the ballots
        const u64 i = it * stride + (u64)blockIdx.x * blockDim.x + tid;
        u32 d0 = 0xFFFFFFFFu, d1 = 0xFFFFFFFFu;
        if (i < npairs) {
            const u64x2 q = *reinterpret_cast<const u64x2*>(keys + 2 * i);
            d0 = fj_owner_of_w1(fj_hash_w1(q.x), nranks);
            d1 = fj_owner_of_w1(fj_hash_w1(q.y), nranks);
        }
        // d = none marks "no key" (top bit set): such lanes only match each other
        { const u32 d = d0 == 0xFFFFFFFFu ? none : d0; const u64 m = fj_match_any(d, nbits);
          if (d < none && lane == (u32)__builtin_ctzll(m)) atomicAdd(&h[d], (u32)__popcll(m)); }
        { const u32 d = d1 == 0xFFFFFFFFu ? none : d1; const u64 m = fj_match_any(d, nbits);
          if (d < none && lane == (u32)__builtin_ctzll(m)) atomicAdd(&h[d], (u32)__popcll(m)); }
    }
    if ((n & 1) && blockIdx.x == 0 && tid == 0) atomicAdd(&h[fj_owner_of_w1(fj_hash_w1(keys[n - 1]), nranks)], 1u);
    __syncthreads();
    if (tid < nranks && h[tid]) atomicAdd(&counts[tid], (unsigned long long)h[tid]);
}

// tile-local counting sort by owner in LDS, then each owner's run is written contiguously at a
// position reserved with ONE global atomic per (tile, owner).
template <bool HAS_VALS>
__global__ __launch_bounds__(512) void fj_owner_scatter_kernel(const u64* __restrict__ keys, const u64* __restrict__ vals,
                                                               u64 n, u32 nranks,
                                                               const unsigned long long* __restrict__ offsets,
                                                               unsigned long long* __restrict__ cursors,
                                                               u64* __restrict__ out_keys, u64* __restrict__ out_vals) {
    constexpr u32 NT = 512, KPT = HAS_VALS ? 4 : 8, T = NT * KPT;
    __shared__ u64 sk[T];
    __shared__ u64 sv[HAS_VALS ? T : 1];
    __shared__ u32 hist[64], toff[65];
    __shared__ u64 gbase[64];
    const u32 tid = threadIdx.x;
    u32 nbits = 1; while ((1u << (nbits - 1)) < nranks) ++nbits;
    const u32 none = 1u << (nbits - 1);
    const u64 ntiles = (n + T - 1) / T;
    for (u64 t = blockIdx.x; t < ntiles; t += gridDim.x) {
        if (tid < 64) hist[tid] = 0;
        __syncthreads();
        u64 k[KPT], v[KPT];
        u32 dr[KPT];
#pragma unroll
        for (u32 i = 0; i < KPT / 2; ++i) {                // 16 B per lane and load (the arrays are 16-B aligned)
            const u64 idx = t * T + ((u64)i * NT + tid) * 2;
            dr[2 * i] = dr[2 * i + 1] = 0xFFFFFFFFu;
            if (idx + 1 < n) {
                const u64x2 q = *reinterpret_cast<const u64x2*>(keys + idx);
                k[2 * i] = q.x; k[2 * i + 1] = q.y;
                if (HAS_VALS) { const u64x2 w = *reinterpret_cast<const u64x2*>(vals + idx); v[2 * i] = w.x; v[2 * i + 1] = w.y; }
            } else if (idx < n) {
                k[2 * i] = keys[idx];
                if (HAS_VALS) v[2 * i] = vals[idx];
            }
#pragma unroll
            for (u32 h = 0; h < 2; ++h) {                    // rank inside the tile: one LDS atomic per (wave, owner), lanes ranked by ballot
                const bool ok = idx + h < n;
                const u32 d = ok ? fj_owner_of_w1(fj_hash_w1(k[2 * i + h]), nranks) : none;
                const u64 m = fj_match_any(d, nbits);          // lanes of this wave going to the same owner
                const u32 leader = (u32)__builtin_ctzll(m), ln = tid & 63;
                u32 base = 0;
                if (ok && ln == leader) base = atomicAdd(&hist[d], (u32)__popcll(m));
                base = __shfl(base, leader, 64);
                if (ok) dr[2 * i + h] = (d << 16) | (base + (u32)__popcll(m & ((1ull << ln) - 1ull)));
            }
        }
        __syncthreads();
        if (tid == 0) {
            u32 run = 0;
            for (u32 d = 0; d < nranks; ++d) { toff[d] = run; run += hist[d]; }
            toff[nranks] = run;
        }
        if (tid < nranks && hist[tid])
            gbase[tid] = offsets[tid] + atomicAdd(&cursors[tid], (unsigned long long)hist[tid]);
        __syncthreads();
#pragma unroll
        for (u32 i = 0; i < KPT; ++i) {
            if (dr[i] != 0xFFFFFFFFu) {
                const u32 s = toff[dr[i] >> 16] + (dr[i] & 0xFFFFu);
                sk[s] = k[i];
                if (HAS_VALS) sv[s] = v[i];
            }
        }
        __syncthreads();
        const u32 total = toff[nranks];
        for (u32 e = tid; e < total; e += NT) {
            u32 d = 0;
            while (e >= toff[d + 1]) ++d;
            const u64 o = gbase[d] + (e - toff[d]);
            out_keys[o] = sk[e];
            if (HAS_VALS) out_vals[o] = sv[e];
        }
        __syncthreads();
    }
}

// =============================== synthetic data ==============================
__global__ void fj_iota_kernel(u64* __restrict__ out, u64 n) {
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = i;
}
__global__ void fj_gen_build_kernel(u64* __restrict__ keys, u64* __restrict__ vals, u64 first, u64 n) {
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        keys[i] = (first + i + 1) * FJ_GOLDEN;
        vals[i] = first + i;
    }
}
__global__ void fj_gen_probe_kernel(u64* __restrict__ keys, u64 first, u64 n, u64 B, u64 seed, u32 hit_bp,
                                    unsigned long long* __restrict__ expected) {
    const u64 stride = (u64)gridDim.x * blockDim.x;
    u32 hits = 0;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const u64 j = first + i;
        const u64 r = 1 + fj_mix(seed, j) % B;
        const bool hit = (fj_mix(seed ^ 1ull, j) % 10000ull) < hit_bp;
        keys[i] = (r + (hit ? 0ull : B)) * FJ_GOLDEN;
        hits += hit ? 1u : 0u;
    }
    hits = fj_wave_sum(hits);
    if ((threadIdx.x & 63) == 0 && hits) atomicAdd(expected, (unsigned long long)hits);
}

}  // namespace

hipError_t fj_launch_lds_join(const FjLdsJoinArgs& a, bool materialize, hipStream_t s, u32* next_item, u32 persistent_min_items) {
    const u32 nb = a.items ? a.items_cap : a.nparts * a.nsplit;       // grid of the one-workgroup-per-item kernels
    if (materialize) {
        // chunk lists on both sides, unique build keys: the cuckoo form (resident workgroups, one per CU, next item prefetched)
        if (a.build.list && a.probe.list && a.items && next_item && nb >= persistent_min_items && !a.dbg && !(a.dbg_flags & ~8u)) {
            const u32 ldsp = sizeof(EkHdr) + 2 * S * 8 + 2 * (JP_META + JB_META) * 4 + 16 + S / 8 + CK_OVF * 8;
            auto pk = a.dedup ? fj_emit_join_persistent<1024, true> : fj_emit_join_persistent<1024, false>;
            hipError_t e = fj_set_max_lds_once(reinterpret_cast<const void*>(pk), ldsp);
            if (e != hipSuccess) return e;
            const u32 grid = nb < 256 ? nb : 256;            // (*next_item is zero: the emitting pass has its own counter word)
            hipLaunchKernelGGL(pk, dim3(grid), dim3(1024), ldsp, s, a, next_item);
            return hipGetLastError();
        }
        const u32 lds = sizeof(JoinHdr) + 2 * S * 8 + S + S / 2 + (JP_META + JB_META) * 4;
        auto kern = (a.build.list && a.probe.list) ? fj_lds_join_kernel<true, 1024, true> : fj_lds_join_kernel<true, 1024, false>;
        hipError_t e = fj_set_max_lds_once(reinterpret_cast<const void*>(kern), lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3(nb), dim3(1024), lds, s, a);
    } else {
        const u32 lds = sizeof(CkHdr) + S * 8 + (JP_META + JB_META) * 4 + S / 8 + CK_OVF * 8;
        const bool lists = a.build.list && a.probe.list;
        // many items: resident workgroups that prefetch the next item's lists and build keys (join -3 % at c3, -3.5 % at
        // 262144 items); few items (c2: 2048): one workgroup per item balances better
        if (lists && next_item && nb >= persistent_min_items && !a.dbg && !a.dbg_flags) {
            const u32 ldsp = sizeof(CkHdr) + S * 8 + 2 * (JP_META + JB_META) * 4 + 16 + S / 8 + CK_OVF * 8;
            auto pk = fj_count_join_persistent<512>;
            hipError_t e = fj_set_max_lds_once(reinterpret_cast<const void*>(pk), ldsp);
            if (e != hipSuccess) return e;
            // (*next_item is zero: cleared with the plan's scalars at the start of the join)
            const u32 grid = nb < 512 ? nb : 512;            // two resident workgroups per CU
            hipLaunchKernelGGL(pk, dim3(grid), dim3(512), ldsp, s, a, next_item);
            return hipGetLastError();
        }
        auto kern = lists ? fj_count_join_kernel<512, true> : fj_count_join_kernel<512, false>;
        hipError_t e = fj_set_max_lds_once(reinterpret_cast<const void*>(kern), lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3(nb), dim3(512), lds, s, a);
    }
    return hipGetLastError();
}

hipError_t fj_launch_emit_single(const FjLdsJoinArgs& a, hipStream_t s, u32* next_item) {
    if (!a.build.list || !a.probe.list || !a.items || !next_item || !a.out_cursor || a.dedup || !a.build.vals) return hipErrorInvalidValue;
    const u32 nb = a.items_cap;
    const u32 ldsp = sizeof(EkHdr) + 2 * S * 8 + 2 * (JP_META + JB_META) * 4 + 16 + S / 8 + CK_OVF * 8;
    auto pk = fj_emit_join_persistent<1024, false, true>;
    hipError_t e = fj_set_max_lds_once(reinterpret_cast<const void*>(pk), ldsp);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(pk, dim3(nb < 256 ? nb : 256), dim3(1024), ldsp, s, a, next_item);
    return hipGetLastError();
}

// emitting pass: the items the cuckoo emit kernel could not place (FJ_STAT_EMIT_RETRY) on the tagged table
hipError_t fj_launch_lds_emit_retry(const FjLdsJoinArgs& a0, hipStream_t s, bool only_marked) {
    FjLdsJoinArgs a = a0;
    a.retry_only = only_marked ? 1u : 0u;
    const u32 nb = a.items ? a.items_cap : a.nparts * a.nsplit;
    const u32 lds = sizeof(JoinHdr) + 2 * S * 8 + S + S / 2 + (JP_META + JB_META) * 4;
    auto kern = (a.build.list && a.probe.list) ? fj_lds_join_kernel<true, 1024, true> : fj_lds_join_kernel<true, 1024, false>;
    hipError_t e = fj_set_max_lds_once(reinterpret_cast<const void*>(kern), lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(nb), dim3(1024), lds, s, a);
    return hipGetLastError();
}

hipError_t fj_launch_lds_join_retry(const FjLdsJoinArgs& a0, hipStream_t s) {
    FjLdsJoinArgs a = a0;
    a.retry_only = 1;
    const u32 nb = a.items ? a.items_cap : a.nparts * a.nsplit;
    const u32 lds = sizeof(JoinHdr) + S * 8 + S + S / 2 + (JP_META + JB_META) * 4;
    auto kern = (a.build.list && a.probe.list) ? fj_lds_join_kernel<false, 1024, true> : fj_lds_join_kernel<false, 1024, false>;
    hipError_t e = fj_set_max_lds_once(reinterpret_cast<const void*>(kern), lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(nb), dim3(1024), lds, s, a);
    return hipGetLastError();
}

hipError_t fj_launch_sample_hits(const FjChunkSet& build, const u64* pk, u64 np, u32 nsamples, u32 shift32, u32 pmask,
                                 unsigned long long* hits, hipStream_t s) {
    if (!build.list || np < nsamples) return hipErrorInvalidValue;
    hipLaunchKernelGGL(fj_sample_hits_kernel, dim3((nsamples * 64 + 255) / 256), dim3(256), 0, s, build, pk, np, nsamples, shift32, pmask, hits);
    return hipGetLastError();
}

hipError_t fj_launch_gt_build(const FjGtArgs& a, hipStream_t s) {
    if (a.nb == 0) return hipSuccess;
    u64 blocks = (a.nb + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(fj_gt_build_kernel, dim3((u32)blocks), dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t fj_launch_gt_probe(const FjGtArgs& a, bool materialize, u32 grid, hipStream_t s) {
    const bool bloom = a.bloom != nullptr;
    if (materialize) {
        if (bloom) hipLaunchKernelGGL((fj_gt_probe_kernel<true, true>), dim3(grid), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((fj_gt_probe_kernel<true, false>), dim3(grid), dim3(256), 0, s, a);
    } else {
        if (bloom) hipLaunchKernelGGL((fj_gt_probe_kernel<false, true>), dim3(grid), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((fj_gt_probe_kernel<false, false>), dim3(grid), dim3(256), 0, s, a);
    }
    return hipGetLastError();
}

hipError_t fj_launch_owner_hist(const u64* keys, u64 n, u32 nranks, unsigned long long* counts, hipStream_t s) {
    if (n == 0) return hipSuccess;
    u64 blocks = (n + 2047) / 2048;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(fj_owner_hist_kernel, dim3((u32)blocks), dim3(256), 0, s, keys, n, nranks, counts);
    return hipGetLastError();
}

hipError_t fj_launch_owner_scatter(const u64* keys, const u64* vals, u64 n, u32 nranks,
                                   const unsigned long long* offsets, unsigned long long* cursors,
                                   u64* out_keys, u64* out_vals, hipStream_t s) {
    if (n == 0) return hipSuccess;
    const u64 tile = vals ? 2048 : 4096;
    u64 blocks = (n + tile - 1) / tile;
    if (blocks > 1024) blocks = 1024;
    if (vals) hipLaunchKernelGGL(fj_owner_scatter_kernel<true>, dim3((u32)blocks), dim3(512), 0, s, keys, vals, n, nranks, offsets, cursors, out_keys, out_vals);
    else hipLaunchKernelGGL(fj_owner_scatter_kernel<false>, dim3((u32)blocks), dim3(512), 0, s, keys, vals, n, nranks, offsets, cursors, out_keys, out_vals);
    return hipGetLastError();
}

hipError_t fj_launch_iota(u64* out, u64 n, hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(fj_iota_kernel, dim3(2048), dim3(256), 0, s, out, n);
    return hipGetLastError();
}
hipError_t fj_launch_gen_build(u64* keys, u64* vals, u64 first, u64 n, hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(fj_gen_build_kernel, dim3(2048), dim3(256), 0, s, keys, vals, first, n);
    return hipGetLastError();
}
hipError_t fj_launch_gen_probe(u64* keys, u64 first, u64 n, u64 build_total, u64 seed, u32 hit_bp,
                               unsigned long long* expected_hits, hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(fj_gen_probe_kernel, dim3(2048), dim3(256), 0, s, keys, first, n, build_total, seed, hit_bp, expected_hits);
    return hipGetLastError();
}
