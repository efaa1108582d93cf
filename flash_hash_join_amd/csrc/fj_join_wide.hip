// fj_join_wide.hip -- counting join for partitions whose build side is NOT thin against the probe side (MI355X, gfx950).
//
// Same role as fj_count_join_persistent (fj_join.hip): insert_local + probe_vectorized of one radix partition
// (hash_join.cpp:112-128, :153-182) inside _hash_join_radix_count (:498-534).  That kernel keeps an 8192-slot cuckoo table per
// workgroup, two workgroups per CU, and its item is a chain of latency-bound steps: table reset -> claims (two dependent LDS
// round trips per KEY) -> barrier -> eviction chains (at ~3800 keys, load 0.47, the longest is 24 dependent exchanges) ->
// barrier -> probe.  With ~3000-3800 build keys against ~3800-4800 probe keys per partition the table build is what the item
// costs (profiles/r04: 0.38 of the HBM peak).  Here ONE 1024-thread workgroup per CU owns a 16384-slot table (128 KiB of the
// CU's 160 KiB) and the item is a software pipeline without a dependent chain in it:
//   * open addressing WITHOUT evictions: a key lives at the first free slot of l1, l2, l3 (three hashed locations), l3+1, ... -
//     slots are claimed with atomic ORs on a slot BITMAP, the table itself only sees plain stores and reads.  At load 0.23
//     1.2 % of the keys go beyond l3.  A lookup reads its three locations unconditionally and walks on only where all three
//     are taken by other keys;
//   * because claims touch the bitmap only, the claims of item k+1 run WHILE item k is probed (two bitmaps, alternating): their
//     LDS round trips hide under the probe's lookups; the slots a thread was assigned stay in its registers;
//   * no table reset: after the probe every thread stores the empty marker over the slots it filled (no evictions: they are
//     exactly the occupied ones), then the new keys go in with plain stores: two short throughput-bound phases;
//   * every wave owns whole 256-key chunks (4 keys per lane and chunk, two 16-B loads), so no lane looks up padding;
//   * nothing in the loop waits for a dependent global load: item descriptors are fetched by one thread five items ahead and
//     parked in LDS, chunk-list entries three items ahead, build keys two, probe keys one (no scalar loads in the loop: they
//     share the LDS wait counter);
//   * DENSE: the build side is not a chunk set but the multi-GPU "build broadcast" wire format (fj_bcast.hip): per source rank
//     one run of keys per final partition, stored as two planes (low word, remaining bits of the high word) behind an offset
//     table; the partition id supplies the top bits.  Nothing is re-partitioned or copied on the receiving side.
// Items are dealt round-robin (item = blockIdx.x + k * gridDim.x): all of a launch's workgroups are resident.  When the probe side
// of a partition is cut into several items (FjWideArgs::group_log > 0: the host sees > ~28 probe chunks per partition - the
// broadcast form at 2 and 4 ranks) the deal is in runs of 2^group_log consecutive items instead, and an item whose predecessor in
// the workgroup belongs to the same partition finds its table built: no claims, no clearing, no stores, no build-side loads.
// Items whose build side does not fit (a claim walks W_MAXWALK slots in vain) are marked FJ_ITEM_RETRY exactly as
// fj_count_join_persistent does; the host's retry / re-partition ladder (radix_join_tail) is unchanged.
#include "fj_internal.h"
#include <type_traits>

namespace {

constexpr int WNT = 1024;
constexpr u32 WSLOG = 14, WS = 1u << WSLOG;
constexpr u32 W_META_P = 32;              // probe-side list entries staged per item = the most an item of this kernel has
constexpr u32 W_META_B = 32;              // build-side list entries staged per item / 2 words per source (DENSE)
constexpr u32 W_UNITS = 32;               // DENSE: 256-slot load units per item (uint4 each: source, first key slot, run begin, run end) + their count
constexpr u32 W_STRIDE = W_META_P + 4 * W_UNITS + 4;
constexpr u32 W_MAXWALK = 48;             // slots a claim walks beyond l3 before the table counts as full
constexpr u32 W_WAVES = WNT / 64;
constexpr u32 W_NOSLOT = 0xFFFFFFFFu;

struct WHdr {
    u32 cnt, pad0[3];
    u32 has_empty[2], full[2];             // per bitmap parity: the item whose claims ran on it
    u32 dring[8][8];                       // item descriptors: {probe list pos, probe chunks, partition, item id, b0, nbc, -, -}
    u64 lo_off[FJ_WIDE_MAXSRC], mid_off[FJ_WIDE_MAXSRC], offs_off[FJ_WIDE_MAXSRC];
};

__device__ __forceinline__ u32 w_l1(u64 h) { return FJ_HW2(h) & (WS - 1); }
__device__ __forceinline__ u32 w_l2(u64 h) { return (FJ_HW2(h) >> WSLOG) & (WS - 1); }
// (third location: the high word's low bits - radix digits come from its TOP - folded with 14 other bits of the low word: three instructions)
__device__ __forceinline__ u32 w_l3(u64 h) { return (FJ_HW1(h) ^ (FJ_HW2(h) >> 18) ^ (FJ_HW2(h) >> 5)) & (WS - 1); }

// MIDB: 0 = chunk-list build side; 2 / 4 = DENSE, the width of the high-word plane's elements (compile-time: the raw planes of a
// batch wait in registers, and one set of registers must do)
template <int MIDB, bool GROUPED>
__global__ __launch_bounds__(WNT) void fj_count_join_wide(FjLdsJoinArgs a, FjWideArgs w) {
    constexpr bool DENSE = MIDB != 0;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    WHdr* hdr = reinterpret_cast<WHdr*>(smem);
    u64* tkeys = reinterpret_cast<u64*>(smem + sizeof(WHdr));
    u32* bits0 = reinterpret_cast<u32*>(tkeys + WS);               // [2][WS / 32] slot bitmaps
    u32* meta = bits0 + 2 * (WS / 32);                             // [4][W_STRIDE]: ring of staged list entries
    const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    u32 item_lo = 0, item_hi = *a.nitems_dev;
    if (w.toff) { item_lo = w.toff[w.part_lo]; item_hi = w.toff[w.part_hi]; }
    // runs of G = 2^group_log consecutive items per workgroup and round (G = 1: plain round-robin)
    // (GROUPED is a compile-time switch: the plain deal's code must not carry the other's registers - it cost the 8-rank join 7 %)
    const u32 glog = GROUPED ? w.group_log : 0u, G = 1u << glog, per_round = gridDim.x << glog;
    constexpr bool grouped = GROUPED;
    if (item_hi <= item_lo) return;
    const u32 nrounds = (item_hi - item_lo) / per_round, rem = (item_hi - item_lo) - nrounds * per_round;
    const u32 nmine = (nrounds << glog) + (rem > (blockIdx.x << glog) ? (rem - (blockIdx.x << glog) < G ? rem - (blockIdx.x << glog) : G) : 0u);     // items of this workgroup
    if (nmine == 0) return;
    auto id_of = [&](u32 q) { return item_lo + (q >> glog) * per_round + (blockIdx.x << glog) + (q & (G - 1u)); };
    auto ring = [&](u32 q, u32 f) -> u32 { return __builtin_amdgcn_readfirstlane(hdr->dring[q & 7][f]); };   // descriptor field, wave-uniform

    // descriptor fetch, one thread: the item-table entry of item q (q >= nmine: a dead item: no chunks on either side)
    auto fetch_items = [&](u32 q) -> uint4 { return q < nmine ? a.items[id_of(q)] : make_uint4(0, 0, 0, 0); };
    auto store_items = [&](u32 q, uint4 it) {
        u32* r = hdr->dring[q & 7];
        r[0] = it.x; r[1] = q < nmine ? it.y : 0u; r[2] = it.z; r[3] = q < nmine ? id_of(q) : 0xFFFFFFFFu; r[4] = 0; r[5] = 0;
        r[6] = it.w;                                               // (every loaded word is used: a half-dead load register gets reused and the reuse waits for ALL loads in flight)
    };
    auto fetch_boff_of = [&](u32 q, u32 part) -> uint2 {           // (chunk-list build side) part: descriptor field 2 of item q
        if (DENSE || q >= nmine) return make_uint2(0, 0);
        return make_uint2(a.build.boff[part], a.build.boff[part + 1]);
    };
    auto fetch_boff = [&](u32 q) -> uint2 { return fetch_boff_of(q, DENSE ? 0u : hdr->dring[q & 7][2]); };      // needs store_items(q) to be visible
    auto store_boff = [&](u32 q, uint2 b) { u32* r = hdr->dring[q & 7]; r[4] = b.x; r[5] = b.y - b.x; };

    // list entries of item q -> registers (one global load per side and thread), later parked in a ring slot
    // (DENSE: the LAST wave - it holds the fewest probe chunks of an item - looks after the build side's entries: lane s of it keeps
    //  source s's plane offsets in registers and loads that source's (run begin, run end) for the item: mb, mb2)
    const bool bwave = DENSE && wave == W_WAVES - 1;
    u32 my_lo16 = 0, my_mid16 = 0; const u32* my_offs = nullptr;
    auto dense_lane_setup = [&]() {                                // after hdr->*_off are visible
        if (bwave && lane < w.nsrc) { my_lo16 = (u32)(hdr->lo_off[lane] >> 4); my_mid16 = (u32)(hdr->mid_off[lane] >> 4); my_offs = reinterpret_cast<const u32*>(w.base + hdr->offs_off[lane]); }
    };
    // (ppos, part, b0: item q's descriptor fields 0, 2, 4 - the caller has them in scalar registers)
    auto request = [&](u32 q, u32 ns, u32 nbc, u32 ppos, u32 part, u32 b0, u32& mp, u32& mb, u32& mb2) {
        mp = 0; mb = 0; mb2 = 0;
        const u32 np0 = ns < W_META_P ? ns : W_META_P;
        if (tid < np0) mp = a.probe.list[ppos + tid];
        if (DENSE) {
            if (bwave && q < nmine && lane < w.nsrc) { mb = my_offs[part]; mb2 = my_offs[part + 1]; }
        } else {
            const u32 nb0 = nbc < W_META_B ? nbc : W_META_B;
            if (tid < nb0) mb = a.build.list[b0 + tid];
        }
    };
    // DENSE: that wave turns the sources' (run begin, run end) pairs into the item's load units: the runs of the partition, one per
    // source, cut into units of 256 key slots that start at a multiple of 4 keys; one 16-byte descriptor per unit, so that a wave
    // fetching a unit reads ONE LDS word (every wave walking the sources' pairs cost 50 LDS instructions per wave and item: as much
    // as the lookups).  No LDS round trip in here (the prefix over the sources runs on lane reads): it sits in front of a barrier.
    auto park = [&](u32* slot, u32 mp, u32 mb, u32 mb2) {
        if (tid < W_META_P) slot[tid] = mp;
        if (!DENSE) { if (tid < W_META_B) slot[W_META_P + tid] = mb; return; }
        if (bwave) {
            const u32 b = mb, e = mb2;                             // lane s: source s
            const bool act = lane < w.nsrc && e > b;
            const u32 a0 = b & ~3u, nu = act ? (e - a0 + 255u) >> 8 : 0u;
            u32 excl = 0, total = 0;
            for (u32 s = 0; s < w.nsrc; ++s) {
                const u32 n_s = (u32)__builtin_amdgcn_readlane((int)nu, (int)s);
                if (lane > s) excl += n_s;
                total += n_s;
            }
            uint4* ud = reinterpret_cast<uint4*>(slot + W_META_P);
            // {low-word plane / 16, high-word plane / 16 (byte offsets into base), first valid key slot (the unit starts at that & ~3), run end}
            for (u32 k = 0; k < nu; ++k) if (excl + k < W_UNITS) ud[excl + k] = make_uint4(my_lo16, my_mid16, k ? a0 + (k << 8) : b, e);
            if (lane == 0) slot[W_META_P + 4 * W_UNITS] = total;
        }
    };
    auto dense_total = [&](const u32* slot) -> u32 { return __builtin_amdgcn_readfirstlane(slot[W_META_P + 4 * W_UNITS]); };   // load units of the item parked in `slot`

    // ---- build keys of one batch -> registers ---------------------------------------------------------------------------
    // chunk lists: wave v takes chunks first + v and first + v + 16 of the `nstaged` staged entries (4 keys per lane each); DENSE:
    // wave v takes the 256-slot units v and v + 16 of the sources' runs (see below).  The loads are
    // unconditional (validity is a mask; a chunk that does not exist is one 16-byte line for the whole wave).  am = the key
    // slots that hold anything for this WAVE (uniform): everything downstream skips the others.
    // DENSE: what a batch's loads return stays RAW (low words; high-word bits) until the batch moves up at the end of the iteration:
    // putting the keys together right behind the loads made every iteration wait for them on the spot (~1.4 us of HBM latency
    // per item: the whole difference to the chunk-list form, whose keys need no assembling)
    typedef typename std::conditional<MIDB == 2, uint2, uint4>::type midv_t;
    struct RawBuild { uint4 lo[2]; midv_t mid[2]; };
    auto assemble = [&](const RawBuild& r, u32 top, u64 (&bk)[8]) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const u32 l[4] = {r.lo[i].x, r.lo[i].y, r.lo[i].z, r.lo[i].w};
            u32 m[4];
            if constexpr (MIDB == 2) { m[0] = r.mid[i].x & 0xFFFFu; m[1] = r.mid[i].x >> 16; m[2] = r.mid[i].y & 0xFFFFu; m[3] = r.mid[i].y >> 16; }
            else if constexpr (MIDB == 4) { m[0] = r.mid[i].x; m[1] = r.mid[i].y; m[2] = r.mid[i].z; m[3] = r.mid[i].w; }
            else { m[0] = m[1] = m[2] = m[3] = 0; }
#pragma unroll
            for (int j = 0; j < 4; ++j) bk[4 * i + j] = ((u64)(top | m[j]) << 32) | l[j];
        }
    };
    auto top_of = [&](u32 part) -> u32 { return w.bits ? part << (32u - w.bits) : 0u; };
    const u32 lo0_16 = DENSE ? (u32)(w.lo_off[0] >> 4) : 0u, mid0_16 = DENSE ? (u32)(w.mid_off[0] >> 4) : 0u;      // (any readable plane: what a wave without a unit loads)
    auto unit_desc = [&](const u32* slot, int i) -> uint4 { return reinterpret_cast<const uint4*>(slot + W_META_P)[wave + (u32)i * W_WAVES]; };     // DENSE: the wave's i-th load unit of the item parked in `slot` (garbage beyond the item's units)
    auto load_build = [&](const u32* slot, u32 part, u32 first, u32 nstaged, u32 total, const uint4 (&ud)[2], u64 (&bk)[8], RawBuild& raw, u32& bok, u32& am) {
        const u32* bm = slot + W_META_P;
        bok = 0; am = 0;
        if (DENSE) {
            // wave v takes load units v and v + 16 (park): per unit one 16-byte load (4 low words per lane) and one 8- or 16-byte load
            // (their high-word bits).  (One 4-byte and one 2-byte load per KEY: the address path charges per load instruction.)
            (void)part; (void)bm;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const u32 u = first + wave + (u32)i * W_WAVES;
                const bool have = u < total && u < W_UNITS;
                const uint4 d4 = ud[i];
                const u32 ulo = have ? __builtin_amdgcn_readfirstlane(d4.x) : lo0_16, umid = have ? __builtin_amdgcn_readfirstlane(d4.y) : mid0_16;
                const u32 ub = __builtin_amdgcn_readfirstlane(d4.z), ue = __builtin_amdgcn_readfirstlane(d4.w), ua0 = ub & ~3u;
                const u32 k0 = ua0 + 4 * lane;
                const bool in = have && k0 < ue;                           // (lanes past the run read its first word: nothing beyond the plane's 16 bytes of padding is touched)
                const u32 kk = in ? k0 : (have ? ua0 : 0u);
                raw.lo[i] = *reinterpret_cast<const uint4*>(w.base + ((u64)ulo << 4) + (u64)kk * 4);
                if constexpr (DENSE) raw.mid[i] = *reinterpret_cast<const midv_t*>(w.base + ((u64)umid << 4) + (u64)kk * (u64)MIDB);
#pragma unroll
                for (int j = 0; j < 4; ++j) if (in && k0 + j >= ub && k0 + j < ue) bok |= 1u << (4 * i + j);
                if (have) am |= 0xFu << (4 * i);
            }
        } else {
            // (both list entries are read before either is used: one LDS round trip, not two)
            u32 e2[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) { const u32 c = first + wave + (u32)i * W_WAVES; e2[i] = bm[c < nstaged ? c : (nstaged ? nstaged - 1 : 0u)]; }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const u32 c = first + wave + (u32)i * W_WAVES;
                const bool have = c < nstaged;
                const u32 e = nstaged ? e2[i] : 0u, cnt = have ? FJ_LIST_CNT(e) : 0u;
                const u64* ck = a.build.keys + (u64)FJ_LIST_ID(e) * FJ_CHUNK + (have ? 2 * lane : 0u);
                const u64x2 q0 = *reinterpret_cast<const u64x2*>(ck);
                const u64x2 q1 = *reinterpret_cast<const u64x2*>(ck + (have ? 128 : 0));
                bk[4 * i] = q0.x; bk[4 * i + 1] = q0.y; bk[4 * i + 2] = q1.x; bk[4 * i + 3] = q1.y;
                bok |= ((2 * lane < cnt ? 1u : 0u) | (2 * lane + 1 < cnt ? 2u : 0u) | (128 + 2 * lane < cnt ? 4u : 0u) | (129 + 2 * lane < cnt ? 8u : 0u)) << (4 * i);
                if (have) am |= (cnt > 128 ? 0xFu : 0x3u) << (4 * i);
            }
        }
        am = __builtin_amdgcn_readfirstlane(am);
    };

    // ---- claims of a batch on bitmap `bits`: sl[j] = the slot key j will be stored in (W_NOSLOT: none) ------------------------
    // per group of 4 key slots (a chunk of the wave / 4096 keys of the workgroup; groups without keys are skipped, wave-uniformly):
    // stage A (l1 of every key: the thread's atomics in flight together), B (l2 of the losers), C (l3), then the rare linear walk
    auto claim4 = [&](auto G, u32* bits, const u64 (&bk8)[8], u32 vm, u32 par, u32 (&sl8)[8]) {
        constexpr int g4 = 4 * decltype(G)::value;
        u32 sl[4]; u64 bk[4];
        u32 lost = 0;                                              // per-lane bit j: key j still has no slot
#pragma unroll
        for (int j = 0; j < 4; ++j) { bk[j] = bk8[g4 + j]; sl[j] = W_NOSLOT; if (bk[j] == FJ_EMPTY_KEY) { if ((vm >> j) & 1u) hdr->has_empty[par] = 1; vm &= ~(1u << j); } }
        auto fin = [&]() {
#pragma unroll
            for (int j = 0; j < 4; ++j) sl8[g4 + j] = ((lost >> j) & 1u) ? W_NOSLOT : sl[j];
        };
        // a stage: every key that still needs a slot tries location loc(key): all of the thread's atomics in flight, then the answers
        auto stage = [&](u32 need, int which) {
            u32 o[4], bit[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                o[j] = 0; bit[j] = 0;
                if ((need >> j) & 1u) {
                    const u32 l = which == 1 ? w_l1(bk[j]) : which == 2 ? w_l2(bk[j]) : w_l3(bk[j]);
                    sl[j] = l; bit[j] = 1u << (l & 31);
                    o[j] = atomicOr(&bits[l >> 5], bit[j]);
                }
            }
            u32 l2 = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) if (o[j] & bit[j]) l2 |= 1u << j;
            return l2;
        };
        lost = stage(vm & 0xFu, 1);
        if (__ballot(lost != 0)) lost = stage(lost, 2);
        if (__ballot(lost != 0)) lost = stage(lost, 3);
        if (__ballot(lost != 0)) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (!((lost >> j) & 1u)) continue;
                u32 s = sl[j];
                bool got = false;
                for (u32 step = 0; step < W_MAXWALK && !got; ++step) {
                    s = (s + 1) & (WS - 1);
                    got = !((atomicOr(&bits[s >> 5], 1u << (s & 31)) >> (s & 31)) & 1u);
                }
                if (got) { sl[j] = s; lost &= ~(1u << j); } else hdr->full[par] = 1;
            }
        }
        fin();
    };
    // (the DENSE instantiation keeps the earlier formulation of the same stages: it compiles to 7.2 ms where this one gives 7.8 - and
    //  the other way round from chunk lists: 5.76 against 5.89 ms; same-box A/Bs, profiles/r05_join_kernel_ab.txt)
    auto claim4_dense = [&](auto G, u32* bits, const u64 (&bk8)[8], u32 vm, u32 par, u32 (&sl8)[8]) {
        constexpr int g4 = 4 * decltype(G)::value;
        u32 o[4], sl[4]; u64 bk[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) bk[j] = bk8[g4 + j];
        auto fin = [&]() {
#pragma unroll
            for (int j = 0; j < 4; ++j) sl8[g4 + j] = sl[j];
        };
#pragma unroll
        for (int j = 0; j < 4; ++j) { sl[j] = W_NOSLOT; o[j] = 0; if (bk[j] == FJ_EMPTY_KEY) { if ((vm >> j) & 1u) hdr->has_empty[par] = 1; vm &= ~(1u << j); } }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const u32 l = w_l1(bk[j]);
            if ((vm >> j) & 1u) { sl[j] = l; o[j] = atomicOr(&bits[l >> 5], 1u << (l & 31)); }
        }
#pragma unroll
        for (int which = 2; which <= 3; ++which) {
            u32 need = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) if (sl[j] != W_NOSLOT && ((o[j] >> (sl[j] & 31)) & 1u)) need |= 1u << j;
            if (__ballot(need != 0) == 0) { fin(); return; }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                o[j] = 0u;                                         // (settled keys read as "won")
                if ((need >> j) & 1u) {
                    const u32 l = which == 2 ? w_l2(bk[j]) : w_l3(bk[j]);
                    sl[j] = l; o[j] = atomicOr(&bits[l >> 5], 1u << (l & 31));
                }
            }
        }
        u32 need = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) if (sl[j] != W_NOSLOT && ((o[j] >> (sl[j] & 31)) & 1u)) need |= 1u << j;
        if (__ballot(need != 0) == 0) { fin(); return; }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (!((need >> j) & 1u)) continue;
            u32 s = sl[j];
            bool got = false;
            for (u32 step = 0; step < W_MAXWALK && !got; ++step) {
                s = (s + 1) & (WS - 1);
                got = !((atomicOr(&bits[s >> 5], 1u << (s & 31)) >> (s & 31)) & 1u);
            }
            if (got) sl[j] = s; else { sl[j] = W_NOSLOT; hdr->full[par] = 1; }
        }
        fin();
    };
    auto claim = [&](u32* bits, const u64 (&bk)[8], u32 bok, u32 am, u32 par, u32 (&sl)[8]) {
        if (am & 0xFu) { if constexpr (DENSE) claim4_dense(std::integral_constant<int, 0>(), bits, bk, bok & 0xFu, par, sl); else claim4(std::integral_constant<int, 0>(), bits, bk, bok & 0xFu, par, sl); }
        else { sl[0] = sl[1] = sl[2] = sl[3] = W_NOSLOT; }
        if (am >> 4) { if constexpr (DENSE) claim4_dense(std::integral_constant<int, 1>(), bits, bk, bok >> 4, par, sl); else claim4(std::integral_constant<int, 1>(), bits, bk, bok >> 4, par, sl); }
        else { sl[4] = sl[5] = sl[6] = sl[7] = W_NOSLOT; }
    };
    auto store_keys = [&](const u64 (&bk)[8], const u32 (&sl)[8], u32 am) {
        if (am & 0xFu) {
#pragma unroll
            for (int j = 0; j < 4; ++j) if (sl[j] != W_NOSLOT) tkeys[sl[j]] = bk[j];
        }
        if (am >> 4) {
#pragma unroll
            for (int j = 4; j < 8; ++j) if (sl[j] != W_NOSLOT) tkeys[sl[j]] = bk[j];
        }
    };
    auto clear_slots = [&](const u32 (&sl)[8], u32 am) {
        if (am & 0xFu) {
#pragma unroll
            for (int j = 0; j < 4; ++j) if (sl[j] != W_NOSLOT) tkeys[sl[j]] = FJ_EMPTY_KEY;
        }
        if (am >> 4) {
#pragma unroll
            for (int j = 4; j < 8; ++j) if (sl[j] != W_NOSLOT) tkeys[sl[j]] = FJ_EMPTY_KEY;
        }
    };
    auto reset_table = [&]() {
        const ulonglong2 e2 = make_ulonglong2(FJ_EMPTY_KEY, FJ_EMPTY_KEY);
        for (u32 i = tid; i < WS / 2; i += WNT) reinterpret_cast<ulonglong2*>(tkeys)[i] = e2;
    };

    // ---- probe side: a wave's chunk -> 4 keys per lane ---------------------------------------------------------------------
    // ---- probe side: a wave's chunk -> 4 keys per lane (whole chunks per wave: units of 64 keys - 8-byte loads, perfectly balanced
    // waves - were 30 % slower, units of 128 keys 7 %: the address path charges per load instruction) --------------------------------
    // a wave's two chunks of an item (c and c + W_WAVES): both list entries in one LDS round trip, then the four loads
    auto load_chunks2 = [&](const u32* pm, u32 nb, u64 (&k0)[4], u32& vm0, u64 (&k1)[4], u32& vm1) {
        const u32 last = nb ? nb - 1 : 0u;
        const u32 c0 = wave, c1 = wave + W_WAVES;
        const u32 ea = pm[c0 < nb ? c0 : last], eb = pm[c1 < nb ? c1 : last];
        auto one = [&](u32 e, bool have, u64 (&k)[4], u32& vm) {
            const u32 cnt = have ? FJ_LIST_CNT(e) : 0u;
            const u64* ck = a.probe.keys + (u64)(nb ? FJ_LIST_ID(e) : 0u) * FJ_CHUNK + (have ? 2 * lane : 0u);
            const u64x2 q0 = *reinterpret_cast<const u64x2*>(ck);
            const u64x2 q1 = *reinterpret_cast<const u64x2*>(ck + (have ? 128 : 0));
            k[0] = q0.x; k[1] = q0.y; k[2] = q1.x; k[3] = q1.y;
            vm = (2 * lane < cnt ? 1u : 0u) | (2 * lane + 1 < cnt ? 2u : 0u) | (128 + 2 * lane < cnt ? 4u : 0u) | (129 + 2 * lane < cnt ? 8u : 0u);
        };
        one(ea, c0 < nb, k0, vm0);
        one(eb, c1 < nb, k1, vm1);
    };
    auto probe2 = [&](u64 k0, u64 k1, u32 vm, u64 he) -> u32 {   // two keys per lane: six lookups in flight
        const u64 k[2] = {k0, k1};
        u64 c1[2], c2[2], c3[2];
        u32 l3[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            l3[i] = w_l3(k[i]);
            c1[i] = tkeys[w_l1(k[i])];
            c2[i] = tkeys[w_l2(k[i])];
            c3[i] = tkeys[l3[i]];
        }
        __builtin_amdgcn_sched_barrier(0);
        u32 hits = 0;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            bool found = (c1[i] == k[i]) | (c2[i] == k[i]) | (c3[i] == k[i]);
            // a key beyond l3 sits behind three occupied slots: walk on only where all three are taken by other keys
            bool walk = !found && c1[i] != FJ_EMPTY_KEY && c2[i] != FJ_EMPTY_KEY && c3[i] != FJ_EMPTY_KEY && ((vm >> i) & 1u) && k[i] != FJ_EMPTY_KEY;
            if (__ballot(walk)) {
                u32 s = l3[i];
                for (u32 step = 0; step < W_MAXWALK && __ballot(walk); ++step) {
                    s = (s + 1) & (WS - 1);
                    if (walk) { const u64 c = tkeys[s]; if (c == k[i]) { found = true; walk = false; } else if (c == FJ_EMPTY_KEY) walk = false; }
                }
            }
            const u64 hit = __ballot(found);
            const u64 ise = __ballot(k[i] == FJ_EMPTY_KEY);        // the empty marker is never stored in the table
            const u64 ok = __ballot((vm >> i) & 1u);
            hits += (u32)__popcll(ok & ((hit & ~ise) | (ise & he)));
        }
        return hits;
    };
    auto probe4 = [&](const u64 (&k)[4], u32 vm, u64 he) -> u32 { return probe2(k[0], k[1], vm, he) + probe2(k[2], k[3], vm >> 2, he); };

    // a partition of more than 32 build chunks / 8192 build keys is not offered to the table at all: its items are marked for the
    // host's retry ladder like any partition the table cannot hold (no rarely-taken loads inside the loop: they would make every
    // wait on the loads in flight conservative)
    // (DENSE: `total` = the partition's 256-slot units over all sources, dense_total)
    auto is_big = [&](u32 nbc, u32 total) -> bool { return DENSE ? total > 2 * W_WAVES : nbc > 2 * W_WAVES; };

    // ---- prologue (synchronous): descriptors 0..4, entries of items 0..2, item 0 built, probe keys of item 0, build keys of item 1
    if (DENSE && tid < w.nsrc) { hdr->lo_off[tid] = w.lo_off[tid]; hdr->mid_off[tid] = w.mid_off[tid]; hdr->offs_off[tid] = w.offs_off[tid]; }
    if (tid < 5) store_items(tid, fetch_items(tid));
    if (tid == 0) { hdr->cnt = 0; hdr->has_empty[0] = hdr->has_empty[1] = 0; hdr->full[0] = hdr->full[1] = 0; }
    reset_table();
    for (u32 i = tid; i < 2 * (WS / 32); i += WNT) bits0[i] = 0;
    __syncthreads();
    if (tid < 4) store_boff(tid, fetch_boff(tid));
    __syncthreads();
    // descriptor fields kept in scalar registers, rotated every iteration: probe chunks of items k .. k+2, build chunks of k+1, k+2
    u32 ns0 = ring(0, 1), ns1 = ring(1, 1), ns2 = ring(2, 1), nbc1 = ring(1, 5), nbc2 = ring(2, 5), part2 = ring(2, 2);
    // grouped deals: partition ids of items k+1 (part1) and k+2 (part2) ride along; sameA / sameB: the item whose build keys
    // bkA / bkB would hold finds its partition's table in place (its predecessor built it)
    u32 part1 = grouped ? ring(1, 2) : 0u;
    bool sameA = grouped && 1 < nmine && part1 == ring(0, 2), sameB = false;
    u32* sl_k = meta, * sl_k1 = meta + W_STRIDE, * sl_k2 = meta + 2 * W_STRIDE, * sl_k3 = meta + 3 * W_STRIDE;
    {
        u32 mp, mb;
        u32 mb2;
        dense_lane_setup();
        request(0, ns0, ring(0, 5), ring(0, 0), ring(0, 2), ring(0, 4), mp, mb, mb2); park(sl_k, mp, mb, mb2);
        request(1, ns1, nbc1, ring(1, 0), ring(1, 2), ring(1, 4), mp, mb, mb2); park(sl_k1, mp, mb, mb2);
        request(2, ns2, nbc2, ring(2, 0), ring(2, 2), ring(2, 4), mp, mb, mb2); park(sl_k2, mp, mb, mb2);
    }
    __syncthreads();
    u64 bkA[8], bkB[8];
    RawBuild rawB;                                                 // DENSE: the planes of the batch bkB stands for
    u32 topB = 0;
    u32 bokA = 0, bokB = 0, amA = 0, amB = 0;
    u32 slots[8], am_cur = 0;                                      // of the item in the table (this thread's keys)
    {
        const u32 nbc0 = ring(0, 5), tot0 = DENSE ? dense_total(sl_k) : 0u;
        const uint4 ud[2] = {unit_desc(sl_k, 0), unit_desc(sl_k, 1)};
        load_build(sl_k, ring(0, 2), 0, nbc0 < W_META_B ? nbc0 : W_META_B, tot0, ud, bkA, rawB, bokA, amA);
        if constexpr (DENSE) assemble(rawB, top_of(ring(0, 2)), bkA);
        claim(bits0, bkA, bokA, amA, 0, slots);
        store_keys(bkA, slots, amA);
        am_cur = amA;
        if (is_big(nbc0, tot0)) hdr->full[0] = 1;
    }
    u64 ka[4], kb[4];
    u32 va = 0, vb = 0;
    {
        const u32 nb = ns0 < W_META_P ? ns0 : W_META_P;
        load_chunks2(sl_k, nb, ka, va, kb, vb);
    }
    u32 tot1 = DENSE ? dense_total(sl_k1) : 0u;
    if (!sameA) {
        const uint4 ud[2] = {unit_desc(sl_k1, 0), unit_desc(sl_k1, 1)};
        load_build(sl_k1, ring(1, 2), 0, nbc1 < W_META_B ? nbc1 : W_META_B, tot1, ud, bkA, rawB, bokA, amA);
        if constexpr (DENSE) assemble(rawB, top_of(ring(1, 2)), bkA);
    }
    __syncthreads();                                               // item 0's keys are in the table

#ifdef FJ_LAB      // diagnostic build (make EXTRA=-DFJ_LAB): where thread 0's time goes, per pipeline stage (FJ_WIDE_STAMPS=1)
    unsigned long long* tacc = reinterpret_cast<unsigned long long*>(meta + 4 * W_STRIDE);    // (the launch adds 72 bytes of LDS)
    if (tid == 0) { for (int i = 0; i < 8; ++i) tacc[i] = 0; tacc[8] = __builtin_amdgcn_s_memrealtime(); }
#define W_STAMP(i) do { if (a.dbg && tid == 0) { const unsigned long long tn_ = __builtin_amdgcn_s_memrealtime(); tacc[i] += tn_ - tacc[8]; tacc[8] = tn_; } } while (0)
#else
#define W_STAMP(i) do { } while (0)
#endif
    for (u32 k = 0; k < nmine; ++k) {
        // at entry: the table holds item k (this thread's keys of it: slots[], am_cur), bitmap k&1 marks them; ka/kb = first probe
        // chunks of item k; bkA = first build batch of item k+1 (tot1 keys if DENSE); ring slots sl_k, sl_k1, sl_k2 hold the entries of
        // items k, k+1, k+2; descriptors complete up to k+3, the item-table part of k+4 is in the ring
        const u32 par = k & 1u, parn = par ^ 1u;
        u32* bitsn = bits0 + parn * (WS / 32);
        // ---- 1. requests: descriptor parts (one thread), entries of k+3, build keys of k+2 ----
        uint4 it5 = make_uint4(0, 0, 0, 0); uint2 bo4 = make_uint2(0, 0);
        // everything the first phases read from LDS in ONE batch - descriptor of item k+3, the flags of the table in place, (DENSE) the
        // unit count and this wave's two unit descriptors of item k+2 - then the values move to scalar registers: each of these
        // used to be a round trip of its own (read, wait, readfirstlane), six in a row at the top of every iteration
        const u32* rq3 = hdr->dring[(k + 3) & 7];
        const uint4 rq_a = *reinterpret_cast<const uint4*>(rq3);          // {probe list pos, probe chunks, partition, item id}
        const uint2 rq_b = *reinterpret_cast<const uint2*>(rq3 + 4);      // {first build-list entry, build chunks}
        const uint4 flg = *reinterpret_cast<const uint4*>(hdr->has_empty);     // has_empty[2], full[2]
        const u32 part4w = DENSE ? 0u : hdr->dring[(k + 4) & 7][2];      // (thread 0's build-list offsets of item k+4 hang on it)
        const u32 tot2w = DENSE ? sl_k2[W_META_P + 4 * W_UNITS] : 0u;
        uint4 ud2[2] = {make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};
        if (DENSE) { ud2[0] = unit_desc(sl_k2, 0); ud2[1] = unit_desc(sl_k2, 1); }
        if (tid == 0) { bo4 = fetch_boff_of(k + 4, part4w); it5 = fetch_items(k + 5); }
        const u32 ns3 = __builtin_amdgcn_readfirstlane(rq_a.y), nbc3 = __builtin_amdgcn_readfirstlane(rq_b.y), part3 = __builtin_amdgcn_readfirstlane(rq_a.z);
        u32 mp, mb;
        u32 mb2;
        request(k + 3, ns3, nbc3, __builtin_amdgcn_readfirstlane(rq_a.x), part3, __builtin_amdgcn_readfirstlane(rq_b.x), mp, mb, mb2);
        const u32 tot2 = DENSE ? __builtin_amdgcn_readfirstlane(tot2w) : 0u;
        sameB = grouped && k + 2 < nmine && part2 == part1;
        if (!sameB) { load_build(sl_k2, part2, 0, nbc2 < W_META_B ? nbc2 : W_META_B, tot2, ud2, bkB, rawB, bokB, amB); topB = DENSE ? top_of(part2) : 0u; }
        else { bokB = 0; amB = 0; }
        W_STAMP(0);
        // ---- 2a. probe item k; then its successor's first probe chunks are requested into the same registers ----
        const bool full = __builtin_amdgcn_readfirstlane(par ? flg.w : flg.z) != 0 || ns0 > 2 * W_WAVES;      // (an item longer than 32 probe chunks - a host-side bug - goes to the retry ladder)
        const u64 he = __builtin_amdgcn_readfirstlane(par ? flg.y : flg.x) ? ~0ull : 0ull;
        u32 wave_hits = 0;
        const bool skip = full || ns0 == 0;
        const u32 nb = ns0 < 2 * W_WAVES ? ns0 : 2 * W_WAVES;      // (items of this kernel have at most 32 probe chunks: the host cuts them so)
        if (!skip && wave < nb) wave_hits += probe4(ka, va, he);
        if (!skip && wave + W_WAVES < nb) wave_hits += probe4(kb, vb, he);
        {
            const u32 nbn = ns1 < W_META_P ? ns1 : W_META_P;
            load_chunks2(sl_k1, nbn, ka, va, kb, vb);
        }
        W_STAMP(1);
        // ---- 2b. claims of item k+1 on the other bitmap (other waves are still probing: the round trips overlap their lookups) ----
        u32 nslots[8];
        if (!sameA) {
            claim(bitsn, bkA, bokA, amA, parn, nslots);
            if (is_big(nbc1, tot1) && tid == 0) hdr->full[parn] = 1;
        }
        W_STAMP(2);
        // ---- 3. park what was requested ----
        park(sl_k3, mp, mb, mb2);
        if (tid == 0) { store_boff(k + 4, bo4); store_items(k + 5, it5); }
        if (lane == 0 && wave_hits) atomicAdd(&hdr->cnt, wave_hits);
        W_STAMP(3);
        __syncthreads();                                         // A: every wave is done with the table and with bitmap parn
        W_STAMP(4);
        // ---- 4. item k's result; the table is emptied by the owners of its keys; bitmap `par` is cleared for item k+2 ----
        if (tid == 0) {
            const u32 cnt = skip ? 0u : hdr->cnt;
            if (sameA) { hdr->has_empty[parn] = hdr->has_empty[par]; hdr->full[parn] = hdr->full[par]; }      // the table stays: so do its flags
            hdr->cnt = 0; hdr->has_empty[par] = 0; hdr->full[par] = 0;
            if (full) atomicOr(a.err, FJ_STAT_RETRY);            // the table could not hold the partition: redone with the tagged table
            a.part_count[id_of(k)] = full ? FJ_ITEM_RETRY : cnt;
            if (cnt) atomicAdd(a.total, (unsigned long long)cnt);
        }
        if (!sameA) clear_slots(slots, am_cur);
        if (tid < WS / 32) bits0[par * (WS / 32) + tid] = 0;
        W_STAMP(5);
        __syncthreads();                                         // B
        // ---- 5. item k+1 goes in ----
        if (!sameA) store_keys(bkA, nslots, amA);
        W_STAMP(6);
        __syncthreads();                                         // C
        W_STAMP(7);
        // ---- 6. rotate ----
        if (!sameA) {
#pragma unroll
            for (int j = 0; j < 8; ++j) slots[j] = nslots[j];
            am_cur = amA;
        }
        if constexpr (DENSE) assemble(rawB, topB, bkA);          // (the loads were issued at the top of the iteration: they are in)
        else {
#pragma unroll
            for (int j = 0; j < 8; ++j) bkA[j] = bkB[j];
        }
        bokA = bokB; amA = amB; tot1 = tot2; sameA = sameB;
        ns0 = ns1; ns1 = ns2; ns2 = ns3; nbc1 = nbc2; nbc2 = nbc3; part1 = part2; part2 = (DENSE || grouped) ? part3 : 0u;
        u32* t = sl_k; sl_k = sl_k1; sl_k1 = sl_k2; sl_k2 = sl_k3; sl_k3 = t;
    }
#ifdef FJ_LAB
    if (a.dbg && tid == 0 && blockIdx.x < 4096) { for (int i = 0; i < 8; ++i) a.dbg[blockIdx.x * 8 + i] = tacc[i]; }
#endif
}

}  // namespace

u32 fj_wide_lds_bytes() { return (u32)(sizeof(WHdr) + WS * 8 + 2 * (WS / 8) + 4 * W_STRIDE * 4 + 80); }

// counting join over the final chunk sets with the 16384-slot table: a.items / a.nitems_dev / a.part_count / a.total / a.err as
// for fj_launch_lds_join.  dense: the build side comes from w (build-broadcast wire format).
hipError_t fj_launch_count_join_wide(const FjLdsJoinArgs& a, const FjWideArgs& w, bool dense, u32 grid, hipStream_t s) {
    if (!a.probe.list || !a.items || (!dense && !a.build.list) || a.want_dups) return hipErrorInvalidValue;
    if (dense && (w.nsrc == 0 || w.nsrc > FJ_WIDE_MAXSRC || !w.base || (w.mid_bytes != 2 && w.mid_bytes != 4) || w.bits > 32)) return hipErrorInvalidValue;
    if (w.group_log > 6) return hipErrorInvalidValue;
    const u32 lds = fj_wide_lds_bytes();
    if (w.group_log && !dense) return hipErrorInvalidValue;        // (chunk-list build sides come with <= 32 probe chunks per partition: fj_plan.hip wide_join_planned)
    auto kern = !dense ? fj_count_join_wide<0, false>
              : w.mid_bytes == 2 ? (w.group_log ? fj_count_join_wide<2, true> : fj_count_join_wide<2, false>)
                                 : (w.group_log ? fj_count_join_wide<4, true> : fj_count_join_wide<4, false>);
    hipError_t e = fj_set_max_lds_once(reinterpret_cast<const void*>(kern), lds);
    if (e != hipSuccess) return e;
    if (grid == 0) grid = 1;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WNT), lds, s, a, w);
    return hipGetLastError();
}
