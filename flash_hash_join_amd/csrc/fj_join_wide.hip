// fj_join_wide.hip -- counting join for partitions whose build side is NOT thin against the probe side (MI355X, gfx950).
//
// Same role as fj_count_join_persistent (fj_join.hip): insert_local + probe_vectorized of one radix partition
// (hash_join.cpp:112-128, :153-182) inside _hash_join_radix_count (:498-534).  ONE 1024-thread workgroup per CU owns a 16384-slot
// table of bare 64-bit keys (128 KiB of the CU's 160 KiB).  Round 5's table (three hashed locations + a walk, slots claimed on a
// bitmap in three stages, owners clearing their slots, plain stores: three barriers and ~36 vector instructions per build key)
// is replaced by a BUCKETED table that needs neither claims nor clearing:
//   * the table is NBK = 16384 / BS buckets of BS consecutive slots (BS = 4: 32 bytes); a key's home bucket is the low bits of
//     its hash word 2.  INSERT = one returning LDS atomic add on the bucket's fill count + one plain store at that position; a
//     key that finds its bucket full (0.35 % of them at load 0.23) tries the next bucket, and so on: linear probing over buckets;
//   * LOOKUP = the home bucket read whole (BS / 2 ds_read_b128, the order of the pieces rotated by a hash bit so that the lanes
//     of a wave spread over all LDS banks), BS 64-bit compares, and on to the next bucket only where this one is FULL;
//   * NOTHING IS EVER CLEARED.  Chunk pools hold the mixed key H, whose top radix bits ARE the partition: an entry that an
//     earlier partition left behind can never equal a probe key of the partition in place, so stale entries are harmless and
//     the table needs no reset and no empty marker - only the fill counts (NBK words) are zeroed, during the previous item's
//     probe phase.  "This bucket is full" is read off the entry in its last slot: it belongs to the partition in place (same
//     partition bits as the probe key, FjWideArgs::pmask) exactly if the bucket's BS-th key went in (a stale same-partition
//     entry - the same partition rebuilt by this workgroup - can only make a lookup walk one bucket further than it had to);
//     the table starts out filled with FJ_EMPTY_KEY (all ones), a key of the LAST partition - no probe key of any other partition
//     can equal it; before the last partition goes in, the table is filled once more, with a key of the FIRST partition.  So
//     every 64-bit value is a legal key on both sides and no value is handled out of band;
//   * an item is two barriers: the returning atomic adds of item k+1's keys are ISSUED (on the fill counts of the other parity - two
//     16-bit counts per 32-bit word - its keys have been in registers for an iteration), item k is probed, the counts of the parity
//     after next are zeroed, the adds are RESOLVED (a key that found its bucket full walks on) | barrier | item k+1's keys are
//     stored | barrier.  Round 5 paid a clear phase, a store phase and three barriers;
//   * every wave owns whole 256-key chunks (4 keys per lane and chunk, two 16-B loads), so no lane looks up padding;
//   * nothing in the loop waits for a dependent global load: item descriptors are fetched by one thread five items ahead and
//     parked in LDS, chunk-list entries three items ahead, build keys two, probe keys one (no scalar loads in the loop: they
//     share the LDS wait counter);
//   * DENSE: the build side is not a chunk set but the multi-GPU "build broadcast" wire format (fj_bcast.hip): per source rank
//     one run of keys per final partition, stored as two planes (low word, remaining bits of the high word) behind an offset
//     table; the partition id supplies the top bits.  Nothing is re-partitioned or copied on the receiving side.
// Items are dealt round-robin (item = blockIdx.x + k * gridDim.x): all of a launch's workgroups are resident.  When the probe side
// of a partition is cut into several items (FjWideArgs::group_log > 0) the deal is in runs of 2^group_log consecutive items
// instead, and an item whose predecessor in the workgroup belongs to the same partition finds its table built: no inserts, no
// build-side loads.  Duplicate build keys simply occupy several slots (a lookup stops at the first match).  Items whose build
// side does not fit (more than 8192 key slots, or a chain of more than W_MAXWALK full buckets: thousands of copies of one key)
// are marked FJ_ITEM_RETRY exactly as fj_count_join_persistent does; the host's retry / re-partition ladder is unchanged.
#include "fj_internal.h"
#include <type_traits>
#ifndef FJ_WIDE_ABLATE      // (timing-only variants, tools/mk_wide_variant.sh <name> -DFJ_WIDE_ABLATE=<1: no lookups | 2: no claims, no stores | 4: no stores, no barrier B | 8: stragglers dropped | sums>; counts wrong on purpose)
#define FJ_WIDE_ABLATE 0
#endif

namespace {

constexpr int WNT = 1024;
#ifndef FJ_WIDE_WSLOG      // (13: an 8192-slot table - what a double-buffered pair of tables would have to live with; experiment)
#define FJ_WIDE_WSLOG 14
#endif
constexpr u32 WSLOG = FJ_WIDE_WSLOG, WS = 1u << WSLOG;
#ifndef FJ_WIDE_BS
#define FJ_WIDE_BS 4
#endif
constexpr u32 BS = FJ_WIDE_BS;            // slots per bucket (4 or 8)
constexpr u32 BSLOG = BS == 4 ? 2 : 3, NBKLOG = WSLOG - BSLOG, NBK = 1u << NBKLOG, NPC = BS / 2;     // buckets; 16-byte pieces per bucket
static_assert(BS == 4 || BS == 8, "bucket size");
constexpr u32 W_META_P = 32;              // probe-side list entries staged per item = the most an item of this kernel has
constexpr u32 W_META_B = 32;              // build-side list entries staged per item / 2 words per source (DENSE)
constexpr u32 W_UNITS = 32;               // DENSE: 256-slot load units per item (uint4 each: source, first key slot, run begin, run end) + their count
constexpr u32 W_STRIDE = W_META_P + 4 * W_UNITS + 4;
constexpr u32 W_MAXWALK = 64;             // full buckets a key walks past before the table counts as full (insert) / the lookup gives up
constexpr u32 W_WAVES = WNT / 64;
constexpr u32 W_NOSLOT = 0xFFFFFFFFu;
// what the table is filled with before the LAST partition goes in: a key of the first partition (high word 0), so no probe key of
// the last partition - whose partition bits are all ones - can equal it
constexpr u64 W_POISON2 = 0x00000000FFFFFFFFull;

// a key's home: hash word 2's low bits name one of the table's WS / 2 16-byte PIECES - the bucket (the bits above the piece-in-bucket
// bits) and the piece a lookup reads first (the order of a bucket's pieces is rotated per key, so that the lanes of a wave whose
// buckets share LDS banks start at different pieces): the first address is one AND and one shift
constexpr u32 NPCLOG = BSLOG - 1;
__device__ __forceinline__ u32 w_bucket(u64 h) { return (FJ_HW2(h) >> NPCLOG) & (NBK - 1u); }
__device__ __forceinline__ u32 w_piece0(u64 h) { return FJ_HW2(h) & (WS / 2u - 1u); }

struct WHdr {
    u32 cnt, pad0[3];
    u32 unused0[2], full[2];               // full: per parity of the item whose keys are / go in the table
    u32 dring[8][8];                       // item descriptors: {probe list pos, probe chunks, partition, item id, b0, nbc, -, -}
    u64 lo_off[FJ_WIDE_MAXSRC], mid_off[FJ_WIDE_MAXSRC], offs_off[FJ_WIDE_MAXSRC];
};

// MIDB: 0 = chunk-list build side; 2 / 4 = DENSE, the width of the high-word plane's elements (compile-time: the raw planes of a
// batch wait in registers, and one set of registers must do)
template <int MIDB, bool GROUPED>
__global__ __launch_bounds__(WNT) void fj_count_join_wide(FjLdsJoinArgs a, FjWideArgs w) {
    constexpr bool DENSE = MIDB != 0;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    WHdr* hdr = reinterpret_cast<WHdr*>(smem);
    u64* tkeys = reinterpret_cast<u64*>(smem + sizeof(WHdr));
    u32* fill = reinterpret_cast<u32*>(tkeys + WS);                // [2][NBK / 2]: per parity of the item, keys that asked for a slot of the bucket (16 bits per bucket, two buckets per word)
    u32* meta = fill + NBK;                                        // [4][W_STRIDE]: ring of staged list entries
    const u32 tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // (the wave id in a SCALAR register: what hangs on it - which chunks and units a wave owns - is then scalar work)
    const u32 pmask = w.pmask;

    u32 item_lo = 0, item_hi = *a.nitems_dev;
    if (w.toff) { item_lo = w.toff[w.part_lo]; item_hi = w.toff[w.part_hi]; }
    // runs of G = 2^group_log consecutive items per workgroup and round (G = 1: plain round-robin)
    // (GROUPED is a compile-time switch: the plain deal's code must not carry the other's registers)
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
    u32 my_lo16 = 0, my_mid16 = 0;                                 // (the source's offset table is found through LDS per item: kept in a register across the loop, its pointer - or its offset - was spilled in the 4-byte-plane kernels, and a scratch reload waits for every load in flight)
    auto dense_lane_setup = [&]() {                                // after hdr->*_off are visible
        if (bwave && lane < w.nsrc) { my_lo16 = (u32)(hdr->lo_off[lane] >> 4); my_mid16 = (u32)(hdr->mid_off[lane] >> 4); }
    };
    // (ppos, part, b0: item q's descriptor fields 0, 2, 4 - the caller has them in scalar registers)
    auto request = [&](u32 q, u32 ns, u32 nbc, u32 ppos, u32 part, u32 b0, u32& mp, u32& mb, u32& mb2) {
        mp = 0; mb = 0; mb2 = 0;
        const u32 np0 = ns < W_META_P ? ns : W_META_P;
        if (tid < np0) mp = a.probe.list[ppos + tid];
        if (DENSE) {
            if (bwave && q < nmine && lane < w.nsrc) {
                u32 z;                                         // (an opaque zero in the index: nothing of this address can be hoisted out of the item loop)
                asm volatile("v_mov_b32 %0, 0" : "=v"(z));
                const u32* my_offs = reinterpret_cast<const u32*>(w.base + hdr->offs_off[lane + z]);
                mb = my_offs[part]; mb2 = my_offs[part + 1];
            }
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
                // the lane's keys k0 .. k0 + 3 that lie in [ub, ue): bits [first, end) of a nibble
                const u32 bfirst = ub > k0 ? ub - k0 : 0u, bend = ue > k0 ? (ue - k0 < 4u ? ue - k0 : 4u) : 0u;      // (bfirst <= 3: the unit starts at ub & ~3)
                if (in) bok |= (((1u << bend) - 1u) & ~((1u << bfirst) - 1u)) << (4 * i);
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

    // ---- slots for a batch: one returning atomic add per key on its home bucket's fill count (the count array of the item's parity:
    // the table itself is not touched, so this runs WHILE the item in place is probed - claim_issue in front of the lookups,
    // claim_resolve behind them, the round trip in between is free); the few keys whose bucket was full walk on, all of a thread's
    // stragglers together.  sl[j] = the slot key j will be stored in once the table is free (W_NOSLOT: none) ----------------------
    auto fill_add = [&](u32* f, u32 b) -> u32 { return (atomicAdd(&f[b >> 1], 1u << ((b & 1u) << 4)) >> ((b & 1u) << 4)) & 0xFFFFu; };
    auto claim_issue = [&](u32* f, const u64 (&bk)[8], u32 bok, u32 am, u32 (&old)[8]) {
#pragma unroll
        for (int g = 0; g < 2; ++g) {
#pragma unroll
            for (int j = 4 * g; j < 4 * g + 4; ++j) old[j] = BS;
            if (!((am >> (4 * g)) & 0xFu)) continue;               // (wave-uniform)
#pragma unroll
            for (int j = 4 * g; j < 4 * g + 4; ++j)
                if ((bok >> j) & 1u) old[j] = fill_add(f, w_bucket(bk[j]));
        }
    };
    // claim_mid (in front of the lookups, the first adds' answers are in): slots of the keys whose bucket had room; a lane's FIRST
    // straggler asks the next bucket at once - that round trip too runs under the lookups.  pend: the lane's keys without a slot;
    // b2 / o2: the bucket its first straggler asked and the answer
    auto claim_mid = [&](u32* f, const u64 (&bk)[8], u32 bok, u32 am, const u32 (&old)[8], u32 (&sl)[8], u32& pend, u32& b2, u32& o2) {
        pend = 0; b2 = 0; o2 = BS;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
#pragma unroll
            for (int j = 4 * g; j < 4 * g + 4; ++j) sl[j] = W_NOSLOT;
            if (!((am >> (4 * g)) & 0xFu)) continue;
#pragma unroll
            for (int j = 4 * g; j < 4 * g + 4; ++j) {
                if (!((bok >> j) & 1u)) continue;
                if (old[j] < BS) sl[j] = (w_bucket(bk[j]) << BSLOG) + old[j];
                else pend |= 1u << j;
            }
        }
        if (__ballot(pend != 0)) {
            const u32 low = pend & (0u - pend);
            u32 b = 0;
#pragma unroll
            for (int t = 0; t < 8; ++t) b |= w_bucket(bk[t]) & (0u - ((low >> t) & 1u));
            b2 = (b + 1u) & (NBK - 1u);
            if (pend) o2 = fill_add(f, b2);
        }
    };
    // claim_resolve (behind the lookups): the first stragglers' answers; whoever is still without a slot (a second straggler of a lane, a
    // straggler whose next bucket was full too: one key in tens of thousands) walks bucket by bucket, a lane's keys one after the other
    auto claim_resolve = [&](u32* f, const u64 (&bk)[8], u32 par, u32 (&sl)[8], u32 pend, u32 b2, u32 o2) {
        if (__ballot(pend != 0)) {
            if (pend != 0 && o2 < BS) {
                const u32 low = pend & (0u - pend), got = (b2 << BSLOG) + o2;
#pragma unroll
                for (int t = 0; t < 8; ++t) sl[t] = ((low >> t) & 1u) ? got : sl[t];
                pend &= pend - 1u;
            }
        }
        if (__ballot(pend != 0)) {
            u32 failed = 0;
            do {
                // the lane's first pending key, picked with bit masks (a chain of selects on the key index becomes an indexed array in
                // scratch memory: eight stores per item whether or not anything is pending).  A key whose second bucket was full starts
                // over from its home: the count of that second bucket goes up once more, which changes nothing (it is full)
                const u32 low = pend & (0u - pend);                // its bit alone
                u32 b = 0;
#pragma unroll
                for (int t = 0; t < 8; ++t) b |= w_bucket(bk[t]) & (0u - ((low >> t) & 1u));
                u32 got = W_NOSLOT, step = 0;
                bool c = pend != 0;
                do {
                    ++step;
                    if (c) {
                        b = (b + 1u) & (NBK - 1u);
                        const u32 o = fill_add(f, b);
                        if (o < BS) { got = (b << BSLOG) + o; c = false; }
                    }
                } while (__ballot(c) && step < W_MAXWALK);
                if (pend != 0 && got == W_NOSLOT) failed = 1;
#pragma unroll
                for (int t = 0; t < 8; ++t) sl[t] = ((low >> t) & 1u) ? got : sl[t];
                pend &= pend - 1u;
            } while (__ballot(pend != 0));
            if (failed) hdr->full[par] = 1;
        }
    };
    auto store_keys = [&](const u64 (&bk)[8], const u32 (&sl)[8], u32 am) {
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            if (!((am >> (4 * g)) & 0xFu)) continue;
#pragma unroll
            for (int j = 4 * g; j < 4 * g + 4; ++j) if (sl[j] != W_NOSLOT) tkeys[sl[j]] = bk[j];
        }
    };
    auto fill_table = [&](u64 v) {
        const ulonglong2 e2 = make_ulonglong2(v, v);
        for (u32 i = tid; i < WS / 2; i += WNT) reinterpret_cast<ulonglong2*>(tkeys)[i] = e2;
    };
    auto zero_fill_counts = [&](u32 par) {                         // one parity: NBK / 2 words
        // (a zero made on the spot: the compiler hoists a constant zero vector out of the item loop and, in the grouped dense kernels,
        // SPILLS it - the reload's s_waitcnt vmcnt(0) then waits for every prefetch in flight, once per item)
        // (... and the store's address hangs on that zero too: hoisted, it was spilled as well)
        u32 z;
        asm volatile("v_mov_b32 %0, 0" : "=v"(z));
        static_assert(NBK / 8 <= WNT, "one 16-byte store per thread");
        if (tid < NBK / 8) reinterpret_cast<uint4*>(fill + par * (NBK / 2))[tid + z] = make_uint4(z, z, z, z);
    };
    // the partition whose keys' partition bits are all ones - the last one - must not find FJ_EMPTY_KEY (what the table starts out
    // with: all ones) in the table: a probe key could equal it, and every untouched bucket would read as full.  Uniform over the
    // workgroup; at most once per launch (the last partition's items are the launch's last).
    auto is_last_part = [&](u32 part) -> bool { return part + 1u == a.nparts; };

    // ---- probe side: a wave's chunk -> 4 keys per lane (whole chunks per wave: units of 64 keys - 8-byte loads, perfectly balanced
    // waves - were 30 % slower, units of 128 keys 7 %: the address path charges per load instruction; round 6 tried again with the
    // bucketed table - chunks 16 .. 31 shared by quarters, every wave 4 + 1 keys per lane instead of four waves 8 and twelve 4: 7 %
    // slower: the kernel is bound by instruction issue, not by its longest wave) ----------------------------------------------------
    // a wave's two chunks of an item (c and c + W_WAVES): both list entries in one LDS round trip, then the four loads; n0 / n1 = the
    // keys the chunks hold (wave-uniform; 0: no such chunk).  Lane l holds keys 2l, 2l + 1, 128 + 2l, 129 + 2l of a chunk.
    auto load_chunks2 = [&](const u32* pm, u32 nb, u64 (&k0)[4], u32& n0, u64 (&k1)[4], u32& n1) {
        const u32 last = nb ? nb - 1 : 0u;
        const u32 c0 = wave, c1 = wave + W_WAVES;
        const u32 ea = __builtin_amdgcn_readfirstlane(pm[c0 < nb ? c0 : last]), eb = __builtin_amdgcn_readfirstlane(pm[c1 < nb ? c1 : last]);
        auto one = [&](u32 e, bool have, u64 (&k)[4], u32& n) {
            n = have ? FJ_LIST_CNT(e) : 0u;
            const u64* ck = a.probe.keys + (u64)(nb ? FJ_LIST_ID(e) : 0u) * FJ_CHUNK + (have ? 2 * lane : 0u);
            const u64x2 q0 = *reinterpret_cast<const u64x2*>(ck);
            const u64x2 q1 = *reinterpret_cast<const u64x2*>(ck + (have ? 128 : 0));
            k[0] = q0.x; k[1] = q0.y; k[2] = q1.x; k[3] = q1.y;
        };
        one(ea, c0 < nb, k0, n0);
        one(eb, c1 < nb, k1, n1);
    };
    // the four keys a lane holds of one chunk of n keys: their home buckets are read whole, all reads in flight; found / walks-on are
    // wave masks in scalar registers (the compares produce them)
    const unsigned char* tb = reinterpret_cast<const unsigned char*>(tkeys);
    const u32 kidx[4] = {2 * lane, 2 * lane + 1, 128 + 2 * lane, 129 + 2 * lane};
    auto probe4 = [&](const u64 (&k)[4], u32 n) -> u32 {
        constexpr int N = 4;
        uint4 pc[N][NPC];
        u32 bkt[N];
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const u32 p0 = w_piece0(k[i]);
            bkt[i] = p0 >> NPCLOG;
#pragma unroll
            for (u32 j = 0; j < NPC; ++j) pc[i][j] = *reinterpret_cast<const uint4*>(tb + ((p0 ^ j) << 4));     // piece (r ^ j) of the bucket, r = p0 & (NPC - 1)
        }
        __builtin_amdgcn_sched_barrier(0);
        u32 hits = 0, cm = 0;                                      // cm: per lane, the keys that must look into the next bucket
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const u32 hi = FJ_HW1(k[i]);
            const u32 r = FJ_HW2(k[i]) & (NPC - 1u);
            const bool ok = kidx[i] < n;
            bool f = false;
#pragma unroll
            for (u32 j = 0; j < NPC; ++j) {
                const u64 e0 = ((u64)pc[i][j].y << 32) | pc[i][j].x, e1 = ((u64)pc[i][j].w << 32) | pc[i][j].z;
                f |= (e0 == k[i]) | (e1 == k[i]);
            }
            // the bucket's LAST slot (piece NPC - 1, read as piece (NPC - 1) ^ r of this lane's order): an entry of the partition in place = full
            // (bitwise operators on the flags: && would put every key's test into a branch of its own)
            u32 last_hi;
            if constexpr (NPC == 2) last_hi = r ? pc[i][0].w : pc[i][1].w;
            else {
                last_hi = pc[i][0].w;
#pragma unroll
                for (u32 j = 1; j < NPC; ++j) last_hi = (((NPC - 1u) ^ r) == j) ? pc[i][j].w : last_hi;
            }
            const bool c = (!f) & (((last_hi ^ hi) & pmask) == 0u) & ok;
            cm |= c ? 1u << i : 0u;
            hits += (u32)__popcll(__ballot(f & ok));
        }
        // the walk (1.5 % of the buckets are full at load 0.23: a lane or two of most chunks): a lane's walking keys one after the other -
        // only the COUNT of the hits matters, so nothing is handed back per key.  (Round 6 also asked the bucket's fill count first - only
        // a count past BS sent a key on, 0.3 % of the buckets - with the counts zeroed behind barrier A instead of under the probe:
        // dense join 4.95 -> 5.05 ms, c4's join 0.51 -> 0.56: the walks were not what the time goes into.)
        if (__ballot(cm != 0)) {
            do {
                const u32 low = cm & (0u - cm);
                u32 klo = 0, khi = 0, b = 0;
#pragma unroll
                for (int t = 0; t < N; ++t) { const u32 m = 0u - ((low >> t) & 1u); klo |= FJ_HW2(k[t]) & m; khi |= FJ_HW1(k[t]) & m; b |= bkt[t] & m; }
                const u64 key = ((u64)khi << 32) | klo;
                bool c = cm != 0, f = false;
                u32 step = 0;
                do {
                    ++step;
                    if (c) {
                        b = (b + 1u) & (NBK - 1u);
                        const unsigned char* bp = tb + ((size_t)b << (BSLOG + 3));
                        u32 lh = 0;
#pragma unroll
                        for (u32 j = 0; j < NPC; ++j) {
                            const uint4 q = *reinterpret_cast<const uint4*>(bp + (j << 4));
                            const u64 e0 = ((u64)q.y << 32) | q.x, e1 = ((u64)q.w << 32) | q.z;
                            f |= (e0 == key) | (e1 == key);
                            lh = q.w;
                        }
                        c = !f && (((lh ^ khi) & pmask) == 0u);
                    }
                } while (__ballot(c) && step < W_MAXWALK);
                hits += (u32)__popcll(__ballot(f));
                cm &= cm - 1u;
            } while (__ballot(cm != 0));
        }
        return hits;
    };

    // a partition of more than 32 build chunks / 8192 build keys is not offered to the table at all: its items are marked for the
    // host's retry ladder like any partition the table cannot hold (no rarely-taken loads inside the loop: they would make every
    // wait on the loads in flight conservative)
    // (DENSE: `total` = the partition's 256-slot units over all sources, dense_total)
    const u32 max_units = DENSE && w.max_units && w.max_units < 2 * W_WAVES ? w.max_units : 2 * W_WAVES;
    auto is_big = [&](u32 nbc, u32 total) -> bool { return DENSE ? total > max_units : nbc > 2 * W_WAVES; };

    // ---- prologue (synchronous): descriptors 0..4, entries of items 0..2, item 0 built, probe keys of item 0, build keys of item 1
    if (DENSE && tid < w.nsrc) { hdr->lo_off[tid] = w.lo_off[tid]; hdr->mid_off[tid] = w.mid_off[tid]; hdr->offs_off[tid] = w.offs_off[tid]; }
    if (tid < 5) store_items(tid, fetch_items(tid));
    if (tid == 0) { hdr->cnt = 0; hdr->unused0[0] = hdr->unused0[1] = 0; hdr->full[0] = hdr->full[1] = 0; }
    fill_table(FJ_EMPTY_KEY);
    zero_fill_counts(0); zero_fill_counts(1);
    __syncthreads();
    if (tid < 4) store_boff(tid, fetch_boff(tid));
    __syncthreads();
    // descriptor fields kept in scalar registers, rotated every iteration: probe chunks of items k .. k+2, build chunks of k+1, k+2
    u32 ns0 = ring(0, 1), ns1 = ring(1, 1), ns2 = ring(2, 1), nbc1 = ring(1, 5), nbc2 = ring(2, 5), part2 = ring(2, 2);
    // partition ids of items k+1 (part1) and k+2 (part2) ride along (the last partition's special case; grouped deals); sameA /
    // sameB: the item whose build keys bkA / bkB would hold finds its partition's table in place (its predecessor built it)
    u32 part1 = ring(1, 2);
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
    if (is_last_part(ring(0, 2))) fill_table(W_POISON2);
    __syncthreads();
    u64 bkA[8], bkB[8];
    RawBuild rawB;                                                 // DENSE: the planes of the batch bkB stands for
    u32 topB = 0;
    u32 bokA = 0, bokB = 0, amA = 0, amB = 0;
    {
        const u32 nbc0 = ring(0, 5), tot0 = DENSE ? dense_total(sl_k) : 0u;
        const uint4 ud[2] = {unit_desc(sl_k, 0), unit_desc(sl_k, 1)};
        load_build(sl_k, ring(0, 2), 0, nbc0 < W_META_B ? nbc0 : W_META_B, tot0, ud, bkA, rawB, bokA, amA);
        if constexpr (DENSE) assemble(rawB, top_of(ring(0, 2)), bkA);
        u32 old[8], sl[8], pend, b2, o2;
        claim_issue(fill, bkA, bokA, amA, old);
        claim_mid(fill, bkA, bokA, amA, old, sl, pend, b2, o2);
        claim_resolve(fill, bkA, 0, sl, pend, b2, o2);
        store_keys(bkA, sl, amA);
        if (is_big(nbc0, tot0)) hdr->full[0] = 1;
    }
    u64 ka[4], kb[4];
    u32 na = 0, nbk = 0;                                           // keys in the two probe chunks this wave holds
    {
        const u32 nb = ns0 < W_META_P ? ns0 : W_META_P;
        load_chunks2(sl_k, nb, ka, na, kb, nbk);
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
    // ... and per WAVE (lane 0 of each): [0] top of the iteration -> probe done, [1] -> arrival at barrier A, [2] barrier A left -> arrival at barrier B
    unsigned long long* wacc = tacc + 9;
    if (tid < 64) wacc[tid] = 0;
    unsigned long long tw_ = __builtin_amdgcn_s_memrealtime();
#define W_WSTAMP(i, restart) do { if (a.dbg) { const unsigned long long tn_ = __builtin_amdgcn_s_memrealtime(); if (!(restart) && lane == 0) wacc[wave * 4 + (i)] += tn_ - tw_; tw_ = tn_; } } while (0)
#else
#define W_STAMP(i) do { } while (0)
#define W_WSTAMP(i, restart) do { } while (0)
#endif
    for (u32 k = 0; k < nmine; ++k) {
        // at entry: the table holds item k (flags of parity k & 1); ka/kb = first probe chunks of item k; bkA = first build batch of
        // item k+1 (tot1 units if DENSE); ring slots sl_k, sl_k1, sl_k2 hold the entries of items k, k+1, k+2; descriptors complete
        // up to k+3, the item-table part of k+4 is in the ring
        const u32 par = k & 1u, parn = par ^ 1u;
        // ---- 0. slots for item k+1 are requested on its parity's fill counts (no lookup reads them; the table is not touched): the
        // answers come back under the requests below ----
        u32* fill_n = fill + parn * (NBK / 2);
        u32 old[8], nslots[8], pend = 0, sb2 = 0, so2 = BS;
        if (!sameA && !(FJ_WIDE_ABLATE & 2)) claim_issue(fill_n, bkA, bokA, amA, old);
        // ---- 1. requests: descriptor parts (one thread), entries of k+3, build keys of k+2 ----
        uint4 it5 = make_uint4(0, 0, 0, 0); uint2 bo4 = make_uint2(0, 0);
        // everything the first phases read from LDS in ONE batch - descriptor of item k+3, the flags of the table in place, (DENSE) the
        // unit count and this wave's two unit descriptors of item k+2 - then the values move to scalar registers: each of these
        // used to be a round trip of its own (read, wait, readfirstlane), six in a row at the top of every iteration
        const u32* rq3 = hdr->dring[(k + 3) & 7];
        const uint4 rq_a = *reinterpret_cast<const uint4*>(rq3);          // {probe list pos, probe chunks, partition, item id}
        const uint2 rq_b = *reinterpret_cast<const uint2*>(rq3 + 4);      // {first build-list entry, build chunks}
        const uint4 flg = *reinterpret_cast<const uint4*>(hdr->unused0);       // -, -, full[2]
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
        // ---- 2. P: the claims' answers are taken, a lane's first straggler asks its next bucket (that round trip runs under the
        // lookups); item k is probed; then its successor's first probe chunks are requested into the same registers; the stragglers
        // are placed; the fill counts of item k's parity are cleared for item k+2 ----
        if (FJ_WIDE_ABLATE & 2) {
#pragma unroll
            for (int j = 0; j < 8; ++j) nslots[j] = W_NOSLOT;
            if (((u32)(bkA[0] ^ bkA[1] ^ bkA[2] ^ bkA[3] ^ bkA[4] ^ bkA[5] ^ bkA[6] ^ bkA[7]) & 0xFFFFFu) == 0x12345u && bokA) nslots[0] = tid;      // (the loads stay)
        } else if (!sameA) claim_mid(fill_n, bkA, bokA, amA, old, nslots, pend, sb2, so2);
        zero_fill_counts(par);
        const bool full = __builtin_amdgcn_readfirstlane(par ? flg.w : flg.z) != 0 || ns0 > 2 * W_WAVES;      // (an item longer than 32 probe chunks - a host-side bug - goes to the retry ladder)
        u32 wave_hits = 0;
        const bool skip = full || ns0 == 0;
        if (!(FJ_WIDE_ABLATE & 1)) {
            if (!skip && na) wave_hits += probe4(ka, na);
            if (!skip && nbk) wave_hits += probe4(kb, nbk);
        } else wave_hits += (u32)(ka[0] ^ ka[1] ^ ka[2] ^ ka[3] ^ kb[0] ^ kb[1] ^ kb[2] ^ kb[3]) & 1u;      // (the loads stay)
        {
            const u32 nbn = ns1 < W_META_P ? ns1 : W_META_P;
            load_chunks2(sl_k1, nbn, ka, na, kb, nbk);
        }
        W_STAMP(1); W_WSTAMP(0, false);
        if (!sameA && !(FJ_WIDE_ABLATE & 2)) {
            if (!(FJ_WIDE_ABLATE & 8)) claim_resolve(fill_n, bkA, parn, nslots, pend, sb2, so2);
            if (is_big(nbc1, tot1) && tid == 0) hdr->full[parn] = 1;
        }
        // ---- 3. park what was requested ----
        park(sl_k3, mp, mb, mb2);
        if (tid == 0) { store_boff(k + 4, bo4); store_items(k + 5, it5); }
        if (lane == 0 && wave_hits) atomicAdd(&hdr->cnt, wave_hits);
        W_STAMP(2); W_WSTAMP(1, false);
        __syncthreads();                                         // A: every wave is done with the table in place; item k+1's slots are settled
        W_STAMP(3); W_WSTAMP(3, true);
        // ---- 4. I: item k's result (one lane of the LAST wave: it holds the fewest probe chunks); item k+1 goes in: plain stores ----
        if (tid == WNT - 64) {
            const u32 cnt = skip ? 0u : hdr->cnt;
            if (sameA) hdr->full[parn] = hdr->full[par];            // the table stays: so does its verdict
            hdr->cnt = 0; hdr->full[par] = 0;
            if (full) atomicOr(a.err, FJ_STAT_RETRY);            // the table could not hold the partition: redone with the tagged table
            a.part_count[id_of(k)] = full ? FJ_ITEM_RETRY : cnt;
            if (cnt) atomicAdd(a.total, (unsigned long long)cnt);
        }
        if (!sameA) {
            if (k + 1 < nmine && is_last_part(part1)) { fill_table(W_POISON2); __syncthreads(); }     // (uniform; once per launch at most)
            if (!(FJ_WIDE_ABLATE & 4)) store_keys(bkA, nslots, amA);
        }
        // the next batch of build keys moves up while the stores drain and the other waves arrive (its loads were issued at the top of
        // the iteration: they are in)
        if constexpr (DENSE) assemble(rawB, topB, bkA);
        else {
#pragma unroll
            for (int j = 0; j < 8; ++j) bkA[j] = bkB[j];
        }
        W_STAMP(4); W_WSTAMP(2, false);
        if (!(FJ_WIDE_ABLATE & 4)) __syncthreads();              // B
        W_STAMP(5); W_WSTAMP(3, true);
        // ---- 5. rotate ----
        bokA = bokB; amA = amB; tot1 = tot2; sameA = sameB;
        ns0 = ns1; ns1 = ns2; ns2 = ns3; nbc1 = nbc2; nbc2 = nbc3; part1 = part2; part2 = part3;
        u32* t = sl_k; sl_k = sl_k1; sl_k1 = sl_k2; sl_k2 = sl_k3; sl_k3 = t;
    }
#ifdef FJ_LAB
    if (a.dbg && tid == 0 && blockIdx.x < 4096) { for (int i = 0; i < 8; ++i) a.dbg[blockIdx.x * 8 + i] = tacc[i]; }
    __syncthreads();
    if (a.dbg && tid < 64 && blockIdx.x < 256) a.dbg[2048 * 8 + blockIdx.x * 64 + tid] = wacc[tid];
#endif
}

}  // namespace

#ifdef FJ_LAB
u32 fj_wide_lds_bytes() { return (u32)(sizeof(WHdr) + WS * 8 + NBK * 4 + 4 * W_STRIDE * 4 + 80 + 512); }     // (+ the stage stamps' accumulators)
#else
u32 fj_wide_lds_bytes() { return (u32)(sizeof(WHdr) + WS * 8 + NBK * 4 + 4 * W_STRIDE * 4 + 80); }
#endif     // (NBK * 4: two parities of NBK 16-bit counts)

// counting join over the final chunk sets with the 16384-slot table: a.items / a.nitems_dev / a.part_count / a.total / a.err as
// for fj_launch_lds_join.  dense: the build side comes from w (build-broadcast wire format).
hipError_t fj_launch_count_join_wide(const FjLdsJoinArgs& a, const FjWideArgs& w, bool dense, u32 grid, hipStream_t s) {
    if (!a.probe.list || !a.items || (!dense && !a.build.list) || a.want_dups) return hipErrorInvalidValue;
    if (dense && (w.nsrc == 0 || w.nsrc > FJ_WIDE_MAXSRC || !w.base || (w.mid_bytes != 2 && w.mid_bytes != 4) || w.bits > 32)) return hipErrorInvalidValue;
    if (w.group_log > 6) return hipErrorInvalidValue;
    if (w.pmask == 0 || a.nparts < 2) return hipErrorInvalidValue;    // (stale table entries are told from live ones by their partition bits: the plan has some)
    const u32 lds = fj_wide_lds_bytes();
    auto kern = !dense ? (w.group_log ? fj_count_join_wide<0, true> : fj_count_join_wide<0, false>)
              : w.mid_bytes == 2 ? (w.group_log ? fj_count_join_wide<2, true> : fj_count_join_wide<2, false>)
                                 : (w.group_log ? fj_count_join_wide<4, true> : fj_count_join_wide<4, false>);
    hipError_t e = fj_set_max_lds_once(reinterpret_cast<const void*>(kern), lds);
    if (e != hipSuccess) return e;
    if (grid == 0) grid = 1;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WNT), lds, s, a, w);
    return hipGetLastError();
}
