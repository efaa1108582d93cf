// fj_api.hip -- host orchestration + C ABI (include/flashjoin.h) for the MI355X hash join.
//
// Plays the role of the reference's join drivers (_hash_join_{radix,scalar}_{count,materialize}
// and adaptive_hash_join_*, hash_join.cpp:315-594): plan the radix passes, run partition ->
// group -> join on one HIP stream, read back the count, and fall back to the global-table path
// when a partition does not fit its LDS table.  No CPU join path exists here: without a HIP
// device every entry point fails with an error string.
#include "fj_internal.h"
#include "../../include/flashjoin.h"

#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

thread_local std::string g_err;
thread_local fj_timings g_last;

int set_err(const char* fmt, ...) {
    char buf[1024];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    g_err = buf;
    return 1;
}
#define HIPCHK(x)                                                                                          \
    do {                                                                                                   \
        hipError_t e_ = (x);                                                                               \
        if (e_ != hipSuccess) return set_err("%s:%d: %s failed: %s", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
    } while (0)

// The C ABI runs on the context's device and leaves the caller's current device as it found it (torch reads the
// runtime's current device: a join on cuda:1 must not move later torch allocations there).
struct DeviceGuard {
    int prev = -1; bool changed = false; hipError_t err = hipSuccess;
    explicit DeviceGuard(int dev) {
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != dev) { err = hipSetDevice(dev); changed = err == hipSuccess; }
    }
    ~DeviceGuard() { if (changed) (void)hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};
#define FJ_ON_DEVICE(dev)                                                                                   \
    DeviceGuard dev_guard_(dev);                                                                            \
    if (dev_guard_.err != hipSuccess) return set_err("selecting HIP device %d failed: %s", (int)(dev), hipGetErrorString(dev_guard_.err))
// Entry of a C-ABI call on context c: calls on one context are serialised (the workspace, the scratch words and the events
// are per context; fj_join_host re-enters through fj_join_device / fj_stream_*: a recursive lock), then the device guard.
#define FJ_ENTER(c)                                                                                         \
    std::lock_guard<std::recursive_mutex> ctx_lock_((c)->mu);                                                \
    FJ_ON_DEVICE((c)->device)

struct Scalars {                       // device scratch words, mirrored in pinned host memory
    unsigned long long total;
    unsigned long long expected;
    u64 empty_val;
    u32 err;
    u32 flags;
    u32 alloc[8];                      // [side*4 + pass] chunk allocators ([side*4 + 3]: the bloom stage's output pool)
    u32 seg_counter[8];                // [side*4 + pass] segment ids
    unsigned long long bloom_survivors;   // probe keys that passed the bloom precheck
    unsigned long long sample_hits;       // sampled probe rows found in the build side (adaptive bloom decision)
    u32 next_item;                     // work counter of the persistent join kernel
    u32 next_emit_item;                // ... and of the persistent emitting kernel (zeroed right before its launch)
    unsigned long long owner_counts[64], owner_cursors[64], owner_offsets[64];
    // owner shuffle, sender side (fj_shuffle_pack: may run while a stream join is open, so it has words of its own)
    u32 own_alloc[64];                 // chunks allocated in each owner's region
    u32 pack_err, pack_seg, pack_alloc_unused, rx_alloc;   // error word / segment counter of the packing pass; chunk count of a received piece
};

enum Slot {
    // [side][pingpong][kind]
    W_POOL_K = 0, W_POOL_V, W_DIR, W_LIST, W_BCHUNKS, W_REL, W_BOFF, W_SEGOFF, W_TOFF, W_TILES, W_KINDS,
    W_SIDE_STRIDE = 2 * W_KINDS,
    W_PART_COUNT = 2 * W_SIDE_STRIDE, W_OUT_OFF, W_GT_KEYS, W_GT_VALS, W_GT_BLOOM, W_WG_COUNT,
    W_H_BK, W_H_BV, W_H_PK, W_H_OK, W_H_OV, W_ROWIDX, W_BKEYS, W_BBASE,
    W_SH_REL, W_SH_SEGOFF, W_SH_BCH,                                   // fj_shuffle_pack scratch
    W_RX_REL, W_RX_LIST, W_RX_SEGOFF, W_RX_BCH, W_RX_BOFF, W_RX_TOFF, W_RX_TILES,   // a received piece as a chunk set
    W_SK_TILES_B, W_SK_TILES_P, W_SK_NT, W_PART_COUNT2, W_OUT_OFF2,                 // re-partitioning of oversized final partitions (skew_join)
    W_NSLOTS
};

struct Buf { void* p = nullptr; size_t bytes = 0; };

enum Ev { E_START = 0, E_BUILD, E_PPART, E_JOIN, E_EMIT0, E_EMIT1, E_SB0, E_SB1, E_BF0, E_BF1, E_H0, E_H1, E_H2, E_PK0, E_NEV = E_PK0 + 8 };

struct Pending {
    bool valid = false;
    int path = 0;
    FjLdsJoinArgs lds{};
    FjGtArgs gt{};
    u32 nitems = 0, gt_grid = 0;
    u64 count = 0;
    // oversized partitions that skew_join re-partitioned: their sub-partitions are a second item set, emitted behind the first
    bool has_second = false; FjLdsJoinArgs lds2{}; u32 nitems2 = 0; u64 count_main = 0; std::vector<u32> flagged;
    // duplicate build keys seen by the counting pass: the emitting pass must pick the FIRST occurrence's value
    bool has_dups = false;
    const u64* bk = nullptr; const u64* bv = nullptr; size_t nb = 0; int top_bits = 64;
};

// caller-provided output buffers large enough for ANY result (>= probe rows): the materialising join may run in one pass
struct SingleOut { u64* keys = nullptr; u64* vals = nullptr; size_t cap = 0; bool done = false; };

// bloom_level: 0 = no bloom precheck; L >= 1 = the probe side's level-L chunk set (output of its L-th pass) is filtered
// against per-bucket Bloom filters of the build side's level L before pass L+1 (csrc/fj_bloom.hip)
struct Plan { int bits = 0, npass = 0; int fan_log[4] = {0, 0, 0, 0}; int bloom_level = 0; };

// iteration state over the plan's passes for one relation (see pass_prepare / pass_launch / pass_complete)
struct PassIter {
    int side = 0; bool has_vals = false; size_t n = 0; Plan plan; int used = 64; u32 parents = 1; u64 lbound = 0;
    u32 tile_chunks = 16; int i = 0;
    FjChunkSet prev{}; bool have_prev = false; const uint4* tiles = nullptr; const u32* ntiles = nullptr; const u32* toff = nullptr;
    FjChunkSet cs{}; u32 Gmax = 1, F = 1, appends = 1;
    size_t piece_rows = 0;               // > 0: no single append of the first pass brings more rows than this (sizes the per-append slack)
    int slot = 0, cs_base = 0;           // ping-pong workspace slot of the next output level / base index of cs's buffers
    // probe side of a join: the final level's consumer is the join kernel; its item table (tiles of the final probe chunk
    // lists) and per-item count array are produced by the final level's bookkeeping launches
    bool want_items = false; u32 items_cap = 0; u32* part_count = nullptr; int part_count_slot = W_PART_COUNT;
    // bloom precheck (probe side): run the filter stage once `bloom_level` passes are complete, against bloom_build
    bool bloom_done = false; const FjChunkSet* bloom_build = nullptr;
    const u32* bloom_prebuilt = nullptr;            // filters shipped by another GPU (sender-side precheck) instead of bloom_build's keys
    unsigned long long* bloom_bucket_keys = nullptr; // [buckets] survivors per bucket (for flattening the survivors)
    // build side: keep a copy of the level the probe side's filter will read
    int save_level = 0; FjChunkSet saved{};
};

}  // namespace

struct StreamState {            // fj_stream_*: a counting join whose relations arrive in pieces
    bool active = false;
    Plan plan; PassIter pit, bit; FjLdsJoinArgs ja{};
    int top_bits = 64, evc = 0;
    size_t np_bound = 0, np_seen = 0, nb_bound = 0, nb_seen = 0;
    u32 p_appends_left = 0, b_appends_left = 0;
    bool probe_done = false, build_done = false;
    // zero-pass plans (build side <= one LDS table): the pieces are joined as flat arrays
    const u64* flat_build = nullptr;
    const u64* flat_probe[64]; size_t flat_np[64]; u32 nflat = 0;
    // every piece appended so far (they stay allocated until fj_stream_finish returns): what the HBM-table fallback reads
    std::vector<std::pair<const u64*, size_t>> bpieces, ppieces;
    // owner shuffle, receiver side (fj_stream_open_shuffled): the pieces are chunk pools that peers filled with the FIRST pass of
    // the global plan; this rank owns level-1 buckets [b_lo, b_lo + nbk) and runs the plan from its second pass on
    bool shuffled = false; u32 b_lo = 0, nbk = 0, nbk_pad = 0;
};

struct fj_ctx {
    int device = 0;
    Buf bufs[W_NSLOTS];
    hipEvent_t ev[E_NEV];
    hipStream_t side = nullptr;        // copy stream of the host-buffer entry (fj_join_host)
    Scalars* d_sc = nullptr;
    Scalars* h_sc = nullptr;
    Pending pend;
    StreamState st;
    size_t ws_bytes = 0;
    u32 num_cus = 256;
    void* stage[3] = {nullptr, nullptr, nullptr};     // pinned staging ring of the host-buffer entry (fj_join_host)
    size_t stage_bytes = 0;
    bool plan_in_flight = false;       // a plan was begun and has not completed (an error in between leaves chunk counts behind)
    bool slot_dirty[W_NSLOTS] = {};    // ... in which case every self-cleaning buffer is re-zeroed IN FULL before its next use (get_zeroed_buf)
    std::recursive_mutex mu;           // one C-ABI call at a time per context (FJ_ENTER)
};

namespace {

// Process-wide dispatch options (fj_set_option; initial values from the environment).
//   radix_threshold  : adaptive joins take the non-partitioned HBM table below this many build rows.  MI355X: the
//                      partitioned driver wins at every build size (<= 4096 rows it runs zero passes: one LDS table per
//                      workgroup over the flat inputs), so the switch point is 0 (tools/sweep_adaptive.py).
//   (schedule: build relation first, then the probe relation, on the caller's stream.  The two-stream and interleaved
//    schedules of rounds 1-2 were measured slower once the level bookkeeping was fused - EXPERIMENTS.md - and are gone.)
//   persistent_min_items : counting joins with at least this many (partition, slice) items run the persistent join
//                      kernel (resident workgroups that prefetch the next item); below it one workgroup per item.
//   scalar_hbm_table : 1 = the reference's "scalar" functions (hash_join*, one table for the whole build side) use the
//                      non-partitioned HBM table at every size; 0 (default) = they use it only as the fallback and
//                      otherwise run the partitioned plan.  One table for B rows means one cache-missing 64-B access
//                      per probe in HBM -- more traffic than the 40 B per probe the two streaming passes + LDS join
//                      move -- so on this machine "scalar" is the slower way to the same result at every size.
struct Options {
    size_t radix_threshold; int scalar_hbm_table; u32 persistent_min_items; u32 plan_target_keys;
    int bloom_variant, bloom_auto, bloom_auto_max_hit_bp, mat_single_pass;
    Options() {
        mat_single_pass = getenv("FJ_MAT_SINGLE_PASS") ? atoi(getenv("FJ_MAT_SINGLE_PASS")) : 1;
        bloom_auto = getenv("FJ_BLOOM_AUTO") ? atoi(getenv("FJ_BLOOM_AUTO")) : 1;
        bloom_auto_max_hit_bp = getenv("FJ_BLOOM_AUTO_MAX_HIT_BP") ? atoi(getenv("FJ_BLOOM_AUTO_MAX_HIT_BP")) : 2300;     // measured break-even at c4 sizes: 24 % hits (profiles/r03_bloom_threshold.csv; round 2: 28 % - the plain plan gained more since)
        const char* bvr = getenv("FJ_BLOOM_VARIANT");
        bloom_variant = bvr ? std::min(2, std::max(0, atoi(bvr))) : 0;
        const char* pt = getenv("FJ_PLAN_TARGET_KEYS");
        plan_target_keys = pt ? (u32)strtoul(pt, nullptr, 10) : FJ_PART_TARGET_KEYS;
        if (plan_target_keys < 16 || plan_target_keys > FJ_PART_TARGET_KEYS) plan_target_keys = FJ_PART_TARGET_KEYS;
        const char* th = getenv("FJ_RADIX_THRESHOLD");
        radix_threshold = th ? (size_t)strtoull(th, nullptr, 10) : (size_t)0;
        const char* sg = getenv("FJ_SCALAR_HBM_TABLE");
        scalar_hbm_table = sg ? atoi(sg) : 0;
        const char* pm = getenv("FJ_PERSISTENT_MIN_ITEMS");
        persistent_min_items = pm ? (u32)strtoul(pm, nullptr, 10) : 8192u;
    }
};
Options& options() { static Options o; return o; }

int get_buf(fj_ctx* c, int slot, size_t bytes, void** out) {
    Buf& b = c->bufs[slot];
    if (bytes == 0) bytes = 16;
    if (b.bytes < bytes) {
        if (b.p) { HIPCHK(hipFree(b.p)); c->ws_bytes -= b.bytes; b.p = nullptr; b.bytes = 0; }
        size_t want = (bytes + 255) & ~(size_t)255;
        hipError_t e = hipMalloc(&b.p, want);
        if (e != hipSuccess) return set_err("hipMalloc(%zu bytes) for workspace slot %d failed: %s", want, slot, hipGetErrorString(e));
        b.bytes = want; c->ws_bytes += want;
    }
    *out = b.p;
    return 0;
}


// A buffer that is zero whenever nobody is using it: its consumer clears what it read (bucket chunk counts: fj_level_scan),
// so a join needs no memset for it.  Zeroed here - the WHOLE allocation, not just the bytes this plan asks for: a later,
// wider plan must not find what an abandoned one left behind - when it is (re)allocated, and when a plan on this context did
// not run to completion since the slot was last cleared (slot_dirty, set for every slot by begin_plan, cleared per slot here).
int get_zeroed_buf(fj_ctx* c, int slot, size_t bytes, void** out, hipStream_t s) {
    const size_t before = c->bufs[slot].bytes;          // (a re-allocation may well return the old address: compare sizes)
    if (get_buf(c, slot, bytes, out)) return 1;
    if (c->bufs[slot].bytes != before || c->slot_dirty[slot]) {
        HIPCHK(hipMemsetAsync(*out, 0, c->bufs[slot].bytes, s));
        c->slot_dirty[slot] = false;
    }
    return 0;
}

// passes for `bits` radix bits: one pass up to 9 bits (256 buckets, 128-B lines, one 1024-thread workgroup per CU; 512 buckets for
// exactly 9 bits: slower per row than an 8-bit pass, far cheaper than two passes), two passes up to 18 bits (8-bit passes
// while they reach, a 9-bit pass beyond 16 bits), three beyond
int plan_npass(int bits) { return bits <= FJ_MAX_FAN_LOG ? (bits > 0 ? 1 : 0) : (bits <= 2 * FJ_MAX_FAN_LOG ? 2 : (bits + FJ_MAX_FAN_LOG - 1) / FJ_MAX_FAN_LOG); }

void plan_passes(Plan& p, bool extra_first) {
    p.npass = plan_npass(p.bits);
    for (int i = 0; i < p.npass; ++i) {
        const int rem = p.bits % p.npass;
        p.fan_log[i] = p.bits / p.npass + ((extra_first ? i < rem : i >= p.npass - rem) ? 1 : 0);
    }
}

Plan make_plan(size_t nb, int top_bits, bool want_bloom = false, u64 target_override = 0) {
    Plan p;
    const u64 target = target_override ? std::min<u64>(target_override, options().plan_target_keys)
                                       : options().plan_target_keys;   // 4096 in production; smaller values make small inputs take deep plans (tests)
    if (nb > target) {
        u64 parts = (nb + target - 1) / target;
        while ((1ull << p.bits) < parts) ++p.bits;
    }
    // Counting joins keep a partition in a 2-location cuckoo table of 8192 slots, reliable to a load of ~0.42 and useless
    // above 0.5 (stash used by < 1 % of the tables at 0.40, by 39 % at 0.48).  Where one more radix bit costs no
    // extra pass the plan takes it once the average partition exceeds FJ_PLAN_BUMP_KEYS (nb at 4096 * 2^k would otherwise
    // put half of the partitions over the table's limit); partitions that still overflow are redone one by one on the
    // tagged table (fj_launch_lds_join_retry), not by re-running the whole join.
    if (!target_override && target == FJ_PART_TARGET_KEYS && (nb >> p.bits) > FJ_PLAN_BUMP_KEYS && p.bits < top_bits - 32) {
        const int nb1 = p.bits == 0 ? 5 : p.bits + 1;
        if (p.bits == 0 || plan_npass(nb1) == plan_npass(p.bits)) p.bits = nb1;
    }
    if (p.bits > 0 && p.bits < 5) p.bits = 5;              // a pass with a tiny fan-out serialises on its per-bucket threads
    if (p.bits > top_bits - 32) p.bits = top_bits - 32;      // radix digits come from hash word 1 (32 bits)
    // extra bits go to the EARLIER passes: a pass over a flat array absorbs a wider fan-out better than one over chunk
    // lists (per-bucket carry work grows with the fan-out), and fewer children per parent leave fewer partial chunks:
    // 8+7 instead of 7+8 bits at c3 is 1.3 % faster end to end (A/B on one box)
    plan_passes(p, true);
    if (want_bloom && p.npass >= 2) {
        // The precheck filters the input of the LAST pass (level npass-1) with an LDS-resident filter per bucket of that
        // level: it needs <= FJ_BLOOM_MAX_KEYS build keys per bucket, and is strong below FJ_BLOOM_GOOD_KEYS.  A two-pass
        // plan moves bits into its first pass (up to 9: the 512-bucket kernel) to get there; the final partitions are the
        // same either way (digits are consecutive bits of hash word 1).
        if (p.npass == 2)
            while (p.fan_log[0] < FJ_MAX_FAN_LOG && p.fan_log[1] > 4 && (nb >> p.fan_log[0]) > FJ_BLOOM_GOOD_KEYS) { ++p.fan_log[0]; --p.fan_log[1]; }
        int lvl_bits = 0;
        for (int i = 0; i + 1 < p.npass; ++i) lvl_bits += p.fan_log[i];
        if ((nb >> lvl_bits) <= FJ_BLOOM_MAX_KEYS) p.bloom_level = p.npass - 1;
    }
    return p;
}

// ---- partition passes over one relation, as a small state machine ------------------------------
// prepare (allocate + clear the output pool of the next pass) -> launch (once, or several times when the
// input arrives in pieces: launches accumulate into the same pool) -> complete (chunk lists, tile table).

void pass_init(PassIter& it, int side, bool has_vals, size_t n, const Plan& plan, int top_bits) {
    it = PassIter();
    it.side = side; it.has_vals = has_vals; it.n = n; it.plan = plan; it.used = top_bits;
    it.lbound = (n + FJ_CHUNK - 1) / FJ_CHUNK;
}

// workgroups for a launch over n rows: enough to fill the chip, but every (workgroup, bucket) pair ends in a
// partial chunk, so keep >= ~128 rows per pair or the consumers drown in tiny chunks (a join item fetches its build rows
// 16 chunks at a time: at B = 1M, 61 workgroups x 256 buckets left 64-row chunks and four fetch rounds per table).
// Swept on c2 (1M x 100M) and 1M x 10M: 64 rows per pair 0.858 / 0.349 ms, 128: 0.789 / 0.304, 256: 0.808 / 0.329,
// 512: 0.872 / 0.427 (the pass itself slows down with fewer workgroups); c3 is not affected.
u32 pass_groups(u64 chunks, u64 rows, u32 tile_chunks, u32 F) {
    const u64 g64 = std::min<u64>(chunks / tile_chunks, rows / ((u64)F * 128));
    return (u32)std::min<u64>(512, std::max<u64>(1, g64));
}

int pass_prepare(fj_ctx* c, PassIter& it, u32 appends, hipStream_t s) {
    const int i = it.i;
    it.F = 1u << it.plan.fan_log[i];
    it.used -= it.plan.fan_log[i];
    it.appends = appends ? appends : 1;
    it.tile_chunks = fj_partition_tile_chunks((u32)it.plan.fan_log[i], it.has_vals);
    it.Gmax = pass_groups(it.lbound, it.n, it.tile_chunks, it.F);
    if (it.piece_rows && it.appends > 1 && !it.have_prev)      // many small appends: each launch has few workgroups, so little slack per append
        it.Gmax = std::min(it.Gmax, pass_groups((it.piece_rows + FJ_CHUNK - 1) / FJ_CHUNK, it.piece_rows, it.tile_chunks, it.F));
    const u32 F = it.F, G = it.Gmax, parents = it.parents;
    const u64 nb_out = (u64)parents * F;
    // chunk ids: the rows' own + per (segment, bucket) one partial chunk and the unused rest of its last run + per workgroup
    // and launch what is left of its last slab
    const u64 cap64 = (it.n / FJ_CHUNK + 1 + (((u64)(G + parents) * F << FJ_RUN_LOG) + (u64)(G + 1) * fj_slab_for(it.appends)) * it.appends + 3) & ~3ull;
    if (cap64 >= (1ull << 24) || nb_out >= (1u << 22))
        return set_err("relation of %zu rows is too large for one GPU's chunk directory", it.n);
    FjChunkSet cs{};
    cs.cap = (u32)cap64; cs.nb = (u32)nb_out; cs.n_flat = 0; cs.fan_mask = F - 1; cs.max_segs = (G + parents + 2) * it.appends;
    // Chunk ids in runs of 4 make the level's bookkeeping cheaper (fj_level_lists: 70 -> 40 us per 1B-row level), but a flat
    // pass of <= 256 buckets writes ~2 % slower with them than with ids handed out densely in the order its tiles open chunks
    // (3.13 -> 3.21 ms per 1B rows; chunk-list passes and the 512-bucket flat pass do not care or gain: EXPERIMENTS.md)
    cs.run_log = (!it.have_prev && it.i == 0 && F <= 256) ? 0u : (u32)FJ_RUN_LOG;      // (i > 0 without a previous level yet: a shuffled stream, whose pieces arrive as chunk lists)
    const int base = it.side * W_SIDE_STRIDE + (it.slot & 1) * W_KINDS;
    it.cs_base = base;
    void* p;
    if (get_buf(c, base + W_POOL_K, cap64 * FJ_CHUNK * 8, &p)) return 1; cs.keys = (u64*)p;
    cs.vals = nullptr;
    if (it.has_vals) { if (get_buf(c, base + W_POOL_V, cap64 * FJ_CHUNK * 8, &p)) return 1; cs.vals = (u64*)p; }
    if (get_buf(c, base + W_DIR, cap64 * 4, &p)) return 1; cs.dir = (u32*)p;
    if (get_buf(c, base + W_REL, cap64 * 8, &p)) return 1; cs.rel = (u64*)p;
    if (get_buf(c, base + W_LIST, cap64 * 4, &p)) return 1; cs.list = (u32*)p;
    if (get_zeroed_buf(c, base + W_BCHUNKS, nb_out * 4, &p, s)) return 1; cs.bchunks = (u32*)p;
    if (get_buf(c, base + W_BOFF, (nb_out + 1) * 4, &p)) return 1; cs.boff = (u32*)p;
    if (get_buf(c, base + W_SEGOFF, (size_t)cs.max_segs * F * 4, &p)) return 1; cs.seg_off = (u32*)p;
    cs.alloc = &c->d_sc->alloc[it.side * 4 + i];
    // No memsets: the directory word of every chunk id below the allocator's high-water mark is written by the workgroup
    // that took the id (unused ids are marked at its exit); bchunks is cleared by its reader; cs.alloc and this pass's
    // segment counter are zero (clear_plan_scalars at the start of the join).
    it.cs = cs;
    return 0;
}

// one launch of the prepared pass: over the previous level (keys == nullptr) or over a flat array of n rows
int pass_launch(fj_ctx* c, PassIter& it, const u64* keys, const u64* vals, size_t n, hipStream_t s, int* ev_cursor) {
    const FjChunkSet& cs = it.cs;
    FjPartArgs a{};
    u32 G = it.Gmax;
    if (it.have_prev) {
        a.in_keys = it.prev.keys; a.in_vals = it.prev.vals; a.in_list = it.prev.list; a.in_dir = it.prev.dir;
        a.in_tiles = it.tiles; a.in_ntiles = it.ntiles; a.n_flat = 0;
    } else {
        a.in_keys = keys; a.in_vals = vals; a.in_list = nullptr; a.in_dir = nullptr; a.in_tiles = nullptr; a.in_ntiles = nullptr; a.n_flat = n;
        G = std::min(G, pass_groups((n + FJ_CHUNK - 1) / FJ_CHUNK, n, it.tile_chunks, it.F));
    }
    a.parent0 = 0;
    a.out_keys = cs.keys; a.out_vals = cs.vals; a.out_dir = cs.dir; a.out_rel = cs.rel; a.seg_off = cs.seg_off;
    a.bchunks = cs.bchunks; a.alloc = cs.alloc; a.seg_counter = &c->d_sc->seg_counter[it.side * 4 + it.i];
    a.cap_chunks = cs.cap; a.max_segs = cs.max_segs;
    a.err = &c->d_sc->err;
    a.shift = (u32)it.used; a.fan_log = (u32)it.plan.fan_log[it.i]; a.slab = fj_slab_for(it.appends); a.run_log = cs.run_log; a.side = (u32)it.side;
    // 128-B lines: 64-B pieces cost ~27 % of scatter bandwidth (tools/ubench_scatter); only a 512-bucket pass that
    // also carries values has to fall back to them (LDS)
    const int line_log = (it.has_vals && it.F > 256) ? 3 : 4;
    if (ev_cursor) HIPCHK(hipEventRecord(c->ev[E_PK0 + 2 * (*ev_cursor)], s));
    HIPCHK(fj_launch_partition(a, it.has_vals, line_log, G, s));
    if (ev_cursor) { HIPCHK(hipEventRecord(c->ev[E_PK0 + 2 * (*ev_cursor) + 1], s)); ++*ev_cursor; }
    return 0;
}

bool bloom_stage_follows(const PassIter& it, int level) { return it.plan.bloom_level == level && it.bloom_build && !it.bloom_done; }

// Work items of the join over the final probe level: tiles of `tc` chunks of the probe chunk lists.  Few partitions
// (< 2048): several slices per partition, each rebuilding the partition's table, so that small builds still fill the chip
// (a slice keeps >= 32 full chunks of probe rows per table build).  Many partitions: one item per partition, except that
// a partition swollen by a hot key is cut into slices of 4x the average (>= 512 chunks).  `bound` over-estimates the
// chunk count (partial chunks), which only makes slices a little longer than planned.
void join_item_geometry(u64 nparts, size_t np, u64 chunk_bound, u32* tc, u64* max_items) {
    const u64 pchunks = (np + FJ_CHUNK - 1) / FJ_CHUNK;
    const u64 bound = std::max<u64>(chunk_bound, pchunks);
    const u64 avg = std::max<u64>(1, bound / nparts);
    u64 want = 1;
    static const u64 target_items = getenv("FJ_JOIN_ITEMS_TARGET") ? strtoull(getenv("FJ_JOIN_ITEMS_TARGET"), nullptr, 10) : 2048;   // (tuning knob)
    if (nparts < target_items) want = std::min<u64>((target_items + nparts - 1) / nparts, std::max<u64>(1, (pchunks / nparts) / 32));
    *tc = (u32)(want > 1 ? std::max<u64>(8, (avg * 9 / 8 + want - 1) / want) : std::max<u64>(512, 4 * avg));
    *max_items = bound / *tc + nparts + 1;
}

// Bookkeeping of the level in it.cs (buffers at it.cs_base), two launches: chunk-list offsets + lists, and the tile table
// of whatever reads the level next - the bloom stage, the next pass, or (probe side, final level) the join's item table
// together with its per-item count array.
int level_finish(fj_ctx* c, PassIter& it, bool final_level, hipStream_t s) {
    const FjChunkSet& cs = it.cs;
    u32 tc = 0; u64 max_tiles = 0; u32* zero_tail = nullptr;
    if (bloom_stage_follows(it, it.i)) tc = fj_bloom_tile_chunks();
    else if (!final_level) tc = fj_partition_tile_chunks((u32)it.plan.fan_log[it.i], it.has_vals);
    else if (it.want_items) join_item_geometry(cs.nb, it.n, it.lbound, &tc, &max_tiles);
    if (tc && !max_tiles) max_tiles = it.lbound / tc + cs.nb + 1;
    if (max_tiles >= (1ull << 31)) return set_err("internal error: tile table too large");
    u32* toff = nullptr; uint4* tiles = nullptr;
    void* p;
    if (tc) {
        if (get_buf(c, it.cs_base + W_TOFF, ((size_t)cs.nb + 1) * 4, &p)) return 1; toff = (u32*)p;
        if (get_buf(c, it.cs_base + W_TILES, max_tiles * sizeof(uint4), &p)) return 1; tiles = (uint4*)p;
        if (final_level) {
            if (get_buf(c, it.part_count_slot, (size_t)max_tiles * 4, &p)) return 1;
            it.part_count = zero_tail = (u32*)p; it.items_cap = (u32)max_tiles;
        }
    }
    HIPCHK(fj_launch_group(cs, tc, toff, tiles, (u32)max_tiles, zero_tail, s));
    it.tiles = tiles; it.ntiles = toff ? toff + cs.nb : nullptr; it.toff = toff;
    return 0;
}

int pass_complete(fj_ctx* c, PassIter& it, hipStream_t s) {
    const FjChunkSet& cs = it.cs;
    it.lbound = it.n / FJ_CHUNK + 1 + (u64)(it.Gmax + it.parents) * it.F * it.appends;
    ++it.i;                                   // (level_finish looks at the stage that follows the level just completed)
    if (level_finish(c, it, it.i == it.plan.npass, s)) return 1;
    it.prev = cs; it.have_prev = true;
    it.parents = cs.nb;
    ++it.slot;
    if (it.save_level == it.i) it.saved = cs;
    return 0;
}

// Bloom precheck between two probe-side passes: it.prev (level bloom_level, tile table built for the filter kernel) ->
// a chunk set with the same buckets that holds only the keys that may be in the build side (csrc/fj_bloom.hip).
int bloom_stage(fj_ctx* c, PassIter& it, hipStream_t s) {
    const FjChunkSet in = it.prev;
    const u32 G = c->num_cus;
    const u32 nw = fj_bloom_waves_per_group();
    const u64 max_segs = (u64)nw * ((u64)in.nb + G) + 16;
    const u64 cap64 = it.n / FJ_CHUNK + 1 + max_segs + (u64)fj_bloom_slab_chunks() * nw * G;
    if (cap64 >= (1ull << 24)) return set_err("relation of %zu rows is too large for one GPU's chunk directory", it.n);
    FjChunkSet cs{};
    cs.cap = (u32)cap64; cs.nb = in.nb; cs.n_flat = 0; cs.fan_mask = 0; cs.max_segs = (u32)max_segs;
    const int base = it.side * W_SIDE_STRIDE + (it.slot & 1) * W_KINDS;
    void* p;
    if (get_buf(c, base + W_POOL_K, cap64 * FJ_CHUNK * 8, &p)) return 1; cs.keys = (u64*)p;
    cs.vals = nullptr;
    if (get_buf(c, base + W_DIR, cap64 * 4, &p)) return 1; cs.dir = (u32*)p;
    if (get_buf(c, base + W_REL, cap64 * 8, &p)) return 1; cs.rel = (u64*)p;
    if (get_buf(c, base + W_LIST, cap64 * 4, &p)) return 1; cs.list = (u32*)p;
    if (get_zeroed_buf(c, base + W_BCHUNKS, (size_t)cs.nb * 4, &p, s)) return 1; cs.bchunks = (u32*)p;
    if (get_buf(c, base + W_BOFF, ((size_t)cs.nb + 1) * 4, &p)) return 1; cs.boff = (u32*)p;
    if (get_buf(c, base + W_SEGOFF, (size_t)cs.max_segs * 4, &p)) return 1; cs.seg_off = (u32*)p;
    cs.alloc = &c->d_sc->alloc[it.side * 4 + 3];
    FjBloomArgs a{};
    a.pkeys = in.keys; a.plist = in.list; a.pnb = in.nb; a.tiles = it.tiles; a.toff = it.toff;
    a.bkeys = it.bloom_build->keys; a.blist = it.bloom_build->list; a.bboff = it.bloom_build->boff;
    a.out_keys = cs.keys; a.out_dir = cs.dir; a.out_rel = cs.rel; a.seg_off = cs.seg_off; a.bchunks = cs.bchunks;
    a.alloc = cs.alloc; a.seg_counter = &c->d_sc->seg_counter[it.side * 4 + 3];
    a.cap_chunks = cs.cap; a.max_segs = cs.max_segs; a.err = &c->d_sc->err; a.survivors = &c->d_sc->bloom_survivors;
    a.prebuilt = it.bloom_prebuilt; a.bucket_keys = it.bloom_bucket_keys;
    HIPCHK(hipEventRecord(c->ev[E_BF0], s));
    HIPCHK(fj_launch_bloom_filter(a, G, options().bloom_variant, s));
    HIPCHK(hipEventRecord(c->ev[E_BF1], s));
    it.cs = cs; it.cs_base = base;
    it.lbound = it.n / FJ_CHUNK + 1 + max_segs;
    it.bloom_done = true;
    if (level_finish(c, it, it.i == it.plan.npass, s)) return 1;   // the next pass reads the survivors (none follows a sender-side precheck)
    it.prev = cs;
    ++it.slot;
    return 0;
}

// run every remaining pass of `it` (input of pass 0: the flat arrays); `out` describes the final level
int run_passes(fj_ctx* c, PassIter& it, const u64* keys, const u64* vals, hipStream_t s, FjChunkSet* out, int* ev_cursor) {
    while (it.i < it.plan.npass) {
        if (bloom_stage_follows(it, it.i) && bloom_stage(c, it, s)) return 1;
        if (pass_prepare(c, it, 1, s)) return 1;
        if (pass_launch(c, it, keys, vals, it.n, s, ev_cursor)) return 1;
        if (pass_complete(c, it, s)) return 1;
    }
    if (!it.have_prev) {    // no pass needed: the join kernel reads the flat arrays as virtual chunks
        it.prev = FjChunkSet();
        it.prev.keys = const_cast<u64*>(keys); it.prev.vals = const_cast<u64*>(vals); it.prev.n_flat = it.n; it.prev.list = nullptr; it.prev.nb = 1;
    }
    *out = it.prev;
    return 0;
}

// result words + every pass's chunk allocator and segment counter, in one fill (each small memset is a ~5 us launch)
int clear_plan_scalars(fj_ctx* c, hipStream_t s) {
    HIPCHK(hipMemsetAsync(c->d_sc, 0, offsetof(Scalars, owner_counts), s));
    return 0;
}

// Bracket of a partitioned plan: the self-cleaning buffers (get_zeroed_buf) are trusted only if the previous plan on this
// context ran all its bookkeeping.  begin_plan before the first pass_prepare, end_plan once the result was read back.
void begin_plan(fj_ctx* c) {
    if (c->plan_in_flight) for (bool& d : c->slot_dirty) d = true;
    c->plan_in_flight = true;
}
void end_plan(fj_ctx* c) { c->plan_in_flight = false; }

int read_scalars(fj_ctx* c, hipStream_t s) {
    HIPCHK(hipMemcpyAsync(c->h_sc, c->d_sc, sizeof(Scalars), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return 0;
}

float ev_ms(fj_ctx* c, int a, int b) { float ms = 0.f; (void)hipEventElapsedTime(&ms, c->ev[a], c->ev[b]); return ms; }

// diagnostic: per-item phase stamps (s_memrealtime, 100 MHz) written by thread 0 of the first 4096 workgroups of a join kernel
int stamps_begin(unsigned long long** dbg, hipStream_t s) {
    static unsigned long long* dbg_buf = nullptr;
    if (!dbg_buf) HIPCHK(hipMalloc((void**)&dbg_buf, 4096 * 8 * 8));
    HIPCHK(hipMemsetAsync(dbg_buf, 0, 4096 * 8 * 8, s));
    *dbg = dbg_buf;
    return 0;
}
int stamps_report(const char* label, const unsigned long long* dbg, u32 nitems, hipStream_t s) {
    std::vector<unsigned long long> h(4096 * 8);
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost));
    double acc[6] = {0, 0, 0, 0, 0, 0}; int n = 0;
    unsigned long long tmin = ~0ull, tmax = 0;
    for (int i = 0; i < 4096 && i < (int)nitems; ++i) {
        const unsigned long long* r = &h[i * 8];
        if (!r[0] || !r[5]) continue;
        for (int j = 1; j <= 5; ++j) acc[j] += (double)(r[j] - r[j - 1]) * 0.01;      // 100 MHz -> us
        if (r[0] < tmin) tmin = r[0];
        if (r[5] > tmax) tmax = r[5];
        ++n;
    }
    if (n == 0) n = 1;
    fprintf(stderr, "[%s] items=%d  meta+init=%.2f  build=%.2f  buildsync=%.2f  probe=%.2f  fin=%.2f us (means); first 4096 items span %.1f us\n",
            label, n, acc[1] / n, acc[2] / n, acc[3] / n, acc[4] / n, acc[5] / n, (double)(tmax - tmin) * 0.01);
    return 0;
}

int emit_pending(fj_ctx* c, u64* d_ok, u64* d_ov, size_t cap, hipStream_t s, fj_timings* t) {
    Pending& pd = c->pend;
    if (!pd.valid) return set_err("fj_emit_pairs: no counted materialising join is pending on this context");
    if (pd.count > cap) return set_err("fj_emit_pairs: output capacity %zu < %llu pairs", cap, (unsigned long long)pd.count);
    HIPCHK(hipEventRecord(c->ev[E_EMIT0], s));
    if (pd.count > 0) {
        if (((uintptr_t)d_ok | (uintptr_t)d_ov) & 7) return set_err("output buffers must be 8-byte aligned");
        void* p;
        if (pd.path == 0) {
            if (pd.has_dups) {
                // duplicate build keys: the reference's radix path keeps the FIRST occurrence (stable partition +
                // insert_local, hash_join.cpp:125).  Re-partition the build side with row indices as payload; the
                // join kernel keeps the smallest index per key and fetches its value from the caller's array.
                if (get_buf(c, W_ROWIDX, pd.nb * 8, &p)) return 1;
                u64* rowidx = (u64*)p;
                HIPCHK(fj_launch_iota(rowidx, pd.nb, s));
                HIPCHK(hipMemsetAsync(&c->d_sc->alloc[0], 0, sizeof(c->d_sc->alloc) + sizeof(c->d_sc->seg_counter), s));   // the build side's passes run again
                PassIter bit;
                pass_init(bit, 0, true, pd.nb, make_plan(pd.nb, pd.top_bits), pd.top_bits);
                begin_plan(c);
                if (run_passes(c, bit, pd.bk, rowidx, s, &pd.lds.build, nullptr)) return 1;
                end_plan(c);                  // (a pool error of these passes surfaces through the emit kernel's missing rows: same sizes as the counted join)
                pd.lds.dedup = 1; pd.lds.orig_vals = pd.bv;
            }
            if (pd.has_second)                  // the items of re-partitioned partitions emit nothing themselves: their sub-partitions do, below
                for (u32 idx : pd.flagged) HIPCHK(hipMemsetAsync(&pd.lds.part_count[idx], 0, 4, s));
            if (get_buf(c, W_OUT_OFF, ((size_t)pd.nitems + 1) * 8, &p)) return 1;
            HIPCHK(fj_launch_scan_u32_to_u64(pd.lds.part_count, (u64*)p, pd.nitems, s));
            pd.lds.out_off = (const u64*)p; pd.lds.out_keys = d_ok; pd.lds.out_vals = d_ov;
            pd.lds.dbg = nullptr;
            if (getenv("FJ_EMIT_STAMPS") && stamps_begin(&pd.lds.dbg, s)) return 1;
            static const bool resident = !getenv("FJ_EMIT_PERSISTENT") || atoi(getenv("FJ_EMIT_PERSISTENT")) != 0;   // (A/B knob)
            if (resident) HIPCHK(hipMemsetAsync(&c->d_sc->next_emit_item, 0, sizeof(u32), s));
            HIPCHK(fj_launch_lds_join(pd.lds, true, s, resident ? &c->d_sc->next_emit_item : nullptr, 1u));
            if (pd.lds.dbg) { if (stamps_report("FJ_EMIT_STAMPS", pd.lds.dbg, pd.nitems, s)) return 1; pd.lds.dbg = nullptr; }
            if (pd.has_second) {                // second item set: the sub-partitions of the oversized partitions, behind the first set's pairs
                if (get_buf(c, W_OUT_OFF2, ((size_t)pd.nitems2 + 1) * 8, &p)) return 1;
                HIPCHK(fj_launch_scan_u32_to_u64(pd.lds2.part_count, (u64*)p, pd.nitems2, s));
                pd.lds2.out_off = (const u64*)p; pd.lds2.out_keys = d_ok + pd.count_main; pd.lds2.out_vals = d_ov + pd.count_main;
                pd.lds2.dedup = 0; pd.lds2.orig_vals = nullptr; pd.lds2.dbg = nullptr;
                HIPCHK(fj_launch_lds_emit_retry(pd.lds2, s, false));  // (the tagged emit kernel over every item of the set: a few hundred items)
            }
        } else if (pd.path == 2) {           // many-to-many: count per item -> scan -> emit
            if (get_buf(c, W_OUT_OFF, ((size_t)pd.nitems + 1) * 8, &p)) return 1;
            HIPCHK(fj_launch_scan_u32_to_u64(pd.lds.part_count, (u64*)p, pd.nitems, s));
            pd.lds.out_off = (const u64*)p; pd.lds.out_keys = d_ok; pd.lds.out_vals = d_ov;
            HIPCHK(fj_launch_mm_join(pd.lds, true, s));
        } else {
            if (get_buf(c, W_OUT_OFF, ((size_t)pd.gt_grid + 1) * 8, &p)) return 1;
            HIPCHK(fj_launch_scan_u32_to_u64(pd.gt.wg_count, (u64*)p, pd.gt_grid, s));
            pd.gt.out_off = (const u64*)p; pd.gt.out_keys = d_ok; pd.gt.out_vals = d_ov;
            HIPCHK(fj_launch_gt_probe(pd.gt, true, pd.gt_grid, s));
        }
    }
    HIPCHK(hipEventRecord(c->ev[E_EMIT1], s));
    if (pd.count > 0) {
        // the emitting kernel can still refuse an item (a table that the counting pass's stricter cuckoo table accepted should
        // never do so, but nothing else enforces that): unwritten output rows must not be handed back with status 0
        if (read_scalars(c, s)) return 1;
        if (pd.path == 0 && (c->h_sc->err & FJ_STAT_EMIT_RETRY)) {
            // the cuckoo emit kernel marked items whose table overflowed its stash: those are redone on the tagged table
            HIPCHK(fj_launch_lds_emit_retry(pd.lds, s));
            HIPCHK(hipEventRecord(c->ev[E_EMIT1], s));
            if (read_scalars(c, s)) return 1;
            if (t) t->lds_retries += 1;
        }
        if (c->h_sc->err & (FJ_ERR_LDS_FULL | FJ_ERR_POOL)) { pd.valid = false; return set_err("fj_emit_pairs: the emitting pass could not place every partition in LDS (device error word 0x%x)", c->h_sc->err); }
    }
    HIPCHK(hipStreamSynchronize(s));
    if (t) { t->emit_ms = ev_ms(c, E_EMIT0, E_EMIT1); t->total_ms += t->emit_ms; t->probe_phase_ms += t->emit_ms; }
    pd.valid = false;
    return 0;
}

// non-partitioned path: one table in HBM (Infinity-Cache / L2 resident when small)
int join_global(fj_ctx* c, int bloom, int materialize, const u64* bk, const u64* bv, size_t nb, const u64* pk, size_t np,
                hipStream_t s, fj_timings* t, u64* out_count) {
    u64 cap = 64;
    while (cap < 2 * (u64)nb) cap <<= 1;
    FjGtArgs a{};
    void* p;
    if (get_buf(c, W_GT_KEYS, cap * 8, &p)) return 1; a.tkeys = (u64*)p;
    if (get_buf(c, W_GT_VALS, cap * 8, &p)) return 1; a.tvals = (u64*)p;
    a.bloom = nullptr;
    if (bloom) { if (get_buf(c, W_GT_BLOOM, cap / 8 * 4, &p)) return 1; a.bloom = (u32*)p; }
    const u64 npairs = (np + 1) / 2;
    const u32 grid = (u32)std::min<u64>(2048, std::max<u64>(1, npairs / 256));
    if (get_buf(c, W_WG_COUNT, (size_t)grid * 4, &p)) return 1; a.wg_count = (u32*)p;
    a.cap_mask = cap - 1; a.flags = &c->d_sc->flags; a.empty_val = &c->d_sc->empty_val;
    a.bk = bk; a.bv = bv; a.nb = nb; a.pk = pk; a.np = np; a.total = &c->d_sc->total;

    HIPCHK(hipEventRecord(c->ev[E_START], s));
    HIPCHK(hipMemsetAsync(c->d_sc, 0, offsetof(Scalars, alloc), s));
    HIPCHK(hipMemsetAsync(a.tkeys, 0xFF, cap * 8, s));
    if (a.bloom) HIPCHK(hipMemsetAsync(a.bloom, 0, cap / 8 * 4, s));
    HIPCHK(fj_launch_gt_build(a, s));
    HIPCHK(hipEventRecord(c->ev[E_BUILD], s));
    HIPCHK(hipEventRecord(c->ev[E_PPART], s));
    if (np > 0) HIPCHK(fj_launch_gt_probe(a, false, grid, s));
    else HIPCHK(hipMemsetAsync(a.wg_count, 0, (size_t)grid * 4, s));
    HIPCHK(hipEventRecord(c->ev[E_JOIN], s));
    if (read_scalars(c, s)) return 1;
    *out_count = c->h_sc->total;
    t->path = 1; t->passes = 0; t->radix_bits = 0; t->partitions = 1;
    t->build_phase_ms = ev_ms(c, E_START, E_BUILD);
    t->join_ms = ev_ms(c, E_PPART, E_JOIN);
    t->probe_phase_ms = t->join_ms;
    t->total_ms = ev_ms(c, E_START, E_JOIN);
    c->pend.valid = false;
    if (materialize) {
        c->pend.valid = true; c->pend.path = 1; c->pend.gt = a; c->pend.gt_grid = grid; c->pend.count = *out_count;
    }
    return 0;
}

// Build-side skew, recovered per partition (the reference maps partitions to threads statically and has no answer to skew,
// hash_join.cpp:507-510; rounds 1-2 re-ran the WHOLE join on one table in HBM, a 4.5x cliff at config-3 sizes for one bad
// partition).  The tagged kernel marked the items whose partition holds more distinct build keys than an LDS table takes
// (FJ_ITEM_TOOBIG); everything else has been joined.  Those partitions - a handful - are re-partitioned by S more radix
// bits of hash word 1 (one more pass over just their chunk lists, both sides: the pass kernel reads any tile table) and
// their sub-partitions are joined by the same kernels; the matches add to the same device total.  *ok = false when that is
// not possible (more than 64 such partitions, no hash bits left, sub-partitions still too large): the caller falls back.
int skew_join(fj_ctx* c, const FjLdsJoinArgs& ja, const Plan& plan, int top_bits, u32 nitems, int probe_slot, int materialize, hipStream_t s, bool* ok,
              u32* nparts_redone, Pending* pend) {
    *ok = false;
    if (!ja.items || !ja.build.list || !ja.probe.list || nitems == 0) return 0;
    std::vector<u32> pc(nitems);
    std::vector<uint4> items(nitems);
    HIPCHK(hipMemcpyAsync(pc.data(), ja.part_count, (size_t)nitems * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(items.data(), ja.items, (size_t)nitems * sizeof(uint4), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    std::vector<u32> parts, flagged;
    for (u32 i = 0; i < nitems; ++i) if (pc[i] == FJ_ITEM_TOOBIG) { parts.push_back(items[i].z); flagged.push_back(i); }
    std::sort(parts.begin(), parts.end());
    parts.erase(std::unique(parts.begin(), parts.end()), parts.end());
    const u32 m = (u32)parts.size();
    if (m == 0 || m > 64) return 0;
    // chunk-list ranges of those partitions on both sides
    std::vector<u32> bo(2 * m), po(2 * m);
    for (u32 j = 0; j < m; ++j) {
        HIPCHK(hipMemcpyAsync(&bo[2 * j], ja.build.boff + parts[j], 8, hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(&po[2 * j], ja.probe.boff + parts[j], 8, hipMemcpyDeviceToHost, s));
    }
    HIPCHK(hipStreamSynchronize(s));
    u64 bchunks = 0, pchunks = 0, bmax = 0;
    for (u32 j = 0; j < m; ++j) { const u64 nb_j = bo[2 * j + 1] - bo[2 * j]; bchunks += nb_j; bmax = std::max(bmax, nb_j); pchunks += po[2 * j + 1] - po[2 * j]; }
    int S = 1;
    while (S < FJ_MAX_FAN_LOG && ((bmax * FJ_CHUNK) >> S) > 2048) ++S;          // aim at half a cuckoo table per sub-partition
    if (((bmax * FJ_CHUNK) >> S) > 6000 || top_bits - plan.bits - S < 32) return 0;
    if (S < 5) S = std::min(5, top_bits - plan.bits - 32);                       // (a pass with a tiny fan-out serialises on its bucket threads)
    if (S < 1) return 0;
    Plan p2; p2.bits = S; p2.npass = 1; p2.fan_log[0] = S;
    auto run_side = [&](PassIter& it, int side, const FjChunkSet& in, const std::vector<u32>& off, u64 chunks, int tiles_slot, u32* d_nt) -> int {
        const bool vals = materialize && side == 0;               // a materialising join's build rows travel with their values
        const u32 tc = fj_partition_tile_chunks((u32)S, vals);
        std::vector<uint4> tiles;
        for (u32 j = 0; j < m; ++j)
            for (u32 pos = off[2 * j]; pos < off[2 * j + 1]; pos += tc) tiles.push_back(make_uint4(pos, std::min(tc, off[2 * j + 1] - pos), j, 0));
        const u32 nt = (u32)tiles.size();
        void* p;
        if (get_buf(c, tiles_slot, std::max<size_t>(1, tiles.size()) * sizeof(uint4), &p)) return 1;
        if (nt) HIPCHK(hipMemcpyAsync(p, tiles.data(), tiles.size() * sizeof(uint4), hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(d_nt, &nt, 4, hipMemcpyHostToDevice, s));
        HIPCHK(hipStreamSynchronize(s));                                         // (`tiles` and `nt` live on this stack frame)
        pass_init(it, side, vals, std::max<u64>(1, chunks * FJ_CHUNK), p2, top_bits - plan.bits);
        it.parents = m; it.lbound = chunks;
        it.slot = side ? probe_slot : plan.npass;            // the ping-pong half that does NOT hold the final level (a bloom stage took a slot of its own on the probe side)
        it.have_prev = true; it.prev = in; it.tiles = (const uint4*)p; it.ntiles = d_nt;
        if (side) { it.want_items = true; it.part_count_slot = W_PART_COUNT2; }
        // the main plan is done with its first pass's allocator and segment counter: they serve this pass
        HIPCHK(hipMemsetAsync(&c->d_sc->alloc[side * 4], 0, 4, s));
        HIPCHK(hipMemsetAsync(&c->d_sc->seg_counter[side * 4], 0, 4, s));
        if (pass_prepare(c, it, 1, s) || pass_launch(c, it, nullptr, nullptr, 0, s, nullptr) || pass_complete(c, it, s)) return 1;
        return 0;
    };
    void* p;
    if (get_buf(c, W_SK_NT, 16, &p)) return 1;
    u32* d_nt = (u32*)p;
    PassIter bit2, pit2;
    const u64 count_main = c->h_sc->total;                                       // what every other partition found
    const bool had_dups = (c->h_sc->err & FJ_STAT_DUPS) != 0;
    HIPCHK(hipMemsetAsync(&c->d_sc->err, 0, 4, s));                              // the main join's status bits have been acted on
    if (run_side(bit2, 0, ja.build, bo, bchunks, W_SK_TILES_B, d_nt)) return 1;
    if (run_side(pit2, 1, ja.probe, po, pchunks, W_SK_TILES_P, d_nt + 1)) return 1;
    FjLdsJoinArgs j2 = ja;
    j2.build = bit2.prev; j2.probe = pit2.prev; j2.nparts = j2.probe.nb; j2.nsplit = 1;
    j2.items = pit2.tiles; j2.nitems_dev = pit2.ntiles; j2.items_cap = pit2.items_cap; j2.part_count = pit2.part_count;
    j2.retry_only = 0; j2.mark_toobig = 0; j2.want_dups = materialize ? 1u : 0u; j2.dbg = nullptr;
    HIPCHK(fj_launch_lds_join(j2, false, s, nullptr, 0xFFFFFFFFu));              // one workgroup per item: a few hundred items
    if (read_scalars(c, s)) return 1;
    if ((c->h_sc->err & FJ_STAT_RETRY) && !(c->h_sc->err & (FJ_ERR_POOL | FJ_ERR_LDS_FULL))) {
        HIPCHK(fj_launch_lds_join_retry(j2, s));
        if (read_scalars(c, s)) return 1;
    }
    if (c->h_sc->err & FJ_ERR_POOL) return set_err("internal error: chunk pool exhausted while re-partitioning a skewed partition");
    if (c->h_sc->err & FJ_ERR_LDS_FULL) return 0;                                // sub-partitions still too large (keys colliding in all of hash word 1)
    if (materialize) {
        // duplicate build keys need the first-occurrence emit path, which re-partitions the whole build side: not combined with this one
        if (had_dups || (c->h_sc->err & FJ_STAT_DUPS)) return 0;
        pend->has_second = true; pend->lds2 = j2; pend->nitems2 = pit2.items_cap; pend->count_main = count_main; pend->flagged = flagged;
    }
    *ok = true; *nparts_redone = m;
    return 0;
}

// launch the per-partition join over the final chunk sets, read back count + error word, fill the timings
int radix_join_tail(fj_ctx* c, int materialize, FjLdsJoinArgs& ja, const Plan& plan, size_t np, const PassIter& pit, hipStream_t s,
                    fj_timings* t, int evc, u64* out_count, bool* lds_full, int top_bits, SingleOut* so = nullptr) {
    c->pend.has_second = false;
    ja.nparts = ja.probe.list ? ja.probe.nb : 1u << plan.bits;      // (an owner of a shuffled join holds a slice of the plan's partitions)
    const u64 pchunks = (np + FJ_CHUNK - 1) / FJ_CHUNK;
    void* p;
    u32 nitems;
    if (ja.probe.list) {
        // work items = tiles of the probe chunk lists, built with the final level's bookkeeping (level_finish)
        ja.items = pit.tiles; ja.nitems_dev = pit.ntiles; ja.items_cap = pit.items_cap; ja.part_count = pit.part_count;
        ja.nsplit = 1;
        nitems = pit.items_cap;
    } else {
        u64 nsplit = 1;
        if (ja.nparts < 2048) {
            nsplit = (2048 + ja.nparts - 1) / ja.nparts;
            const u64 per_part = pchunks / ja.nparts;
            nsplit = std::min<u64>(nsplit, std::max<u64>(1, per_part / 32));
        }
        ja.nsplit = (u32)nsplit; ja.items = nullptr; ja.nitems_dev = nullptr; ja.items_cap = 0;
        nitems = ja.nparts * ja.nsplit;
        if (get_buf(c, W_PART_COUNT, (size_t)nitems * 4, &p)) return 1; ja.part_count = (u32*)p;
    }
    ja.total = &c->d_sc->total; ja.err = &c->d_sc->err;
    ja.want_dups = materialize ? 1u : 0u; ja.dedup = 0; ja.orig_vals = nullptr; ja.retry_only = 0;
    ja.dbg = nullptr;
    ja.dbg_flags = getenv("FJ_JOIN_ABLATE") ? (u32)atoi(getenv("FJ_JOIN_ABLATE")) : 0u;
    if (so && materialize && ja.probe.list && ja.build.list && ja.build.vals && ja.items && !ja.dbg_flags) {
        // Single-pass materialising join: every item is probed ONCE; a probe round reserves its pairs' range on a device cursor
        // (the plan's `total` word) and writes them - no counting pass, no scan, no second read of the probe side (c3 sizes:
        // 13.4 -> ~12 ms).  It serves unique build keys; duplicates (reported exactly), a partition beyond the cuckoo table or
        // an output buffer that turns out too small leave the partitions in place and the two-pass path below takes over.
        FjLdsJoinArgs js = ja;
        js.out_cursor = &c->d_sc->total; js.out_capacity = so->cap; js.out_keys = so->keys; js.out_vals = so->vals; js.out_off = nullptr;
        HIPCHK(hipMemsetAsync(&c->d_sc->next_emit_item, 0, sizeof(u32), s));
        HIPCHK(fj_launch_emit_single(js, s, &c->d_sc->next_emit_item));
        HIPCHK(hipEventRecord(c->ev[E_JOIN], s));
        if (read_scalars(c, s)) return 1;
        if (c->h_sc->err & FJ_ERR_POOL) return set_err("internal error: chunk pool exhausted during a partition pass");
        if (!(c->h_sc->err & (FJ_STAT_DUPS | FJ_STAT_EMIT_RETRY | FJ_ERR_LDS_FULL | FJ_ERR_OUTCAP))) {
            end_plan(c);
            t->path = 0; t->passes = plan.npass; t->radix_bits = plan.bits; t->partitions = ja.nparts; t->lds_retries = 0;
            t->build_phase_ms = ev_ms(c, E_START, E_BUILD);
            t->join_ms = ev_ms(c, E_PPART, E_JOIN);
            t->probe_phase_ms = ev_ms(c, E_BUILD, E_JOIN);
            t->total_ms = ev_ms(c, E_START, E_JOIN);
            for (int i = 0; i < evc && i < 4; ++i) t->probe_part_kernel_ms[i] = ev_ms(c, E_PK0 + 2 * i, E_PK0 + 2 * i + 1);
            t->bloom_level = plan.bloom_level;
            if (plan.bloom_level > 0) { t->filter_ms = ev_ms(c, E_BF0, E_BF1); t->filter_survivors = c->h_sc->bloom_survivors; }
            *out_count = c->h_sc->total;
            c->pend.valid = false;
            so->done = true;
            return 0;
        }
        // not this time: clear what the attempt left in the scalars (the item counts are rewritten by the counting pass)
        HIPCHK(hipMemsetAsync(&c->d_sc->total, 0, sizeof(unsigned long long), s));
        HIPCHK(hipMemsetAsync(&c->d_sc->err, 0, sizeof(u32), s));
    }
    if (getenv("FJ_JOIN_STAMPS") && stamps_begin(&ja.dbg, s)) return 1;
    HIPCHK(fj_launch_lds_join(ja, false, s, &c->d_sc->next_item, options().persistent_min_items));
    if (ja.dbg) { if (stamps_report("FJ_JOIN_STAMPS", ja.dbg, nitems, s)) return 1; ja.dbg = nullptr; }
    HIPCHK(hipEventRecord(c->ev[E_JOIN], s));
    if (read_scalars(c, s)) return 1;
    t->lds_retries = 0;
    if ((c->h_sc->err & FJ_STAT_RETRY) && !(c->h_sc->err & (FJ_ERR_POOL | FJ_ERR_LDS_FULL))) {
        // some partitions overflowed the cuckoo table (load above ~0.45): those items run again on the tagged table; a
        // partition beyond that table too is marked (counting joins over chunk lists) and re-partitioned alone below
        ja.retry_only = 1; ja.mark_toobig = ja.items ? 1u : 0u;
        HIPCHK(fj_launch_lds_join_retry(ja, s));
        ja.retry_only = 0;
        HIPCHK(hipEventRecord(c->ev[E_JOIN], s));
        if (read_scalars(c, s)) return 1;
        t->lds_retries = 1;
        if ((c->h_sc->err & FJ_STAT_TOOBIG) && !(c->h_sc->err & (FJ_ERR_POOL | FJ_ERR_LDS_FULL))) {
            bool ok = false; u32 redone = 0;
            if (skew_join(c, ja, plan, top_bits, nitems, pit.slot, materialize, s, &ok, &redone, &c->pend)) return 1;
            HIPCHK(hipEventRecord(c->ev[E_JOIN], s));
            if (read_scalars(c, s)) return 1;
            if (ok) t->lds_retries = 1 + (int)redone;            // 1 + the partitions that were re-partitioned
            else c->h_sc->err |= FJ_ERR_LDS_FULL;                // not recoverable this way: the caller's whole-join fallback
        }
    }
    if (c->h_sc->err & FJ_ERR_POOL) return set_err("internal error: chunk pool exhausted during a partition pass");
    end_plan(c);                              // every prepared pass ran its bookkeeping: the self-cleaning buffers are clean
    t->path = 0; t->passes = plan.npass; t->radix_bits = plan.bits; t->partitions = ja.nparts;
    // one-shot joins: build_phase_ms = the build relation's passes, probe_phase_ms = first probe-side pass .. end of the join
    // (disjoint intervals of one stream).  Streamed joins overwrite both in fj_stream_finish.
    t->build_phase_ms = ev_ms(c, E_START, E_BUILD);
    t->join_ms = ev_ms(c, E_PPART, E_JOIN);
    t->probe_phase_ms = ev_ms(c, E_BUILD, E_JOIN);
    t->total_ms = ev_ms(c, E_START, E_JOIN);
    for (int i = 0; i < evc && i < 4; ++i) t->probe_part_kernel_ms[i] = ev_ms(c, E_PK0 + 2 * i, E_PK0 + 2 * i + 1);
    t->bloom_level = plan.bloom_level;
    if (plan.bloom_level > 0) {
        t->filter_ms = ev_ms(c, E_BF0, E_BF1); t->filter_survivors = c->h_sc->bloom_survivors;
    }
    if (c->h_sc->err & FJ_ERR_LDS_FULL) { *lds_full = true; return 0; }
    *out_count = c->h_sc->total;
    c->pend.valid = false;
    if (materialize) {
        c->pend.valid = true; c->pend.path = 0; c->pend.lds = ja; c->pend.nitems = nitems; c->pend.count = *out_count;
        c->pend.has_dups = (c->h_sc->err & FJ_STAT_DUPS) != 0;
    }
    return 0;
}

// radix path: partition both relations, then one LDS-table join per final partition
// bloom: 0 = no precheck, 1 = precheck whenever the plan allows one (the *_bloom functions), 2 = decide from a sample
// of the probe side (the adaptive_* functions): SURVEY 8(f) "bloom auto-enable by sampled hit rate"
int join_radix(fj_ctx* c, int materialize, int bloom, const u64* bk, const u64* bv, size_t nb, const u64* pk, size_t np, int top_bits,
               hipStream_t s, fj_timings* t, u64* out_count, bool* lds_full, SingleOut* so = nullptr) {
    Plan plan = make_plan(nb, top_bits, bloom != 0);
    *lds_full = false;
    t->sampled_hit_bp = -1;
    // a sample only pays where the precheck could: a filterable plan and a probe side that dominates the work
    if (bloom == 2 && (plan.bloom_level == 0 || np < 4 * nb || np < (1u << 24) || !options().bloom_auto)) {
        bloom = 0; plan = make_plan(nb, top_bits, false);
    }
    begin_plan(c);
    HIPCHK(hipEventRecord(c->ev[E_START], s));
    if (clear_plan_scalars(c, s)) return 1;
    FjLdsJoinArgs ja{};
    PassIter bit, pit;
    // a counting join never looks at a value: its build side moves keys only (half the build-phase bytes)
    pass_init(bit, 0, materialize != 0, nb, plan, top_bits);
    int evc = 0;
    if (plan.bloom_level > 0) bit.save_level = plan.bloom_level;
    // build relation first, then the probe relation, on the caller's stream (the build-side filter of a bloom plan needs
    // the whole build side anyway)
    if (run_passes(c, bit, bk, materialize ? bv : nullptr, s, &ja.build, nullptr)) return 1;
    HIPCHK(hipEventRecord(c->ev[E_BUILD], s));
    Plan pplan = plan;
    if (bloom == 2) {
        // Decide from a sample.  The build relation was partitioned with the filterable plan (its final partitions are the
        // same under either plan: digits are consecutive hash bits); FJ_SAMPLE_KEYS probe rows, evenly spaced, are looked
        // up in their final build partitions (a wave scans the partition's ~3000 keys: 25 MB of reads in all); the host
        // reads the hit count and picks the probe side's plan (~50 us).
        const u32 nsamp = FJ_SAMPLE_KEYS;
        HIPCHK(fj_launch_sample_hits(ja.build, pk, np, nsamp, (u32)(top_bits - 32 - plan.bits), (1u << plan.bits) - 1u, &c->d_sc->sample_hits, s));
        HIPCHK(hipMemcpyAsync(&c->h_sc->sample_hits, &c->d_sc->sample_hits, sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        const u32 hit_bp = (u32)(c->h_sc->sample_hits * 10000ull / nsamp);
        t->sampled_hit_bp = (int)hit_bp;
        if (hit_bp > (u32)options().bloom_auto_max_hit_bp) { pplan = make_plan(nb, top_bits, false); plan.bloom_level = 0; plan.npass = pplan.npass; }
    }
    pass_init(pit, 1, false, np, pplan, top_bits);
    pit.want_items = true;
    // bloom precheck: the probe side's level `bloom_level` is filtered against the build side's same level
    if (plan.bloom_level > 0) pit.bloom_build = &bit.saved;
    if (run_passes(c, pit, pk, nullptr, s, &ja.probe, &evc)) return 1;
    HIPCHK(hipEventRecord(c->ev[E_PPART], s));
    if (radix_join_tail(c, materialize, ja, plan, np, pit, s, t, evc, out_count, lds_full, top_bits, so)) return 1;
    if (c->pend.valid) { c->pend.bk = bk; c->pend.bv = bv; c->pend.nb = nb; c->pend.top_bits = top_bits; }
    return 0;
}

// EXTENSION: many-to-many inner join on the partitioned plan (csrc/fj_many.hip).  Build relation first (with its values when
// materialising), then the probe relation, then one workgroup per work item; no bloom stage, no fallback: a partition of
// more than 4096 build rows is an error.
int join_many(fj_ctx* c, int materialize, const u64* bk, const u64* bv, size_t nb, const u64* pk, size_t np, int top_bits,
              hipStream_t s, fj_timings* t, u64* out_count) {
    const Plan plan = make_plan(nb, top_bits, false, 2048);          // aim at half of the kernel's 4096 rows per partition
    begin_plan(c);
    HIPCHK(hipEventRecord(c->ev[E_START], s));
    if (clear_plan_scalars(c, s)) return 1;
    FjLdsJoinArgs ja{};
    PassIter bit, pit;
    pass_init(bit, 0, materialize != 0, nb, plan, top_bits);
    pass_init(pit, 1, false, np, plan, top_bits);
    pit.want_items = true;
    int evc = 0;
    if (run_passes(c, bit, bk, materialize ? bv : nullptr, s, &ja.build, nullptr)) return 1;
    HIPCHK(hipEventRecord(c->ev[E_BUILD], s));
    if (run_passes(c, pit, pk, nullptr, s, &ja.probe, &evc)) return 1;
    HIPCHK(hipEventRecord(c->ev[E_PPART], s));
    ja.nparts = 1u << plan.bits;
    u32 nitems;
    void* p;
    if (ja.probe.list) {
        ja.items = pit.tiles; ja.nitems_dev = pit.ntiles; ja.items_cap = pit.items_cap; ja.part_count = pit.part_count; ja.nsplit = 1;
        nitems = pit.items_cap;
    } else {
        const u64 pchunks = (np + FJ_CHUNK - 1) / FJ_CHUNK;
        ja.nsplit = (u32)std::min<u64>(2048, std::max<u64>(1, pchunks / 32)); ja.items = nullptr; ja.nitems_dev = nullptr; ja.items_cap = 0;
        nitems = ja.nparts * ja.nsplit;
        if (get_buf(c, W_PART_COUNT, (size_t)nitems * 4, &p)) return 1; ja.part_count = (u32*)p;
    }
    ja.total = &c->d_sc->total; ja.err = &c->d_sc->err;
    HIPCHK(fj_launch_mm_join(ja, false, s));
    HIPCHK(hipEventRecord(c->ev[E_JOIN], s));
    if (read_scalars(c, s)) return 1;
    if (c->h_sc->err & FJ_ERR_POOL) return set_err("internal error: chunk pool exhausted during a partition pass");
    end_plan(c);
    if (c->h_sc->err & FJ_ERR_LDS_FULL)
        return set_err("many-to-many join: a final partition holds more than 4096 build rows (a build key with thousands of duplicates?); not supported");
    *out_count = c->h_sc->total;
    t->path = 0; t->passes = plan.npass; t->radix_bits = plan.bits; t->partitions = ja.nparts;
    t->build_phase_ms = ev_ms(c, E_START, E_BUILD);
    t->join_ms = ev_ms(c, E_PPART, E_JOIN);
    t->probe_phase_ms = ev_ms(c, E_BUILD, E_JOIN);
    t->total_ms = ev_ms(c, E_START, E_JOIN);
    for (int i = 0; i < evc && i < 4; ++i) t->probe_part_kernel_ms[i] = ev_ms(c, E_PK0 + 2 * i, E_PK0 + 2 * i + 1);
    c->pend.valid = false;
    if (materialize) { c->pend.valid = true; c->pend.path = 2; c->pend.lds = ja; c->pend.nitems = nitems; c->pend.count = *out_count; c->pend.has_dups = false; }
    return 0;
}

fj_ctx* g_host_ctx = nullptr;


}  // namespace

void fj_set_error_string(const char* msg) { g_err = msg ? msg : ""; }     // (csrc/fj_dist.hip reports through the same thread-local string)

extern "C" {

const char* fj_last_error(void) { return g_err.c_str(); }
const char* fj_version(void) { return "flash_hash_join_amd 0.2 (gfx950)"; }

int fj_set_option(const char* name, long long value) {
    if (!name) return set_err("fj_set_option: null name");
    if (!strcmp(name, "radix_threshold")) { if (value < 0) return set_err("fj_set_option: radix_threshold must be >= 0"); options().radix_threshold = (size_t)value; return 0; }
    if (!strcmp(name, "scalar_hbm_table")) { options().scalar_hbm_table = value != 0; return 0; }
    if (!strcmp(name, "plan_target_keys")) { if (value < 16 || value > (long long)FJ_PART_TARGET_KEYS) return set_err("fj_set_option: plan_target_keys must be 16..%u", FJ_PART_TARGET_KEYS); options().plan_target_keys = (u32)value; return 0; }
    if (!strcmp(name, "bloom_auto")) { options().bloom_auto = value != 0; return 0; }
    if (!strcmp(name, "mat_single_pass")) { options().mat_single_pass = value != 0; return 0; }
    if (!strcmp(name, "bloom_auto_max_hit_bp")) { if (value < 0 || value > 10000) return set_err("fj_set_option: bloom_auto_max_hit_bp must be 0..10000"); options().bloom_auto_max_hit_bp = (int)value; return 0; }
    if (!strcmp(name, "bloom_variant")) { if (value < 0 || value > 2) return set_err("fj_set_option: bloom_variant must be 0..2"); options().bloom_variant = (int)value; return 0; }
    if (!strcmp(name, "persistent_min_items")) { if (value < 0) return set_err("fj_set_option: persistent_min_items must be >= 0"); options().persistent_min_items = (u32)std::min<long long>(value, 0xFFFFFFFFll); return 0; }
    return set_err("fj_set_option: unknown option '%s'", name);
}

long long fj_get_option(const char* name) {
    if (name && !strcmp(name, "radix_threshold")) return (long long)options().radix_threshold;
    if (name && !strcmp(name, "scalar_hbm_table")) return options().scalar_hbm_table;
    if (name && !strcmp(name, "persistent_min_items")) return options().persistent_min_items;
    if (name && !strcmp(name, "plan_target_keys")) return options().plan_target_keys;
    if (name && !strcmp(name, "bloom_variant")) return options().bloom_variant;
    if (name && !strcmp(name, "bloom_auto")) return options().bloom_auto;
    if (name && !strcmp(name, "mat_single_pass")) return options().mat_single_pass;
    if (name && !strcmp(name, "bloom_auto_max_hit_bp")) return options().bloom_auto_max_hit_bp;
    set_err("fj_get_option: unknown option '%s'", name ? name : "(null)");
    return -1;
}

uint64_t fj_key_mix64(uint64_t key) { return fj_key_mix(key); }
uint64_t fj_key_unmix64(uint64_t mixed) { return fj_key_unmix(mixed); }

int fj_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int fj_initialize(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) return set_err("fj_initialize: no usable HIP device (%s)", e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    return 0;
}

fj_ctx* fj_ctx_create(int device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || device < 0 || device >= n) { set_err("fj_ctx_create: HIP device %d not available (%d devices)", device, n); return nullptr; }
    DeviceGuard guard(device);
    if (guard.err != hipSuccess) { set_err("fj_ctx_create: hipSetDevice(%d) failed", device); return nullptr; }
    fj_ctx* c = new fj_ctx();
    c->device = device;
    { int ncu = 0; if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && ncu > 0) c->num_cus = (u32)ncu; }
    bool ok = hipMalloc((void**)&c->d_sc, sizeof(Scalars)) == hipSuccess &&
              hipHostMalloc((void**)&c->h_sc, sizeof(Scalars), hipHostMallocDefault) == hipSuccess &&
              hipMemset(c->d_sc, 0, sizeof(Scalars)) == hipSuccess;
    for (int i = 0; ok && i < E_NEV; ++i) ok = hipEventCreate(&c->ev[i]) == hipSuccess;
    ok = ok && hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking) == hipSuccess;
    if (!ok) { set_err("fj_ctx_create: allocating context scratch failed: %s", hipGetErrorString(hipGetLastError())); delete c; return nullptr; }
    return c;
}

void fj_ctx_destroy(fj_ctx* c) {
    if (!c) return;
    DeviceGuard guard(c->device);
    for (auto& b : c->bufs) if (b.p) (void)hipFree(b.p);
    for (int i = 0; i < E_NEV; ++i) (void)hipEventDestroy(c->ev[i]);
    if (c->side) (void)hipStreamDestroy(c->side);
    for (void* p : c->stage) if (p) (void)hipHostFree(p);
    if (c->d_sc) (void)hipFree(c->d_sc);
    if (c->h_sc) (void)hipHostFree(c->h_sc);
    delete c;
}

size_t fj_ctx_workspace_bytes(const fj_ctx* c) { return c ? c->ws_bytes : 0; }

// Give the cached workspace (chunk pools, directories, tables: tens of GB after a 1B-row join) back to the device; the
// context stays usable and grows again on demand.  A pending emit (fj_emit_pairs not yet called) is dropped.
int fj_ctx_trim(fj_ctx* c) {
    if (!c) c = g_host_ctx;                       // NULL: the context behind fj_join_host (nothing to do before its first call)
    if (!c) return 0;
    if (c->st.active) return set_err("fj_ctx_trim: a stream join is open on this context (fj_stream_finish it first)");
    FJ_ENTER(c);
    HIPCHK(hipDeviceSynchronize());               // kernels of earlier joins may still read the buffers
    c->pend.valid = false;
    for (auto& b : c->bufs) if (b.p) { HIPCHK(hipFree(b.p)); b.p = nullptr; b.bytes = 0; }
    c->ws_bytes = 0;
    return 0;
}

int fj_join_device(fj_ctx* c, int algo, int bloom, int materialize,
                   const uint64_t* d_bk, const uint64_t* d_bv, size_t nb, const uint64_t* d_pk, size_t np,
                   void* stream, int hash_top_bits, uint64_t* out_count,
                   uint64_t* d_out_keys, uint64_t* d_out_vals, size_t out_capacity, fj_timings* timings) {
    if (!c) return set_err("fj_join_device: null context");
    const bool many = algo >= 0 && (algo & FJ_ALGO_MANY_TO_MANY) != 0;
    if (many) algo &= ~FJ_ALGO_MANY_TO_MANY;
    if (algo < 0 || algo > 2) return set_err("fj_join_device: unknown algo %d", algo);
    if (hash_top_bits != 64 && hash_top_bits != 48) return set_err("fj_join_device: hash_top_bits must be 64 or 48");
    if (c->st.active) return set_err("fj_join_device: a stream join is open on this context (fj_stream_finish it first)");
    if ((nb && (!d_bk || !d_bv)) || (np && !d_pk)) return set_err("fj_join_device: null input pointer");
    if (((uintptr_t)d_bk | (uintptr_t)d_bv | (uintptr_t)d_pk) & 15) return set_err("fj_join_device: input pointers must be 16-byte aligned");
    FJ_ENTER(c);
    hipStream_t s = (hipStream_t)stream;
    fj_timings t; memset(&t, 0, sizeof t);
    t.sampled_hit_bp = -1;
    u64 count = 0;
    c->pend.valid = false;
    const Options& opt = options();
    bool use_radix = algo == FJ_ALGO_RADIX || (algo == FJ_ALGO_ADAPTIVE && nb >= opt.radix_threshold) ||
                     (algo == FJ_ALGO_SCALAR && !opt.scalar_hbm_table);
    if (nb == 0 || np == 0) {                   // empty side: (0, t), hash_join.cpp behaviour for empty inputs
        count = 0;
    } else if (many) {
        if (join_many(c, materialize, d_bk, d_bv, nb, d_pk, np, hash_top_bits, s, &t, &count)) return 1;
    } else if (use_radix) {
        bool lds_full = false;
        // adaptive_*: the precheck is decided from a sample of the probe side; *_bloom by name: always on; otherwise off
        const int bloom_mode = algo == FJ_ALGO_ADAPTIVE ? (options().bloom_auto ? 2 : (bloom ? 1 : 0)) : (bloom ? 1 : 0);
        SingleOut so;
        so.keys = (u64*)d_out_keys; so.vals = (u64*)d_out_vals; so.cap = out_capacity;
        const bool try_single = materialize && d_out_keys && d_out_vals && out_capacity >= np && options().mat_single_pass &&
                                !(((uintptr_t)d_out_keys | (uintptr_t)d_out_vals) & 7);
        if (join_radix(c, materialize, bloom_mode, d_bk, d_bv, nb, d_pk, np, hash_top_bits, s, &t, &count, &lds_full, try_single ? &so : nullptr)) return 1;
        if (lds_full) {
            fj_timings t2; memset(&t2, 0, sizeof t2);
            if (join_global(c, bloom, materialize, d_bk, d_bv, nb, d_pk, np, s, &t2, &count)) return 1;
            t2.total_ms += t.total_ms; t2.fell_back = 1; t2.sampled_hit_bp = t.sampled_hit_bp; t = t2;
        }
    } else {
        if (join_global(c, bloom, materialize, d_bk, d_bv, nb, d_pk, np, s, &t, &count)) return 1;
    }
    if (out_count) *out_count = count;
    if (materialize && d_out_keys && d_out_vals && c->pend.valid) {
        if (emit_pending(c, d_out_keys, d_out_vals, out_capacity, s, &t)) return 1;
    }
    if (timings) *timings = t;
    g_last = t;
    return 0;
}

int fj_emit_pairs(fj_ctx* c, uint64_t* d_out_keys, uint64_t* d_out_vals, size_t out_capacity, void* stream, fj_timings* timings) {
    if (!c) return set_err("fj_emit_pairs: null context");
    FJ_ENTER(c);
    fj_timings t = g_last;
    if (emit_pending(c, d_out_keys, d_out_vals, out_capacity, (hipStream_t)stream, &t)) return 1;
    if (timings) *timings = t;
    g_last = t;
    return 0;
}

int fj_owner_split(fj_ctx* c, const uint64_t* d_keys, const uint64_t* d_vals, size_t n, int nranks,
                   uint64_t* d_out_keys, uint64_t* d_out_vals, uint64_t* h_counts, void* stream) {
    if (!c) return set_err("fj_owner_split: null context");
    if (nranks < 1 || nranks > 64) return set_err("fj_owner_split: nranks must be 1..64");
    FJ_ENTER(c);
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(hipMemsetAsync(c->d_sc->owner_counts, 0, sizeof(unsigned long long) * 128, s));   // counts + cursors
    HIPCHK(fj_launch_owner_hist(d_keys, n, (u32)nranks, c->d_sc->owner_counts, s));
    HIPCHK(hipMemcpyAsync(c->h_sc->owner_counts, c->d_sc->owner_counts, sizeof(unsigned long long) * 64, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    unsigned long long run = 0;
    for (int r = 0; r < nranks; ++r) { c->h_sc->owner_offsets[r] = run; run += c->h_sc->owner_counts[r]; h_counts[r] = c->h_sc->owner_counts[r]; }
    if (run != n) return set_err("fj_owner_split: histogram covers %llu of %zu rows", run, n);
    HIPCHK(hipMemcpyAsync(c->d_sc->owner_offsets, c->h_sc->owner_offsets, sizeof(unsigned long long) * 64, hipMemcpyHostToDevice, s));
    HIPCHK(fj_launch_owner_scatter(d_keys, d_vals, n, (u32)nranks, c->d_sc->owner_offsets, c->d_sc->owner_cursors, d_out_keys, d_out_vals, s));
    HIPCHK(hipStreamSynchronize(s));
    return 0;
}

// ---- a counting join whose relations arrive in pieces (multi-GPU: pieces of an exchange) -----------------
// open: plan + first-pass pools of both sides; append_*: one first-pass launch per piece (launches accumulate
// into the same chunk pool); advance_probe: the probe side's remaining passes (so that they can overlap an
// exchange of the build side); finish: whatever remains + join.
namespace {

int stream_flat_join(fj_ctx* c, StreamState& st, const u64* d_pk, size_t n, hipStream_t s) {
    FjLdsJoinArgs ja = st.ja;
    ja.probe = FjChunkSet(); ja.probe.keys = const_cast<u64*>(d_pk); ja.probe.n_flat = n; ja.probe.nb = 1;
    ja.nparts = 1;
    ja.nsplit = (u32)std::min<u64>(2048, std::max<u64>(1, ((n + FJ_CHUNK - 1) / FJ_CHUNK) / 32));
    void* p;
    if (get_buf(c, W_PART_COUNT, (size_t)ja.nsplit * 4, &p)) return 1; ja.part_count = (u32*)p;
    ja.total = &c->d_sc->total; ja.err = &c->d_sc->err; ja.dbg = nullptr; ja.dbg_flags = 0;
    ja.want_dups = 0; ja.dedup = 0; ja.orig_vals = nullptr; ja.retry_only = 0;
    HIPCHK(fj_launch_lds_join(ja, false, s));
    return 0;
}

// counting join of the appended pieces over one table in HBM (fj_gt_*): the fallback of a streamed join
int stream_global_count(fj_ctx* c, StreamState& st, hipStream_t s, fj_timings* t, u64* out_count) {
    u64 cap = 64;
    while (cap < 2 * (u64)st.nb_seen) cap <<= 1;
    FjGtArgs a{};
    void* p;
    if (get_buf(c, W_GT_KEYS, cap * 8, &p)) return 1; a.tkeys = (u64*)p;
    if (get_buf(c, W_GT_VALS, cap * 8, &p)) return 1; a.tvals = (u64*)p;
    a.bloom = nullptr;
    if (get_buf(c, W_WG_COUNT, (size_t)2048 * 4, &p)) return 1; a.wg_count = (u32*)p;
    a.cap_mask = cap - 1; a.flags = &c->d_sc->flags; a.empty_val = &c->d_sc->empty_val; a.total = &c->d_sc->total;
    HIPCHK(hipEventRecord(c->ev[E_START], s));
    HIPCHK(hipMemsetAsync(c->d_sc, 0, offsetof(Scalars, alloc), s));
    HIPCHK(hipMemsetAsync(a.tkeys, 0xFF, cap * 8, s));
    for (const auto& bp : st.bpieces) {
        a.bk = bp.first; a.bv = bp.first; a.nb = bp.second;           // a counting join never reads the values
        HIPCHK(fj_launch_gt_build(a, s));
    }
    HIPCHK(hipEventRecord(c->ev[E_BUILD], s));
    for (const auto& pp : st.ppieces) {
        a.pk = pp.first; a.np = pp.second;
        const u32 grid = (u32)std::min<u64>(2048, std::max<u64>(1, ((pp.second + 1) / 2) / 256));
        if (pp.second) HIPCHK(fj_launch_gt_probe(a, false, grid, s));
    }
    HIPCHK(hipEventRecord(c->ev[E_JOIN], s));
    if (read_scalars(c, s)) return 1;
    *out_count = c->h_sc->total;
    t->path = 1; t->passes = 0; t->radix_bits = 0; t->partitions = 1;
    t->build_phase_ms = ev_ms(c, E_START, E_BUILD);
    t->join_ms = ev_ms(c, E_BUILD, E_JOIN);
    t->probe_phase_ms = t->join_ms;
    t->total_ms = ev_ms(c, E_START, E_JOIN);
    return 0;
}

// the build side is complete: run its remaining passes (or fix the flat table input of a zero-pass plan)
int stream_flush_build(fj_ctx* c, StreamState& st, hipStream_t s) {
    if (st.build_done) return 0;
    st.build_done = true;
    HIPCHK(hipEventRecord(c->ev[E_SB0], s));
    if (st.plan.npass > 0) {
        if (st.nb_seen > 0) {
            if (pass_complete(c, st.bit, s)) return 1;
            if (run_passes(c, st.bit, nullptr, nullptr, s, &st.ja.build, nullptr)) return 1;
        } else { st.ja.build = FjChunkSet(); st.ja.build.n_flat = 0; }
    } else {
        st.ja.build = FjChunkSet();
        st.ja.build.keys = const_cast<u64*>(st.flat_build); st.ja.build.n_flat = st.nb_seen; st.ja.build.list = nullptr; st.ja.build.nb = 1;
    }
    HIPCHK(hipEventRecord(c->ev[E_SB1], s));
    HIPCHK(hipEventRecord(c->ev[E_BUILD], s));
    return 0;
}

int stream_open(fj_ctx* c, size_t nb_bound, int build_appends, size_t np_bound, int probe_appends, hipStream_t s, int top_bits,
                size_t probe_piece_rows = 0) {
    StreamState& st = c->st;
    st = StreamState();
    c->pend.valid = false;
    st.plan = make_plan(nb_bound, top_bits);
    st.top_bits = top_bits; st.np_bound = np_bound; st.nb_bound = nb_bound;
    st.p_appends_left = (u32)probe_appends; st.b_appends_left = (u32)build_appends;
    begin_plan(c);
    HIPCHK(hipEventRecord(c->ev[E_START], s));
    if (clear_plan_scalars(c, s)) return 1;
    if (st.plan.npass > 0) {
        pass_init(st.bit, 0, false, std::max<size_t>(nb_bound, 1), st.plan, top_bits);     // count only: keys
        if (pass_prepare(c, st.bit, (u32)build_appends, s)) return 1;
        pass_init(st.pit, 1, false, std::max<size_t>(np_bound, 1), st.plan, top_bits);
        st.pit.want_items = true;
        st.pit.piece_rows = probe_piece_rows;
        if (pass_prepare(c, st.pit, (u32)probe_appends, s)) return 1;
    }
    st.active = true;
    return 0;
}

int stream_append_build(fj_ctx* c, const u64* d_bk, size_t n, hipStream_t s) {
    StreamState& st = c->st;
    if (n == 0) return 0;
    if (st.build_done) return set_err("fj_stream_append_build: the build side is already closed");
    if (!d_bk || ((uintptr_t)d_bk & 15)) return set_err("fj_stream_append_build: build piece must be a 16-byte aligned device pointer");
    if (st.b_appends_left == 0) return set_err("fj_stream_append_build: more pieces than build_appends");
    if (st.nb_seen + n > st.nb_bound) return set_err("fj_stream_append_build: more build rows than nb_bound");
    --st.b_appends_left; st.nb_seen += n;
    st.bpieces.emplace_back(d_bk, n);
    if (st.plan.npass > 0) return pass_launch(c, st.bit, d_bk, nullptr, n, s, nullptr);
    if (st.flat_build) return set_err("fj_stream_append_build: a build side of <= %d rows must arrive in one piece", (int)FJ_PART_TARGET_KEYS);
    st.flat_build = d_bk;
    return 0;
}

}  // namespace

int fj_stream_open(fj_ctx* c, size_t nb_bound, int build_appends, size_t np_bound, int probe_appends, void* stream, int hash_top_bits) {
    if (!c) return set_err("fj_stream_open: null context");
    if (hash_top_bits != 64 && hash_top_bits != 48) return set_err("fj_stream_open: hash_top_bits must be 64 or 48");
    if (build_appends < 1 || build_appends > 64 || probe_appends < 1 || probe_appends > 64)
        return set_err("fj_stream_open: build_appends and probe_appends must be 1..64");
    FJ_ENTER(c);
    return stream_open(c, nb_bound, build_appends, np_bound, probe_appends, (hipStream_t)stream, hash_top_bits);
}

int fj_stream_append_build(fj_ctx* c, const uint64_t* d_bk, size_t n, void* stream) {
    if (!c || !c->st.active) return set_err("fj_stream_append_build: no stream join is open on this context");
    FJ_ENTER(c);
    return stream_append_build(c, (const u64*)d_bk, n, (hipStream_t)stream);
}

int fj_stream_begin(fj_ctx* c, const uint64_t* d_bk, const uint64_t* d_bv, size_t nb, size_t np_bound, int max_appends,
                    void* stream, int hash_top_bits) {
    if (!c) return set_err("fj_stream_begin: null context");
    if (hash_top_bits != 64 && hash_top_bits != 48) return set_err("fj_stream_begin: hash_top_bits must be 64 or 48");
    if (max_appends < 1 || max_appends > 64) return set_err("fj_stream_begin: max_appends must be 1..64");
    if (nb && (!d_bk || !d_bv)) return set_err("fj_stream_begin: null input pointer");
    if (((uintptr_t)d_bk | (uintptr_t)d_bv) & 15) return set_err("fj_stream_begin: input pointers must be 16-byte aligned");
    FJ_ENTER(c);
    hipStream_t s = (hipStream_t)stream;
    if (stream_open(c, nb, 1, np_bound, max_appends, s, hash_top_bits)) return 1;
    if (stream_append_build(c, (const u64*)d_bk, nb, s)) return 1;
    return stream_flush_build(c, c->st, s);
}

int fj_stream_append_probe(fj_ctx* c, const uint64_t* d_pk, size_t n, void* stream) {
    if (!c || !c->st.active) return set_err("fj_stream_append_probe: no stream join is open on this context");
    StreamState& st = c->st;
    if (n == 0) return 0;
    if (st.probe_done) return set_err("fj_stream_append_probe: the probe side is already closed");
    if (!d_pk || ((uintptr_t)d_pk & 15)) return set_err("fj_stream_append_probe: probe piece must be a 16-byte aligned device pointer");
    if (st.p_appends_left == 0) return set_err("fj_stream_append_probe: more pieces than probe_appends");
    if (st.np_seen + n > st.np_bound) return set_err("fj_stream_append_probe: more probe rows than np_bound");
    FJ_ENTER(c);
    hipStream_t s = (hipStream_t)stream;
    --st.p_appends_left; st.np_seen += n;
    st.ppieces.emplace_back((const u64*)d_pk, n);
    if (st.plan.npass > 0) return pass_launch(c, st.pit, d_pk, nullptr, n, s, st.evc < 4 ? &st.evc : nullptr);
    // zero-pass plan (tiny build side): join this piece right away when the build side is known, else at finish
    if (st.build_done) return st.ja.build.n_flat == 0 ? 0 : stream_flat_join(c, st, d_pk, n, s);
    st.flat_probe[st.nflat] = d_pk; st.flat_np[st.nflat] = n; ++st.nflat;
    return 0;
}

int fj_stream_advance_probe(fj_ctx* c, void* stream) {
    if (!c || !c->st.active) return set_err("fj_stream_advance_probe: no stream join is open on this context");
    StreamState& st = c->st;
    if (st.probe_done) return 0;
    FJ_ENTER(c);
    hipStream_t s = (hipStream_t)stream;
    st.probe_done = true;
    if (st.plan.npass > 0 && st.np_seen > 0) {
        if (pass_complete(c, st.pit, s)) return 1;
        if (run_passes(c, st.pit, nullptr, nullptr, s, &st.ja.probe, nullptr)) return 1;
    }
    return 0;
}

// Drop an open stream join without a result (an error on the caller's side between two appends): the context is free for
// other joins again; the buffers the abandoned passes left half-filled are re-zeroed by the next plan (plan_in_flight).
int fj_stream_abort(fj_ctx* c) {
    if (!c) return set_err("fj_stream_abort: null context");
    FJ_ENTER(c);
    if (!c->st.active) return 0;
    HIPCHK(hipDeviceSynchronize());               // launched passes still read the caller's pieces
    c->st.active = false;
    return 0;
}

int fj_stream_finish(fj_ctx* c, void* stream, uint64_t* out_count, fj_timings* timings) {
    if (!c || !c->st.active) return set_err("fj_stream_finish: no stream join is open on this context");
    StreamState& st = c->st;
    st.active = false;
    FJ_ENTER(c);
    hipStream_t s = (hipStream_t)stream;
    fj_timings t; memset(&t, 0, sizeof t);
    u64 count = 0;
    if (stream_flush_build(c, st, s)) return 1;
    if (st.plan.npass > 0 && st.nb_seen > 0 && st.np_seen > 0) {
        if (!st.probe_done) {
            st.probe_done = true;
            if (pass_complete(c, st.pit, s)) return 1;
            if (run_passes(c, st.pit, nullptr, nullptr, s, &st.ja.probe, nullptr)) return 1;
        }
        HIPCHK(hipEventRecord(c->ev[E_PPART], s));
        bool lds_full = false;
        if (radix_join_tail(c, 0, st.ja, st.plan, st.np_seen, st.pit, s, &t, st.evc, &count, &lds_full, st.top_bits)) return 1;
        if (lds_full && st.shuffled)
            return set_err("shuffled stream join: a final partition holds more than 8128 distinct build keys (skewed build side); no fallback for chunk pieces");
        if (lds_full) {
            // a partition of more than 8128 distinct build keys: count over ONE table in HBM, piece by piece (the streamed
            // join's own fallback; the one-shot join has the same one)
            fj_timings t2; memset(&t2, 0, sizeof t2); t2.sampled_hit_bp = -1;
            if (stream_global_count(c, st, s, &t2, &count)) return 1;
            t2.total_ms += t.total_ms; t2.fell_back = 1; t = t2;
        }
    } else {
        if (st.plan.npass == 0 && st.nb_seen > 0)
            for (u32 i = 0; i < st.nflat; ++i) if (stream_flat_join(c, st, st.flat_probe[i], st.flat_np[i], s)) return 1;
        HIPCHK(hipEventRecord(c->ev[E_PPART], s));
        HIPCHK(hipEventRecord(c->ev[E_JOIN], s));
        if (read_scalars(c, s)) return 1;
        if (st.plan.npass == 0) end_plan(c);          // (a partitioned plan with an empty side never ran its bookkeeping: stays "in flight")
        if (c->h_sc->err & (FJ_ERR_LDS_FULL | FJ_STAT_RETRY)) {
            // the one table of a zero-pass plan overflowed (a build side of ~3900 rows at a bad moment): HBM-table fallback
            fj_timings t2; memset(&t2, 0, sizeof t2); t2.sampled_hit_bp = -1;
            if (stream_global_count(c, st, s, &t2, &count)) return 1;
            t2.fell_back = 1;
            if (out_count) *out_count = count;
            if (timings) *timings = t2;
            g_last = t2;
            return 0;
        }
        count = c->h_sc->total;
        t.path = 0; t.passes = 0; t.partitions = 1;
        t.total_ms = ev_ms(c, E_START, E_JOIN);
    }
    if (!t.fell_back) {
        t.build_phase_ms = ev_ms(c, E_SB0, E_SB1);              // the two sides may have run in either order
        t.probe_phase_ms = t.total_ms - t.build_phase_ms;
    }
    if (out_count) *out_count = count;
    if (timings) *timings = t;
    g_last = t;
    return 0;
}

// ---- owner shuffle in the shape of SURVEY 8(e): the first radix pass of the GLOBAL plan is the owner split ------------------
// Every rank plans for the TOTAL build side (all ranks' rows): pass 1 of that plan has F0 = 256 or 512 buckets, bucket b
// belongs to owner GPU (b * nranks) >> log2(F0).  A sender runs that pass over its local rows in the owner-grouped form of
// the partition kernel (fj_shuffle_pack): the chunks of owner r's buckets land in region r of one pool, so what goes to a peer
// is one contiguous piece (whole 2-KiB chunks + one directory word per chunk; ~3 % of partial chunks and abandoned slab
// ids travel along).  The owner appends what it received as level-1 chunk sets (fj_stream_append_*_chunks: directory words ->
// chunk lists, then the plan's SECOND pass over them) and finishes like any stream join.  No separate owner histogram, no
// owner scatter, no first pass at the receiver: three passes over every probe row per rank become two.
namespace {
int shuffle_plan(size_t nb_total, int nranks, Plan* out) {
    if (nranks < 1 || nranks > 64) return set_err("owner shuffle: nranks must be 1..64");
    const Plan p = make_plan(nb_total, 64);
    if (p.npass < 2) return set_err("owner shuffle: a build side of %zu rows in all has a %d-pass plan (the chunk form needs two or more: use fj_owner_split)", nb_total, p.npass);
    if ((1 << p.fan_log[0]) < nranks) return set_err("owner shuffle: %d ranks but only %d first-pass buckets", nranks, 1 << p.fan_log[0]);
    *out = p;
    return 0;
}
// workgroups of the packing pass: every (workgroup, owner) pair abandons half a slab of chunk ids on average and every
// (workgroup, bucket) pair ends in a partial chunk - all of which travels - so a small piece gets few workgroups (<= 1/8 of
// its chunks lost to abandoned ids; config 5's 312M-row pieces still get one workgroup per CU)
u32 shuffle_groups(size_t n, u32 tile_chunks, u32 F, u32 nranks, u32 slab) {
    const u64 chunks = (n + FJ_CHUNK - 1) / FJ_CHUNK;
    const u64 by_slabs = std::max<u64>(1, chunks / ((u64)nranks * 4 * slab));
    return (u32)std::min<u64>(std::min<u64>(256u, by_slabs), pass_groups(chunks, n, tile_chunks, F));
}
}  // namespace

int fj_shuffle_plan(size_t nb_total, int nranks, int* fan_log0, int* npass) {
    Plan p;
    if (shuffle_plan(nb_total, nranks, &p)) return 1;
    if (fan_log0) *fan_log0 = p.fan_log[0];
    if (npass) *npass = p.npass;
    return 0;
}

size_t fj_shuffle_region_chunks(size_t n, size_t nb_total, int nranks, int with_vals) {
    Plan p;
    if (shuffle_plan(nb_total, nranks, &p)) return 0;
    const u32 F = 1u << p.fan_log[0], tc = fj_partition_tile_chunks((u32)p.fan_log[0], with_vals != 0);
    const u64 FO = (F + nranks - 1) / nranks + 1, slab = fj_own_slab((u32)p.fan_log[0], with_vals != 0, (u32)nranks);
    const u64 G = shuffle_groups(n, tc, F, (u32)nranks, (u32)slab);
    // an even share of the rows + 25 % for the hash's imbalance, the partial chunk of every (workgroup, bucket) pair at either
    // end of a segment, and one abandoned slab remainder per (workgroup, owner)
    const u64 share = (n / FJ_CHUNK + nranks - 1) / nranks;
    return (size_t)(share + share / 4 + 2 * G * FO + (G + 1) * slab + 64);
}

int fj_shuffle_pack(fj_ctx* c, const uint64_t* d_keys, const uint64_t* d_vals, size_t n, size_t nb_total, int nranks,
                    uint64_t* d_out_keys, uint64_t* d_out_vals, uint32_t* d_out_dir, size_t region_chunks, uint64_t* h_used, void* stream) {
    if (!c) return set_err("fj_shuffle_pack: null context");
    if (!h_used || (n && (!d_keys || !d_out_keys || !d_out_dir)) || (d_vals && !d_out_vals)) return set_err("fj_shuffle_pack: null pointer");
    if (((uintptr_t)d_keys | (uintptr_t)d_vals | (uintptr_t)d_out_keys | (uintptr_t)d_out_vals) & 15) return set_err("fj_shuffle_pack: pointers must be 16-byte aligned");
    Plan plan;
    if (shuffle_plan(nb_total, nranks, &plan)) return 1;
    for (int r = 0; r < nranks; ++r) h_used[r] = 0;
    if (n == 0) return 0;
    if (region_chunks * (size_t)nranks >= (1ull << 24)) return set_err("fj_shuffle_pack: %d regions of %zu chunks exceed one chunk directory (send the relation in pieces)", nranks, region_chunks);
    FJ_ENTER(c);
    hipStream_t s = (hipStream_t)stream;
    const bool vals = d_vals != nullptr;
    const u32 fan_log = (u32)plan.fan_log[0], F = 1u << fan_log, tc = fj_partition_tile_chunks(fan_log, vals);
    const u32 slab = fj_own_slab(fan_log, vals, (u32)nranks), G = shuffle_groups(n, tc, F, (u32)nranks, slab);
    if (region_chunks < (size_t)2 * slab) return set_err("fj_shuffle_pack: region of %zu chunks is too small", region_chunks);
    const u32 cap = (u32)(region_chunks * nranks), max_segs = G + 3;
    void* p;
    FjPartArgs a{};
    if (get_buf(c, W_SH_REL, (size_t)cap * 8, &p)) return 1; a.out_rel = (u64*)p;
    if (get_buf(c, W_SH_SEGOFF, (size_t)max_segs * F * 4, &p)) return 1; a.seg_off = (u32*)p;
    if (get_buf(c, W_SH_BCH, (size_t)F * 4, &p)) return 1; a.bchunks = (u32*)p;
    a.in_keys = (const u64*)d_keys; a.in_vals = (const u64*)d_vals; a.n_flat = n; a.parent0 = 0;
    a.out_keys = (u64*)d_out_keys; a.out_vals = (u64*)d_out_vals; a.out_dir = d_out_dir;
    a.alloc = &c->d_sc->pack_alloc_unused; a.seg_counter = &c->d_sc->pack_seg; a.cap_chunks = cap; a.max_segs = max_segs; a.err = &c->d_sc->pack_err;
    a.shift = 64u - fan_log; a.fan_log = fan_log; a.side = vals ? 0u : 1u; a.slab = slab;
    a.own_nranks = (u32)nranks; a.own_region = (u32)region_chunks; a.own_alloc = c->d_sc->own_alloc;
    HIPCHK(hipMemsetAsync(c->d_sc->own_alloc, 0, sizeof(u32) * 67, s));          // allocators + error word + segment counter + unused allocator
    const int line_log = (vals && F > 256) ? 3 : 4;
    HIPCHK(fj_launch_partition(a, vals, line_log, G, s));
    HIPCHK(hipMemcpyAsync(c->h_sc->own_alloc, c->d_sc->own_alloc, sizeof(u32) * 67, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (c->h_sc->pack_err & FJ_ERR_POOL)
        return set_err("fj_shuffle_pack: an owner's region of %zu chunks overflowed (skewed keys: more than 1.25x an even share go to one GPU)", region_chunks);
    for (int r = 0; r < nranks; ++r) h_used[r] = std::min<u64>(c->h_sc->own_alloc[r], region_chunks);
    return 0;
}

int fj_stream_open_shuffled(fj_ctx* c, size_t nb_total, int nranks, int rank, size_t nb_bound, int build_appends, size_t np_bound, int probe_appends,
                            void* stream) {
    if (!c) return set_err("fj_stream_open_shuffled: null context");
    if (rank < 0 || rank >= nranks) return set_err("fj_stream_open_shuffled: rank %d of %d", rank, nranks);
    if (build_appends < 1 || build_appends > 64 || probe_appends < 1 || probe_appends > 64) return set_err("fj_stream_open_shuffled: build_appends and probe_appends must be 1..64");
    Plan plan;
    if (shuffle_plan(nb_total, nranks, &plan)) return 1;
    FJ_ENTER(c);
    hipStream_t s = (hipStream_t)stream;
    StreamState& st = c->st;
    st = StreamState();
    c->pend.valid = false;
    st.plan = plan; st.top_bits = 64; st.shuffled = true;
    const u32 F0 = 1u << plan.fan_log[0];
    st.b_lo = (u32)(((u64)rank * F0 + nranks - 1) / nranks);                     // first bucket b with (b * nranks) >> log2(F0) == rank
    st.nbk = (u32)(((u64)(rank + 1) * F0 + nranks - 1) / nranks) - st.b_lo;
    st.nbk_pad = (st.nbk + 3u) & ~3u;                                            // (fj_level_scan works in 16-B pieces)
    st.np_bound = np_bound; st.nb_bound = nb_bound;
    st.p_appends_left = (u32)probe_appends; st.b_appends_left = (u32)build_appends;
    begin_plan(c);
    HIPCHK(hipEventRecord(c->ev[E_START], s));
    if (clear_plan_scalars(c, s)) return 1;
    auto init = [&](PassIter& it, int side, size_t n, u32 appends) -> int {
        pass_init(it, side, false, std::max<size_t>(n, 1), plan, 64);
        it.i = 1; it.used = 64 - plan.fan_log[0]; it.parents = st.nbk_pad; it.slot = 1;     // pass 1 of the plan ran at the senders
        it.lbound = it.n / FJ_CHUNK + 1 + (u64)appends * (2ull * 256 * F0 + 4096);          // their partial chunks and abandoned ids arrive too
        return pass_prepare(c, it, appends, s);
    };
    if (init(st.bit, 0, nb_bound, (u32)build_appends)) return 1;
    if (init(st.pit, 1, np_bound, (u32)probe_appends)) return 1;
    st.pit.want_items = true;
    st.active = true;
    return 0;
}

namespace {
// one received piece (whole chunks + their directory words) -> chunk lists + tile table -> the plan's second pass over it
int stream_append_chunks(fj_ctx* c, int side, const u64* d_chunks, u32* d_dir, size_t nchunks, hipStream_t s) {
    StreamState& st = c->st;
    PassIter& it = side ? st.pit : st.bit;
    if (nchunks >= (1ull << 24)) return set_err("fj_stream_append_*_chunks: a piece of %zu chunks exceeds one chunk directory", nchunks);
    const u32 n = (u32)nchunks, nblocks = (n + 4095u) / 4096u;
    u32 fan = 4; while (fan < st.nbk_pad) fan <<= 1;
    const u32 tc = fj_partition_tile_chunks((u32)st.plan.fan_log[1], false);
    const u64 max_tiles = n / tc + st.nbk_pad + 1;
    FjChunkSet cs{};
    void* p;
    cs.keys = const_cast<u64*>(d_chunks); cs.vals = nullptr; cs.dir = d_dir; cs.cap = n; cs.nb = st.nbk_pad; cs.fan_mask = fan - 1; cs.max_segs = nblocks;
    if (get_buf(c, W_RX_REL, (size_t)n * 8, &p)) return 1; cs.rel = (u64*)p;
    if (get_buf(c, W_RX_LIST, (size_t)n * 4, &p)) return 1; cs.list = (u32*)p;
    if (get_buf(c, W_RX_SEGOFF, (size_t)nblocks * fan * 4, &p)) return 1; cs.seg_off = (u32*)p;
    if (get_zeroed_buf(c, W_RX_BCH, (size_t)fan * 4, &p, s)) return 1; cs.bchunks = (u32*)p;
    if (get_buf(c, W_RX_BOFF, ((size_t)st.nbk_pad + 1) * 4, &p)) return 1; cs.boff = (u32*)p;
    if (get_buf(c, W_RX_TOFF, ((size_t)st.nbk_pad + 1) * 4, &p)) return 1; u32* toff = (u32*)p;
    if (get_buf(c, W_RX_TILES, (size_t)max_tiles * sizeof(uint4), &p)) return 1; uint4* tiles = (uint4*)p;
    cs.alloc = &c->d_sc->rx_alloc;
    HIPCHK(fj_launch_dir_rank(d_dir, n, st.b_lo, st.nbk, fan, cs.rel, cs.seg_off, cs.bchunks, cs.alloc, s));
    HIPCHK(fj_launch_group(cs, tc, toff, tiles, (u32)max_tiles, nullptr, s));
    it.prev = cs; it.have_prev = true; it.tiles = tiles; it.ntiles = toff + st.nbk_pad; it.toff = toff;
    return pass_launch(c, it, nullptr, nullptr, 0, s, side && st.evc < 4 ? &st.evc : nullptr);
}
}  // namespace

int fj_stream_append_build_chunks(fj_ctx* c, const uint64_t* d_chunks, uint32_t* d_dir, size_t nchunks, void* stream) {
    if (!c || !c->st.active || !c->st.shuffled) return set_err("fj_stream_append_build_chunks: no shuffled stream join is open on this context");
    StreamState& st = c->st;
    if (st.build_done) return set_err("fj_stream_append_build_chunks: the build side is already closed");
    if (st.b_appends_left == 0) return set_err("fj_stream_append_build_chunks: more pieces than build_appends");
    if (nchunks && (!d_chunks || !d_dir || ((uintptr_t)d_chunks & 15))) return set_err("fj_stream_append_build_chunks: null or misaligned piece");
    FJ_ENTER(c);
    hipStream_t s = (hipStream_t)stream;
    --st.b_appends_left;
    if (nchunks) { st.nb_seen += nchunks * FJ_CHUNK; if (stream_append_chunks(c, 0, (const u64*)d_chunks, d_dir, nchunks, s)) return 1; }
    if (st.b_appends_left == 0) return stream_flush_build(c, st, s);          // the build side is complete: its remaining passes run now
    return 0;
}

int fj_stream_append_probe_chunks(fj_ctx* c, const uint64_t* d_chunks, uint32_t* d_dir, size_t nchunks, void* stream) {
    if (!c || !c->st.active || !c->st.shuffled) return set_err("fj_stream_append_probe_chunks: no shuffled stream join is open on this context");
    StreamState& st = c->st;
    if (st.probe_done) return set_err("fj_stream_append_probe_chunks: the probe side is already closed");
    if (st.p_appends_left == 0) return set_err("fj_stream_append_probe_chunks: more pieces than probe_appends");
    if (nchunks && (!d_chunks || !d_dir || ((uintptr_t)d_chunks & 15))) return set_err("fj_stream_append_probe_chunks: null or misaligned piece");
    FJ_ENTER(c);
    --st.p_appends_left;
    if (nchunks == 0) return 0;
    st.np_seen += nchunks * FJ_CHUNK;
    return stream_append_chunks(c, 1, (const u64*)d_chunks, d_dir, nchunks, (hipStream_t)stream);
}

// split fj_owner_split into its two halves so that a caller can size its exchange before scattering
int fj_owner_hist(fj_ctx* c, const uint64_t* d_keys, size_t n, int nranks, uint64_t* h_counts, void* stream) {
    if (!c) return set_err("fj_owner_hist: null context");
    if (nranks < 1 || nranks > 64) return set_err("fj_owner_hist: nranks must be 1..64");
    FJ_ENTER(c);
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(hipMemsetAsync(c->d_sc->owner_counts, 0, sizeof(unsigned long long) * 64, s));
    HIPCHK(fj_launch_owner_hist(d_keys, n, (u32)nranks, c->d_sc->owner_counts, s));
    HIPCHK(hipMemcpyAsync(c->h_sc->owner_counts, c->d_sc->owner_counts, sizeof(unsigned long long) * 64, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    unsigned long long run = 0;
    for (int r = 0; r < nranks; ++r) { h_counts[r] = c->h_sc->owner_counts[r]; run += h_counts[r]; }
    if (run != n) return set_err("fj_owner_hist: histogram covers %llu of %zu rows", run, n);
    return 0;
}

int fj_owner_scatter(fj_ctx* c, const uint64_t* d_keys, const uint64_t* d_vals, size_t n, int nranks, const uint64_t* h_counts,
                     uint64_t* d_out_keys, uint64_t* d_out_vals, void* stream) {
    if (!c) return set_err("fj_owner_scatter: null context");
    if (nranks < 1 || nranks > 64) return set_err("fj_owner_scatter: nranks must be 1..64");
    FJ_ENTER(c);
    hipStream_t s = (hipStream_t)stream;
    // the offsets travel in a pinned slot that a previous asynchronous scatter may still be reading: drain first
    HIPCHK(hipStreamSynchronize(s));
    unsigned long long run = 0;
    for (int r = 0; r < nranks; ++r) { c->h_sc->owner_offsets[r] = run; run += h_counts[r]; }
    if (run != n) return set_err("fj_owner_scatter: counts cover %llu of %zu rows", run, n);
    HIPCHK(hipMemcpyAsync(c->d_sc->owner_offsets, c->h_sc->owner_offsets, sizeof(unsigned long long) * 64, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemsetAsync(c->d_sc->owner_cursors, 0, sizeof(unsigned long long) * 64, s));
    HIPCHK(fj_launch_owner_scatter(d_keys, d_vals, n, (u32)nranks, c->d_sc->owner_offsets, c->d_sc->owner_cursors, d_out_keys, d_out_vals, s));
    return 0;                                         // asynchronous: ordered on `stream`
}

// ---- sender-side bloom precheck of the owner shuffle (no reference counterpart) -----------------------------------------
// An owner GPU partitions its build keys by FJ_PREFILTER_BITS radix bits (at the hash_top_bits it joins with) and exports one
// LDS-sized Bloom filter per bucket; a peer partitions the probe rows it is about to send by the same bits, tests them against
// the owner's filters (the bloom stage of the partitioned plan, csrc/fj_bloom.hip, with the filters read from HBM) and sends
// only the survivors.
size_t fj_bloom_filter_words(void) { return ((size_t)1 << FJ_PREFILTER_BITS) * FJ_BLOOM_WORDS + 4; }   // + header (variant)

int fj_bloom_export(fj_ctx* c, const uint64_t* d_build_keys, size_t nb, int hash_top_bits, uint32_t* d_filters, void* stream) {
    if (!c) return set_err("fj_bloom_export: null context");
    if (hash_top_bits != 64 && hash_top_bits != 48) return set_err("fj_bloom_export: hash_top_bits must be 64 or 48");
    if (c->st.active) return set_err("fj_bloom_export: a stream join is open on this context (fj_stream_finish it first)");
    if (!d_filters || (nb && !d_build_keys) || ((uintptr_t)d_build_keys & 15) || ((uintptr_t)d_filters & 15)) return set_err("fj_bloom_export: null or misaligned pointer");
    FJ_ENTER(c);
    hipStream_t s = (hipStream_t)stream;
    c->pend.valid = false;                                 // the passes below reuse the chunk pools a pending emit would read
    const u32 nbuckets = 1u << FJ_PREFILTER_BITS;
    if (nb == 0) {                                          // empty filters reject everything
        HIPCHK(hipMemsetAsync(d_filters, 0, fj_bloom_filter_words() * 4, s));
        HIPCHK(hipMemsetD32Async((hipDeviceptr_t)(d_filters + (size_t)nbuckets * FJ_BLOOM_WORDS), (int)(FJ_BLOOM_HDR_MAGIC | (u32)options().bloom_variant), 1, s));
        return 0;
    }
    Plan plan; plan.bits = FJ_PREFILTER_BITS; plan_passes(plan, true);
    begin_plan(c);
    if (clear_plan_scalars(c, s)) return 1;
    PassIter bit;
    pass_init(bit, 0, false, nb, plan, hash_top_bits);
    FjChunkSet cs{};
    if (run_passes(c, bit, (const u64*)d_build_keys, nullptr, s, &cs, nullptr)) return 1;
    HIPCHK(fj_launch_bloom_export(cs, d_filters, c->num_cus, options().bloom_variant, s));
    if (read_scalars(c, s)) return 1;
    if (c->h_sc->err & FJ_ERR_POOL) return set_err("internal error: chunk pool exhausted during a partition pass");
    end_plan(c);
    return 0;
}

int fj_bloom_prefilter(fj_ctx* c, const uint64_t* d_probe_keys, size_t n, int hash_top_bits, const uint32_t* d_filters,
                       uint64_t* d_out_keys, size_t out_capacity, uint64_t* out_n, void* stream) {
    if (!c) return set_err("fj_bloom_prefilter: null context");
    if (hash_top_bits != 64 && hash_top_bits != 48) return set_err("fj_bloom_prefilter: hash_top_bits must be 64 or 48");
    if (c->st.active) return set_err("fj_bloom_prefilter: a stream join is open on this context (fj_stream_finish it first)");
    if (!d_filters || !out_n || (n && (!d_probe_keys || !d_out_keys)) || ((uintptr_t)d_probe_keys & 15) || ((uintptr_t)d_filters & 15) || ((uintptr_t)d_out_keys & 7))
        return set_err("fj_bloom_prefilter: null or misaligned pointer");
    if (out_capacity < n) return set_err("fj_bloom_prefilter: output capacity %zu < %zu input rows", out_capacity, n);
    *out_n = 0;
    if (n == 0) return 0;
    FJ_ENTER(c);
    hipStream_t s = (hipStream_t)stream;
    c->pend.valid = false;                                 // the passes below reuse the chunk pools a pending emit would read
    Plan plan; plan.bits = FJ_PREFILTER_BITS; plan_passes(plan, true);
    plan.bloom_level = 1;
    begin_plan(c);
    if (clear_plan_scalars(c, s)) return 1;
    const u32 nbuckets = 1u << FJ_PREFILTER_BITS;
    void* p;
    if (get_buf(c, W_BKEYS, (size_t)nbuckets * 8, &p)) return 1; unsigned long long* bkeys = (unsigned long long*)p;
    if (get_buf(c, W_BBASE, ((size_t)nbuckets + 1) * 8, &p)) return 1; unsigned long long* bbase = (unsigned long long*)p;
    HIPCHK(hipMemsetAsync(bkeys, 0, (size_t)nbuckets * 8, s));
    static const FjChunkSet no_build{};                     // (the filters are prebuilt: the build side's chunks are not here)
    PassIter pit;
    pass_init(pit, 1, false, n, plan, hash_top_bits);
    pit.bloom_build = &no_build; pit.bloom_prebuilt = (const u32*)d_filters; pit.bloom_bucket_keys = bkeys;
    FjChunkSet cs{};
    if (run_passes(c, pit, (const u64*)d_probe_keys, nullptr, s, &cs, nullptr)) return 1;      // the pass; the filter stage follows it:
    if (bloom_stage(c, pit, s)) return 1;
    cs = pit.prev;
    HIPCHK(fj_launch_flatten(cs, bkeys, bbase, (u64*)d_out_keys, s));
    if (read_scalars(c, s)) return 1;
    HIPCHK(hipMemcpyAsync(&c->h_sc->expected, &bbase[nbuckets], sizeof(unsigned long long), hipMemcpyDeviceToHost, s));   // (pinned scratch word)
    HIPCHK(hipStreamSynchronize(s));
    if (c->h_sc->err & FJ_ERR_POOL) return set_err("internal error: chunk pool exhausted during a partition pass");
    end_plan(c);
    if (c->h_sc->err & FJ_ERR_VARIANT)
        return set_err("fj_bloom_prefilter: these filters were not exported with bloom_variant %d (every rank must use the same FJ_BLOOM_VARIANT)", options().bloom_variant);
    if (c->h_sc->expected != c->h_sc->bloom_survivors) return set_err("internal error: prefilter flattened %llu of %llu survivors", c->h_sc->expected, c->h_sc->bloom_survivors);
    *out_n = c->h_sc->expected;
    return 0;
}

int fj_generate_build(fj_ctx* c, uint64_t* d_keys, uint64_t* d_vals, uint64_t first, size_t n, void* stream) {
    if (!c) return set_err("fj_generate_build: null context");
    FJ_ENTER(c);
    HIPCHK(fj_launch_gen_build(d_keys, d_vals, first, n, (hipStream_t)stream));
    return 0;
}

int fj_generate_probe(fj_ctx* c, uint64_t* d_keys, uint64_t first, size_t n, uint64_t build_total, uint64_t seed,
                      uint32_t hit_bp, uint64_t* h_expected_hits, void* stream) {
    if (!c) return set_err("fj_generate_probe: null context");
    if (build_total == 0) return set_err("fj_generate_probe: build_total must be > 0");
    FJ_ENTER(c);
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(hipMemsetAsync(&c->d_sc->expected, 0, sizeof(unsigned long long), s));
    HIPCHK(fj_launch_gen_probe(d_keys, first, n, build_total, seed, hit_bp, &c->d_sc->expected, s));
    HIPCHK(hipMemcpyAsync(&c->h_sc->expected, &c->d_sc->expected, sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (h_expected_hits) *h_expected_hits = c->h_sc->expected;
    return 0;
}

// Diagnostic: run `total_bits` of radix partitioning over a flat relation and linearise the final
// chunk lists on the host (bucket by bucket).  Used by the tests to check the partition pass in
// isolation: output must be a permutation of the input with every row in the bucket its hash names.
int fj_debug_partition(fj_ctx* c, const uint64_t* d_keys, const uint64_t* d_vals, size_t n, int total_bits,
                       int hash_top_bits, void* stream, uint64_t* h_out_keys, uint64_t* h_out_vals,
                       uint32_t* h_bucket_of, uint64_t* h_nvalid) {
    if (!c) return set_err("fj_debug_partition: null context");
    if (total_bits < 2 || total_bits > 24) return set_err("fj_debug_partition: total_bits must be 2..24");
    FJ_ENTER(c);
    hipStream_t s = (hipStream_t)stream;
    Plan plan; plan.bits = total_bits;
    plan_passes(plan, true);
    begin_plan(c);
    if (clear_plan_scalars(c, s)) return 1;
    FjChunkSet cs{};
    PassIter dit;
    pass_init(dit, d_vals ? 0 : 1, d_vals != nullptr, n, plan, hash_top_bits);
    if (run_passes(c, dit, d_keys, d_vals, s, &cs, nullptr)) return 1;
    if (read_scalars(c, s)) return 1;
    if (c->h_sc->err) return set_err("fj_debug_partition: device error word 0x%x", c->h_sc->err);
    end_plan(c);
    std::vector<u32> dir(cs.cap), list(cs.cap), boff(cs.nb + 1);
    HIPCHK(hipMemcpy(dir.data(), cs.dir, (size_t)cs.cap * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(list.data(), cs.list, (size_t)cs.cap * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(boff.data(), cs.boff, (size_t)(cs.nb + 1) * 4, hipMemcpyDeviceToHost));
    std::vector<u64> ck(FJ_CHUNK), cv(FJ_CHUNK);
    u64 w = 0;
    for (u32 b = 0; b < cs.nb; ++b) {
        for (u32 i = boff[b]; i < boff[b + 1]; ++i) {
            const u32 id = FJ_LIST_ID(list[i]);
            if (id >= cs.cap) return set_err("fj_debug_partition: list entry %u out of range", id);
            const u32 e = dir[id], cnt = e & FJ_DIR_CNT_MASK;
            if ((e >> FJ_DIR_CNT_BITS) != b) return set_err("fj_debug_partition: chunk %u listed under bucket %u but tagged %u", id, b, e >> FJ_DIR_CNT_BITS);
            if (cnt == 0 || cnt > FJ_CHUNK || cnt != FJ_LIST_CNT(list[i])) return set_err("fj_debug_partition: chunk %u has count %u (list says %u)", id, cnt, FJ_LIST_CNT(list[i]));
            if (w + cnt > n) return set_err("fj_debug_partition: more than %zu rows in the chunk lists", n);
            HIPCHK(hipMemcpy(ck.data(), cs.keys + (size_t)id * FJ_CHUNK, cnt * 8, hipMemcpyDeviceToHost));
            for (u32 j = 0; j < cnt; ++j) h_out_keys[w + j] = fj_key_unmix(ck[j]);       // chunk pools hold mixed keys (fj_common.h)
            if (d_vals && h_out_vals) {
                HIPCHK(hipMemcpy(cv.data(), cs.vals + (size_t)id * FJ_CHUNK, cnt * 8, hipMemcpyDeviceToHost));
                memcpy(h_out_vals + w, cv.data(), cnt * 8);
            }
            for (u32 j = 0; j < cnt; ++j) h_bucket_of[w + j] = b;
            w += cnt;
        }
    }
    *h_nvalid = w;
    return 0;
}

int fj_device_malloc(void** p, size_t bytes) { HIPCHK(hipMalloc(p, bytes ? bytes : 16)); return 0; }
int fj_device_free(void* p) { if (p) HIPCHK(hipFree(p)); return 0; }
int fj_memcpy_h2d(void* d, const void* h, size_t bytes) { if (bytes) HIPCHK(hipMemcpy(d, h, bytes, hipMemcpyHostToDevice)); return 0; }
int fj_memcpy_d2h(void* h, const void* d, size_t bytes) { if (bytes) HIPCHK(hipMemcpy(h, d, bytes, hipMemcpyDeviceToHost)); return 0; }

void fj_free_host(void* p) { free(p); }
int fj_last_timings(fj_timings* out) { if (!out) return set_err("fj_last_timings: null"); *out = g_last; return 0; }

// ---- host-buffer entry: pageable NumPy memory -> pinned staging ring -> HBM, pipelined with the join's first pass ----
}  // extern "C"
namespace {

// memcpy by a few persistent threads: one core copies pageable -> pinned memory at 10-15 GB/s, PCIe Gen5 x16 moves ~55
class CopyPool {
    std::vector<std::thread> th_;
    std::mutex m_;
    std::condition_variable work_, done_;
    const char* src_ = nullptr; char* dst_ = nullptr; size_t n_ = 0;
    unsigned gen_ = 0, remaining_ = 0;
    bool stop_ = false;
    void slice(unsigned id, unsigned parts, size_t* off, size_t* len) const {
        const size_t per = ((n_ / parts) + 4095) & ~(size_t)4095;
        *off = std::min(n_, per * id);
        *len = id + 1 == parts ? n_ - *off : std::min(per, n_ - *off);
    }
    void worker(unsigned id) {
        unsigned seen = 0;
        for (;;) {
            std::unique_lock<std::mutex> l(m_);
            work_.wait(l, [&] { return stop_ || gen_ != seen; });
            if (stop_) return;
            seen = gen_;
            size_t off, len; slice(id + 1, (unsigned)th_.size() + 1, &off, &len);
            const char* s = src_; char* d = dst_;
            l.unlock();
            if (len) memcpy(d + off, s + off, len);
            l.lock();
            if (--remaining_ == 0) done_.notify_one();
        }
    }
public:
    explicit CopyPool(unsigned nthreads) { for (unsigned i = 0; i + 1 < nthreads; ++i) th_.emplace_back([this, i] { worker(i); }); }
    ~CopyPool() { { std::lock_guard<std::mutex> l(m_); stop_ = true; } work_.notify_all(); for (auto& t : th_) t.join(); }
    void copy(void* dst, const void* src, size_t n) {
        if (n < (4u << 20) || th_.empty()) { memcpy(dst, src, n); return; }
        { std::lock_guard<std::mutex> l(m_); src_ = (const char*)src; dst_ = (char*)dst; n_ = n; remaining_ = (unsigned)th_.size(); ++gen_; }
        work_.notify_all();
        size_t off, len; slice(0, (unsigned)th_.size() + 1, &off, &len);
        if (len) memcpy((char*)dst + off, (const char*)src + off, len);
        std::unique_lock<std::mutex> l(m_);
        done_.wait(l, [&] { return remaining_ == 0; });
    }
};
CopyPool& copy_pool() {
    static CopyPool pool([] {
        if (const char* e = getenv("FJ_HOST_COPY_THREADS")) return (unsigned)std::max(1, atoi(e));
        const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
        return std::min(12u, std::max(2u, hw / 4));
    }());
    return pool;
}

// src (pageable host memory) -> dst (device), `piece` bytes at a time through the context's pinned ring; the copy of piece
// i+1 into the ring overlaps the DMA of piece i.  on_piece(offset, bytes, event) is called once a piece's DMA is enqueued on
// the context's copy stream (the event fires when it has landed).
int h2d_pipelined(fj_ctx* c, void* dst, const void* src, size_t bytes, size_t piece, unsigned* cursor,
                  const std::function<int(size_t, size_t, hipEvent_t)>& on_piece) {
    for (size_t off = 0; off < bytes; off += piece) {
        const size_t n = std::min(piece, bytes - off);
        const unsigned k = (*cursor)++ % 3u;
        HIPCHK(hipEventSynchronize(c->ev[E_H0 + k]));                         // the ring slot's previous DMA has left it
        copy_pool().copy(c->stage[k], (const char*)src + off, n);
        HIPCHK(hipMemcpyAsync((char*)dst + off, c->stage[k], n, hipMemcpyHostToDevice, c->side));
        HIPCHK(hipEventRecord(c->ev[E_H0 + k], c->side));
        if (on_piece && on_piece(off, n, c->ev[E_H0 + k])) return 1;
    }
    return 0;
}

}  // namespace
extern "C" {

int fj_join_host(int algo, int bloom, int materialize,
                 const uint64_t* bk, const uint64_t* bv, size_t nb, const uint64_t* pk, size_t np,
                 uint64_t* out_count, double* out_seconds, uint64_t** out_keys, uint64_t** out_vals) {
    if (out_keys) *out_keys = nullptr;
    if (out_vals) *out_vals = nullptr;
    const bool many_host = algo >= 0 && (algo & FJ_ALGO_MANY_TO_MANY) != 0;
    if (algo < 0 || (algo & ~FJ_ALGO_MANY_TO_MANY) > 2) return set_err("fj_join_host: unknown algo %d", algo);
    {
        static std::mutex create_mu;
        std::lock_guard<std::mutex> lk(create_mu);
        if (!g_host_ctx) {
            int dev = 0;
            if (const char* d = getenv("FJ_DEVICE")) dev = atoi(d);
            g_host_ctx = fj_ctx_create(dev);
            if (!g_host_ctx) return 1;
        }
    }
    fj_ctx* c = g_host_ctx;
    FJ_ENTER(c);
    if (c->st.active) {            // an earlier streamed call failed between stream_open and fj_stream_finish: nobody else can
        HIPCHK(hipDeviceSynchronize());   // abort a stream join on this internal context, so drop it here
        c->st.active = false;
    }
    void *dbk, *dbv, *dpk;
    if (get_buf(c, W_H_BK, nb * 8, &dbk) || get_buf(c, W_H_BV, nb * 8, &dbv) || get_buf(c, W_H_PK, np * 8, &dpk)) return 1;
    // pieces: >= 16 MiB (the ring's DMA and memcpy run at full rate), at most 48 of them for the probe side (the streamed
    // join takes <= 64 appends), a multiple of 4 KiB
    size_t piece = std::max<size_t>(16u << 20, (np * 8 + 47) / 48);
    piece = (piece + 4095) & ~(size_t)4095;
    if (c->stage_bytes < piece) {
        for (void*& p : c->stage) { if (p) { (void)hipHostFree(p); p = nullptr; } }
        c->stage_bytes = 0;
        for (void*& p : c->stage) HIPCHK(hipHostMalloc(&p, piece, hipHostMallocDefault));
        c->stage_bytes = piece;
    }
    const Options& opt = options();
    const bool use_radix = algo == FJ_ALGO_RADIX || (algo == FJ_ALGO_ADAPTIVE && nb >= opt.radix_threshold) ||
                           (algo == FJ_ALGO_SCALAR && !opt.scalar_hbm_table);
    // A counting join of the partitioned plan starts on the first piece: the build side is copied and partitioned, then every
    // probe piece gets its first partition pass while the next one crosses PCIe (the join hides under the copy; the bloom
    // precheck is skipped here - it saves device time the copy does not leave on the critical path).
    const bool streamed = use_radix && !materialize && nb > 0 && np > 0 && !many_host;
    hipStream_t js = nullptr;
    auto t0 = std::chrono::steady_clock::now();
    unsigned cursor = 0;
    fj_timings t; memset(&t, 0, sizeof t); t.sampled_hit_bp = -1;
    u64 count = 0;
    bool joined = false;
    if (h2d_pipelined(c, dbk, bk, nb * 8, piece, &cursor, nullptr)) return 1;
    if (!streamed) {
        if (h2d_pipelined(c, dbv, bv, nb * 8, piece, &cursor, nullptr)) return 1;
        if (h2d_pipelined(c, dpk, pk, np * 8, piece, &cursor, nullptr)) return 1;
        HIPCHK(hipStreamSynchronize(c->side));
    } else {
        HIPCHK(hipStreamSynchronize(c->side));                                  // build keys are in HBM
        const int appends = (int)((np * 8 + piece - 1) / piece);
        if (stream_open(c, nb, 1, np, appends, js, 64, piece / 8)) return 1;
        if (stream_append_build(c, (const u64*)dbk, nb, js)) return 1;
        if (stream_flush_build(c, c->st, js)) return 1;
        auto on_piece = [&](size_t off, size_t n, hipEvent_t landed) -> int {
            HIPCHK(hipStreamWaitEvent(js, landed, 0));
            return fj_stream_append_probe(c, (const u64*)((const char*)dpk + off), n / 8, js);
        };
        if (h2d_pipelined(c, dpk, pk, np * 8, piece, &cursor, on_piece)) return 1;
        uint64_t cnt = 0;
        if (fj_stream_finish(c, js, &cnt, &t)) return 1;          // (a partition beyond the LDS tables: it falls back to the HBM table by itself)
        count = cnt; joined = true;
    }
    const double h2d = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (!joined) {
        if (fj_join_device(c, algo, bloom, materialize, (const u64*)dbk, (const u64*)dbv, nb, (const u64*)dpk, np, js, 64,
                           &count, nullptr, nullptr, 0, &t)) return 1;
    }
    double d2h = 0;
    if (materialize && c->pend.valid) {
        void *dok, *dov;
        if (get_buf(c, W_H_OK, count * 8, &dok) || get_buf(c, W_H_OV, count * 8, &dov)) return 1;
        if (emit_pending(c, (u64*)dok, (u64*)dov, count, js, &t)) return 1;
        if (out_keys && out_vals) {
            u64* hk = (u64*)malloc(std::max<size_t>(count, 1) * 8);
            u64* hv = (u64*)malloc(std::max<size_t>(count, 1) * 8);
            if (!hk || !hv) { free(hk); free(hv); return set_err("fj_join_host: out of host memory for %llu pairs", (unsigned long long)count); }
            auto t1 = std::chrono::steady_clock::now();
            if (count) { HIPCHK(hipMemcpy(hk, dok, count * 8, hipMemcpyDeviceToHost)); HIPCHK(hipMemcpy(hv, dov, count * 8, hipMemcpyDeviceToHost)); }
            d2h = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count();
            *out_keys = hk; *out_vals = hv;
        }
    }
    // h2d_ms: wall time from the first byte copied to the last piece enqueued + joined when the join was streamed under the
    // copy (then total_ms, the device-resident time, lies INSIDE it), else the copies alone
    t.h2d_ms = h2d; t.d2h_ms = d2h;
    t.host_streamed = joined ? 1 : 0;
    g_last = t;
    if (out_count) *out_count = count;
    if (out_seconds) *out_seconds = t.total_ms * 1e-3;
    return 0;
}

}  // extern "C"
