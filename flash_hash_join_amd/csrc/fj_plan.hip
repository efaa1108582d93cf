// fj_plan.hip -- workspace, plans and the partition-pass state machine of the host side; contexts, options, diagnostics.
// (Split out of fj_api.hip in round 4; see fj_host.h for the map.)
#include "fj_host.h"

namespace fjh {

namespace { thread_local std::string g_err; thread_local fj_timings g_last; }

int set_err(const char* fmt, ...) {
    char buf[1024];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    g_err = buf;
    return 1;
}
fj_timings& last_timings() { return g_last; }

Options& options() { static Options o; return o; }
static int set_option_value(Options& o, const char* name, long long value);
Options::Options() {
    // Until round 5 every option had an environment variable of its own; they are gone (FJ_OPTIONS carries them all, test hooks are
    // bits of "lab_hooks").  A job that still sets one would be ignored silently: say so, once.
    static const char* const legacy[] = {"FJ_RADIX_THRESHOLD", "FJ_SCALAR_HBM_TABLE", "FJ_BLOOM_AUTO", "FJ_BLOOM_AUTO_MAX_HIT_BP", "FJ_BLOOM_VARIANT", "FJ_PLAN_TARGET_KEYS",
                                         "FJ_MAT_SINGLE_PASS", "FJ_PERSISTENT_MIN_ITEMS", "FJ_JOIN_ITEMS_TARGET", "FJ_JOIN_WIDE", "FJ_DIST_LOOPBACK", "FJ_DIST_INJECT_FAIL",
                                         "FJ_DIST_RESERVE_ALWAYS", "FJ_DIST_SPLIT_ALWAYS", "FJ_DIST_ONE_COMM", "FJ_EMIT_TAGGED"};
    for (const char* name : legacy)
        if (getenv(name)) fprintf(stderr, "flash_hash_join_amd: the environment variable %s is no longer read (use FJ_OPTIONS=\"name=value,...\" or fj_set_option; see include/flashjoin.h)\n", name);
    const char* e = getenv("FJ_OPTIONS");
    if (!e) return;
    std::string str(e);
    size_t pos = 0;
    while (pos < str.size()) {
        size_t end = str.find(',', pos);
        if (end == std::string::npos) end = str.size();
        const std::string item = str.substr(pos, end - pos);
        pos = end + 1;
        if (item.empty()) continue;
        const size_t eq = item.find('=');
        char* tail = nullptr;
        const long long value = eq == std::string::npos ? 0 : strtoll(item.c_str() + eq + 1, &tail, 10);
        if (eq == std::string::npos || eq == 0 || tail == item.c_str() + eq + 1 || *tail != '\0')      // (name=abc used to read as name=0)
            fprintf(stderr, "flash_hash_join_amd: FJ_OPTIONS: ignoring '%s' (want name=integer)\n", item.c_str());
        else if (set_option_value(*this, item.substr(0, eq).c_str(), value))
            fprintf(stderr, "flash_hash_join_amd: FJ_OPTIONS: ignoring '%s' (%s)\n", item.c_str(), fj_last_error());
    }
}

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

Plan make_plan(size_t nb, int top_bits, bool want_bloom, u64 target_override) {
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
    if (c->reserve_cus) it.Gmax = std::min<u32>(it.Gmax, c->reserve_cus < c->num_cus ? c->num_cus - c->reserve_cus : 1u);   // (fj_ctx_reserve_cus)
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
    if (get_zeroed_buf(c, base + W_BCHUNKS, nb_out * 4 + 16, &p, s)) return 1; cs.bchunks = (u32*)p;      // (+ the completion counter of fj_level_lists_binned behind the counts)
    if (get_buf(c, base + W_BOFF, (nb_out + 1) * 4, &p)) return 1; cs.boff = (u32*)p;
    if (get_buf(c, base + W_SEGOFF, (size_t)cs.max_segs * F * 4, &p)) return 1; cs.seg_off = (u32*)p;
    cs.alloc = it.alloc_word ? it.alloc_word : &c->d_sc->alloc[it.side * 4 + i];
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
    a.bchunks = cs.bchunks; a.alloc = cs.alloc; a.seg_counter = it.seg_word ? it.seg_word : &c->d_sc->seg_counter[it.side * 4 + it.i];
    a.cap_chunks = cs.cap; a.max_segs = cs.max_segs;
    a.err = it.err_word ? it.err_word : &c->d_sc->err;
    if (it.have_prev && it.in_pk7) { a.in_pk7 = 1; a.in_b0 = it.in_b0; a.in_top_shift = it.in_top_shift; }
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
    const u64 target_items = options().join_items_target;
    if (nparts < target_items) want = std::min<u64>((target_items + nparts - 1) / nparts, std::max<u64>(1, (pchunks / nparts) / 32));
    *tc = (u32)(want > 1 ? std::max<u64>(8, (avg * 9 / 8 + want - 1) / want) : std::max<u64>(512, 4 * avg));
    *max_items = bound / *tc + nparts + 1;
}

// Counting joins whose build side is not small against the probe side run the bucketed 16384-slot kernel (fj_join_wide.hip): its
// table build costs a third of the cuckoo kernel's (one atomic add and one store per key, nothing to clear), its lookups somewhat
// more (a whole 32-byte bucket per key).  Measured on one box, join kernel ms, cuckoo / wide: 800M x 1B 4.62 / 3.96, 400M x 1B
// 2.80 / 2.57, config 4 behind its filter (100M x ~120M survivors) 0.60 / 0.51 - and 200M x 1B 1.81 / 2.29, 100M x 1B 1.48 / 2.06,
// 125M x 1.25B 1.84 / 2.75: the switch is at ~3 probe rows per build row (profiles/r06_wide_by_shape.txt).
// np_eff: the probe rows that reach the join (behind a filter: an estimate).  Decided before the probe side's passes: the final
// level's bookkeeping cuts the items to 32 chunks.
bool wide_join_planned(bool materialize, size_t nb, size_t np_eff, int bits) {
    if (materialize || bits <= 0 || options().join_wide == 0) return false;
    if (options().join_wide == 1) return true;
    return np_eff <= 3 * nb;
}

// Bookkeeping of the level in it.cs (buffers at it.cs_base), two launches: chunk-list offsets + lists, and the tile table
// of whatever reads the level next - the bloom stage, the next pass, or (probe side, final level) the join's item table
// together with its per-item count array.
int level_finish(fj_ctx* c, PassIter& it, bool final_level, hipStream_t s) {
    const FjChunkSet& cs = it.cs;
    u32 tc = 0; u64 max_tiles = 0; u32* zero_tail = nullptr;
    if (bloom_stage_follows(it, it.i)) tc = fj_bloom_tile_chunks();
    else if (!final_level) tc = fj_partition_tile_chunks((u32)it.plan.fan_log[it.i], it.has_vals);
    else if (it.want_items) {
        join_item_geometry(cs.nb, it.n, it.lbound, &tc, &max_tiles);
        if (it.item_tc_max && tc > it.item_tc_max) { tc = it.item_tc_max; max_tiles = std::max<u64>(it.lbound, (it.n + FJ_CHUNK - 1) / FJ_CHUNK) / tc + cs.nb + 1; }
    }
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
    if (get_zeroed_buf(c, base + W_BCHUNKS, (size_t)cs.nb * 4 + 16, &p, s)) return 1; cs.bchunks = (u32*)p;
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

static int set_option_value(Options& o, const char* name, long long value) {
    if (!strcmp(name, "radix_threshold")) { if (value < 0) return set_err("fj_set_option: radix_threshold must be >= 0"); o.radix_threshold = (size_t)value; return 0; }
    if (!strcmp(name, "scalar_hbm_table")) { o.scalar_hbm_table = value != 0; return 0; }
    if (!strcmp(name, "plan_target_keys")) { if (value < 16 || value > (long long)FJ_PART_TARGET_KEYS) return set_err("fj_set_option: plan_target_keys must be 16..%u", FJ_PART_TARGET_KEYS); o.plan_target_keys = (u32)value; return 0; }
    if (!strcmp(name, "bloom_auto")) { o.bloom_auto = value != 0; return 0; }
    if (!strcmp(name, "mat_single_pass")) { o.mat_single_pass = value != 0; return 0; }
    if (!strcmp(name, "bloom_auto_max_hit_bp")) { if (value < 0 || value > 10000) return set_err("fj_set_option: bloom_auto_max_hit_bp must be 0..10000"); o.bloom_auto_max_hit_bp = (int)value; return 0; }
    if (!strcmp(name, "bloom_variant")) { if (value < 0 || value > 2) return set_err("fj_set_option: bloom_variant must be 0..2"); o.bloom_variant = (int)value; return 0; }
    if (!strcmp(name, "join_wide")) { if (value < 0 || value > 2) return set_err("fj_set_option: join_wide must be 0..2"); o.join_wide = (int)value; return 0; }
    if (!strcmp(name, "persistent_min_items")) { if (value < 0) return set_err("fj_set_option: persistent_min_items must be >= 0"); o.persistent_min_items = (u32)std::min<long long>(value, 0xFFFFFFFFll); return 0; }
    if (!strcmp(name, "lab_hooks")) { if (value < 0 || value > 0xFFFF) return set_err("fj_set_option: lab_hooks is a mask of FJ_HOOK_* bits"); o.lab_hooks = (u32)value; return 0; }
    if (!strcmp(name, "join_items_target")) { if (value < 1 || value > (1 << 20)) return set_err("fj_set_option: join_items_target must be 1..1048576"); o.join_items_target = (u32)value; return 0; }
    return set_err("fj_set_option: unknown option '%s'", name);
}

}  // namespace fjh
using namespace fjh;

void fj_set_error_string(const char* msg) { fjh::g_err = msg ? msg : ""; }     // (csrc/fj_dist.hip reports through the same thread-local string)

extern "C" {

const char* fj_last_error(void) { return fjh::g_err.c_str(); }
const char* fj_version(void) { return "flash_hash_join_amd 0.6 (gfx950)"; }
int fj_abi_version(void) { return FJ_ABI_VERSION; }

int fj_set_option(const char* name, long long value) {
    if (!name) return set_err("fj_set_option: null name");
    return fjh::set_option_value(options(), name, value);
}

long long fj_get_option(const char* name) {
    const Options& o = options();
    if (name && !strcmp(name, "radix_threshold")) return (long long)o.radix_threshold;
    if (name && !strcmp(name, "scalar_hbm_table")) return o.scalar_hbm_table;
    if (name && !strcmp(name, "persistent_min_items")) return o.persistent_min_items;
    if (name && !strcmp(name, "join_wide")) return o.join_wide;
    if (name && !strcmp(name, "plan_target_keys")) return o.plan_target_keys;
    if (name && !strcmp(name, "bloom_variant")) return o.bloom_variant;
    if (name && !strcmp(name, "bloom_auto")) return o.bloom_auto;
    if (name && !strcmp(name, "mat_single_pass")) return o.mat_single_pass;
    if (name && !strcmp(name, "bloom_auto_max_hit_bp")) return o.bloom_auto_max_hit_bp;
    if (name && !strcmp(name, "lab_hooks")) return o.lab_hooks;
    if (name && !strcmp(name, "join_items_target")) return o.join_items_target;
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
    ok = ok && hipEventCreateWithFlags(&c->pk_ev, hipEventDisableTiming) == hipSuccess && hipHostMalloc((void**)&c->pk_h, 67 * 8, hipHostMallocDefault) == hipSuccess;
    if (!ok) { set_err("fj_ctx_create: allocating context scratch failed: %s", hipGetErrorString(hipGetLastError())); delete c; return nullptr; }
    return c;
}

void fj_ctx_destroy(fj_ctx* c) {
    if (!c) return;
    DeviceGuard guard(c->device);
    for (auto& b : c->bufs) if (b.p) (void)hipFree(b.p);
    for (int i = 0; i < E_NEV; ++i) (void)hipEventDestroy(c->ev[i]);
    if (c->side) (void)hipStreamDestroy(c->side);
    if (c->pk_ev) (void)hipEventDestroy(c->pk_ev);
    if (c->pk_h) (void)hipHostFree(c->pk_h);
    for (void* p : c->stage) if (p) (void)hipHostFree(p);
    if (c->d_sc) (void)hipFree(c->d_sc);
    if (c->h_sc) (void)hipHostFree(c->h_sc);
    delete c;
}

size_t fj_ctx_workspace_bytes(const fj_ctx* c) { return c ? c->ws_bytes : 0; }
void fj_ctx_reserve_cus(fj_ctx* c, unsigned n) { if (c) c->reserve_cus = n; }

// Give the cached workspace (chunk pools, directories, tables: tens of GB after a 1B-row join) back to the device; the
// context stays usable and grows again on demand.  A pending emit (fj_emit_pairs not yet called) is dropped.
int fj_ctx_trim(fj_ctx* c) {
    if (!c) c = host_ctx();                       // NULL: the context behind fj_join_host (nothing to do before its first call)
    if (!c) return 0;
    if (c->st.active) return set_err("fj_ctx_trim: a stream join is open on this context (fj_stream_finish it first)");
    FJ_ENTER(c);
    HIPCHK(hipDeviceSynchronize());               // kernels of earlier joins may still read the buffers
    c->pend.valid = false;
    for (auto& b : c->bufs) if (b.p) { HIPCHK(hipFree(b.p)); b.p = nullptr; b.bytes = 0; }
    c->ws_bytes = 0;
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
int fj_memcpy_d2d(void* dst, const void* src, size_t bytes) { if (bytes) HIPCHK(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToDevice)); return 0; }

void fj_free_host(void* p) { free(p); }
int fj_last_timings(fj_timings* out) { if (!out) return set_err("fj_last_timings: null"); *out = last_timings(); return 0; }

}  // extern "C"
