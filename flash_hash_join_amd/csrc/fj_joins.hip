// fj_joins.hip -- the one-shot joins behind fj_join_device: the role of the reference's drivers _hash_join_{radix,scalar}_{count,
// materialize} and adaptive_hash_join_* (hash_join.cpp:315-594); emit, owner split, bloom export / prefilter.
// (Split out of fj_api.hip in round 4; see fj_host.h for the map.)
#include "fj_host.h"

namespace fjh {

static int skew_side(fj_ctx* c, PassIter& it, int side, bool vals, const FjChunkSet& in, const std::vector<u32>& off, u64 chunks, int S, int bits_left,
                     int slot, int tiles_slot, u32* d_nt, hipStream_t s);

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
                if (pd.has_second) {
                    // ... and the oversized partitions once more from THAT level: their sub-partitions' build rows carry row indices too
                    const u32 m = (u32)pd.sk_parts.size();
                    std::vector<u32> bo(2 * m);
                    for (u32 j = 0; j < m; ++j) HIPCHK(hipMemcpyAsync(&bo[2 * j], pd.lds.build.boff + pd.sk_parts[j], 8, hipMemcpyDeviceToHost, s));
                    HIPCHK(hipStreamSynchronize(s));
                    u64 bchunks = 0;
                    for (u32 j = 0; j < m; ++j) bchunks += bo[2 * j + 1] - bo[2 * j];
                    if (get_buf(c, W_SK_NT, 16, &p)) return 1;
                    PassIter bit2;
                    begin_plan(c);
                    if (skew_side(c, bit2, 0, true, pd.lds.build, bo, bchunks, pd.sk_bits, pd.top_bits - pd.sk_plan_bits, pd.sk_npass, W_SK_TILES_B, (u32*)p, s)) return 1;
                    end_plan(c);
                    pd.lds2.build = bit2.prev;
                }
            }
            if (pd.has_second)                  // the items of re-partitioned partitions emit nothing themselves: their sub-partitions do, below
                for (u32 idx : pd.flagged) HIPCHK(hipMemsetAsync(&pd.lds.part_count[idx], 0, 4, s));
            if (get_buf(c, W_OUT_OFF, ((size_t)pd.nitems + 1) * 8, &p)) return 1;
            HIPCHK(fj_launch_scan_u32_to_u64(pd.lds.part_count, (u64*)p, pd.nitems, s));
            pd.lds.out_off = (const u64*)p; pd.lds.out_keys = d_ok; pd.lds.out_vals = d_ov;
            pd.lds.dbg = nullptr;
#ifdef FJ_LAB
            if (getenv("FJ_EMIT_STAMPS") && stamps_begin(&pd.lds.dbg, s)) return 1;
#endif
            const bool resident = !(options().lab_hooks & FJ_HOOK_EMIT_TAGGED);   // (A/B knob)
            if (resident) HIPCHK(hipMemsetAsync(&c->d_sc->next_emit_item, 0, sizeof(u32), s));
            HIPCHK(fj_launch_lds_join(pd.lds, true, s, resident ? &c->d_sc->next_emit_item : nullptr, 1u));
            if (pd.lds.dbg) { if (stamps_report("FJ_EMIT_STAMPS", pd.lds.dbg, pd.nitems, s)) return 1; pd.lds.dbg = nullptr; }
            if (pd.has_second) {                // second item set: the sub-partitions of the oversized partitions, behind the first set's pairs
                if (get_buf(c, W_OUT_OFF2, ((size_t)pd.nitems2 + 1) * 8, &p)) return 1;
                HIPCHK(fj_launch_scan_u32_to_u64(pd.lds2.part_count, (u64*)p, pd.nitems2, s));
                pd.lds2.out_off = (const u64*)p; pd.lds2.out_keys = d_ok + pd.count_main; pd.lds2.out_vals = d_ov + pd.count_main;
                pd.lds2.dedup = pd.has_dups ? 1u : 0u; pd.lds2.orig_vals = pd.has_dups ? pd.bv : nullptr; pd.lds2.dbg = nullptr;
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

// one side of the per-partition skew recovery: the chunk lists of partitions `parts` of `in` (their list ranges in off[2j], off[2j+1]) go
// through one more pass of S radix bits (the pass kernel reads any tile table); vals: the side's payload travels along (values, or -
// first-occurrence emit of duplicate build keys - row indices).  The sub-partitions land in the ping-pong half `slot` does not hold.
static int skew_side(fj_ctx* c, PassIter& it, int side, bool vals, const FjChunkSet& in, const std::vector<u32>& off, u64 chunks, int S, int bits_left,
                     int slot, int tiles_slot, u32* d_nt, hipStream_t s) {
    const u32 m = (u32)(off.size() / 2);
    Plan p2; p2.bits = S; p2.npass = 1; p2.fan_log[0] = S;
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
    pass_init(it, side, vals, std::max<u64>(1, chunks * FJ_CHUNK), p2, bits_left);
    it.parents = m; it.lbound = chunks;
    it.slot = slot;                                          // the ping-pong half that does NOT hold the final level
    it.have_prev = true; it.prev = in; it.tiles = (const uint4*)p; it.ntiles = d_nt;
    if (side) { it.want_items = true; it.part_count_slot = W_PART_COUNT2; }
    // the main plan is done with its first pass's allocator and segment counter: they serve this pass
    HIPCHK(hipMemsetAsync(&c->d_sc->alloc[side * 4], 0, 4, s));
    HIPCHK(hipMemsetAsync(&c->d_sc->seg_counter[side * 4], 0, 4, s));
    if (pass_prepare(c, it, 1, s) || pass_launch(c, it, nullptr, nullptr, 0, s, nullptr) || pass_complete(c, it, s)) return 1;
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
    // A partition's probe side may be cut into MANY items (items of <= 32 chunks for the bucketed kernel; slices of a probe side
    // swollen by a hot key), and the tagged kernel's verdict is per item: under millions of copies of one build key its racing
    // inserts overflow a group in one item and not in the next.  The partitions below are joined again WHOLE, so whatever their
    // other items have already counted comes off the total (and those items count as flagged: an emitting pass must not write
    // their pairs twice).  Found by tools/r6_wide_fuzz.py: 5 distinct build keys x 2.3M copies counted one key's probe rows twice.
    u64 already = 0;
    {
        u32 nlive = nitems;
        HIPCHK(hipMemcpy(&nlive, ja.nitems_dev, 4, hipMemcpyDeviceToHost));
        if (nlive > nitems) nlive = nitems;
        for (u32 i = 0; i < nlive; ++i) {
            if (pc[i] == FJ_ITEM_TOOBIG || pc[i] == FJ_ITEM_RETRY || !std::binary_search(parts.begin(), parts.end(), items[i].z)) continue;
            already += pc[i];
            flagged.push_back(i);
            HIPCHK(hipMemsetAsync(&ja.part_count[i], 0, 4, s));
        }
    }
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
    void* p;
    if (get_buf(c, W_SK_NT, 16, &p)) return 1;
    u32* d_nt = (u32*)p;
    PassIter bit2, pit2;
    const u64 count_main = c->h_sc->total - already;                             // what every other partition found
    if (already) {
        HIPCHK(hipMemcpyAsync(&c->d_sc->total, &count_main, 8, hipMemcpyHostToDevice, s));
        HIPCHK(hipStreamSynchronize(s));                                         // (count_main lives on this stack frame)
    }
    const bool had_dups = (c->h_sc->err & FJ_STAT_DUPS) != 0;
    HIPCHK(hipMemsetAsync(&c->d_sc->err, 0, 4, s));                              // the main join's status bits have been acted on
    if (skew_side(c, bit2, 0, materialize != 0, ja.build, bo, bchunks, S, top_bits - plan.bits, plan.npass, W_SK_TILES_B, d_nt, s)) return 1;
    if (skew_side(c, pit2, 1, false, ja.probe, po, pchunks, S, top_bits - plan.bits, probe_slot, W_SK_TILES_P, d_nt + 1, s)) return 1;
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
        // (duplicate build keys: the first-occurrence emit path re-partitions the whole build side with row indices, and then these
        //  partitions again - emit_pending; what it needs to do that is kept here)
        pend->has_second = true; pend->lds2 = j2; pend->nitems2 = pit2.items_cap; pend->count_main = count_main; pend->flagged = flagged;
        pend->sk_parts = parts; pend->sk_bits = S; pend->sk_plan_bits = plan.bits; pend->sk_npass = plan.npass;
        pend->dups_main = had_dups;                          // (the main join's verdict: its status word was cleared above)
    }
    *ok = true; *nparts_redone = m;
    return 0;
}

// launch the per-partition join over the final chunk sets, read back count + error word, fill the timings
int radix_join_tail(fj_ctx* c, int materialize, FjLdsJoinArgs& ja, const Plan& plan, size_t np, const PassIter& pit, hipStream_t s,
                    fj_timings* t, int evc, u64* out_count, bool* lds_full, int top_bits, SingleOut* so) {
    c->pend.has_second = false; c->pend.dups_main = false;
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
    ja.dbg_flags = (options().lab_hooks & FJ_HOOK_EMIT_RETRY_7TH) ? 8u : 0u;
#ifdef FJ_LAB      // (make EXTRA=-DFJ_LAB: ablations - 1 skip lookups, 2 skip inserts, 4 no output stores: results wrong on purpose - and phase stamps)
    if (getenv("FJ_JOIN_ABLATE")) ja.dbg_flags = (u32)atoi(getenv("FJ_JOIN_ABLATE"));
#endif
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
#ifdef FJ_LAB
    if (getenv("FJ_JOIN_STAMPS") && stamps_begin(&ja.dbg, s)) return 1;
#endif
    // (wide_join_planned: the probe side's final bookkeeping cut the items for the 16384-slot kernel, fj_join_wide.hip)
    const bool wide = pit.item_tc_max == 32 && !materialize && !ja.dbg && !ja.dbg_flags && ja.probe.list && ja.build.list && ja.items;
    if (wide) {
        FjWideArgs wa{};
        wa.pmask = fj_wide_pmask(plan.bits, top_bits);
        // a partition's probe side cut into several items (items hold <= 32 chunks): dealt in runs of 8, a run's items of one partition share one table build
        wa.group_log = np / ((size_t)1 << plan.bits) > 7000 ? 3u : 0u;
        FjLdsJoinArgs jw = ja;
#ifdef FJ_LAB
        if (getenv("FJ_WIDE_STAMPS") && stamps_begin(&jw.dbg, s)) return 1;      // (diagnostic: where a workgroup's time goes, per pipeline stage)
#endif
        const u32 grid = std::min<u32>(nitems, c->num_cus);
        HIPCHK(fj_launch_count_join_wide(jw, wa, false, grid, s));
        if (jw.dbg) {
            std::vector<unsigned long long> h(4096 * 8);
            HIPCHK(hipStreamSynchronize(s));
            HIPCHK(hipMemcpy(h.data(), jw.dbg, h.size() * 8, hipMemcpyDeviceToHost));
            double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (u32 g = 0; g < grid && g < 4096; ++g) for (int i = 0; i < 8; ++i) acc[i] += (double)h[g * 8 + i] * 0.01;
            u32 nit = 0;
            HIPCHK(hipMemcpy(&nit, jw.nitems_dev, 4, hipMemcpyDeviceToHost));           // (nitems is the table's capacity)
            const double per = (double)std::min<u32>(grid, 4096) * ((double)nit / grid);
            fprintf(stderr, "[FJ_WIDE_STAMPS] us per item (thread 0): rotate+requests %.3f  claim_issue+zero %.3f  barB %.3f  probe %.3f  resolve+park %.3f  barA %.3f  stores %.3f\n",
                    acc[0] / per, acc[5] / per, acc[6] / per, acc[1] / per, acc[2] / per, acc[3] / per, acc[4] / per);
            // per wave: top of the iteration -> probe done -> arrival at barrier A; barrier A left -> arrival at barrier B
            for (int ph = 0; ph < 3; ++ph) {
                fprintf(stderr, "[FJ_WIDE_STAMPS] wave 0..15, %s:", ph == 0 ? "requests + claims issued + probe" : ph == 1 ? "claims resolved + park (to barrier A)" : "stores (barrier A to B)");
                for (int w = 0; w < 16; ++w) {
                    double sacc = 0;
                    for (u32 g = 0; g < grid && g < 256; ++g) sacc += (double)h[2048 * 8 + g * 64 + w * 4 + ph] * 0.01;
                    fprintf(stderr, " %.2f", sacc / ((double)std::min<u32>(grid, 256) * ((double)nit / grid)));
                }
                fprintf(stderr, "\n");
            }
        }
    } else
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
        c->pend.has_dups = (c->h_sc->err & FJ_STAT_DUPS) != 0 || c->pend.dups_main;
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
    {
        // probe rows that will reach the join: behind the filter the sampled hit rate + what the filter lets through (~10 % of the misses);
        // a filter that was asked for by name (no sample) is taken to be there for a reason: 15 %
        size_t np_eff = np;
        if (plan.bloom_level > 0) np_eff = (size_t)((double)np * (t->sampled_hit_bp >= 0 ? std::min(10000, t->sampled_hit_bp + 1000) : 1500) / 10000.0);
        if (wide_join_planned(materialize != 0, nb, np_eff, plan.bits)) pit.item_tc_max = 32;
    }
    // bloom precheck: the probe side's level `bloom_level` is filtered against the build side's same level
    if (plan.bloom_level > 0) pit.bloom_build = &bit.saved;
    if (run_passes(c, pit, pk, nullptr, s, &ja.probe, &evc)) return 1;
    HIPCHK(hipEventRecord(c->ev[E_PPART], s));
    ja.avg_build_keys = (u32)std::min<u64>(0xFFFFFFFFu, (u64)nb >> plan.bits);
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


}  // namespace fjh
using namespace fjh;

extern "C" {

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
    last_timings() = t;
    return 0;
}

int fj_emit_pairs(fj_ctx* c, uint64_t* d_out_keys, uint64_t* d_out_vals, size_t out_capacity, void* stream, fj_timings* timings) {
    if (!c) return set_err("fj_emit_pairs: null context");
    if (c->bc.mat_ready) return fj_bcast_emit(c, d_out_keys, d_out_vals, out_capacity, stream);      // a materialising build-broadcast step (csrc/fj_bcast.hip)
    FJ_ENTER(c);
    fj_timings t = last_timings();
    if (emit_pending(c, d_out_keys, d_out_vals, out_capacity, (hipStream_t)stream, &t)) return 1;
    if (timings) *timings = t;
    last_timings() = t;
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
}  // extern "C"
