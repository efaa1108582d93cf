// fj_stream.hip -- joins whose relations arrive in pieces (fj_stream_*), and the owner shuffle's sender / receiver sides
// (fj_shuffle_*, fj_stream_open_shuffled, fj_stream_append_*_chunks).  No reference counterpart: the reference is one process
// (hash_join.cpp:318); what is exploited is that radix partitions are independent join units (:340-356, :515-525).
// (Split out of fj_api.hip in round 4; see fj_host.h for the map.)
#include "fj_host.h"
using namespace fjh;

extern "C" {
}  // extern "C"
namespace fjh {

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
                size_t probe_piece_rows) {
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

}  // namespace fjh
extern "C" {

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
        const int mat = st.shuffled && st.with_vals ? 1 : 0;     // counting pass of a materialising join: the pairs follow with fj_emit_pairs
        st.ja.avg_build_keys = st.ja.build.list && st.ja.build.nb ? (u32)std::min<u64>(0xFFFFFFFFu, (u64)st.nb_seen / st.ja.build.nb) : 0u;
        if (radix_join_tail(c, mat, st.ja, st.plan, st.np_seen, st.pit, s, &t, st.evc, &count, &lds_full, st.top_bits)) return 1;
        if (mat && c->pend.valid && (c->pend.has_dups || c->pend.has_second)) {
            // first-occurrence semantics for duplicate build keys (and the re-partitioning of an oversized partition) need the
            // caller's flat build arrays, which a shuffled stream never sees: the caller takes the owner-scatter form instead
            c->pend.valid = false;
            return set_err("shuffled materialising join: duplicate build keys or an oversized partition (use the owner-scatter form)");
        }
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
            last_timings() = t2;
            return 0;
        }
        count = c->h_sc->total;
        t.path = 0; t.passes = 0; t.partitions = 1;
        t.total_ms = ev_ms(c, E_START, E_JOIN);
        // a materialising shuffled join with an empty side has counted zero pairs: the fj_emit_pairs that follows finds that
        if (st.shuffled && st.with_vals) { c->pend = Pending(); c->pend.valid = true; c->pend.count = 0; }
    }
    if (!t.fell_back) {
        t.build_phase_ms = ev_ms(c, E_SB0, E_SB1);              // the two sides may have run in either order
        t.probe_phase_ms = t.total_ms - t.build_phase_ms;
    }
    if (out_count) *out_count = count;
    if (timings) *timings = t;
    last_timings() = t;
    return 0;
}

// ---- owner shuffle in the shape of SURVEY 8(e): the first radix pass of the GLOBAL plan is the owner split ------------------
// Every rank plans for the TOTAL build side (all ranks' rows): pass 1 of that plan has F0 = 2^fan_log0 buckets, bucket b belongs
// to owner GPU (b * nranks) >> fan_log0.  A sender runs that pass - the ordinary partition pass - over its local rows and
// rewrites the result for the wire (csrc/fj_pack.hip: dense 256-key chunks, bucket after bucket, 7 bytes per key when
// F0 >= 256, one directory word per chunk): what goes to a peer is ONE contiguous range, sized exactly, written where the
// caller wants it (its own share straight into its receive buffer).  The owner appends what it received as level-1 chunk sets
// (fj_stream_append_*_chunks: directory words -> chunk lists, then the plan's SECOND pass, which unpacks the wire format in
// registers) and finishes like any stream join.  Per rank: two passes + one 15-byte-per-key copy over every probe row, and
// 7.02 bytes per key on the links (8.24 with the padded 8-byte chunks of round 3).
namespace {
int shuffle_plan(size_t nb_total, int nranks, Plan* out) {
    if (nranks < 1 || nranks > 64) return set_err("owner shuffle: nranks must be 1..64");
    const Plan p = make_plan(nb_total, 64);
    if (p.npass < 2) return set_err("owner shuffle: a build side of %zu rows in all has a %d-pass plan (the chunk form needs two or more: use fj_owner_split)", nb_total, p.npass);
    if ((1 << p.fan_log[0]) < nranks) return set_err("owner shuffle: %d ranks but only %d first-pass buckets", nranks, 1 << p.fan_log[0]);
    *out = p;
    return 0;
}
bool wire7(const Plan& p) { return p.fan_log[0] >= 8; }
}  // namespace

int fj_shuffle_plan(size_t nb_total, int nranks, int* fan_log0, int* npass) {
    Plan p;
    if (shuffle_plan(nb_total, nranks, &p)) return 1;
    if (fan_log0) *fan_log0 = p.fan_log[0];
    if (npass) *npass = p.npass;
    return 0;
}

size_t fj_shuffle_chunk_bytes(size_t nb_total, int nranks) {
    Plan p;
    if (shuffle_plan(nb_total, nranks, &p)) return 0;
    return wire7(p) ? FJ_WIRE7_BYTES : FJ_CHUNK * 8u;
}

size_t fj_shuffle_part_filter_bytes(void) { return FJ_PFILT_BYTES; }

int fj_shuffle_part_filter_range(size_t nb_total, int nranks, int rank, size_t* first_part, size_t* n_parts, size_t* total_parts) {
    Plan p;
    if (shuffle_plan(nb_total, nranks, &p)) return 1;
    if (rank < 0 || rank >= nranks) return set_err("fj_shuffle_part_filter_range: rank %d of %d", rank, nranks);
    const u64 F0 = 1ull << p.fan_log[0], rest = 1ull << (p.bits - p.fan_log[0]);
    const u64 lo = ((u64)rank * F0 + nranks - 1) / nranks, hi = ((u64)(rank + 1) * F0 + nranks - 1) / nranks;
    if (first_part) *first_part = (size_t)(lo * rest);
    if (n_parts) *n_parts = (size_t)((hi - lo) * rest);
    if (total_parts) *total_parts = (size_t)1 << p.bits;
    return 0;
}

namespace {
// second half of a packing pass: (precheck,) keys per bucket, output chunk ranges, descriptors, counts on their way to the host
int pack_plan_tail(fj_ctx* c, const u64* filters, hipStream_t s) {
    PackState& pk = c->pk;
    PassIter& it = pk.it;
    HIPCHK(hipMemsetAsync(&c->d_sc->pack_kept, 0, sizeof(unsigned long long) + sizeof(c->d_sc->pack_xcd), s));     // (+ pack_xcd, right behind it)
    if (filters)                                                                           // four workgroups per CU take batches of chunks as they become free
        HIPCHK(fj_launch_part_filter_inplace(it.cs, filters, pk.part_shift, &c->d_sc->pack_kept, c->d_sc->pack_xcd, 4u * c->num_cus, s));
    HIPCHK(fj_launch_pack_plan(pk.args, s));
    HIPCHK(hipMemcpyAsync(c->pk_h, c->d_sc->pack_used, 64 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(c->pk_h + 64, &c->d_sc->pack_err, sizeof(u32), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(c->pk_h + 65, &c->d_sc->pack_kept, sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    HIPCHK(hipEventRecord(c->pk_ev, s));
    pk.deferred = false; pk.begun = true;
    return 0;
}
}  // namespace

int fj_shuffle_pack_begin(fj_ctx* c, const uint64_t* d_keys, const uint64_t* d_vals, size_t n, size_t nb_total, int nranks, int defer_plan, void* stream) {
    if (!c) return set_err("fj_shuffle_pack_begin: null context");
    if (n && !d_keys) return set_err("fj_shuffle_pack_begin: null pointer");
    if (((uintptr_t)d_keys | (uintptr_t)d_vals) & 15) return set_err("fj_shuffle_pack_begin: pointers must be 16-byte aligned");
    if (defer_plan && d_vals) return set_err("fj_shuffle_pack_begin: the precheck is for probe pieces (keys only)");
    Plan plan;
    if (shuffle_plan(nb_total, nranks, &plan)) return 1;
    FJ_ENTER(c);
    hipStream_t s = (hipStream_t)stream;
    PackState& pk = c->pk;
    pk = PackState();
    pk.nranks = nranks;
    pk.part_shift = 64u - (u32)plan.bits;
    const bool vals = d_vals != nullptr;
    const u32 fan_log = (u32)plan.fan_log[0], F = 1u << fan_log;
    Plan p1; p1.bits = (int)fan_log; p1.npass = 1; p1.fan_log[0] = (int)fan_log;
    PassIter& it = pk.it;
    pass_init(it, 2, vals, std::max<size_t>(n, 1), p1, 64);
    it.alloc_word = &c->d_sc->pack_alloc; it.seg_word = &c->d_sc->pack_seg; it.err_word = &c->d_sc->pack_err;
    HIPCHK(hipMemsetAsync(&c->d_sc->pack_alloc, 0, 3 * sizeof(u32), s));
    if (pass_prepare(c, it, 1, s)) return 1;
    HIPCHK(hipMemsetAsync(it.cs.bchunks, 0, (size_t)F * 4, s));                  // (this side's counts are not covered by the plan bracket)
    if (n && pass_launch(c, it, (const u64*)d_keys, (const u64*)d_vals, n, s, nullptr)) return 1;
    HIPCHK(fj_launch_group(it.cs, 0, nullptr, nullptr, 0, nullptr, s));           // chunk lists: the copy reads bucket after bucket
    FjPackArgs a{};
    void* p;
    a.keys = it.cs.keys; a.vals = it.cs.vals; a.list = it.cs.list; a.boff = it.cs.boff;
    a.nb = F; a.fan_log = fan_log; a.nranks = (u32)nranks; a.wire7 = wire7(plan) ? 1u : 0u;
    if (get_buf(c, W_PK_FI, (size_t)it.cs.cap * (sizeof(uint4) + 4), &p)) return 1; a.fi = (uint4*)p; a.fb = (u32*)(a.fi + it.cs.cap);
    if (get_buf(c, W_PK_BKEYS, (size_t)F * 4, &p)) return 1; a.bkeys = (u32*)p;
    if (get_buf(c, W_PK_OBASE, ((size_t)F + 1 + 64) * 4, &p)) return 1; a.obase = (u32*)p;
    a.used = c->d_sc->pack_used;
    pk.args = a;
    if (defer_plan) { pk.deferred = true; return 0; }
    return pack_plan_tail(c, nullptr, s);
}

int fj_shuffle_pack_filter(fj_ctx* c, const void* d_part_filters, void* stream) {
    if (!c) return set_err("fj_shuffle_pack_filter: null context");
    if (!c->pk.deferred) return set_err("fj_shuffle_pack_filter: no fj_shuffle_pack_begin(defer_plan = 1) is pending on this context");
    if ((uintptr_t)d_part_filters & 7) return set_err("fj_shuffle_pack_filter: misaligned filters");
    FJ_ENTER(c);
    return pack_plan_tail(c, (const u64*)d_part_filters, (hipStream_t)stream);
}

int fj_part_filter_sample(fj_ctx* c, const uint64_t* d_raw_keys, size_t n, size_t stride, const void* d_part_filters, size_t nb_total, int nranks, void* stream, uint64_t* kept) {
    if (!c || !kept || (n && (!d_raw_keys || !d_part_filters)) || stride == 0) return set_err("fj_part_filter_sample: bad argument");
    Plan plan;
    if (shuffle_plan(nb_total, nranks, &plan)) return 1;
    FJ_ENTER(c);
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(hipMemsetAsync(&c->d_sc->pack_kept, 0, sizeof(unsigned long long), s));
    HIPCHK(fj_launch_part_filter_sample((const u64*)d_raw_keys, n, stride, (const u64*)d_part_filters, 64u - (u32)plan.bits, &c->d_sc->pack_kept, s));
    HIPCHK(hipMemcpyAsync(c->pk_h + 66, &c->d_sc->pack_kept, sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    *kept = c->pk_h[66];
    return 0;
}

int fj_shuffle_pack_counts(fj_ctx* c, uint64_t* h_used) {
    if (!c || !h_used) return set_err("fj_shuffle_pack_counts: null argument");
    FJ_ENTER(c);
    if (!c->pk.begun) return set_err("fj_shuffle_pack_counts: no fj_shuffle_pack_begin is pending on this context");
    HIPCHK(hipEventSynchronize(c->pk_ev));
    if ((u32)c->pk_h[64] & FJ_ERR_POOL) { c->pk.begun = false; return set_err("fj_shuffle_pack: internal error: the packing pass exhausted its chunk pool"); }
    for (int r = 0; r < c->pk.nranks; ++r) h_used[r] = c->pk_h[r];
    return 0;
}

uint64_t fj_shuffle_pack_kept(fj_ctx* c) { return c && c->pk_h ? c->pk_h[65] : 0; }   // after fj_shuffle_pack_counts: keys the precheck kept

int fj_shuffle_pack_finish(fj_ctx* c, void* const* d_dst_chunks, uint64_t* const* d_dst_vals, uint32_t* const* d_dst_dir, void* stream) {
    if (!c || !d_dst_chunks || !d_dst_dir) return set_err("fj_shuffle_pack_finish: null argument");
    FJ_ENTER(c);
    PackState& pk = c->pk;
    if (!pk.begun) return set_err("fj_shuffle_pack_finish: no fj_shuffle_pack_begin is pending on this context");
    HIPCHK(hipEventSynchronize(c->pk_ev));                                        // (a caller that skipped fj_shuffle_pack_counts)
    FjPackArgs a = pk.args;
    if (a.vals && !d_dst_vals) return set_err("fj_shuffle_pack_finish: the piece carries values but no value buffers were given");
    u64 total = 0;
    for (int r = 0; r < pk.nranks; ++r) {
        const u64 u = c->pk_h[r];
        total += u;
        a.dst_k[r] = (unsigned char*)d_dst_chunks[r]; a.dst_d[r] = d_dst_dir[r]; a.dst_v[r] = a.vals ? (u64*)d_dst_vals[r] : nullptr;
        if (u && (!a.dst_k[r] || !a.dst_d[r] || (a.vals && !a.dst_v[r]) || ((uintptr_t)a.dst_k[r] & 15) || ((uintptr_t)a.dst_v[r] & 15)))
            return set_err("fj_shuffle_pack_finish: null or misaligned destination for owner %d (%llu chunks)", r, (unsigned long long)u);
    }
    pk.begun = false;
    if (total == 0) return 0;
    const u32 grid = (u32)std::min<u64>(512u * c->num_cus, (total + 3) / 4);      // 256-thread workgroups, four chunks per step, one or two steps each
    HIPCHK(fj_launch_pack_squeeze(a, grid, (hipStream_t)stream));
    return 0;
}

}  // extern "C"
namespace {
// one side of a shuffled stream join: the plan from its second pass on, over pieces that arrive as chunk lists (pass 1 ran at the senders)
int shuffled_side_init(fj_ctx* c, StreamState& st, int side, size_t n, u32 appends, hipStream_t s) {
    PassIter& it = side ? st.pit : st.bit;
    const Plan& plan = st.plan;
    const u32 F0 = 1u << plan.fan_log[0];
    pass_init(it, side, side == 0 && st.with_vals, std::max<size_t>(n, 1), plan, 64);
    it.i = 1; it.used = 64 - plan.fan_log[0]; it.parents = st.nbk_pad; it.slot = 1;
    it.lbound = it.n / FJ_CHUNK + 1 + (u64)appends * ((u64)64 * F0 + 64);               // one partial chunk per (sender, bucket, piece)
    it.in_pk7 = wire7(plan); it.in_b0 = st.b_lo; it.in_top_shift = (u32)std::max(0, plan.fan_log[0] - 8);
    if (side) it.want_items = true;
    return pass_prepare(c, it, appends, s);
}
}  // namespace
extern "C" {

int fj_stream_open_shuffled(fj_ctx* c, size_t nb_total, int nranks, int rank, size_t nb_bound, int build_appends, size_t np_bound, int probe_appends,
                            int with_vals, void* stream) {
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
    st.plan = plan; st.top_bits = 64; st.shuffled = true; st.with_vals = with_vals != 0;
    const u32 F0 = 1u << plan.fan_log[0];
    st.b_lo = (u32)(((u64)rank * F0 + nranks - 1) / nranks);                     // first bucket b with (b * nranks) >> log2(F0) == rank
    st.nbk = (u32)(((u64)(rank + 1) * F0 + nranks - 1) / nranks) - st.b_lo;
    st.nbk_pad = (st.nbk + 3u) & ~3u;                                            // (fj_level_scan works in 16-B pieces)
    st.np_bound = np_bound; st.nb_bound = nb_bound;
    st.p_appends_left = (u32)probe_appends; st.b_appends_left = (u32)build_appends;
    begin_plan(c);
    HIPCHK(hipEventRecord(c->ev[E_START], s));
    if (clear_plan_scalars(c, s)) return 1;
    if (shuffled_side_init(c, st, 0, nb_bound, (u32)build_appends, s)) return 1;
    // the probe side's pools are sized when its first piece arrives (fj_stream_append_probe_chunks): from the bound given here, or -
    // for an owner of hot probe keys, whose share is far above the mean the bound was derived from - from 1.25x the first piece's rate
    st.probe_prepared = false; st.probe_appends = (u32)probe_appends;
    st.active = true;
    return 0;
}

namespace {
// one received piece (wire-format chunks + their directory words) -> chunk lists + tile table -> the plan's second pass over it
int stream_append_chunks(fj_ctx* c, int side, const void* d_chunks, const u64* d_vals, u32* d_dir, size_t nchunks, hipStream_t s) {
    StreamState& st = c->st;
    PassIter& it = side ? st.pit : st.bit;
    if (nchunks >= (1ull << 24)) return set_err("fj_stream_append_*_chunks: a piece of %zu chunks exceeds one chunk directory", nchunks);
    const u32 n = (u32)nchunks, nblocks = (n + 4095u) / 4096u;
    u32 fan = 4; while (fan < st.nbk_pad) fan <<= 1;
    const u32 tc = fj_partition_tile_chunks((u32)st.plan.fan_log[1], it.has_vals);
    const u64 max_tiles = n / tc + st.nbk_pad + 1;
    FjChunkSet cs{};
    void* p;
    cs.keys = (u64*)const_cast<void*>(d_chunks); cs.vals = const_cast<u64*>(d_vals); cs.dir = d_dir; cs.cap = n; cs.nb = st.nbk_pad; cs.fan_mask = fan - 1; cs.max_segs = nblocks;
    if (get_buf(c, W_RX_REL, (size_t)n * 8, &p)) return 1; cs.rel = (u64*)p;
    if (get_buf(c, W_RX_LIST, (size_t)n * 4, &p)) return 1; cs.list = (u32*)p;
    if (get_buf(c, W_RX_SEGOFF, (size_t)nblocks * fan * 4, &p)) return 1; cs.seg_off = (u32*)p;
    if (get_zeroed_buf(c, W_RX_BCH, (size_t)fan * 4 + 16, &p, s)) return 1; cs.bchunks = (u32*)p;
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

int fj_stream_append_build_chunks(fj_ctx* c, const void* d_chunks, const uint64_t* d_vals, uint32_t* d_dir, size_t nchunks, void* stream) {
    if (!c || !c->st.active || !c->st.shuffled) return set_err("fj_stream_append_build_chunks: no shuffled stream join is open on this context");
    StreamState& st = c->st;
    if (st.build_done) return set_err("fj_stream_append_build_chunks: the build side is already closed");
    if (st.b_appends_left == 0) return set_err("fj_stream_append_build_chunks: more pieces than build_appends");
    if (nchunks && (!d_chunks || !d_dir || ((uintptr_t)d_chunks & 15) || ((uintptr_t)d_vals & 15))) return set_err("fj_stream_append_build_chunks: null or misaligned piece");
    if (nchunks && st.with_vals != (d_vals != nullptr)) return set_err("fj_stream_append_build_chunks: the stream was opened %s values", st.with_vals ? "with" : "without");
    FJ_ENTER(c);
    hipStream_t s = (hipStream_t)stream;
    --st.b_appends_left;
    if (nchunks) { st.nb_seen += nchunks * FJ_CHUNK; if (stream_append_chunks(c, 0, d_chunks, (const u64*)d_vals, d_dir, nchunks, s)) return 1; }
    if (st.b_appends_left == 0) return stream_flush_build(c, st, s);          // the build side is complete: its remaining passes run now
    return 0;
}

int fj_stream_export_part_filters(fj_ctx* c, void* d_out, void* stream) {
    if (!c || !c->st.active || !c->st.shuffled) return set_err("fj_stream_export_part_filters: no shuffled stream join is open on this context");
    StreamState& st = c->st;
    if (!st.build_done) return set_err("fj_stream_export_part_filters: the build side is not complete yet");
    if (!d_out || ((uintptr_t)d_out & 7)) return set_err("fj_stream_export_part_filters: null or misaligned output");
    FJ_ENTER(c);
    hipStream_t s = (hipStream_t)stream;
    const u64 rest = 1ull << (st.plan.bits - st.plan.fan_log[0]);
    const u64 nparts = (u64)st.nbk * rest;                                     // (the padding buckets of the owner's range have no partitions to speak for)
    if (st.nb_seen == 0 || !st.ja.build.list) { HIPCHK(hipMemsetAsync(d_out, 0, (size_t)nparts * FJ_PFILT_BYTES, s)); return 0; }
    FjChunkSet b = st.ja.build;
    b.nb = (u32)nparts;
    HIPCHK(fj_launch_part_filter_export(b, (u64*)d_out, 8u * c->num_cus, s));
    return 0;
}

int fj_stream_append_probe_chunks(fj_ctx* c, const void* d_chunks, uint32_t* d_dir, size_t nchunks, void* stream) {
    if (!c || !c->st.active || !c->st.shuffled) return set_err("fj_stream_append_probe_chunks: no shuffled stream join is open on this context");
    StreamState& st = c->st;
    if (st.probe_done) return set_err("fj_stream_append_probe_chunks: the probe side is already closed");
    if (st.p_appends_left == 0) return set_err("fj_stream_append_probe_chunks: more pieces than probe_appends");
    if (nchunks && (!d_chunks || !d_dir || ((uintptr_t)d_chunks & 15))) return set_err("fj_stream_append_probe_chunks: null or misaligned piece");
    FJ_ENTER(c);
    --st.p_appends_left;
    if (nchunks == 0) return 0;
    if (!st.probe_prepared) {
        const size_t seen = (size_t)(1.25 * (double)nchunks * FJ_CHUNK * (st.p_appends_left + 1)) + ((size_t)1 << 20);
        st.np_bound = std::max(st.np_bound, seen);
        if (shuffled_side_init(c, st, 1, st.np_bound, st.probe_appends, (hipStream_t)stream)) return 1;
        st.probe_prepared = true;
    }
    st.np_seen += nchunks * FJ_CHUNK;
    return stream_append_chunks(c, 1, d_chunks, nullptr, d_dir, nchunks, (hipStream_t)stream);
}

}  // extern "C"
