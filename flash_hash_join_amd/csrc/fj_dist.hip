// fj_dist.hip -- the multi-GPU counting radix join behind the C ABI: relations block-distributed over the ranks of a
// communicator (BASELINE configs[4]: 1B x 10B rows over 8 MI355X), ONE driver for every transport.
//
// No reference counterpart: the reference is one process (hash_join.cpp:318).  What is exploited is that radix partitions
// are independent join units (hash_join.cpp:340-356, :515-525).  Two forms of the step, one driver (fj_dist_comm_set_form; by
// default a cost model over the all-gathered sizes picks, fj_dist_model):
//   OWNER SHUFFLE - the first radix pass of the plan for the TOTAL build side is the owner split, the exchange is an
//     all-to-all of dense wire-format chunks (7 bytes per key, fj_pack.hip), every owner runs the rest of the plan on what it
//     received (the protocol sketched below);
//   BUILD BROADCAST - the probe rows stay where they are; every rank's build rows, partitioned by the whole plan and packed to 6 bytes
//     per key, go to every peer, and every rank joins its own probe partitions against all of them (dist_join_bcast, fj_bcast.hip).
//
//   sizes all-gathered -> plan -> build side as one piece -> probe side in `pieces` pieces; per piece:
//     pack_begin (first pass, pack stream)  ->  counts to the host, all-gathered (control channel)  ->  buffers, agreed on  ->
//     pack_finish (the copy into the wire format: own share straight into the receive buffer, the rest into a send pool)  ->
//     exchange (exchange stream)  ->  append (second pass, join stream)
//   piece c+1's first pass is queued behind piece c's copy, piece c's append is enqueued before the host waits for piece
//   c+1's counts: the three streams stay busy and the host blocks once per piece.
//   -> finish -> all-reduce of (count, failure flag).
//
// Two seams, so that the same state machine runs everywhere:
//   Net    - RCCL (grouped ncclSend / ncclRecv on an exchange stream, control collectives on a second communicator so that a
//            count exchange does not queue behind the previous piece's sends), or a caller's blocking callbacks (gloo, MPI,
//            a test harness: fj_dist_transport);
//   Engine - the HIP engine (fj_shuffle_pack_* / fj_stream_*_chunks of include/flashjoin.h), or a caller's stand-in
//            (fj_dist_engine_ops: the CPU test-suite drives this file over gloo without a GPU).
// A failure of one rank's packing or allocation is agreed on before anybody posts an exchange; a failure of its local join - or
// of the copy into the wire format, behind a piece's agreement points: the shares then travel marked unusable - never takes it
// out of step (the rank keeps taking part, the ranks agree in the final all-reduce).  NOT covered: a failure of the transport
// itself behind the agreement points - ncclSend returns an error - makes that rank return while its peers have posted the
// matching receives: they wait in the transport, and the job has to be torn down from outside (RCCL's own behaviour for a rank
// that dies mid-collective; a watchdog around the step is the host's business).  RCCL is bound at run
// time (dlopen of librccl.so.1 - in a PyTorch process that is the copy torch already loaded) through locally declared
// prototypes: the library neither links against RCCL nor needs its headers.  RCCL moves wrong data when one point-to-point
// message exceeds 4 GiB (tools/rccl_large_message_check.py): messages are cut into rounds of <= 1 GiB.
#include "fj_host.h"

#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

void fj_set_error_string(const char* msg);       // fj_plan.hip: the thread-local string fj_last_error() returns

namespace {

// ---- the few RCCL declarations this file needs (rccl.h: stable ABI since NCCL 2.7) ----
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;                         // ncclSuccess == 0
constexpr int ncclUint8 = 1, ncclUint64 = 5, ncclSum = 0;

struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommSplit)(ncclComm_t, int, int, ncclComm_t*, void*) = nullptr;      // optional (NCCL >= 2.18)
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string err;
};

Rccl* rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* name : {"librccl.so.1", "librccl.so"}) {
            r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.lib) break;
        }
        if (!r.lib) { r.err = std::string("cannot load librccl.so.1: ") + (dlerror() ? dlerror() : "?"); return; }
        auto sym = [&](const char* n, bool optional = false) -> void* { void* p = dlsym(r.lib, n); if (!p && !optional && r.err.empty()) r.err = std::string("librccl has no ") + n; return p; };
        r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
        r.CommSplit = (decltype(r.CommSplit))sym("ncclCommSplit", true);
        r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
        r.CommCount = (decltype(r.CommCount))sym("ncclCommCount");
        r.CommUserRank = (decltype(r.CommUserRank))sym("ncclCommUserRank");
        r.Send = (decltype(r.Send))sym("ncclSend");
        r.Recv = (decltype(r.Recv))sym("ncclRecv");
        r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
        r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
        r.AllReduce = (decltype(r.AllReduce))sym("ncclAllReduce");
        r.AllGather = (decltype(r.AllGather))sym("ncclAllGather");
        r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
    });
    return &r;
}

int derr(const char* fmt, ...) {
    char buf[1024];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    fj_set_error_string(buf);
    return 1;
}
#define DHIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return derr("%s:%d: %s failed: %s", __FILE__, __LINE__, #x, hipGetErrorString(e_)); } while (0)
#define DNCCL(x) do { ncclResult_t r_ = (x); if (r_ != 0) return derr("%s:%d: %s failed: %s", __FILE__, __LINE__, #x, rccl()->GetErrorString ? rccl()->GetErrorString(r_) : "?"); } while (0)

constexpr int MAX_PIECES = 16;
constexpr size_t MAX_MSG_BYTES = (size_t)1 << 30;          // 1 GiB per point-to-point message and round
constexpr unsigned long long FAIL = 1ull << 60;            // in a counts vector: this rank could not pack the piece
typedef void* Token;                                       // a hipEvent_t (work ordered on a stream) or null (already done)

// ---- Engine: what a rank does to its own rows ---------------------------------------------------------------------------
struct Engine {
    virtual ~Engine() {}
    virtual int plan(size_t nb_total, int nranks, size_t* chunk_bytes) = 0;     // same verdict on every rank (same arguments)
    virtual void* alloc(size_t bytes) = 0;
    virtual void release(void* p) = 0;
    // asynchronous (ordered behind the previous pack_finish); vals: null = keys only; filters: null, or the partition filters the
    // piece is prechecked against once `filters_ready` has happened (the first pass does not wait for it)
    virtual int pack_begin(const void* rows, const void* vals, size_t n, size_t nb_total, int nranks, const void* filters, Token filters_ready) = 0;
    virtual unsigned long long pack_kept() { return 0; }                        // after pack_counts: rows the precheck kept
    virtual bool has_precheck() const { return false; }
    // sender-side precheck: where rank r's partition filters sit; this owner's filters; a sample of raw rows
    virtual int filter_range(size_t, int, int, size_t*, size_t*, size_t*, size_t*) { return derr("fj_dist: this engine has no sender-side precheck"); }
    virtual int export_filters(void*, Token*) { return derr("fj_dist: this engine has no sender-side precheck"); }
    virtual int sample(const void*, size_t, size_t, const void*, size_t, int, Token, unsigned long long*) { return derr("fj_dist: this engine has no sender-side precheck"); }
    virtual int pack_counts(unsigned long long* used) = 0;                      // blocks until the counts are known
    virtual int pack_finish(void* const* dst_chunks, uint64_t* const* dst_vals, uint32_t* const* dst_dir, Token after, Token* done) = 0;
    // pack_finish failed: the directory words of every share of the piece say "no chunk" (where the engine can write them: device memory of ours)
    virtual void mark_unusable(uint32_t* const*, const unsigned long long*, int) {}
    virtual int open(size_t nb_total, int nranks, int rank, size_t nb_bound, size_t np_bound, int pieces, bool with_vals) = 0;
    virtual int append(int side, const void* chunks, const uint64_t* vals, uint32_t* dir, size_t nchunks, Token after) = 0;
    virtual int finish(uint64_t* count, fj_timings* lt) = 0;
    virtual void abort() = 0;
    virtual int drain() = 0;                                                    // everything this engine enqueued has finished
    virtual void reserve(unsigned) {}                                           // leave n CUs to the transport's own kernels (HIP engine over RCCL); 0: none
    // ---- build-broadcast form (csrc/fj_bcast.hip) ----
    virtual bool has_bcast() const { return false; }
    virtual size_t bc_region_bytes(size_t, size_t, bool /*with values*/ = false) { return 0; }               // 0: no such plan
    virtual int bc_span(size_t, size_t, size_t, size_t, int, size_t*, size_t*) { return derr("fj_dist: this engine has no build-broadcast form"); }
    virtual int bc_nparts(size_t, uint32_t*) { return derr("fj_dist: this engine has no build-broadcast form"); }
    virtual int bc_pack(const void*, const void* /*values*/, bool /*a materialising join: the values travel*/, size_t, size_t, void*, int, Token*) { return derr("fj_dist: this engine has no build-broadcast form"); }   // asynchronous
    virtual int bc_bounds(int, unsigned long long*) { return derr("fj_dist: this engine has no build-broadcast form"); }                      // blocks
    virtual int bc_probe(const void*, size_t, size_t) { return derr("fj_dist: this engine has no build-broadcast form"); }
    virtual int bc_join(const void*, int, const uint64_t*, const uint64_t*, uint32_t, uint32_t, Token) { return derr("fj_dist: this engine has no build-broadcast form"); }
    virtual int bc_finish(uint64_t*, fj_timings*) { return derr("fj_dist: this engine has no build-broadcast form"); }
    virtual void bc_abort() {}
};

struct HipEngine : Engine {
    fj_ctx* ctx; hipStream_t js = nullptr, ps = nullptr;                        // join stream (the caller's), pack stream
    hipEvent_t ev_pack[2] = {nullptr, nullptr}, ev_filt = nullptr; int evi = 0;
    explicit HipEngine(fj_ctx* c) : ctx(c) {}
    int setup() {
        DHIP(hipStreamCreateWithFlags(&ps, hipStreamNonBlocking));
        for (auto& e : ev_pack) DHIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        DHIP(hipEventCreateWithFlags(&ev_filt, hipEventDisableTiming));
        return 0;
    }
    ~HipEngine() override { for (auto& e : ev_pack) if (e) (void)hipEventDestroy(e); if (ev_filt) (void)hipEventDestroy(ev_filt); if (ps) (void)hipStreamDestroy(ps); }
    int plan(size_t nb_total, int nranks, size_t* cb) override {
        int f0 = 0, np = 0;
        if (fj_shuffle_plan(nb_total, nranks, &f0, &np)) return 1;
        *cb = fj_shuffle_chunk_bytes(nb_total, nranks);
        return 0;
    }
    void* alloc(size_t bytes) override { void* p = nullptr; if (hipMalloc(&p, bytes) != hipSuccess) { (void)hipGetLastError(); return nullptr; } return p; }
    void release(void* p) override { if (p) (void)hipFree(p); }
    int pack_begin(const void* rows, const void* vals, size_t n, size_t nb_total, int nranks, const void* filters, Token ready) override {
        if (fj_shuffle_pack_begin(ctx, (const uint64_t*)rows, (const uint64_t*)vals, n, nb_total, nranks, filters ? 1 : 0, ps)) return 1;
        if (!filters) return 0;
        if (ready) DHIP(hipStreamWaitEvent(ps, (hipEvent_t)ready, 0));
        return fj_shuffle_pack_filter(ctx, filters, ps);
    }
    unsigned long long pack_kept() override { return fj_shuffle_pack_kept(ctx); }
    bool has_precheck() const override { return true; }
    int filter_range(size_t nb_total, int nranks, int r, size_t* first, size_t* count, size_t* total, size_t* bytes_each) override {
        *bytes_each = fj_shuffle_part_filter_bytes();
        return fj_shuffle_part_filter_range(nb_total, nranks, r, first, count, total);
    }
    int export_filters(void* dst, Token* done) override {
        if (fj_stream_export_part_filters(ctx, dst, js)) return 1;
        DHIP(hipEventRecord(ev_filt, js));
        *done = ev_filt;
        return 0;
    }
    int sample(const void* rows, size_t n, size_t stride, const void* filters, size_t nb_total, int nranks, Token after, unsigned long long* kept) override {
        if (after) DHIP(hipStreamWaitEvent(ps, (hipEvent_t)after, 0));
        uint64_t k = 0;
        if (fj_part_filter_sample(ctx, (const uint64_t*)rows, n, stride, filters, nb_total, nranks, ps, &k)) return 1;
        *kept = k;
        return 0;
    }
    int pack_counts(unsigned long long* used) override { return fj_shuffle_pack_counts(ctx, (uint64_t*)used); }
    int pack_finish(void* const* dk, uint64_t* const* dv, uint32_t* const* dd, Token after, Token* done) override {
        if (after) DHIP(hipStreamWaitEvent(ps, (hipEvent_t)after, 0));
        if (fj_shuffle_pack_finish(ctx, dk, dv, dd, ps)) return 1;
        hipEvent_t e = ev_pack[evi ^= 1];
        DHIP(hipEventRecord(e, ps));
        *done = e;
        return 0;
    }
    void mark_unusable(uint32_t* const* dd, const unsigned long long* used, int n) override {
        for (int d = 0; d < n; ++d) if (used[d] && dd[d]) (void)hipMemsetAsync(dd[d], 0xFF, (size_t)used[d] * 4, ps);      // FJ_DIR_INVALID
        (void)hipStreamSynchronize(ps);
    }
    int open(size_t nb_total, int nranks, int rank, size_t nb_bound, size_t np_bound, int pieces, bool with_vals) override {
        return fj_stream_open_shuffled(ctx, nb_total, nranks, rank, nb_bound, 1, np_bound, pieces, with_vals ? 1 : 0, js);
    }
    int append(int side, const void* chunks, const uint64_t* vals, uint32_t* dir, size_t n, Token after) override {
        if (side && (fj_get_option("lab_hooks") & FJ_HOOK_INJECT_FAIL)) return derr("injected failure of a local append (test hook: lab_hooks & FJ_HOOK_INJECT_FAIL)");
        if (after) DHIP(hipStreamWaitEvent(js, (hipEvent_t)after, 0));
        return side ? fj_stream_append_probe_chunks(ctx, chunks, dir, n, js) : fj_stream_append_build_chunks(ctx, chunks, vals, dir, n, js);
    }
    int finish(uint64_t* count, fj_timings* lt) override { return fj_stream_finish(ctx, js, count, lt); }
    void abort() override { (void)fj_stream_abort(ctx); }
    int drain() override { DHIP(hipStreamSynchronize(ps)); DHIP(hipStreamSynchronize(js)); return 0; }
    // RCCL's kernels are resident for the length of an exchange: the partition passes and the wide join (one persistent workgroup per
    // CU, static tile shares) can leave them room - fj_ctx_reserve_cus.  How many CUs, if any, is measured per communicator
    // (reserve_for_step below): a pass that does not get every CU it was launched for takes twice as long, a pass launched on 224
    // CUs runs ~10 % slower than on 256.
    void reserve(unsigned n) override { fj_ctx_reserve_cus(ctx, n); }
    bool has_bcast() const override { return true; }
    size_t bc_region_bytes(size_t nb_total, size_t nkeys, bool vals) override { return fj_bcast_region_bytes(nb_total, nkeys, vals ? 1 : 0); }
    int bc_span(size_t nb_total, size_t nkeys, size_t lo, size_t hi, int part, size_t* off, size_t* bytes) override { return fj_bcast_piece_span(nb_total, nkeys, lo, hi, part, off, bytes); }
    int bc_nparts(size_t nb_total, uint32_t* n) override { return fj_bcast_plan(nb_total, nullptr, n, nullptr); }
    int bc_pack(const void* rows, const void* vals, bool with_vals, size_t n, size_t nb_total, void* region, int pieces, Token* packed) override {
        if (fj_bcast_pack(ctx, (const uint64_t*)rows, (const uint64_t*)vals, n, nb_total, region, pieces, with_vals ? 1 : 0, js)) return 1;
        DHIP(hipEventRecord(ev_filt, js));
        *packed = ev_filt;
        return 0;
    }
    int bc_bounds(int pieces, unsigned long long* b) override { (void)pieces; return fj_bcast_pack_bounds(ctx, (uint64_t*)b); }
    int bc_probe(const void* rows, size_t n, size_t nb_total) override { return fj_bcast_probe(ctx, (const uint64_t*)rows, n, nb_total, js); }
    int bc_join(const void* base, int nsrc, const uint64_t* off, const uint64_t* nk, uint32_t lo, uint32_t hi, Token after) override {
        if (after) DHIP(hipStreamWaitEvent(js, (hipEvent_t)after, 0));
        return fj_bcast_join(ctx, base, nsrc, off, nk, lo, hi, js);
    }
    int bc_finish(uint64_t* count, fj_timings* lt) override { return fj_bcast_finish(ctx, js, count, lt); }
    void bc_abort() override { fj_bcast_abort(ctx); }
};

struct CallbackEngine : Engine {                             // a caller's stand-in (tests): everything is synchronous, tokens are null
    fj_dist_engine_ops o;
    explicit CallbackEngine(const fj_dist_engine_ops& ops) { memset(&o, 0, sizeof o); memcpy(&o, &ops, std::min(sizeof o, ops.struct_size)); }   // (a caller built against a shorter struct: the callbacks it does not know read as NULL)
    int fail(const char* what) { return derr("%s: %s", what, o.error ? o.error(o.user) : "engine callback failed"); }
    int plan(size_t nb_total, int nranks, size_t* cb) override { if (o.plan(o.user, nb_total, nranks)) return fail("plan"); *cb = o.chunk_bytes; return 0; }
    void* alloc(size_t bytes) override { return o.alloc(o.user, bytes); }
    void release(void* p) override { if (p) o.release(o.user, p); }
    uint64_t kept_ = 0;
    bool has_precheck() const override { return o.filter_range && o.export_filters && o.pack_filter && o.sample; }
    int pack_begin(const void* rows, const void* vals, size_t n, size_t nb_total, int nranks, const void* filters, Token) override {
        if (vals) return derr("fj_dist: the stand-in engine carries no values");
        if (filters && !has_precheck()) return derr("fj_dist: this stand-in engine has no precheck");
        if (o.pack_begin(o.user, rows, n, nb_total, nranks)) return fail("pack_begin");
        kept_ = n;
        return filters && o.pack_filter(o.user, filters, &kept_) ? fail("pack_filter") : 0;
    }
    unsigned long long pack_kept() override { return kept_; }
    int filter_range(size_t nb_total, int nranks, int r, size_t* first, size_t* count, size_t* total, size_t* bytes_each) override {
        if (!has_precheck()) return derr("fj_dist: this stand-in engine has no precheck");
        uint64_t f = 0, c = 0, t = 0, b = 0;
        if (o.filter_range(o.user, nb_total, nranks, r, &f, &c, &t, &b)) return fail("filter_range");
        *first = f; *count = c; *total = t; *bytes_each = b;
        return 0;
    }
    int export_filters(void* dst, Token* done) override { *done = nullptr; return o.export_filters(o.user, dst) ? fail("export_filters") : 0; }
    int sample(const void* rows, size_t n, size_t stride, const void* filters, size_t nb_total, int nranks, Token, unsigned long long* kept) override {
        uint64_t k = 0;
        if (o.sample(o.user, rows, n, stride, filters, nb_total, nranks, &k)) return fail("sample");
        *kept = k;
        return 0;
    }
    int pack_counts(unsigned long long* used) override { return o.pack_counts(o.user, (uint64_t*)used) ? fail("pack_counts") : 0; }
    int pack_finish(void* const* dk, uint64_t* const*, uint32_t* const* dd, Token, Token* done) override { *done = nullptr; return o.pack_finish(o.user, dk, dd) ? fail("pack_finish") : 0; }
    int open(size_t nb_total, int nranks, int rank, size_t nb_bound, size_t np_bound, int pieces, bool) override { return o.open(o.user, nb_total, nranks, rank, nb_bound, np_bound, pieces) ? fail("open") : 0; }
    int append(int side, const void* chunks, const uint64_t*, uint32_t* dir, size_t n, Token) override { return o.append(o.user, side, chunks, dir, n) ? fail("append") : 0; }
    int finish(uint64_t* count, fj_timings* lt) override { memset(lt, 0, sizeof *lt); return o.finish(o.user, count) ? fail("finish") : 0; }
    void abort() override { if (o.abort) o.abort(o.user); }
    int drain() override { return 0; }
    bool has_bcast() const override { return o.bc_region_bytes && o.bc_span && o.bc_nparts && o.bc_pack && o.bc_probe && o.bc_join && o.bc_finish; }
    size_t bc_region_bytes(size_t nb_total, size_t nkeys, bool vals) override { return has_bcast() && !vals ? (size_t)o.bc_region_bytes(o.user, nb_total, nkeys) : 0; }      // (a stand-in carries no values)
    int bc_span(size_t nb_total, size_t nkeys, size_t lo, size_t hi, int part, size_t* off, size_t* bytes) override {
        uint64_t a = 0, b = 0;
        if (o.bc_span(o.user, nb_total, nkeys, lo, hi, part, &a, &b)) return fail("bc_span");
        *off = a; *bytes = b;
        return 0;
    }
    int bc_nparts(size_t nb_total, uint32_t* n) override { return o.bc_nparts(o.user, nb_total, n) ? fail("bc_nparts") : 0; }
    std::vector<uint64_t> bounds_;
    int bc_pack(const void* rows, const void*, bool with_vals, size_t n, size_t nb_total, void* region, int pieces, Token* packed) override {
        if (with_vals) return derr("fj_dist: the stand-in engine carries no values");
        bounds_.assign((size_t)pieces + 1, 0);
        *packed = nullptr;
        return o.bc_pack(o.user, rows, n, nb_total, region, pieces, bounds_.data()) ? fail("bc_pack") : 0;
    }
    int bc_bounds(int pieces, unsigned long long* b) override { for (int q = 0; q <= pieces; ++q) b[q] = bounds_[q]; return 0; }
    int bc_probe(const void* rows, size_t n, size_t nb_total) override { return o.bc_probe(o.user, rows, n, nb_total) ? fail("bc_probe") : 0; }
    int bc_join(const void* base, int nsrc, const uint64_t* off, const uint64_t* nk, uint32_t lo, uint32_t hi, Token) override {
        return o.bc_join(o.user, base, nsrc, off, nk, lo, hi) ? fail("bc_join") : 0;
    }
    int bc_finish(uint64_t* count, fj_timings* lt) override { memset(lt, 0, sizeof *lt); return o.bc_finish(o.user, count) ? fail("bc_finish") : 0; }
};

// ---- Net: what moves between ranks --------------------------------------------------------------------------------------
struct Net {
    int nranks = 1, rank = 0;
    virtual ~Net() {}
    virtual int all_gather(const unsigned long long* v, int n, unsigned long long* out) = 0;     // blocking
    virtual int all_reduce(unsigned long long* v, int n) = 0;                                    // blocking, sum
    // nparts parts (chunks, directory words[, values]): sp[p * nranks + r] / sb[...] = what goes to rank r, rp / rb = what arrives from it.
    // Starts after `after`; *done = the data have landed (null: they have when the call returns).
    // `largest` = the largest single message anywhere in the group (every rank passes the same value: both ends of a message
    // must cut it into the same rounds).
    virtual int exchange(int nparts, const void* const* sp, const size_t* sb, void* const* rp, const size_t* rb, size_t largest, Token after, Token* done, int slot) = 0;
    virtual int drain() = 0;
    virtual bool loopback() const { return false; }          // test hook: a rank's own share travels through the transport too
    virtual bool shares_the_gpu() const { return false; }    // the transport runs kernels of its own on this GPU during an exchange (RCCL)
    virtual void begin_step() {}
};

struct RcclNet : Net {
    ncclComm_t data = nullptr, ctl = nullptr;
    bool own_data = false, own_ctl = false, loop = false;
    hipStream_t xs = nullptr, cs = nullptr;                  // exchange stream, control stream
    bool split() const { return own_ctl; }
    hipEvent_t ev_x[MAX_PIECES + 2];                          // build piece, probe pieces, partition filters
    unsigned long long* d_w = nullptr; unsigned long long* h_w = nullptr;     // scratch words of the control collectives (+ pinned mirror)
    static constexpr size_t WORDS = 72 + 64 * 65 + 16;
    int setup() {
        for (auto& e : ev_x) e = nullptr;
        DHIP(hipStreamCreateWithFlags(&xs, hipStreamNonBlocking));
        DHIP(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
        for (auto& e : ev_x) DHIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        DHIP(hipMalloc((void**)&d_w, WORDS * 8));
        DHIP(hipHostMalloc((void**)&h_w, WORDS * 8, hipHostMallocDefault));
        // control collectives on a communicator of their own: on ONE communicator RCCL runs operations in the order they were
        // issued whatever their streams, and a count exchange must not wait for the previous piece's sends
        Rccl* R = rccl();
        const long long hooks = fj_get_option("lab_hooks");
        if (R->CommSplit && (nranks > 1 || (hooks & FJ_HOOK_SPLIT_ALWAYS)) && !(hooks & FJ_HOOK_ONE_COMM)) {      // (FJ_HOOK_SPLIT_ALWAYS: test hook, a 1-rank communicator is split too)
            ncclComm_t c2 = nullptr;
            if (R->CommSplit(data, 0, rank, &c2, nullptr) == 0 && c2) { ctl = c2; own_ctl = true; }
        }
        if (!ctl) ctl = data;
        return 0;
    }
    ~RcclNet() override {
        (void)hipDeviceSynchronize();
        if (d_w) (void)hipFree(d_w);
        if (h_w) (void)hipHostFree(h_w);
        for (auto& e : ev_x) if (e) (void)hipEventDestroy(e);
        if (xs) (void)hipStreamDestroy(xs);
        if (cs) (void)hipStreamDestroy(cs);
        Rccl* R = rccl();
        if (own_ctl && ctl && R->CommDestroy) (void)R->CommDestroy(ctl);
        if (own_data && data && R->CommDestroy) (void)R->CommDestroy(data);
    }
    int all_gather(const unsigned long long* v, int n, unsigned long long* out) override {
        if (n > 65) return derr("fj_dist: control vector of %d words", n);
        Rccl* R = rccl();
        memcpy(h_w, v, 8 * (size_t)n);
        DHIP(hipMemcpyAsync(d_w, h_w, 8 * (size_t)n, hipMemcpyHostToDevice, cs));
        DNCCL(R->AllGather(d_w, d_w + 72, (size_t)n, ncclUint64, ctl, cs));
        DHIP(hipMemcpyAsync(h_w + 72, d_w + 72, 8 * (size_t)n * nranks, hipMemcpyDeviceToHost, cs));
        DHIP(hipStreamSynchronize(cs));
        memcpy(out, h_w + 72, 8 * (size_t)n * nranks);
        return 0;
    }
    int all_reduce(unsigned long long* v, int n) override {
        if (n > 8) return derr("fj_dist: control vector of %d words", n);
        Rccl* R = rccl();
        unsigned long long* d = d_w + 72 + 64 * 65;
        memcpy(h_w, v, 8 * (size_t)n);
        DHIP(hipMemcpyAsync(d, h_w, 8 * (size_t)n, hipMemcpyHostToDevice, cs));
        DNCCL(R->AllReduce(d, d + 8, (size_t)n, ncclUint64, ncclSum, ctl, cs));
        DHIP(hipMemcpyAsync(h_w + 8, d + 8, 8 * (size_t)n, hipMemcpyDeviceToHost, cs));
        DHIP(hipStreamSynchronize(cs));
        memcpy(v, h_w + 8, 8 * (size_t)n);
        return 0;
    }
    bool loopback() const override { return loop; }
    bool shares_the_gpu() const override { return true; }
    void begin_step() override { loop = (fj_get_option("lab_hooks") & FJ_HOOK_LOOPBACK) != 0; }      // test hook: a rank's own share travels through ncclSend / ncclRecv too
    bool is_peer(int r) const { return r != rank || loop; }
    int exchange(int nparts, const void* const* sp, const size_t* sb, void* const* rp, const size_t* rb, size_t largest, Token after, Token* done, int slot) override {
        Rccl* R = rccl();
        if (after) DHIP(hipStreamWaitEvent(xs, (hipEvent_t)after, 0));
        bool any = false;
        for (int i = 0; i < nparts * nranks; ++i) if (is_peer(i % nranks)) any = any || sb[i] || rb[i];
        const size_t rounds = std::max<size_t>(1, (largest + MAX_MSG_BYTES - 1) / MAX_MSG_BYTES);
        auto cut = [&](size_t bytes, size_t r) { return bytes / rounds * r + std::min(r, bytes % rounds); };     // round r of a message = [cut(r), cut(r + 1))
        for (size_t r = 0; any && r < rounds; ++r) {
            DNCCL(R->GroupStart());
            for (int i = 0; i < nparts * nranks; ++i) {
                const int d = i % nranks;
                if (is_peer(d) && cut(sb[i], r + 1) > cut(sb[i], r)) DNCCL(R->Send((const char*)sp[i] + cut(sb[i], r), cut(sb[i], r + 1) - cut(sb[i], r), ncclUint8, d, data, xs));
            }
            for (int i = 0; i < nparts * nranks; ++i) {
                const int q = i % nranks;
                if (is_peer(q) && cut(rb[i], r + 1) > cut(rb[i], r)) DNCCL(R->Recv((char*)rp[i] + cut(rb[i], r), cut(rb[i], r + 1) - cut(rb[i], r), ncclUint8, q, data, xs));
            }
            DNCCL(R->GroupEnd());
        }
        hipEvent_t e = ev_x[slot % (MAX_PIECES + 2)];
        DHIP(hipEventRecord(e, xs));
        *done = e;
        return 0;
    }
    int drain() override { DHIP(hipStreamSynchronize(xs)); return 0; }
};

struct CallbackNet : Net {                                   // a caller's blocking transport
    fj_dist_transport t;
    explicit CallbackNet(const fj_dist_transport& tr) : t(tr) { nranks = tr.nranks; rank = tr.rank; }
    int all_gather(const unsigned long long* v, int n, unsigned long long* out) override {
        return t.all_gather_u64(t.user, (const uint64_t*)v, n, (uint64_t*)out) ? derr("fj_dist: the transport's all_gather_u64 failed") : 0;
    }
    int all_reduce(unsigned long long* v, int n) override {
        return t.all_reduce_sum_u64(t.user, (uint64_t*)v, n) ? derr("fj_dist: the transport's all_reduce_sum_u64 failed") : 0;
    }
    int exchange(int nparts, const void* const* sp, const size_t* sb, void* const* rp, const size_t* rb, size_t, Token after, Token* done, int) override {
        if (after) DHIP(hipEventSynchronize((hipEvent_t)after));          // the transport reads the buffers from the host side
        std::vector<uint64_t> s64(sb, sb + nparts * nranks), r64(rb, rb + nparts * nranks);
        *done = nullptr;
        return t.all_to_all_bytes(t.user, nparts, sp, s64.data(), rp, r64.data()) ? derr("fj_dist: the transport's all_to_all_bytes failed") : 0;
    }
    int drain() override { return 0; }
};

struct DBuf { void* p = nullptr; size_t bytes = 0; };

// what the step measured -> the caller's struct, as far as the caller's struct goes (fj_dist_timings::struct_size, checked at entry)
void write_timings(fj_dist_timings* out, const fj_dist_timings& T) {
    const size_t n = std::min(out->struct_size, sizeof T);
    memcpy((char*)out + sizeof(size_t), (const char*)&T + sizeof(size_t), n - sizeof(size_t));
}

}  // namespace

struct fj_dist_comm {
    std::unique_ptr<Net> net;
    std::unique_ptr<Engine> eng;
    HipEngine* hip = nullptr;                               // == eng.get() when the engine is the HIP one
    DBuf pool_k[2], pool_d[2], pool_v;                      // send pools (alternating per piece; values: the build piece only)
    DBuf recv_k[MAX_PIECES + 1], recv_d[MAX_PIECES + 1], recv_v;   // what arrives: [0] build side, [1 + c] probe piece c
    DBuf filt;                                              // every final partition's Bloom filter (sender-side precheck)
    DBuf bcast;                                             // build-broadcast form: the regions of all ranks, rank after rank
    int form = FJ_DIST_FORM_AUTO; double link_rate = 0.0;   // fj_dist_comm_set_form
    // the CU reserve, measured: successful steps so far; this rank's probe-side pass time (ms) of the step that ran with / without it
    // (per join shape: a step of other sizes - a pre-flight check in front of the real joins - starts the measurement afresh)
    int rs_step = 0; double rs_with_ms = 0.0, rs_without_ms = 0.0; unsigned long long rs_nb = 0, rs_np = 0;
    int grow(DBuf& b, size_t bytes) {
        if (bytes == 0) bytes = 16;
        if (b.bytes >= bytes) return 0;
        if (b.p) { eng->release(b.p); b.p = nullptr; b.bytes = 0; }
        const size_t want = (bytes + 4095) & ~(size_t)4095;
        b.p = eng->alloc(want);
        if (!b.p) return derr("fj_dist: allocating %zu bytes of device memory failed", want);
        b.bytes = want;
        return 0;
    }
    void free_all() {
        for (auto* arr : {pool_k, pool_d}) for (int i = 0; i < 2; ++i) if (arr[i].p) { eng->release(arr[i].p); arr[i] = DBuf(); }
        for (auto* arr : {recv_k, recv_d}) for (int i = 0; i <= MAX_PIECES; ++i) if (arr[i].p) { eng->release(arr[i].p); arr[i] = DBuf(); }
        for (DBuf* b : {&pool_v, &recv_v, &filt, &bcast}) if (b->p) { eng->release(b->p); *b = DBuf(); }
    }
};

// CUs the step's passes and joins leave to the transport (and how the number came about: fj_dist_timings::reserve_how).  tot_with /
// tot_without: every rank's measured pass time summed (the same on every rank: they travelled in the step's first all-gather).
static constexpr unsigned RESERVE_CUS = 32;
static unsigned reserve_for_step(const fj_dist_comm* dc, bool applies, double tot_with, double tot_without, int* how) {
    *how = 0;
    if (!applies) return 0;
    if (const char* e = getenv("FJ_DIST_RESERVE_CUS")) { *how = 1; return (unsigned)atoi(e); }
    if (dc->rs_step <= 1) { *how = 2; return RESERVE_CUS; }            // step 0 (pools grow, communicators connect) and the measuring step WITH the reserve
    if (dc->rs_step == 2) { *how = 2; return 0; }                      // the measuring step WITHOUT it
    *how = 3;
    return tot_with > 0 && tot_without > 0 && tot_without < tot_with ? 0u : RESERVE_CUS;
}
static void reserve_account(fj_dist_comm* dc, const fj_timings& lt, int how) {         // after a successful step
    double ms = 0;
    for (int i = 0; i < 4; ++i) ms += lt.probe_part_kernel_ms[i];
    if (how == 2 && dc->rs_step == 1) dc->rs_with_ms = ms;             // (only measuring steps measure: a pinned reserve records nothing)
    if (how == 2 && dc->rs_step == 2) dc->rs_without_ms = ms;
    ++dc->rs_step;
}


// ---- the step in its BUILD-BROADCAST form ----------------------------------------------------------------------------------------
// The probe rows never move.  Every rank packs its build rows into its region (fj_bcast_pack: both passes of the plan for the
// TOTAL build side, then one run per final partition, 6 bytes per key), queues both passes over its own probe rows behind that,
// and - while those run - learns where the pieces of its region (consecutive partitions) begin, tells the others, and sends every
// piece to every peer (grouped send / recv: one link per peer on an xGMI mesh).  The join of partition range q against the runs of
// ALL ranks is queued behind piece q's arrival.  One blocking point per step (the bounds), two small control collectives, one
// agreement on the buffers; a rank-local failure is agreed on like in the shuffle form.
static int dist_join_bcast(fj_dist_comm* dc, const uint64_t* d_build_keys, const uint64_t* d_build_vals, bool mat, size_t nb, const uint64_t* d_probe_keys, size_t np, int pieces,
                           const std::vector<uint64_t>& nb_of, unsigned long long nb_total, uint64_t* out_global_count, uint64_t* out_local_count,
                           fj_dist_timings* timings, unsigned reserve_n, int reserve_how, double tot_with, double tot_without) {
    Net& net = *dc->net; Engine& eng = *dc->eng;
    const int N = net.nranks, me = net.rank;
    const auto t0 = std::chrono::steady_clock::now();
    auto ms_since = [](std::chrono::steady_clock::time_point a) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count(); };
    uint32_t nparts = 0;
    if (eng.bc_nparts(nb_total, &nparts)) return 1;
    if ((uint32_t)pieces > nparts) pieces = (int)nparts;
    std::vector<uint64_t> roff(N + 1, 0);
    // (mat: a materialising step - the regions carry the values as a fourth part; a rank without build rows may hold no value pointer)
    const int nparts_x = mat ? 4 : 3;
    for (int r = 0; r < N; ++r) roff[r + 1] = roff[r] + ((eng.bc_region_bytes(nb_total, (size_t)nb_of[r], mat) + 255) & ~(size_t)255);
    net.begin_step();
    const bool reserve = reserve_n > 0;
    struct ReserveGuard { Engine& e; bool on; ~ReserveGuard() { if (on) e.reserve(0); } } reserve_guard{eng, reserve};
    if (reserve) eng.reserve(reserve_n);
    bool begun = false;
    auto bail = [&](const std::string& why) { (void)net.drain(); (void)eng.drain(); if (begun) eng.bc_abort(); return derr("%s", why.c_str()); };

    // buffers, agreed on
    unsigned long long bad = dc->grow(dc->bcast, (size_t)roff[N]) ? 1 : 0;
    const std::string why0 = bad ? fj_last_error() : "";
    const bool mine_bad = bad != 0;
    if (net.all_reduce(&bad, 1)) return 1;
    if (bad) { char b[1200]; snprintf(b, sizeof b, "fj_dist_join_count: the broadcast buffer could not be allocated on %llu rank(s)%s%s", bad, mine_bad ? "; this rank: " : "", why0.c_str()); return derr("%s", b); }
    char* base = (char*)dc->bcast.p;

    // pack + probe passes queued; the host then waits for the pack's bounds only
    Token packed = nullptr;
    bool ok = eng.bc_pack(d_build_keys, d_build_vals, mat, nb, nb_total, base + roff[me], pieces, &packed) == 0;
    begun = ok;
    std::string why = ok ? "" : fj_last_error();
    if (ok && eng.bc_probe(d_probe_keys, np, nb_total)) { ok = false; why = fj_last_error(); }
    const auto tb = std::chrono::steady_clock::now();
    std::vector<unsigned long long> v((size_t)pieces + 2, 0), all((size_t)N * (pieces + 2));
    if (ok && eng.bc_bounds(pieces, v.data())) { ok = false; why = fj_last_error(); }
    const double split_ms = ms_since(tb);
    v[pieces + 1] = ok ? 0 : FAIL;
    if (net.all_gather(v.data(), pieces + 2, all.data())) return bail(fj_last_error());
    int nfail = 0;
    for (int r = 0; r < N; ++r) if (all[(size_t)r * (pieces + 2) + pieces + 1] >= FAIL) ++nfail;
    if (nfail) { char b[1200]; snprintf(b, sizeof b, "fj_dist_join_count: packing the build side failed on %d rank(s)%s%s", nfail, ok ? "" : "; this rank: ", why.c_str()); return bail(b); }
    auto bound = [&](int r, int q) { return (size_t)all[(size_t)r * (pieces + 2) + q]; };

    // piece q of every region to every peer; the join of range q behind its arrival
    std::string failed;
    auto guarded = [&](int rc) { if (rc && failed.empty()) failed = fj_last_error(); return rc; };
    size_t wire_sent = 0;
    std::vector<const void*> sp(4 * N); std::vector<void*> rp(4 * N); std::vector<size_t> sb(4 * N), rb(4 * N);
    for (int q = 0; q < pieces; ++q) {
        std::fill(sp.begin(), sp.end(), nullptr); std::fill(rp.begin(), rp.end(), nullptr); std::fill(sb.begin(), sb.end(), 0); std::fill(rb.begin(), rb.end(), 0);
        size_t largest = 0;
        bool spans_ok = true;
        for (int r = 0; r < N; ++r) {
            if (r == me) continue;
            for (int part = q == 0 ? 0 : 1; part < nparts_x; ++part) {
                size_t off = 0, bytes = 0;
                spans_ok = spans_ok && eng.bc_span(nb_total, (size_t)nb_of[me], bound(me, q), bound(me, q + 1), part, &off, &bytes) == 0;     // what I send to r
                sp[part * N + r] = base + roff[me] + off; sb[part * N + r] = bytes; wire_sent += bytes;
                largest = std::max(largest, bytes);
                spans_ok = spans_ok && eng.bc_span(nb_total, (size_t)nb_of[r], bound(r, q), bound(r, q + 1), part, &off, &bytes) == 0;       // what r sends to me
                rp[part * N + r] = base + roff[r] + off; rb[part * N + r] = bytes;
                largest = std::max(largest, bytes);
            }
        }
        if (!spans_ok) return bail(fj_last_error());               // (same arguments on every rank: everybody bails)
        // `largest` must be the same on both ends of every message: the largest piece of ANY region
        for (int r = 0; r < N; ++r) for (int part = 0; part < nparts_x; ++part) {
            size_t off = 0, bytes = 0;
            if (eng.bc_span(nb_total, (size_t)nb_of[r], bound(r, q), bound(r, q + 1), part, &off, &bytes) == 0) largest = std::max(largest, bytes);
        }
        Token done = nullptr;
        if (net.exchange(nparts_x, sp.data(), sb.data(), rp.data(), rb.data(), largest, packed, &done, q)) return bail(fj_last_error());
        if (failed.empty()) {
            // (the last range's join starts when the last piece has landed: the transport's kernels are gone from this GPU by then)
            if (reserve && q + 1 == pieces) eng.reserve(0);
            const uint32_t plo = (uint32_t)(((uint64_t)nparts * q) / pieces), phi = (uint32_t)(((uint64_t)nparts * (q + 1)) / pieces);
            guarded(eng.bc_join(base, N, roff.data(), nb_of.data(), plo, phi, done ? done : packed));
        }
    }
    const auto t2 = std::chrono::steady_clock::now();
    uint64_t local = 0;
    fj_timings lt; memset(&lt, 0, sizeof lt);
    if (failed.empty()) { guarded(eng.bc_finish(&local, &lt)); if (failed.empty()) begun = false; }
    if (!failed.empty()) { local = 0; if (begun) eng.bc_abort(); begun = false; }
    unsigned long long res[2] = {local, failed.empty() ? 0ull : 1ull};
    if (net.all_reduce(res, 2)) return bail(fj_last_error());
    if (net.drain() || eng.drain()) return 1;
    if (res[1]) return derr("fj_dist_join_count: the local join failed on %llu rank(s)%s%s", res[1], failed.empty() ? "" : "; this rank: ", failed.c_str());
    if (out_global_count) *out_global_count = res[0];
    if (out_local_count) *out_local_count = local;
    reserve_account(dc, lt, reserve_how);
    if (timings) {
        fj_dist_timings T; memset(&T, 0, sizeof T);
        T.reserve_cus = (int)reserve_n; T.reserve_how = reserve_how; T.reserve_with_ms = tot_with; T.reserve_without_ms = tot_without;
        T.total_ms = ms_since(t0); T.split_ms = split_ms; T.join_ms = ms_since(t2);
        T.exchange_ms = std::max(0.0, T.total_ms - T.join_ms - split_ms);
        T.local_count = local; T.pieces = pieces; T.nranks = N; T.local = lt;
        T.form = FJ_DIST_FORM_BROADCAST; T.wire_bytes_sent = wire_sent; T.prefilter_sampled = -1.0;
        T.probe_rows_kept = np;
        T.local_build_chunks = nb_total; T.local_probe_chunks = np;       // (this form: ROWS this rank joined - every rank's build rows, its own probe rows)
        write_timings(timings, T);
    }
    return 0;
}

extern "C" {

int fj_dist_unique_id(char* out128) {
    Rccl* R = rccl();
    if (!R->err.empty()) return derr("fj_dist_unique_id: %s", R->err.c_str());
    if (!out128) return derr("fj_dist_unique_id: null buffer");
    ncclUniqueId id;
    DNCCL(R->GetUniqueId(&id));
    memcpy(out128, &id, sizeof id);
    return 0;
}

static fj_dist_comm* comm_over_rccl(fj_ctx* ctx, ncclComm_t nccl, bool own, int nranks, int rank) {
    fj_dist_comm* dc = new fj_dist_comm();
    RcclNet* net = new RcclNet();
    dc->net.reset(net);
    net->data = nccl; net->own_data = own; net->nranks = nranks; net->rank = rank;
    HipEngine* he = new HipEngine(ctx);
    dc->eng.reset(he); dc->hip = he;
    if (net->setup() || he->setup()) { delete dc; return nullptr; }
    return dc;
}

fj_dist_comm* fj_dist_comm_create(fj_ctx* ctx, const char* unique_id128, int nranks, int rank) {
    Rccl* R = rccl();
    if (!R->err.empty()) { derr("fj_dist_comm_create: %s", R->err.c_str()); return nullptr; }
    if (!ctx || !unique_id128 || nranks < 1 || nranks > 64 || rank < 0 || rank >= nranks) { derr("fj_dist_comm_create: bad arguments (1..64 ranks)"); return nullptr; }
    ncclUniqueId id; memcpy(&id, unique_id128, sizeof id);
    ncclComm_t nccl = nullptr;
    ncclResult_t r = R->CommInitRank(&nccl, nranks, id, rank);          // (the caller has selected this rank's device)
    if (r != 0) { derr("fj_dist_comm_create: ncclCommInitRank failed: %s", R->GetErrorString(r)); return nullptr; }
    return comm_over_rccl(ctx, nccl, true, nranks, rank);
}

fj_dist_comm* fj_dist_comm_from_nccl(fj_ctx* ctx, void* nccl_comm) {
    Rccl* R = rccl();
    if (!R->err.empty()) { derr("fj_dist_comm_from_nccl: %s", R->err.c_str()); return nullptr; }
    if (!ctx || !nccl_comm) { derr("fj_dist_comm_from_nccl: null argument"); return nullptr; }
    int nranks = 0, rank = 0;
    if (R->CommCount((ncclComm_t)nccl_comm, &nranks) != 0 || R->CommUserRank((ncclComm_t)nccl_comm, &rank) != 0 || nranks < 1 || nranks > 64) {
        derr("fj_dist_comm_from_nccl: not a usable communicator (1..64 ranks)"); return nullptr;
    }
    return comm_over_rccl(ctx, (ncclComm_t)nccl_comm, false, nranks, rank);
}

fj_dist_comm* fj_dist_comm_from_transport(fj_ctx* ctx, const fj_dist_transport* transport, const fj_dist_engine_ops* engine) {
    if (!transport || !transport->all_gather_u64 || !transport->all_reduce_sum_u64 || !transport->all_to_all_bytes ||
        transport->nranks < 1 || transport->nranks > 64 || transport->rank < 0 || transport->rank >= transport->nranks) {
        derr("fj_dist_comm_from_transport: incomplete transport (three callbacks, 1..64 ranks)"); return nullptr;
    }
    if (!ctx && !engine) { derr("fj_dist_comm_from_transport: a context or a stand-in engine is needed"); return nullptr; }
    if (engine && engine->struct_size < offsetof(fj_dist_engine_ops, filter_range)) { derr("fj_dist_comm_from_transport: fj_dist_engine_ops.struct_size is not set (ABI %d: the struct begins with its own size)", FJ_ABI_VERSION); return nullptr; }
    if (engine && (!engine->plan || !engine->alloc || !engine->release || !engine->pack_begin || !engine->pack_counts || !engine->pack_finish || !engine->open ||
                   !engine->append || !engine->finish || engine->chunk_bytes == 0)) { derr("fj_dist_comm_from_transport: incomplete engine"); return nullptr; }
    fj_dist_comm* dc = new fj_dist_comm();
    dc->net.reset(new CallbackNet(*transport));
    if (engine) dc->eng.reset(new CallbackEngine(*engine));
    else { HipEngine* he = new HipEngine(ctx); dc->eng.reset(he); dc->hip = he; if (he->setup()) { delete dc; return nullptr; } }
    return dc;
}

void fj_dist_comm_destroy(fj_dist_comm* dc) {
    if (!dc) return;
    if (dc->hip) (void)hipDeviceSynchronize();
    dc->free_all();
    delete dc;
}

// The cost model behind FJ_DIST_FORM_AUTO, as plain arithmetic (also what tools/scale_model.py prints): seconds of one counting step in
// either form = max(bytes per link / link rate, kernel seconds per rank) + what cannot overlap.  Kernel seconds per row measured on
// one MI355X at config 5's per-rank sizes (round 6: profiles/r06_bcast_one_rank_2_4_8.txt; shuffle: r04_c5_one_rank_kernel_stats.csv):
// broadcast - pack 1.41 ms per 125M build rows; two probe-side passes 7.92 / 8.33 / 8.60 ms per 1.25B rows under the 16- / 17- / 18-bit
// plans of 2 / 4 / 8 ranks; dense join (the bucketed table) 2.0 ps per build key of ALL ranks + 2.2 ps per local probe key = 3.25 / 3.75 /
// 4.75 ms modelled, 3.25 / 4.05 / 4.77 measured at 2 / 4 / 8 ranks (round 5's table: 3.85 / 4.8 / 6.2), all of it kernels of this rank
// (the wire overlaps everything behind the pack); shuffle - 13.4 ms of kernels per 1.375B rows of both relations + ~2.5 ms of head and
// tail outside the overlap.  region_max: the largest fj_bcast_region_bytes of any rank (0: computed from nb_max and the plan).
// the broadcast form's terms (seconds): bytes on one link, this rank's pack, probe-side passes, dense join
static void bcast_terms(uint64_t nb_max, uint64_t np_max, uint64_t nb_total, uint64_t region_max, double rate, int nranks, double* wire, double* pack, double* passes, double* join) {
    const int bits = fjh::make_plan((size_t)nb_total, 64).bits;
    if (region_max == 0) region_max = nb_max * (bits >= 16 ? 6u : 8u) + 4ull * ((1ull << bits) + 1) + 64;      // (fj_bcast_region_bytes: offset table + the two planes)
    *pack = (double)nb_max * 11.3e-12;
    *passes = (double)np_max * (6.34e-12 + 0.27e-12 * (bits > 16 ? std::min(bits, 18) - 16 : 0));
    *join = (double)nb_total * 2.0e-12 + (double)np_max * 2.2e-12;
    *wire = nranks > 1 ? (double)region_max / rate : 0.0;
}
// ... in p pieces: the last of p partition ranges is joined after the wire is done; a range join beyond the fourth costs ~20 us of
// ramp and tail (measured as one rank of 8: join 4.76 / 4.84 / 4.92 ms in 4 / 8 / 16 ranges)
static double bcast_time(double wire, double pack, double passes, double join, int p) {
    return std::max(wire + pack + join / p, pack + passes + join + 20e-6 * (p > 4 ? p - 4 : 0));
}
// pieces = 0 ("the driver decides"): 8 where the wire bounds the broadcast step (the tail behind the last piece is an eighth of the
// join instead of a quarter), else 4; the shuffle: 4 (the measured default)
static int auto_pieces(bool bcast, int nranks, uint64_t nb_max, uint64_t np_max, uint64_t nb_total, uint64_t region_max, double rate) {
    if (!bcast || nranks < 2) return 4;
    double wire, pack, passes, join;
    bcast_terms(nb_max, np_max, nb_total, region_max, rate, nranks, &wire, &pack, &passes, &join);
    return bcast_time(wire, pack, passes, join, 8) < bcast_time(wire, pack, passes, join, 4) ? 8 : 4;
}
int fj_dist_model(int nranks, uint64_t nb_max, uint64_t np_max, uint64_t nb_total, uint64_t np_global, uint64_t region_max, double link_bytes_per_s,
                  double* t_shuffle, double* t_broadcast) {
    const double rate = link_bytes_per_s > 0 ? link_bytes_per_s : 55e9, N = nranks < 1 ? 1 : nranks;
    double wire_b, pack, passes, join;
    bcast_terms(nb_max, np_max, nb_total, region_max, rate, nranks, &wire_b, &pack, &passes, &join);
    const double t_b = std::min(bcast_time(wire_b, pack, passes, join, 4), bcast_time(wire_b, pack, passes, join, 8));      // (the piece count fj_dist_join takes by itself)
    const double rows = (double)nb_max + (double)np_max;
    const double wire_s = N > 1 ? 7.02 * ((double)np_global + (double)nb_total) / (N * N) / rate : 0.0;
    const double t_s = std::max(wire_s, rows * 9.75e-12) + 2.5e-3 * rows / 1.375e9;
    if (t_shuffle) *t_shuffle = t_s;
    if (t_broadcast) *t_broadcast = t_b;
    return t_b < t_s ? FJ_DIST_FORM_BROADCAST : FJ_DIST_FORM_SHUFFLE;
}

int fj_dist_comm_set_form(fj_dist_comm* dc, int form, double link_bytes_per_s) {
    if (!dc) return derr("fj_dist_comm_set_form: null communicator");
    if (form != FJ_DIST_FORM_AUTO && form != FJ_DIST_FORM_SHUFFLE && form != FJ_DIST_FORM_BROADCAST) return derr("fj_dist_comm_set_form: unknown form %d", form);
    dc->form = form; dc->link_rate = link_bytes_per_s;
    return 0;
}

int fj_dist_comm_rank(const fj_dist_comm* dc) { return dc ? dc->net->rank : -1; }
int fj_dist_comm_size(const fj_dist_comm* dc) { return dc ? dc->net->nranks : 0; }

int fj_dist_join_count(fj_dist_comm* dc, const uint64_t* d_build_keys, size_t nb, const uint64_t* d_probe_keys, size_t np, int pieces,
                       void* stream, uint64_t* out_global_count, fj_dist_timings* timings) {
    return fj_dist_join(dc, d_build_keys, nullptr, nb, d_probe_keys, np, pieces, 0, 0.0, stream, out_global_count, nullptr, timings);
}

int fj_dist_join(fj_dist_comm* dc, const uint64_t* d_build_keys, const uint64_t* d_build_vals, size_t nb, const uint64_t* d_probe_keys, size_t np, int pieces,
                 int materialize, double prefilter_below, void* stream, uint64_t* out_global_count, uint64_t* out_local_count, fj_dist_timings* timings) {
    if (!dc) return derr("fj_dist_join_count: null communicator");
    if (timings && timings->struct_size < offsetof(fj_dist_timings, local)) return derr("fj_dist_join: fj_dist_timings.struct_size is not set (ABI %d: the struct begins with its own size)", FJ_ABI_VERSION);

    if (materialize && (!dc->hip || (nb && !d_build_vals) || ((uintptr_t)d_build_vals & 15))) return derr("fj_dist_join: a materialising join needs the HIP engine and 16-byte aligned build values");
    const bool mat = materialize != 0;
    if (pieces < 0 || pieces > MAX_PIECES) return derr("fj_dist_join_count: pieces must be 0 (the driver decides) or 1..%d", MAX_PIECES);
    if ((nb && !d_build_keys) || (np && !d_probe_keys) || (((uintptr_t)d_build_keys | (uintptr_t)d_probe_keys) & 15)) return derr("fj_dist_join_count: null or misaligned input");
    Net& net = *dc->net; Engine& eng = *dc->eng;
    const int N = net.nranks, me = net.rank;
    if (dc->hip) dc->hip->js = (hipStream_t)stream;
    const auto t0 = std::chrono::steady_clock::now();
    auto ms_since = [](std::chrono::steady_clock::time_point a) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count(); };
    double split_ms = 0;
    // FJ_DIST_TRACE=1: host-side time stamps of this rank's step on stderr (where does the host spend its time between the launches)
    const bool trace = getenv("FJ_DIST_TRACE") != nullptr;
    std::vector<std::pair<const char*, double>> marks;
    auto mark = [&](const char* what) { if (trace) marks.emplace_back(what, ms_since(t0)); };

    // relation sizes of every rank: one plan for everybody
    std::vector<unsigned long long> m((size_t)N * std::max(N + 1, 24));
    // (... and one precheck threshold, one form, one link rate, one piece count: rank 0's - what is exchanged, and in how many rounds,
    //  must not depend on the rank)
    {
        const double pb = prefilter_below > 0 ? std::min(prefilter_below, 4.0) : 0.0;
        const unsigned long long v[7] = {nb, np, (unsigned long long)(pb * 1e9), (unsigned long long)dc->form | ((unsigned long long)pieces << 8),
                                         (unsigned long long)(dc->link_rate > 0 ? dc->link_rate : 0.0),
                                         (unsigned long long)(dc->rs_with_ms * 1e3), (unsigned long long)(dc->rs_without_ms * 1e3)};      // (this rank's measured pass times, us)
        if (net.all_gather(v, 7, m.data())) return 1;
    }
    unsigned long long nb_total = 0, np_global = 0, np_min = ~0ull, nb_max = 0, np_max = 0;
    std::vector<uint64_t> nb_of(N);
    {   // the CU reserve is measured per join shape (sizes are the same on every rank after the all-gather: so is this reset)
        unsigned long long nbt = 0, npt = 0;
        for (int r = 0; r < N; ++r) { nbt += m[7 * r]; npt += m[7 * r + 1]; }
        if (nbt != dc->rs_nb || npt != dc->rs_np) {
            dc->rs_nb = nbt; dc->rs_np = npt; dc->rs_step = 0; dc->rs_with_ms = dc->rs_without_ms = 0.0;
            for (int r = 0; r < N; ++r) m[7 * r + 5] = m[7 * r + 6] = 0;       // (what travelled belongs to the previous shape)
        }
    }
    double tot_with = 0, tot_without = 0;                     // the ranks' measured probe-side pass times, added up (ms): the CU reserve's evidence
    for (int r = 0; r < N; ++r) {
        nb_of[r] = m[7 * r]; nb_total += m[7 * r]; np_global += m[7 * r + 1]; np_min = std::min(np_min, m[7 * r + 1]);
        nb_max = std::max(nb_max, m[7 * r]); np_max = std::max(np_max, m[7 * r + 1]);
        tot_with += (double)m[7 * r + 5] * 1e-3; tot_without += (double)m[7 * r + 6] * 1e-3;
    }
    int reserve_how = 0;
    const unsigned reserve_n = reserve_for_step(dc, net.shares_the_gpu() && (N > 1 || (fj_get_option("lab_hooks") & FJ_HOOK_RESERVE_ALWAYS)), tot_with, tot_without, &reserve_how);
    prefilter_below = (double)m[2] * 1e-9;
    pieces = (int)(m[3] >> 8);
    if (pieces < 0 || pieces > MAX_PIECES) return derr("fj_dist_join_count: rank 0 asks for %d pieces (0..%d)", pieces, MAX_PIECES);
    const bool pieces_auto = pieces == 0;                     // (decided below, from what every rank knows: the same number everywhere)
    if (pieces_auto) pieces = 4;
    const int form_req = (int)(m[3] & 0xFF);
    const double link_rate = m[4] > 0 ? (double)m[4] : 55e9;
    const bool want_pf = prefilter_below > 0;
    if (want_pf && !eng.has_precheck()) return derr("fj_dist_join: this engine has no sender-side precheck");
    // ---- which form (same inputs, hence the same verdict, on every rank) ----
    if (!want_pf && form_req != FJ_DIST_FORM_SHUFFLE && (!mat || form_req == FJ_DIST_FORM_BROADCAST)) {      // (a materialising join broadcasts when asked to: the caller's model, distributed.py)
        bool can = eng.has_bcast() && N <= (int)FJ_WIDE_MAXSRC && nb_total > 0;
        size_t region_max = 0;
        for (int r = 0; can && r < N; ++r) { const size_t b = eng.bc_region_bytes(nb_total, (size_t)nb_of[r], mat); can = b != 0; region_max = std::max(region_max, b); }
        bool bcast = can && form_req == FJ_DIST_FORM_BROADCAST;
        if (can && form_req == FJ_DIST_FORM_AUTO && N > 1)
            bcast = fj_dist_model(N, nb_max, np_max, nb_total, np_global, region_max, link_rate, nullptr, nullptr) == FJ_DIST_FORM_BROADCAST;
        if (form_req == FJ_DIST_FORM_BROADCAST && !can)
            return derr("fj_dist_join_count: the build-broadcast form needs an engine that has it, <= %u ranks and a total build side with a partitioned plan (%llu rows)", FJ_WIDE_MAXSRC, nb_total);
        if (bcast && pieces_auto && !mat) pieces = auto_pieces(true, N, nb_max, np_max, nb_total, region_max, link_rate);
        if (bcast) return dist_join_bcast(dc, d_build_keys, mat ? d_build_vals : nullptr, mat, nb, d_probe_keys, np, pieces, nb_of, nb_total, out_global_count, out_local_count, timings, reserve_n, reserve_how, tot_with, tot_without);
    }
    size_t CB = 0;
    if (eng.plan(nb_total, N, &CB)) return 1;                 // (same verdict on every rank: same arguments)
    if (np_min < 2ull * pieces) pieces = 1;
    net.begin_step();
    const bool loop = net.loopback();
    const bool reserve = reserve_n > 0;
    struct ReserveGuard { Engine& e; bool on; ~ReserveGuard() { if (on) e.reserve(0); } } reserve_guard{eng, reserve};
    if (reserve) eng.reserve(reserve_n);

    std::string failed;                                       // this rank's first local join failure: later engine calls are skipped, collectives go on
    auto guarded = [&](int rc) { if (rc && failed.empty()) failed = fj_last_error(); return rc; };
    bool opened = false;
    // every exit after the first enqueue drains this rank's streams and drops an open stream join
    auto bail = [&](const std::string& why) { (void)net.drain(); (void)eng.drain(); if (opened) eng.abort(); return derr("%s", why.c_str()); };

    struct Piece { void* rk = nullptr; uint32_t* rd = nullptr; uint64_t* rv = nullptr; size_t chunks = 0; Token done = nullptr; };
    std::vector<unsigned long long> used(N), recv_n(N);
    size_t sent_chunks = 0;                                   // chunks this rank put on the links (all pieces)
    Token pool_free[2] = {nullptr, nullptr};                  // the exchange that last read send pool [slot]
    // counts -> agreement -> buffers -> agreement -> copy into the wire format -> exchange   (pack_begin has been issued)
    auto finish_piece = [&](bool begun_ok, int pslot, int rslot, bool vals, Piece* out) -> int {
        const auto tp = std::chrono::steady_clock::now();
        mark("finish_piece");
        bool ok = begun_ok && eng.pack_counts(used.data()) == 0;
        const std::string why = ok ? "" : fj_last_error();
        split_ms += ms_since(tp);
        mark(" counts known");
        std::vector<unsigned long long> v(N + 1);
        for (int r = 0; r < N; ++r) v[r] = ok ? used[r] : 0;
        v[N] = ok ? 0 : FAIL;
        if (net.all_gather(v.data(), N + 1, m.data())) return bail(fj_last_error());
        mark(" all-gathered");
        int nfail = 0;
        size_t total = 0, out_chunks = 0, largest = 0;
        for (int q = 0; q < N; ++q) {
            if (m[(size_t)q * (N + 1) + N] >= FAIL) ++nfail;
            recv_n[q] = m[(size_t)q * (N + 1) + me];
            total += (size_t)recv_n[q];
            for (int d = 0; d < N; ++d) if (m[(size_t)q * (N + 1) + d] < FAIL) largest = std::max<size_t>(largest, (size_t)m[(size_t)q * (N + 1) + d] * CB);
        }
        if (nfail) { char b[1200]; snprintf(b, sizeof b, "fj_dist_join_count: packing a piece failed on %d rank(s)%s%s", nfail, ok ? "" : "; this rank: ", why.c_str()); return bail(b); }
        for (int d = 0; d < N; ++d) if (d != me || loop) out_chunks += (size_t)used[d];
        sent_chunks += out_chunks;
        unsigned long long bad = (dc->grow(dc->recv_k[rslot], total * CB) || dc->grow(dc->recv_d[rslot], total * 4) ||
                                  dc->grow(dc->pool_k[pslot], out_chunks * CB) || dc->grow(dc->pool_d[pslot], out_chunks * 4) ||
                                  (vals && (dc->grow(dc->recv_v, total * FJ_CHUNK * 8) || dc->grow(dc->pool_v, out_chunks * FJ_CHUNK * 8)))) ? 1 : 0;
        const std::string why2 = bad ? fj_last_error() : "";
        const bool mine_bad = bad != 0;
        mark(" buffers");
        if (net.all_reduce(&bad, 1)) return bail(fj_last_error());
        mark(" agreed");
        if (bad) { char b[1200]; snprintf(b, sizeof b, "fj_dist_join_count: buffers for a piece could not be allocated on %llu rank(s)%s%s", bad, mine_bad ? "; this rank: " : "", why2.c_str()); return bail(b); }
        // where every owner's share goes: this rank's own straight into its receive buffer, the others' into the send pool
        const int nparts = vals ? 3 : 2;                       // chunks, directory words, values
        std::vector<void*> dk(N), sp(3 * N, nullptr), rp(3 * N, nullptr);
        std::vector<uint32_t*> dd(N);
        std::vector<uint64_t*> dv(N, nullptr);
        std::vector<size_t> sb(3 * N, 0), rb(3 * N, 0);
        std::vector<size_t> roff(N + 1, 0);
        for (int q = 0; q < N; ++q) roff[q + 1] = roff[q] + (size_t)recv_n[q];
        char* rk = (char*)dc->recv_k[rslot].p; uint32_t* rd = (uint32_t*)dc->recv_d[rslot].p;
        uint64_t* rv = vals ? (uint64_t*)dc->recv_v.p : nullptr;
        size_t po = 0;
        for (int d = 0; d < N; ++d) {
            if (d == me && !loop) { dk[d] = rk + roff[me] * CB; dd[d] = rd + roff[me]; if (vals) dv[d] = rv + roff[me] * FJ_CHUNK; }
            else {
                dk[d] = (char*)dc->pool_k[pslot].p + po * CB; dd[d] = (uint32_t*)dc->pool_d[pslot].p + po;
                sp[d] = dk[d]; sb[d] = (size_t)used[d] * CB; sp[N + d] = dd[d]; sb[N + d] = (size_t)used[d] * 4;
                if (vals) { dv[d] = (uint64_t*)dc->pool_v.p + po * FJ_CHUNK; sp[2 * N + d] = dv[d]; sb[2 * N + d] = (size_t)used[d] * FJ_CHUNK * 8; }
                po += (size_t)used[d];
            }
            rp[d] = rk + roff[d] * CB; rp[N + d] = rd + roff[d];
            if (vals) rp[2 * N + d] = rv + roff[d] * FJ_CHUNK;
            if (d != me || loop) { rb[d] = (size_t)recv_n[d] * CB; rb[N + d] = (size_t)recv_n[d] * 4; if (vals) rb[2 * N + d] = (size_t)recv_n[d] * FJ_CHUNK * 8; }
        }
        if (vals) largest = largest / CB * (FJ_CHUNK * 8);     // (the value part of a message is its largest: 2048 bytes per chunk)
        Token packed = nullptr, done = nullptr;
        if (eng.pack_finish(dk.data(), vals ? dv.data() : nullptr, dd.data(), pool_free[pslot], &packed)) {
            // behind the agreement points: the peers post this piece's receives whatever happens here.  The rank stays in step - its
            // shares travel with directory words that say "no chunk", its later engine calls are skipped - and the failure is agreed
            // on in the final all-reduce like any failure of a local join
            guarded(1);
            eng.mark_unusable(dd.data(), used.data(), N);
            packed = nullptr;
        }
        mark(" copy enqueued");
        if (net.exchange(nparts, sp.data(), sb.data(), rp.data(), rb.data(), largest, packed, &done, rslot)) return bail(fj_last_error());
        mark(" exchange enqueued");
        pool_free[pslot] = done;
        out->rk = rk; out->rd = rd; out->rv = rv; out->chunks = total; out->done = done ? done : packed;
        return 0;
    };

    // ---- build side: one piece ----
    Piece B;
    {
        const auto tp = std::chrono::steady_clock::now();
        const bool ok = eng.pack_begin(d_build_keys, mat ? d_build_vals : nullptr, nb, nb_total, N, nullptr, nullptr) == 0;
        split_ms += ms_since(tp);
        if (finish_piece(ok, 0, 0, mat, &B)) return 1;
    }
    const size_t np_bound = (size_t)(1.5 * (double)np_global / N) + ((size_t)1 << 22) + (size_t)FJ_CHUNK * 512 * N * pieces;
    // ---- sender-side precheck: the owners' partition filters, all-gathered; a sample decides whether they are used ----
    bool pf = false;
    double sampled = -1.0;
    Token filt_ready = nullptr;
    size_t filter_bytes = 0;
    unsigned long long kept_rows = 0;
    if (want_pf) {
        opened = guarded(eng.open(nb_total, N, me, B.chunks * FJ_CHUNK, np_bound, pieces, mat)) == 0;
        if (failed.empty()) guarded(eng.append(0, B.rk, B.rv, B.rd, B.chunks, B.done));
        std::vector<size_t> first(N), cnt(N);
        size_t total = 0, FB = 0, largest = 0;
        for (int r = 0; r < N && failed.empty(); ++r) guarded(eng.filter_range(nb_total, N, r, &first[r], &cnt[r], &total, &FB));
        if (failed.empty()) guarded(dc->grow(dc->filt, total * FB));
        Token exported = nullptr;
        if (failed.empty()) guarded(eng.export_filters((char*)dc->filt.p + first[me] * FB, &exported));
        unsigned long long bad = failed.empty() ? 0 : 1;
        if (net.all_reduce(&bad, 1)) return bail(fj_last_error());
        if (bad) { char b[1200]; snprintf(b, sizeof b, "fj_dist_join: the partition filters could not be prepared on %llu rank(s)%s%s", bad, failed.empty() ? "" : "; this rank: ", failed.c_str()); return bail(b); }
        std::vector<const void*> sp(N, nullptr); std::vector<void*> rp(N, nullptr);
        std::vector<size_t> sb(N, 0), rb(N, 0);
        for (int r = 0; r < N; ++r) {
            largest = std::max(largest, cnt[r] * FB);
            if (r == me) continue;                             // (a rank's own filters are where they belong)
            sp[r] = (char*)dc->filt.p + first[me] * FB; sb[r] = cnt[me] * FB;
            rp[r] = (char*)dc->filt.p + first[r] * FB; rb[r] = cnt[r] * FB;
            filter_bytes += rb[r];
        }
        if (net.exchange(1, sp.data(), sb.data(), rp.data(), rb.data(), largest, exported, &filt_ready, MAX_PIECES + 1)) return bail(fj_last_error());
        if (!filt_ready) filt_ready = exported;
        mark("filters on their way");
        if (prefilter_below >= 2.0) pf = true;
        else {
            const size_t ns = std::min<size_t>(np, (size_t)1 << 20), stride = ns ? np / ns : 1;
            unsigned long long k = 0;
            const bool ok = eng.sample(d_probe_keys, ns, stride, dc->filt.p, nb_total, N, filt_ready, &k) == 0;
            const std::string why = ok ? "" : fj_last_error();
            unsigned long long v[3] = {k, ns, ok ? 0ull : 1ull};
            if (net.all_reduce(v, 3)) return bail(fj_last_error());
            if (v[2]) { char b[1200]; snprintf(b, sizeof b, "fj_dist_join: sampling the probe rows failed on %llu rank(s)%s%s", v[2], ok ? "" : "; this rank: ", why.c_str()); return bail(b); }
            sampled = v[1] ? (double)v[0] / (double)v[1] : 1.0;
            pf = sampled < prefilter_below;
            mark("sampled");
        }
    }
    const void* filters = pf ? dc->filt.p : nullptr;
    // ---- probe side: piece c+1's first pass is queued behind piece c's copy; piece c is appended before the host waits for c+1 ----
    std::vector<Piece> P(pieces);
    auto bounds = [&](int c, size_t* lo, size_t* hi) { *lo = (np * (size_t)c / pieces) & ~(size_t)1; *hi = c + 1 == pieces ? np : ((np * (size_t)(c + 1) / pieces) & ~(size_t)1); };
    size_t lo, hi;
    bounds(0, &lo, &hi);
    auto tp0 = std::chrono::steady_clock::now();
    bool begun = eng.pack_begin(d_probe_keys + lo, nullptr, hi - lo, nb_total, N, filters, filt_ready) == 0;
    split_ms += ms_since(tp0);
    size_t rows_recv_chunks = 0;
    for (int c = 0; c < pieces; ++c) {
        if (finish_piece(begun, c & 1, c + 1, false, &P[c])) return 1;
        kept_rows += pf ? eng.pack_kept() : (unsigned long long)(hi - lo);
        if (c + 1 < pieces) {
            bounds(c + 1, &lo, &hi);
            const auto tp = std::chrono::steady_clock::now();
            begun = eng.pack_begin(d_probe_keys + lo, nullptr, hi - lo, nb_total, N, filters, filt_ready) == 0;
            split_ms += ms_since(tp);
        }
        if (c == 0 && !want_pf) {
            // the owner's stream join opens once the first probe piece's size is known: an owner of hot probe keys (its share far above
            // np_global / N) sizes its pools from what actually arrives - 1.25x the first piece's rate - instead of failing later
            const size_t seen = (size_t)(1.25 * (double)P[0].chunks * FJ_CHUNK * pieces) + ((size_t)1 << 20);
            opened = guarded(eng.open(nb_total, N, me, B.chunks * FJ_CHUNK, std::max(np_bound, seen), pieces, mat)) == 0;
            if (failed.empty()) guarded(eng.append(0, B.rk, B.rv, B.rd, B.chunks, B.done));
        }
        mark(" next pass enqueued");
        rows_recv_chunks += P[c].chunks;
        if (failed.empty()) guarded(eng.append(1, P[c].rk, nullptr, P[c].rd, P[c].chunks, P[c].done));
        mark(" append enqueued");
    }
    const auto t2 = std::chrono::steady_clock::now();
    uint64_t local = 0;
    fj_timings lt; memset(&lt, 0, sizeof lt);
    mark("finish");
    if (failed.empty()) { guarded(eng.finish(&local, &lt)); if (failed.empty()) opened = false; }
    mark(" finished");
    if (!failed.empty()) { local = 0; if (opened) eng.abort(); opened = false; }

    // ---- global count + agreement ----
    unsigned long long res[2] = {local, failed.empty() ? 0ull : 1ull};
    if (net.all_reduce(res, 2)) return bail(fj_last_error());
    if (net.drain() || eng.drain()) return 1;
    mark("done");
    if (trace) { for (auto& mk : marks) fprintf(stderr, "[fj_dist rank %d] %9.3f ms  %s\n", me, mk.second, mk.first); }
    if (res[1]) return derr("fj_dist_join_count: the local join failed on %llu rank(s)%s%s", res[1], failed.empty() ? "" : "; this rank: ", failed.c_str());
    if (out_global_count) *out_global_count = res[0];
    if (out_local_count) *out_local_count = local;          // materialising: fj_emit_pairs(ctx, ...) then writes this rank's `local` pairs (they stay with the owner)
    reserve_account(dc, lt, reserve_how);
    if (timings) {
        fj_dist_timings T; memset(&T, 0, sizeof T);
        T.reserve_cus = (int)reserve_n; T.reserve_how = reserve_how; T.reserve_with_ms = tot_with; T.reserve_without_ms = tot_without;
        T.total_ms = ms_since(t0); T.split_ms = split_ms; T.join_ms = ms_since(t2);
        T.exchange_ms = std::max(0.0, T.total_ms - T.join_ms - split_ms);
        T.local_count = local; T.local_build_chunks = B.chunks; T.local_probe_chunks = rows_recv_chunks;
        T.pieces = pieces; T.nranks = N; T.fan_log0 = 0; T.local = lt;
        T.wire_chunk_bytes = (int)CB; T.sent_chunks = sent_chunks;
        T.form = FJ_DIST_FORM_SHUFFLE; T.wire_bytes_sent = (uint64_t)sent_chunks * (CB + 4);
        T.prefilter = pf ? 1 : 0; T.prefilter_sampled = sampled; T.probe_rows_kept = kept_rows; T.filter_bytes = filter_bytes;
        write_timings(timings, T);
    }
    return 0;
}

}  // extern "C"
