// fj_dist.hip -- native multi-GPU entry behind the C ABI: the counting radix join of relations that are block-distributed
// over the ranks of an RCCL communicator (BASELINE configs[4]: 1B x 10B rows over 8 MI355X).
//
// No reference counterpart: the reference is one process (hash_join.cpp:318).  What is exploited is that radix partitions
// are independent join units (hash_join.cpp:340-356, :515-525): the first radix pass of the plan for the TOTAL build side is
// the owner split, the exchange is an all-to-all of whole 2-KiB chunks over xGMI, and every owner runs the
// rest of the plan on what it received.  This file is the protocol of flash_hash_join_amd/distributed.py
// (_chunk_shuffle_count) for a host that has nothing but include/flashjoin.h and an ncclComm_t:
//
//   sizes all-gathered -> plan -> build side: fj_shuffle_pack, counts all-gathered, grouped ncclSend / ncclRecv of every
//   owner's region (keys + directory words), fj_stream_open_shuffled + fj_stream_append_build_chunks -> probe side in
//   `pieces` rounds: piece c is packed (join stream) and put on the wire (exchange stream) while piece c-1 arrives and gets
//   its second radix pass -> fj_stream_finish -> ncclAllReduce of (count, failure flag).
//
// A failure of one rank's local work never takes it out of step: the rank keeps taking part in the collectives, and the
// ranks agree on the outcome in the final all-reduce (a packing failure travels in the counts vector).  RCCL is bound at
// run time (dlopen of librccl.so.1 - in a PyTorch process that is the copy torch already loaded), so the library has no
// link-time dependency on it and single-GPU users never touch it.  RCCL moves wrong data when one point-to-point message
// exceeds 4 GiB (tools/rccl_large_message_check.py): messages are cut into rounds of <= 1 GiB.
#include "fj_internal.h"
#include "../../include/flashjoin.h"

#include <rccl/rccl.h>
#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

void fj_set_error_string(const char* msg);       // fj_api.hip: the thread-local string fj_last_error() returns

namespace {

struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
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
        auto sym = [&](const char* n) -> void* { void* p = dlsym(r.lib, n); if (!p && r.err.empty()) r.err = std::string("librccl has no ") + n; return p; };
        r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
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
#define DNCCL(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) return derr("%s:%d: %s failed: %s", __FILE__, __LINE__, #x, rccl()->GetErrorString ? rccl()->GetErrorString(r_) : "?"); } while (0)

struct DBuf { void* p = nullptr; size_t bytes = 0; };
int grow(DBuf& b, size_t bytes) {
    if (bytes == 0) bytes = 16;
    if (b.bytes >= bytes) return 0;
    if (b.p) { DHIP(hipFree(b.p)); b.p = nullptr; b.bytes = 0; }
    const size_t want = (bytes + 4095) & ~(size_t)4095;
    hipError_t e = hipMalloc(&b.p, want);
    if (e != hipSuccess) return derr("fj_dist: hipMalloc(%zu bytes) failed: %s", want, hipGetErrorString(e));
    b.bytes = want;
    return 0;
}

constexpr int MAX_PIECES = 16;
constexpr size_t MAX_MSG_ELEMS = (size_t)1 << 27;         // 1 GiB of int64 per point-to-point message and round
constexpr unsigned long long FAIL = 1ull << 60;            // in a counts vector: this rank could not pack the piece

}  // namespace

struct fj_dist_comm {
    fj_ctx* ctx = nullptr;
    ncclComm_t nccl = nullptr;
    bool own_comm = false;
    int nranks = 1, rank = 0, device = 0;
    hipStream_t xs = nullptr;                              // exchange stream: the sends / receives of a piece
    hipEvent_t ev_x[MAX_PIECES + 1];                       // piece c's exchange has finished
    unsigned long long* d_cnt = nullptr;                   // scratch words for the small collectives (comm_setup)
    unsigned long long* h_cnt = nullptr;                   // pinned mirror
    DBuf pool_k[2], pool_d[2];                             // what fj_shuffle_pack writes (alternating per piece)
    DBuf recv_k[MAX_PIECES + 1], recv_d[MAX_PIECES + 1];   // what arrives: [0] build side, [1 + c] probe piece c
};

namespace {

// every rank's vector v[0 .. n) -> matrix m[rank][0 .. n) on every rank (device round trip through the comm's scratch words)
int all_gather_u64(fj_dist_comm* dc, const unsigned long long* v, int n, unsigned long long* m, hipStream_t s) {
    Rccl* R = rccl();
    unsigned long long* d_send = dc->d_cnt, *d_recv = dc->d_cnt + 72;
    memcpy(dc->h_cnt, v, sizeof(unsigned long long) * n);
    DHIP(hipMemcpyAsync(d_send, dc->h_cnt, sizeof(unsigned long long) * n, hipMemcpyHostToDevice, s));
    DNCCL(R->AllGather(d_send, d_recv, (size_t)n, ncclUint64, dc->nccl, s));
    DHIP(hipMemcpyAsync(dc->h_cnt + 72, d_recv, sizeof(unsigned long long) * n * dc->nranks, hipMemcpyDeviceToHost, s));
    DHIP(hipStreamSynchronize(s));
    memcpy(m, dc->h_cnt + 72, sizeof(unsigned long long) * n * dc->nranks);
    return 0;
}

// the sends and receives of one packed piece, grouped, on the exchange stream: owner r gets chunks [r * region, r * region + used[r])
// of the pool; what source q sends lands at chunk offset roff[q] of the receive buffers
int exchange_piece(fj_dist_comm* dc, const u64* pool_k, const u32* pool_d, size_t region, const unsigned long long* used,
                   u64* recv_k, u32* recv_d, const unsigned long long* recv_n, size_t largest_chunks) {
    Rccl* R = rccl();
    const int N = dc->nranks;
    const size_t rounds = std::max<size_t>(1, (largest_chunks * FJ_CHUNK + MAX_MSG_ELEMS - 1) / MAX_MSG_ELEMS);
    std::vector<size_t> roff(N + 1, 0);
    for (int q = 0; q < N; ++q) roff[q + 1] = roff[q] + (size_t)recv_n[q];
    // this rank's own region never touches the network: a device-to-device copy on the exchange stream (RCCL would move it
    // through its channel kernels at a fraction of the copy rate)
    if (used[dc->rank]) {
        DHIP(hipMemcpyAsync(recv_k + roff[dc->rank] * FJ_CHUNK, pool_k + (size_t)dc->rank * region * FJ_CHUNK, (size_t)used[dc->rank] * FJ_CHUNK * 8, hipMemcpyDeviceToDevice, dc->xs));
        DHIP(hipMemcpyAsync(recv_d + roff[dc->rank], pool_d + (size_t)dc->rank * region, (size_t)used[dc->rank] * 4, hipMemcpyDeviceToDevice, dc->xs));
    }
    if (N == 1) return 0;
    for (size_t r = 0; r < rounds; ++r) {
        DNCCL(R->GroupStart());
        for (int d = 0; d < N; ++d) {
            const size_t lo = (size_t)used[d] * r / rounds, hi = (size_t)used[d] * (r + 1) / rounds;
            if (hi > lo && d != dc->rank) {
                DNCCL(R->Send(pool_k + ((size_t)d * region + lo) * FJ_CHUNK, (hi - lo) * FJ_CHUNK, ncclUint64, d, dc->nccl, dc->xs));
                DNCCL(R->Send(pool_d + (size_t)d * region + lo, hi - lo, ncclUint32, d, dc->nccl, dc->xs));
            }
        }
        for (int q = 0; q < N; ++q) {
            const size_t lo = (size_t)recv_n[q] * r / rounds, hi = (size_t)recv_n[q] * (r + 1) / rounds;
            if (hi > lo && q != dc->rank) {
                DNCCL(R->Recv(recv_k + (roff[q] + lo) * FJ_CHUNK, (hi - lo) * FJ_CHUNK, ncclUint64, q, dc->nccl, dc->xs));
                DNCCL(R->Recv(recv_d + roff[q] + lo, hi - lo, ncclUint32, q, dc->nccl, dc->xs));
            }
        }
        DNCCL(R->GroupEnd());
    }
    return 0;
}

}  // namespace

extern "C" {

int fj_dist_unique_id(char* out128) {
    Rccl* R = rccl();
    if (!R->err.empty()) return derr("fj_dist_unique_id: %s", R->err.c_str());
    if (!out128) return derr("fj_dist_unique_id: null buffer");
    ncclUniqueId id;
    DNCCL(R->GetUniqueId(&id));
    static_assert(sizeof id == 128, "ncclUniqueId is 128 bytes");
    memcpy(out128, &id, sizeof id);
    return 0;
}

static int comm_setup(fj_dist_comm* dc) {
    DHIP(hipGetDevice(&dc->device));
    DHIP(hipStreamCreateWithFlags(&dc->xs, hipStreamNonBlocking));
    for (auto& e : dc->ev_x) DHIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    const size_t words = 72 + (size_t)64 * 65 + 16;          // send vector (<= 65 words) | gathered matrix (<= 64 x 65) | result words
    DHIP(hipMalloc((void**)&dc->d_cnt, words * 8));
    DHIP(hipHostMalloc((void**)&dc->h_cnt, words * 8, hipHostMallocDefault));
    return 0;
}

fj_dist_comm* fj_dist_comm_create(fj_ctx* ctx, const char* unique_id128, int nranks, int rank) {
    Rccl* R = rccl();
    if (!R->err.empty()) { derr("fj_dist_comm_create: %s", R->err.c_str()); return nullptr; }
    if (!ctx || !unique_id128 || nranks < 1 || nranks > 64 || rank < 0 || rank >= nranks) { derr("fj_dist_comm_create: bad arguments (1..64 ranks)"); return nullptr; }
    fj_dist_comm* dc = new fj_dist_comm();
    dc->ctx = ctx; dc->nranks = nranks; dc->rank = rank; dc->own_comm = true;
    ncclUniqueId id; memcpy(&id, unique_id128, sizeof id);
    ncclResult_t r = R->CommInitRank(&dc->nccl, nranks, id, rank);       // (the caller has selected this rank's device)
    if (r != ncclSuccess) { derr("fj_dist_comm_create: ncclCommInitRank failed: %s", R->GetErrorString(r)); delete dc; return nullptr; }
    if (comm_setup(dc)) { fj_dist_comm_destroy(dc); return nullptr; }
    return dc;
}

fj_dist_comm* fj_dist_comm_from_nccl(fj_ctx* ctx, void* nccl_comm) {
    Rccl* R = rccl();
    if (!R->err.empty()) { derr("fj_dist_comm_from_nccl: %s", R->err.c_str()); return nullptr; }
    if (!ctx || !nccl_comm) { derr("fj_dist_comm_from_nccl: null argument"); return nullptr; }
    fj_dist_comm* dc = new fj_dist_comm();
    dc->ctx = ctx; dc->nccl = (ncclComm_t)nccl_comm; dc->own_comm = false;
    if (R->CommCount(dc->nccl, &dc->nranks) != ncclSuccess || R->CommUserRank(dc->nccl, &dc->rank) != ncclSuccess || dc->nranks > 64) {
        derr("fj_dist_comm_from_nccl: not a usable communicator (1..64 ranks)"); delete dc; return nullptr;
    }
    if (comm_setup(dc)) { fj_dist_comm_destroy(dc); return nullptr; }
    return dc;
}

void fj_dist_comm_destroy(fj_dist_comm* dc) {
    if (!dc) return;
    (void)hipDeviceSynchronize();
    for (auto* arr : {dc->pool_k, dc->pool_d}) for (int i = 0; i < 2; ++i) if (arr[i].p) (void)hipFree(arr[i].p);
    for (auto* arr : {dc->recv_k, dc->recv_d}) for (int i = 0; i <= MAX_PIECES; ++i) if (arr[i].p) (void)hipFree(arr[i].p);
    if (dc->d_cnt) (void)hipFree(dc->d_cnt);
    if (dc->h_cnt) (void)hipHostFree(dc->h_cnt);
    for (auto& e : dc->ev_x) if (e) (void)hipEventDestroy(e);
    if (dc->xs) (void)hipStreamDestroy(dc->xs);
    if (dc->own_comm && dc->nccl && rccl()->CommDestroy) (void)rccl()->CommDestroy(dc->nccl);
    delete dc;
}

int fj_dist_comm_rank(const fj_dist_comm* dc) { return dc ? dc->rank : -1; }
int fj_dist_comm_size(const fj_dist_comm* dc) { return dc ? dc->nranks : 0; }

int fj_dist_join_count(fj_dist_comm* dc, const uint64_t* d_build_keys, size_t nb, const uint64_t* d_probe_keys, size_t np, int pieces,
                       void* stream, uint64_t* out_global_count, fj_dist_timings* timings) {
    if (!dc) return derr("fj_dist_join_count: null communicator");
    if (pieces < 1 || pieces > MAX_PIECES) return derr("fj_dist_join_count: pieces must be 1..%d", MAX_PIECES);
    if ((nb && !d_build_keys) || (np && !d_probe_keys) || (((uintptr_t)d_build_keys | (uintptr_t)d_probe_keys) & 15)) return derr("fj_dist_join_count: null or misaligned input");
    Rccl* R = rccl();
    const int N = dc->nranks, me = dc->rank;
    hipStream_t s = (hipStream_t)stream;
    const auto t0 = std::chrono::steady_clock::now();
    auto ms_since = [](std::chrono::steady_clock::time_point a) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count(); };
    double split_ms = 0;

    // relation sizes of every rank: one plan for everybody
    std::vector<unsigned long long> m((size_t)N * std::max(N + 1, 2));
    { const unsigned long long v[2] = {nb, np}; if (all_gather_u64(dc, v, 2, m.data(), s)) return 1; }
    unsigned long long nb_total = 0, np_global = 0, np_min = ~0ull;
    for (int r = 0; r < N; ++r) { nb_total += m[2 * r]; np_global += m[2 * r + 1]; np_min = std::min(np_min, m[2 * r + 1]); }
    int fan_log0 = 0, npass = 0;
    if (fj_shuffle_plan(nb_total, N, &fan_log0, &npass)) return 1;          // (same verdict on every rank: same arguments)
    if (np_min < 2ull * pieces) pieces = 1;

    std::string failed;                                // this rank's first local failure: later engine calls are skipped, collectives go on
    auto guarded = [&](int rc) { if (rc && failed.empty()) failed = fj_last_error(); return rc; };

    // pack one piece, agree on its counts; *largest = the largest message of the piece anywhere, in chunks
    std::vector<unsigned long long> used(N), recv_n(N);
    auto pack_and_count = [&](const u64* rows, size_t n, int slot, size_t* region, size_t* largest) -> int {
        const auto tp = std::chrono::steady_clock::now();
        *region = fj_shuffle_region_chunks(n, nb_total, N, 0);
        bool ok = *region != 0 && !grow(dc->pool_k[slot], (size_t)N * *region * FJ_CHUNK * 8) && !grow(dc->pool_d[slot], (size_t)N * *region * 4);
        if (ok) ok = fj_shuffle_pack(dc->ctx, rows, nullptr, n, nb_total, N, (uint64_t*)dc->pool_k[slot].p, nullptr, (uint32_t*)dc->pool_d[slot].p, *region,
                                     (uint64_t*)used.data(), s) == 0;
        const std::string why = ok ? "" : fj_last_error();
        split_ms += ms_since(tp);
        std::vector<unsigned long long> v(N + 1);
        for (int r = 0; r < N; ++r) v[r] = ok ? used[r] : 0;
        v[N] = ok ? 0 : FAIL;
        if (all_gather_u64(dc, v.data(), N + 1, m.data(), s)) return 1;
        *largest = 0;
        int nfail = 0;
        for (int q = 0; q < N; ++q) {
            if (m[(size_t)q * (N + 1) + N] >= FAIL) ++nfail;
            recv_n[q] = m[(size_t)q * (N + 1) + me];
            for (int d = 0; d < N; ++d) *largest = std::max<size_t>(*largest, (size_t)m[(size_t)q * (N + 1) + d]);
        }
        if (nfail) return derr("fj_dist_join_count: packing a piece failed on %d rank(s)%s%s", nfail, ok ? "" : "; this rank: ", why.c_str());   // every rank returns here
        return 0;
    };
    auto recv_total = [&]() { size_t t = 0; for (int q = 0; q < N; ++q) t += (size_t)recv_n[q]; return t; };

    // ---- build side: one piece ----
    size_t region = 0, largest = 0;
    if (pack_and_count((const u64*)d_build_keys, nb, 0, &region, &largest)) return 1;
    const size_t nbc = recv_total();
    if (grow(dc->recv_k[0], nbc * FJ_CHUNK * 8) || grow(dc->recv_d[0], nbc * 4)) return 1;
    if (exchange_piece(dc, (const u64*)dc->pool_k[0].p, (const u32*)dc->pool_d[0].p, region, used.data(), (u64*)dc->recv_k[0].p, (u32*)dc->recv_d[0].p,
                       recv_n.data(), largest)) return 1;
    DHIP(hipEventRecord(dc->ev_x[0], dc->xs));
    DHIP(hipStreamWaitEvent(s, dc->ev_x[0], 0));
    const size_t np_bound = (size_t)(1.5 * (double)np_global / N) + ((size_t)1 << 22) + (size_t)2 * FJ_CHUNK * 512 * N * pieces;
    guarded(fj_stream_open_shuffled(dc->ctx, nb_total, N, me, nbc * FJ_CHUNK, 1, np_bound, pieces, s));
    if (failed.empty()) guarded(fj_stream_append_build_chunks(dc->ctx, (const uint64_t*)dc->recv_k[0].p, (uint32_t*)dc->recv_d[0].p, nbc, s));

    // ---- probe side: piece c is packed and put on the wire while piece c-1 arrives and gets its second pass ----
    size_t chunks_in[MAX_PIECES + 1] = {0};
    size_t rows_recv_chunks = 0;
    int rc = 0;
    for (int c = 0; c <= pieces && !rc; ++c) {
        if (c < pieces) {
            const size_t lo = (np * (size_t)c / pieces) & ~(size_t)1, hi = c + 1 == pieces ? np : ((np * (size_t)(c + 1) / pieces) & ~(size_t)1);
            const int slot = c & 1;
            if (c >= 2) DHIP(hipStreamWaitEvent(s, dc->ev_x[c - 1], 0));      // the pool this piece is packed into was last read by piece c-2's sends
            if ((rc = pack_and_count((const u64*)d_probe_keys + lo, hi - lo, slot, &region, &largest))) break;
            chunks_in[c + 1] = recv_total();
            if ((rc = grow(dc->recv_k[c + 1], chunks_in[c + 1] * FJ_CHUNK * 8) || grow(dc->recv_d[c + 1], chunks_in[c + 1] * 4))) break;
            if ((rc = exchange_piece(dc, (const u64*)dc->pool_k[slot].p, (const u32*)dc->pool_d[slot].p, region, used.data(), (u64*)dc->recv_k[c + 1].p,
                                     (u32*)dc->recv_d[c + 1].p, recv_n.data(), largest))) break;
            DHIP(hipEventRecord(dc->ev_x[c + 1], dc->xs));
        }
        if (c >= 1) {
            DHIP(hipStreamWaitEvent(s, dc->ev_x[c], 0));
            rows_recv_chunks += chunks_in[c];
            if (failed.empty()) guarded(fj_stream_append_probe_chunks(dc->ctx, (const uint64_t*)dc->recv_k[c].p, (uint32_t*)dc->recv_d[c].p, chunks_in[c], s));
        }
    }
    if (rc) {                                          // a failure every rank has seen at the same point (or one nobody recovers from)
        const std::string why = fj_last_error();
        (void)hipStreamSynchronize(dc->xs);
        (void)fj_stream_abort(dc->ctx);
        return derr("%s", why.c_str());
    }
    const auto t2 = std::chrono::steady_clock::now();
    uint64_t local = 0;
    fj_timings lt; memset(&lt, 0, sizeof lt);
    if (failed.empty()) guarded(fj_stream_finish(dc->ctx, s, &local, &lt));
    if (!failed.empty()) { local = 0; (void)fj_stream_abort(dc->ctx); }

    // ---- global count + agreement ----
    unsigned long long* d_res = dc->d_cnt + 72 + 64 * 65;
    dc->h_cnt[0] = local; dc->h_cnt[1] = failed.empty() ? 0 : 1;
    DHIP(hipMemcpyAsync(d_res, dc->h_cnt, 16, hipMemcpyHostToDevice, s));
    DNCCL(R->AllReduce(d_res, d_res + 2, 2, ncclUint64, ncclSum, dc->nccl, s));
    DHIP(hipMemcpyAsync(dc->h_cnt + 2, d_res + 2, 16, hipMemcpyDeviceToHost, s));
    DHIP(hipStreamSynchronize(s));
    DHIP(hipStreamSynchronize(dc->xs));
    if (dc->h_cnt[3]) return derr("fj_dist_join_count: the local join failed on %llu rank(s)%s%s", dc->h_cnt[3], failed.empty() ? "" : "; this rank: ", failed.c_str());
    if (out_global_count) *out_global_count = dc->h_cnt[2];
    if (timings) {
        memset(timings, 0, sizeof *timings);
        timings->total_ms = ms_since(t0); timings->split_ms = split_ms; timings->join_ms = ms_since(t2);
        timings->exchange_ms = std::max(0.0, timings->total_ms - timings->join_ms - split_ms);
        timings->local_count = local; timings->local_build_chunks = nbc; timings->local_probe_chunks = rows_recv_chunks;
        timings->pieces = pieces; timings->nranks = N; timings->fan_log0 = fan_log0; timings->local = lt;
    }
    return 0;
}

}  // extern "C"
