// fj_hostentry.hip -- the host-buffer (NumPy) entry fj_join_host: pageable memory -> pinned staging ring -> HBM, pipelined
// with the join's first pass.  Replaces all twelve pybind entry points hash_join.cpp:603-637 for host arrays.
// (Split out of fj_api.hip in round 4; see fj_host.h for the map.)
#include "fj_host.h"

#include <atomic>
#include <memory>
#include <pthread.h>
#include <sched.h>
#include <sys/syscall.h>
#include <unistd.h>
using namespace fjh;

namespace fjh { fj_ctx*& host_ctx() { static fj_ctx* c = nullptr; return c; } }

namespace {

// ---- where the copy threads run ---------------------------------------------------------------------------------------------
// A box of this pool has two sockets; a thread that reads the caller's array from the other socket's memory copies at a
// fraction of the local rate, and round 3's unbound threads landed wherever the scheduler put them (the same entry took 16.6 ms
// on one box and 23.0 on another).  FJ_HOST_COPY_BIND: 2 (default) = every piece is copied by threads bound to the NUMA node
// that HOLDS the piece's source pages (move_pages query of the piece's first page; one thread group per node, created on
// demand), 1 = threads bound to the GPU's node, 0 = unbound.
std::vector<int> parse_cpulist(const std::string& s) {
    std::vector<int> out;
    size_t i = 0;
    while (i < s.size()) {
        size_t j = s.find(',', i); if (j == std::string::npos) j = s.size();
        const std::string tok = s.substr(i, j - i);
        const size_t d = tok.find('-');
        const int a = atoi(tok.c_str()), b = d == std::string::npos ? a : atoi(tok.c_str() + d + 1);
        for (int c = a; c <= b && tok.size(); ++c) out.push_back(c);
        i = j + 1;
    }
    return out;
}
std::vector<int> node_cpus(int node) {                       // CPUs of a NUMA node that this process may run on
    std::vector<int> out;
    char path[128]; snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
    FILE* f = fopen(path, "r");
    if (!f) return out;
    char buf[4096] = {0};
    if (fgets(buf, sizeof buf, f)) {
        std::string s(buf); while (!s.empty() && (s.back() == '\n' || s.back() == ' ')) s.pop_back();
        cpu_set_t aff; CPU_ZERO(&aff);
        const bool have = sched_getaffinity(0, sizeof aff, &aff) == 0;
        for (int c : parse_cpulist(s)) if (!have || (c < CPU_SETSIZE && CPU_ISSET(c, &aff))) out.push_back(c);
    }
    fclose(f);
    return out;
}
int node_of_address(const void* p) {                         // NUMA node of the page that holds p (-1: unknown)
    void* page = (void*)((uintptr_t)p & ~(uintptr_t)4095);
    int status = -1;
    if (syscall(SYS_move_pages, 0, 1ul, &page, nullptr, &status, 0) != 0) return -1;
    return status;
}
int node_of_device(int device) {
    char bdf[64] = {0};
    if (hipDeviceGetPCIBusId(bdf, sizeof bdf, device) != hipSuccess) return -1;
    for (char* q = bdf; *q; ++q) *q = (char)tolower(*q);
    char path[160]; snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", bdf);
    FILE* f = fopen(path, "r");
    if (!f) return -1;
    int n = -1; if (fscanf(f, "%d", &n) != 1) n = -1;
    fclose(f);
    return n;
}

// memcpy by a few persistent threads: one core copies pageable -> pinned memory at 10-15 GB/s, PCIe Gen5 x16 moves ~55.  The
// threads spin for the next piece (a condition-variable wake-up per 16-MiB piece cost as much as a fifth of the piece's copy)
// and fall asleep only after ~2 ms without work.
class CopyPool {
    std::vector<std::thread> th_;
    std::mutex m_;
    std::condition_variable work_;
    const char* src_ = nullptr; char* dst_ = nullptr; size_t n_ = 0;
    std::atomic<unsigned> gen_{0}, remaining_{0};
    std::atomic<bool> stop_{false};
    void slice(unsigned id, unsigned parts, size_t* off, size_t* len) const {
        const size_t per = ((n_ / parts) + 4095) & ~(size_t)4095;
        *off = std::min(n_, per * id);
        *len = id + 1 == parts ? n_ - *off : std::min(per, n_ - *off);
    }
    void worker(unsigned id, std::vector<int> cpus) {
        if (!cpus.empty()) {
            cpu_set_t set; CPU_ZERO(&set);
            for (int c : cpus) if (c < CPU_SETSIZE) CPU_SET(c, &set);
            (void)pthread_setaffinity_np(pthread_self(), sizeof set, &set);
        }
        unsigned seen = 0;
        for (;;) {
            const auto t0 = std::chrono::steady_clock::now();
            unsigned spins = 0;
            while (gen_.load(std::memory_order_acquire) == seen && !stop_.load(std::memory_order_relaxed)) {
                if ((++spins & 1023u) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) {
                    std::unique_lock<std::mutex> l(m_);
                    work_.wait(l, [&] { return stop_.load() || gen_.load() != seen; });
                    break;
                }
                __builtin_ia32_pause();
            }
            if (stop_.load()) return;
            seen = gen_.load(std::memory_order_acquire);
            size_t off, len; slice(id + 1, (unsigned)th_.size() + 1, &off, &len);
            if (len) memcpy(dst_ + off, src_ + off, len);
            remaining_.fetch_sub(1, std::memory_order_acq_rel);
        }
    }
public:
    CopyPool(unsigned nthreads, const std::vector<int>& cpus) { for (unsigned i = 0; i + 1 < nthreads; ++i) th_.emplace_back([this, i, cpus] { worker(i, cpus); }); }
    ~CopyPool() { stop_.store(true); { std::lock_guard<std::mutex> l(m_); } work_.notify_all(); for (auto& t : th_) t.join(); }
    unsigned threads() const { return (unsigned)th_.size() + 1; }
    void copy(void* dst, const void* src, size_t n) {
        if (n < (4u << 20) || th_.empty()) { memcpy(dst, src, n); return; }
        src_ = (const char*)src; dst_ = (char*)dst; n_ = n;
        remaining_.store((unsigned)th_.size(), std::memory_order_relaxed);
        gen_.fetch_add(1, std::memory_order_release);
        { std::lock_guard<std::mutex> l(m_); }                 // (a worker between its predicate check and its sleep)
        work_.notify_all();
        size_t off, len; slice(0, (unsigned)th_.size() + 1, &off, &len);
        if (len) memcpy((char*)dst + off, (const char*)src + off, len);
        while (remaining_.load(std::memory_order_acquire) != 0) __builtin_ia32_pause();
    }
};

int copy_bind_mode() { static const int m = getenv("FJ_HOST_COPY_BIND") ? atoi(getenv("FJ_HOST_COPY_BIND")) : 2; return m; }
unsigned copy_threads_for(size_t ncpus) {
    if (const char* e = getenv("FJ_HOST_COPY_THREADS")) return (unsigned)std::max(1, atoi(e));
    // from the CPUs the threads may actually use (the affinity mask, or one node's share of it): a quarter of them, at most 6.  One
    // core copies pageable -> pinned memory at ~14 GB/s and PCIe Gen5 x16 takes ~57: four to six threads keep up, and MORE is worse
    // (profiles/r04_host_entry_probe.txt: 1.11-1.15x a pinned copy with 4-16 threads, 1.37-1.41x with 24-32)
    return (unsigned)std::min<size_t>(6, std::max<size_t>(2, ncpus / 4));
}
// the pool that copies a piece whose source starts at `src`, for the context's device
CopyPool& copy_pool_for(const void* src, int device) {
    static std::mutex mu;
    static std::vector<std::pair<int, std::unique_ptr<CopyPool>>> pools;      // (node or -1, pool)
    int node = -1;
    const int mode = copy_bind_mode();
    if (mode == 2) node = node_of_address(src);
    else if (mode == 1) node = node_of_device(device);
    std::lock_guard<std::mutex> l(mu);
    for (auto& p : pools) if (p.first == node) return *p.second;
    std::vector<int> cpus = node >= 0 ? node_cpus(node) : std::vector<int>();
    size_t ncpus = cpus.size();
    if (ncpus == 0) { cpu_set_t aff; CPU_ZERO(&aff); ncpus = sched_getaffinity(0, sizeof aff, &aff) == 0 ? (size_t)CPU_COUNT(&aff) : std::max(1u, std::thread::hardware_concurrency()); }
    pools.emplace_back(node, std::unique_ptr<CopyPool>(new CopyPool(copy_threads_for(ncpus), cpus)));
    return *pools.back().second;
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
        copy_pool_for((const char*)src + off, c->device).copy(c->stage[k], (const char*)src + off, n);
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
        if (!host_ctx()) {
            int dev = 0;
            if (const char* d = getenv("FJ_DEVICE")) dev = atoi(d);
            host_ctx() = fj_ctx_create(dev);
            if (!host_ctx()) return 1;
        }
    }
    fj_ctx* c = host_ctx();
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
    last_timings() = t;
    if (out_count) *out_count = count;
    if (out_seconds) *out_seconds = t.total_ms * 1e-3;
    return 0;
}

}  // extern "C"