// fj_host.h -- host-side state shared by the translation units behind the C ABI (include/flashjoin.h):
//   fj_plan.hip   workspace, plans, the partition-pass state machine, contexts, options, diagnostics
//   fj_joins.hip  one-shot joins (the reference's drivers, hash_join.cpp:315-594), emit, owner split, bloom export / prefilter
//   fj_stream.hip joins whose relations arrive in pieces, the owner shuffle's pack / append side
//   fj_hostentry.hip  the NumPy (host-buffer) entry
// (one file, fj_api.hip, until round 4).
#pragma once
#include "fj_internal.h"
#include "../../include/flashjoin.h"
#include "../../include/flashjoin_lab.h"

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

struct StreamState;

namespace fjh {

int set_err(const char* fmt, ...);
fj_timings& last_timings();               // thread-local: what fj_last_timings() returns
#define HIPCHK(x)                                                                                          \
    do {                                                                                                   \
        hipError_t e_ = (x);                                                                               \
        if (e_ != hipSuccess) return fjh::set_err("%s:%d: %s failed: %s", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
    } while (0)

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
    fjh::DeviceGuard dev_guard_(dev);                                                                            \
    if (dev_guard_.err != hipSuccess) return fjh::set_err("selecting HIP device %d failed: %s", (int)(dev), hipGetErrorString(dev_guard_.err))
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
    // owner shuffle, sender side (fj_shuffle_pack_*: runs while a stream join is open and on another stream, so it has words of
    // its own, outside what clear_plan_scalars and the plan's error handling touch)
    u32 pack_alloc, pack_seg, pack_err, rx_alloc;   // chunk allocator / segment counter / error word of the packing pass; chunk count of a received piece
    unsigned long long pack_used[64];  // wire-format chunks per owner GPU (fj_pack_offsets)
    unsigned long long pack_kept;      // probe keys a piece kept after the sender-side precheck (fj_part_filter_inplace) / passing keys of a sample
    u32 pack_xcd[8 * FJ_PF_COUNTERS];  // ... and its per-XCD work counters (directly behind pack_kept: one memset clears both)
    u32 bc_bounds[20];                 // build broadcast (fj_bcast_pack): key index at which piece q of the rank's region starts
};

enum Slot {
    // [side][pingpong][kind]
    W_POOL_K = 0, W_POOL_V, W_DIR, W_LIST, W_BCHUNKS, W_REL, W_BOFF, W_SEGOFF, W_TOFF, W_TILES, W_KINDS,
    W_SIDE_STRIDE = 2 * W_KINDS,                 // sides: 0 = build relation, 1 = probe relation, 2 = the owner shuffle's packing pass
    W_PART_COUNT = 3 * W_SIDE_STRIDE, W_OUT_OFF, W_GT_KEYS, W_GT_VALS, W_GT_BLOOM, W_WG_COUNT,
    W_H_BK, W_H_BV, W_H_PK, W_H_OK, W_H_OV, W_ROWIDX, W_BKEYS, W_BBASE,
    W_PK_FI, W_PK_BKEYS, W_PK_OBASE,                                   // fj_shuffle_pack_*: output-chunk index, keys per bucket, output chunks before a bucket
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
    bool dups_main = false; std::vector<u32> sk_parts; int sk_bits = 0, sk_plan_bits = 0, sk_npass = 0;     // ... which partitions, by how many more bits (the first-occurrence emit path repeats it with row indices)
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
    u32 item_tc_max = 0;                 // > 0: the join's items hold at most this many probe chunks (32: the 16384-slot counting kernel will run, fj_join_wide.hip)
    bool want_items = false; u32 items_cap = 0; u32* part_count = nullptr; int part_count_slot = W_PART_COUNT;
    // bloom precheck (probe side): run the filter stage once `bloom_level` passes are complete, against bloom_build
    bool bloom_done = false; const FjChunkSet* bloom_build = nullptr;
    const u32* bloom_prebuilt = nullptr;            // filters shipped by another GPU (sender-side precheck) instead of bloom_build's keys
    unsigned long long* bloom_bucket_keys = nullptr; // [buckets] survivors per bucket (for flattening the survivors)
    // build side: keep a copy of the level the probe side's filter will read
    int save_level = 0; FjChunkSet saved{};
    // scalars of the pass when they are not the plan's (the owner shuffle's packing pass runs beside an open stream join)
    u32* alloc_word = nullptr; u32* seg_word = nullptr; u32* err_word = nullptr;
    // the previous level arrived from other GPUs in the 7-byte wire format (FjPartArgs::in_pk7)
    bool in_pk7 = false; u32 in_b0 = 0, in_top_shift = 0;
};

}  // namespace fjh

struct PackState {              // fj_shuffle_pack_begin .. _finish: one piece on its way into the wire format
    bool begun = false, deferred = false;      // deferred: the first pass is queued, fj_shuffle_pack_filter queues the rest
    u32 part_shift = 0;                        // mixed key >> part_shift = final partition of the global plan
    fjh::PassIter it;
    FjPackArgs args{};
    int nranks = 0;
};

struct StreamState {            // fj_stream_*: a counting join whose relations arrive in pieces
    bool active = false;
    fjh::Plan plan; fjh::PassIter pit, bit; FjLdsJoinArgs ja{};
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
    bool probe_prepared = true; u32 probe_appends = 0;      // shuffled: the probe side's pools are sized when its first piece is seen (an owner of hot probe keys receives far more than the mean)
    bool with_vals = false;         // ... and the build side carries values: a materialising join (pairs stay with the owner: fj_emit_pairs after the finish)
};

struct BcastState {             // fj_bcast_*: one step of the multi-GPU build-broadcast join on this rank (csrc/fj_bcast.hip)
    bool packed = false, probed = false;
    size_t nb_total = 0, nb = 0, np = 0;
    int pieces = 1, evc = 0;
    fjh::Plan plan; fjh::PassIter pit; FjLdsJoinArgs ja{};
    // a materialising step (the regions carry the build values): counted by fj_bcast_join, the pairs written by fj_emit_pairs afterwards
    bool with_vals = false, mat_ready = false;
    const void* mat_base = nullptr; int mat_nsrc = 0; uint64_t mat_off[FJ_WIDE_MAXSRC] = {}, mat_nk[FJ_WIDE_MAXSRC] = {};
    u64 mat_count = 0; u32 mat_items = 0;
};

struct fj_ctx {
    int device = 0;
    fjh::Buf bufs[fjh::W_NSLOTS];
    hipEvent_t ev[fjh::E_NEV];
    hipStream_t side = nullptr;        // copy stream of the host-buffer entry (fj_join_host)
    fjh::Scalars* d_sc = nullptr;
    fjh::Scalars* h_sc = nullptr;
    fjh::Pending pend;
    StreamState st;
    PackState pk;
    BcastState bc;
    hipEvent_t pk_ev = nullptr;        // the packing pass's counts have landed in pk_h
    unsigned long long* pk_h = nullptr;   // pinned: [64] chunks per owner, [64] = the pass's error word, [65] = keys kept by the precheck, [66] = sample result
    size_t ws_bytes = 0;
    u32 num_cus = 256;
    u32 reserve_cus = 0;               // CUs the partition passes leave free (set by the multi-GPU driver while RCCL's kernels share the GPU: fj_ctx_reserve_cus)
    void* stage[3] = {nullptr, nullptr, nullptr};     // pinned staging ring of the host-buffer entry (fj_join_host)
    size_t stage_bytes = 0;
    bool plan_in_flight = false;       // a plan was begun and has not completed (an error in between leaves chunk counts behind)
    bool slot_dirty[fjh::W_NSLOTS] = {};    // ... in which case every self-cleaning buffer is re-zeroed IN FULL before its next use (get_zeroed_buf)
    std::recursive_mutex mu;           // one C-ABI call at a time per context (FJ_ENTER)
};

namespace fjh {

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
// test / measurement hooks, one bit each in the option "lab_hooks" (fj_set_option; nothing reads the environment for them):
#define FJ_HOOK_LOOPBACK 1        // multi-GPU driver: a rank's own share travels through ncclSend / ncclRecv too (the code path every peer's share takes at N > 1)
#define FJ_HOOK_INJECT_FAIL 2     // multi-GPU driver: this rank's probe-side appends fail (agreed failures, reruns)
#define FJ_HOOK_SPLIT_ALWAYS 4    // a 1-rank communicator gets its own control communicator too
#define FJ_HOOK_ONE_COMM 8        // control collectives on the data communicator
#define FJ_HOOK_RESERVE_ALWAYS 16 // the CU reserve of an N > 1 step on one rank too (measurement)
#define FJ_HOOK_EMIT_TAGGED 32    // emitting pass on the tagged-table kernel (A/B)
#define FJ_HOOK_EMIT_RETRY_7TH 64 // every 7th item of the cuckoo emit kernel takes its retry path (results exact)
struct Options {
    size_t radix_threshold = 0; int scalar_hbm_table = 0; u32 persistent_min_items = 8192; u32 plan_target_keys = FJ_PART_TARGET_KEYS;
    int bloom_variant = 0, bloom_auto = 1, bloom_auto_max_hit_bp = 2300, mat_single_pass = 1;     // (2300: measured break-even at c4 sizes is 24 % hits, profiles/r03_bloom_threshold.csv)
    int join_wide = 2;                 // counting joins on the bucketed 16384-slot table (fj_join_wide.hip): 0 never, 1 whenever eligible, 2 (default) when at most ~3 probe rows per build row reach the join (wide_join_planned)
    u32 lab_hooks = 0;                 // FJ_HOOK_* bits
    u32 join_items_target = 2048;      // work items the join of a plan with few partitions is cut into (tuning knob)
    Options();                         // initial values: FJ_OPTIONS="name=value,name=value" (the names of fj_set_option), the ONE environment variable behind all of them
};
Options& options();

// ---- workspace and plans (fj_plan.hip) ----
int get_buf(fj_ctx* c, int slot, size_t bytes, void** out);
int get_zeroed_buf(fj_ctx* c, int slot, size_t bytes, void** out, hipStream_t s);
int plan_npass(int bits);
void plan_passes(Plan& p, bool extra_first);
Plan make_plan(size_t nb, int top_bits, bool want_bloom = false, u64 target_override = 0);
void pass_init(PassIter& it, int side, bool has_vals, size_t n, const Plan& plan, int top_bits);
u32 pass_groups(u64 chunks, u64 rows, u32 tile_chunks, u32 F);
int pass_prepare(fj_ctx* c, PassIter& it, u32 appends, hipStream_t s);
int pass_launch(fj_ctx* c, PassIter& it, const u64* keys, const u64* vals, size_t n, hipStream_t s, int* ev_cursor);
bool bloom_stage_follows(const PassIter& it, int level);
void join_item_geometry(u64 nparts, size_t np, u64 chunk_bound, u32* tc, u64* max_items);
bool wide_join_planned(bool materialize, size_t nb, size_t np_eff, int bits);
int level_finish(fj_ctx* c, PassIter& it, bool final_level, hipStream_t s);
int pass_complete(fj_ctx* c, PassIter& it, hipStream_t s);
int bloom_stage(fj_ctx* c, PassIter& it, hipStream_t s);
int run_passes(fj_ctx* c, PassIter& it, const u64* keys, const u64* vals, hipStream_t s, FjChunkSet* out, int* ev_cursor);
int clear_plan_scalars(fj_ctx* c, hipStream_t s);
void begin_plan(fj_ctx* c);
void end_plan(fj_ctx* c);
int read_scalars(fj_ctx* c, hipStream_t s);
float ev_ms(fj_ctx* c, int a, int b);
int stamps_begin(unsigned long long** dbg, hipStream_t s);
int stamps_report(const char* label, const unsigned long long* dbg, u32 nitems, hipStream_t s);

// ---- one-shot joins (fj_joins.hip) ----
int emit_pending(fj_ctx* c, u64* d_ok, u64* d_ov, size_t cap, hipStream_t s, fj_timings* t);
int radix_join_tail(fj_ctx* c, int materialize, FjLdsJoinArgs& ja, const Plan& plan, size_t np, const PassIter& pit, hipStream_t s,
                    fj_timings* t, int evc, u64* out_count, bool* lds_full, int top_bits, SingleOut* so = nullptr);

// ---- streamed joins (fj_stream.hip) ----
int stream_open(fj_ctx* c, size_t nb_bound, int build_appends, size_t np_bound, int probe_appends, hipStream_t s, int top_bits,
                size_t probe_piece_rows = 0);
int stream_append_build(fj_ctx* c, const u64* d_bk, size_t n, hipStream_t s);
int stream_flush_build(fj_ctx* c, StreamState& st, hipStream_t s);

// ---- host-buffer entry (fj_hostentry.hip) ----
fj_ctx*& host_ctx();                      // the internal context of fj_join_host (null before its first call)

}  // namespace fjh
