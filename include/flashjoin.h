/*
 * flashjoin.h -- C ABI of libflashjoin_hip.so, the MI355X (gfx950) drop-in for the hot path of
 * conanhujinming/flash_hash_join.
 *
 * The reference exposes this path only through a pybind11 module (hash_join.cpp:598-640):
 * twelve join functions that all take (build_keys, build_values, probe_keys) as uint64 arrays
 * and all return (total_results, core_duration_sec), plus initialize().  A C ABI for it does not
 * exist in the reference; this header is what a binding for the path (ctypes / cgo / JNI / a
 * pybind11 shim) binds instead of the reference's C++ templates.  Each entry point names the
 * reference interface it replaces.
 *
 * Conventions: every function returns 0 on success, non-zero on failure; fj_last_error() then
 * returns a thread-local, NUL-terminated description.  No torch / pybind types cross this ABI:
 * plain pointers and sizes only.  Keys and values are 64-bit words; int64 inputs are
 * reinterpreted bit-for-bit (the reference does the same through array_t<uint64_t>'s cast).
 */
#ifndef FLASHJOIN_H
#define FLASHJOIN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__)
#pragma GCC visibility push(default)       /* the library is built with -fvisibility=hidden and linked with csrc/exports.map: these 40 entry points are all it exports */
#endif

/* algo: which of the reference's drivers the call stands for */
#define FJ_ALGO_ADAPTIVE 0 /* adaptive_hash_join_{count,materialize}   hash_join.cpp:576-594 */
#define FJ_ALGO_SCALAR 1   /* _hash_join_scalar_{count,materialize}     hash_join.cpp:383-496, :536-567 */
#define FJ_ALGO_RADIX 2    /* _hash_join_radix_{count,materialize}      hash_join.cpp:315-381, :498-534 */

/* EXTENSION (no reference counterpart: the reference deduplicates build keys, hash_join.cpp:125): OR this into `algo` for a
 * many-to-many inner join - every build row is kept, a probe row yields one pair per build row with its key, the count is the
 * number of pairs.  Partitioned plan only; fails when a final partition would hold more than 4096 build rows (a key with
 * thousands of duplicates).  `bloom` is ignored. */
#define FJ_ALGO_MANY_TO_MANY 0x10

typedef struct fj_ctx fj_ctx;

/* Per-call device timings (milliseconds, HIP events on the caller's stream) and plan facts. */
typedef struct fj_timings {
    double total_ms;             /* whole device-resident join: what core_duration_sec reports */
    double build_phase_ms;       /* everything that touches only the build relation            */
    double probe_phase_ms;       /* probe-side partition passes + per-partition build/probe    */
    double join_ms;              /* the join kernel(s) alone (part of probe_phase_ms)          */
    double emit_ms;              /* second (writing) pass of a materialising join              */
    double probe_part_kernel_ms[4]; /* each probe-side partition kernel launch                 */
    double h2d_ms, d2h_ms;       /* host-buffer entry only                                     */
    int path;                    /* 0 = radix + LDS tables, 1 = global table                   */
    int passes;                  /* radix passes run on each relation (k)                      */
    int radix_bits;              /* total partition bits                                       */
    int fell_back;               /* 1 if the radix path overflowed an LDS table and the global path re-ran */
    uint64_t partitions;
    int overlapped;              /* always 0: build_phase_ms and probe_phase_ms are disjoint intervals (the two-stream schedules of
                                    rounds 1-2 are gone; the field keeps the struct layout)                              */
    int lds_retries;             /* 1 if some partitions overflowed the counting join's cuckoo table and were redone on the tagged table */
    /* bloom precheck of the partitioned plan (the *_bloom functions): */
    double filter_ms;            /* the filter kernel between the probe side's passes (part of probe_phase_ms)              */
    uint64_t filter_survivors;   /* probe keys that passed it (hits + false positives); 0 when bloom_level == 0             */
    int bloom_level;             /* 0: no precheck ran; L: the probe side was filtered after its L-th partition pass         */
    int sampled_hit_bp;          /* adaptive_* joins: hit rate (basis points) of the probe-side sample that decided on the precheck; -1: no sample taken */
    int host_streamed;           /* fj_join_host: 1 if the join ran piece by piece under the PCIe copy (h2d_ms then contains it) */
    int reserved3;
} fj_timings;

/* replaces: flash_join.initialize() / initialize_memory_system (hash_join.cpp:596, :639).
 * Selects nothing, allocates nothing; verifies a HIP device is usable. */
int fj_initialize(void);
const char* fj_last_error(void);
int fj_device_count(void);
const char* fj_version(void);
/* The binary interface's version.  It changes whenever a struct of this header changes layout or an entry point its signature; a
 * binding checks it once after loading the library (flash_hash_join_amd/_lib.py does).  Structs the library fills or reads on a
 * caller's behalf (fj_dist_timings, fj_dist_engine_ops) start with a struct_size word the caller sets to sizeof(its struct): the
 * library touches no byte beyond it. */
#define FJ_ABI_VERSION 6
int fj_abi_version(void);

/* Process-wide dispatch options (no reference counterpart; the reference hard-codes 1'000'000 at hash_join.cpp:576).  Initial values:
 * the ONE environment variable FJ_OPTIONS="name=value,name=value" (read once):
 *   "radix_threshold"  - adaptive_* joins use the non-partitioned HBM table below this many build rows (default 0:
 *                        the partitioned driver wins at every size on MI355X);
 *   "scalar_hbm_table" - (the HBM-table path is NOT a fast path: probe 0.05-0.07 of the HBM peak, build by global CAS 20 ms per
 *                        100M rows, profiles/README.md; it exists as the literal form of the reference's scalar algorithm
 *                        and as the fallback for a partition of more than 8128 distinct keys)
 *                        1: the "scalar" functions (FJ_ALGO_SCALAR: hash_join*, hash_join.cpp:383-496, :536-567) keep
 *                        ONE table for the whole build side in HBM, as the reference does in DRAM; 0 (default): they
 *                        run the same partitioned plan as the radix functions (identical results, 2-4x faster here)
 *                        and the HBM table is only the overflow fallback.
 *   "persistent_min_items" - counting joins whose plan has at least this many (partition, probe slice) work items use
 *                        the persistent join kernel (default 8192; a tuning/testing knob).
 *   "bloom_auto"       - 1 (default): the adaptive_* functions decide on the bloom precheck of the partitioned plan from a
 *                        sample of the probe side (4096 rows looked up in the partitioned build side): on when at most
 *                        "bloom_auto_max_hit_bp" (default 2300 = 23 %; measured break-even 24 %) of them hit; 0: adaptive_*_bloom filter, adaptive_* do
 *                        not, as named.  The explicit hash_join*_bloom
 *                        functions always run the precheck when the plan has two or more passes; hash_join* never do.
 *   "bloom_variant"    - hash / bit layout of the filter, 0..2 (csrc/fj_bloom_dev.h; default 2).  Must be
 *                        the same on every rank of a multi-GPU job (fj_bloom_prefilter checks it against the exporter's).
 *   "mat_single_pass"  - 1 (default): a materialising join whose output buffers hold >= np pairs runs single-pass (above);
 *                        0: always count, scan, emit.
 *   "plan_target_keys" - average build keys per final partition the plan aims for (default and maximum 4096 = half an LDS
 *                        table).  A testing knob: small values make small inputs take the deep
 *                        (two- and three-pass, bloom-filtered) plans that production only uses for >1M-row build sides.
 *   "join_wide"        - counting joins on the bucketed 16384-slot LDS table kernel (csrc/fj_join_wide.hip): 0 never, 1 whenever
 *                        eligible, 2 (default) when at most ~3 probe rows per build row reach the join (behind a filter: an estimate).
 *   "join_items_target" - work items the join of a plan with few partitions is cut into (default 2048; a tuning knob).
 *   "lab_hooks"        - test / measurement hooks, one bit each (csrc/fj_host.h FJ_HOOK_*; default 0): 1 a rank's own share of a
 *                        multi-GPU exchange travels through ncclSend / ncclRecv too, 2 injected failure of a local append, 4 split a
 *                        1-rank communicator, 8 one communicator, 16 the CU reserve on one rank too, 32 emitting pass on the tagged
 *                        kernel, 64 every 7th emit item takes its retry path.
 * fj_get_option returns -1 for an unknown name. */
int fj_set_option(const char* name, long long value);
long long fj_get_option(const char* name);

/* The device's key mixer (no reference counterpart; the reference hashes with CRC32C * const, hash_join.cpp:40-44, and
 * join results are hash-independent): a BIJECTION of 64-bit words.  Chunk pools, LDS tables and the owner shuffle's wire
 * format hold fj_key_mix64(key) instead of key - equality is preserved, radix digits are its top bits, table slots its low
 * bits, and a chunk of one radix bucket need not store the digits the bucket implies (7 bytes per key on the wire).
 * Host functions, no GPU needed: for tests and for hosts that want to predict the owner GPU of a key
 * (owner = ((mix >> 48) * nranks) >> 16 in the owner-scatter form; first-pass bucket * nranks >> fan_log0 in the chunk form). */
uint64_t fj_key_mix64(uint64_t key);
uint64_t fj_key_unmix64(uint64_t mixed);

/* One context per (process, device): owns the grow-only workspace, events and scratch words. */
fj_ctx* fj_ctx_create(int device);
void fj_ctx_destroy(fj_ctx* ctx);
size_t fj_ctx_workspace_bytes(const fj_ctx* ctx);
int fj_ctx_trim(fj_ctx* ctx);                 /* free the cached workspace; the context stays usable (grows again on demand).
                                                 NULL = the internal context of fj_join_host */

/*
 * replaces: all twelve pybind entry points hash_join.cpp:603-637 for HOST (NumPy) buffers.
 *   algo x bloom x materialize selects the function, e.g.
 *     hash_join_count_radix_bloom == (FJ_ALGO_RADIX, 1, 0); adaptive_join == (FJ_ALGO_ADAPTIVE, 0, 1).
 * build_vals must hold nb words (the reference never checks this; here it is the caller's
 * contract).  out_seconds receives the device-resident time (fj_timings.total_ms / 1e3); copies
 * over PCIe are reported separately through fj_last_timings().  When materialize != 0 and
 * out_keys/out_vals are non-NULL they receive malloc'ed host arrays of *out_count
 * (probe_key, build_value) pairs, to be released with fj_free_host(); pass NULL to drop them as
 * the reference does (hash_join.cpp:365-380).
 */
int fj_join_host(int algo, int bloom, int materialize,
                 const uint64_t* build_keys, const uint64_t* build_vals, size_t nb,
                 const uint64_t* probe_keys, size_t np,
                 uint64_t* out_count, double* out_seconds,
                 uint64_t** out_keys, uint64_t** out_vals);
void fj_free_host(void* p);
int fj_last_timings(fj_timings* out);

/*
 * Device-resident form of the same twelve functions (inputs already in HBM, 16-byte aligned).
 * stream is a hipStream_t (NULL = default stream).  hash_top_bits is 64 for a single-GPU join and
 * 48 after fj_owner_split (the top 16 hash bits chose the owner GPU).
 * Materialising joins: with d_out_keys == NULL the call counts only and keeps its partitions
 * resident; fj_emit_pairs() then writes exactly *out_count pairs into caller-allocated buffers.
 * With d_out_keys != NULL and out_capacity >= count both steps happen in this call.  With out_capacity >= np - room for
 * ANY result, what the reference allocates too (hash_join.cpp:330-334) - a partitioned join with unique build keys runs in
 * ONE pass over the probe side (no counting pass; option "mat_single_pass", default 1): the first *out_count rows of the
 * buffers are the pairs.
 */
int fj_join_device(fj_ctx* ctx, int algo, int bloom, int materialize,
                   const uint64_t* d_build_keys, const uint64_t* d_build_vals, size_t nb,
                   const uint64_t* d_probe_keys, size_t np,
                   void* stream, int hash_top_bits,
                   uint64_t* out_count,
                   uint64_t* d_out_keys, uint64_t* d_out_vals, size_t out_capacity,
                   fj_timings* timings);
int fj_emit_pairs(fj_ctx* ctx, uint64_t* d_out_keys, uint64_t* d_out_vals, size_t out_capacity,
                  void* stream, fj_timings* timings);

/*
 * Multi-GPU building block (no reference counterpart: the reference is single-process).
 * Splits n local rows by owner GPU = (top 16 hash bits * nranks) >> 16 into nranks contiguous
 * segments of d_out_keys (and d_out_vals when d_vals != NULL); h_counts[r] = rows for rank r.
 * The caller exchanges the segments (RCCL all-to-all) and joins what it receives with
 * hash_top_bits = 48.
 */
int fj_owner_split(fj_ctx* ctx, const uint64_t* d_keys, const uint64_t* d_vals, size_t n, int nranks,
                   uint64_t* d_out_keys, uint64_t* d_out_vals, uint64_t* h_counts, void* stream);

/* error recovery: drops a stream join that a failed multi-GPU step may have left open on the context (no-op when none is open) */
int fj_stream_abort(fj_ctx* ctx);

/* Plan queries of the two multi-GPU forms: what fj_dist_join can do for a total build side of nb_total rows (no GPU work).
 *   fj_shuffle_plan - 0 if the owner shuffle's chunk form applies (a plan of two or more passes and at least nranks first-pass
 *                     buckets: build sides above ~2M rows in all); *fan_log0 = log2 of the first pass's buckets, *npass = passes
 *   fj_bcast_plan   - 0 if the build-broadcast form applies (the plan has a pass): radix bits, final partitions, bytes per key of
 *                     the high-word plane (2 from 134M build rows in all: 6 bytes per key on the wire) */
int fj_shuffle_plan(size_t nb_total, int nranks, int* fan_log0, int* npass);
int fj_bcast_plan(size_t nb_total, int* bits, uint32_t* nparts, int* mid_bytes);

/*
 * Native multi-GPU entry (no reference counterpart: the reference is one process, hash_join.cpp:318; SURVEY.md 5 row
 * "Distributed communication backend", 7.1 dist/alltoall): the counting radix join of relations whose rows are block-distributed
 * over the ranks of a communicator - BASELINE configs[4].  One process per GPU; every rank calls fj_dist_join_count with its
 * LOCAL rows and gets the GLOBAL match count.  ONE driver (csrc/fj_dist.hip) runs the owner shuffle in chunk form (above) over
 * whatever moves the bytes: per piece the first radix pass, the per-owner chunk counts all-gathered on a control channel, the
 * copy into the 7-byte wire format (a rank's own share straight into its receive buffer), the exchange, the owner's second
 * pass - on three streams, the host blocking once per piece; a failure on any rank is reported by every rank.
 *   fj_dist_unique_id           - rank 0: 128 bytes for the other ranks (any out-of-band channel), as ncclGetUniqueId
 *   fj_dist_comm_create         - every rank, after selecting its device and creating its fj_ctx: ncclCommInitRank inside.
 *                                 RCCL is bound at run time (dlopen of librccl.so.1): a host that never calls this needs neither
 *                                 the library nor its headers.  Payload: grouped ncclSend / ncclRecv on an exchange stream;
 *                                 control collectives on a second communicator (ncclCommSplit) so that they do not queue behind
 *                                 the previous piece's sends.
 *   fj_dist_comm_from_nccl      - or wrap an ncclComm_t the host already has (not destroyed by fj_dist_comm_destroy)
 *   fj_dist_comm_from_transport - or bring your own transport: three blocking callbacks (gloo, MPI, a test harness).  With
 *                                 engine == NULL the rank's work runs on ctx's GPU; a stand-in engine (tests) replaces it,
 *                                 then ctx may be NULL and "device" pointers are whatever the stand-in's alloc returns.
 *   (a communicator serves ONE join at a time - it owns the exchange buffers and streams of the step; use one per thread)
 *   The CU reserve: over RCCL with more than one rank, the transport's send / receive kernels are resident on the GPU for the
 *   length of an exchange, and the partition passes and the wide join (one persistent workgroup per CU, static tile shares) must not
 *   find CUs taken.  Whether leaving 32 CUs free beats sharing all 256 is MEASURED per communicator: its second step runs with the
 *   reserve, its third without, both times the probe-side passes are timed (HIP events, an exchange in flight), the ranks add their
 *   times up in the next step's first all-gather, and every later step uses the faster setting (fj_dist_timings.reserve_*).
 *   FJ_DIST_RESERVE_CUS=n pins the number instead.
 *   fj_dist_join_count          - collective.  pieces: rounds of the probe exchange / partition ranges of a broadcast step; 0 = the
 *                                 driver decides (4, the measured default; 8 for a broadcast step that its model finds wire-bound:
 *                                 the tail behind the last piece is an eighth of the join), rank 0's value counts.  A build side
 *                                 of less than ~2M rows in all is refused (one-pass plan: join it on one GPU).
 *   fj_dist_join                - the same step, optionally materialising (materialize != 0: the build rows travel with their
 *                                 values, 16 bytes per build row on the wire): *out_local_count = pairs this rank owns; the
 *                                 caller then allocates them and calls fj_emit_pairs(ctx, ...) - before the next join on ctx.
 *                                 prefilter_below: the sender-side precheck (above) runs when a sample of the probe rows says that
 *                                 less than this share of them would travel - 0 = never (nothing is exported or sampled), >= 2 =
 *                                 always (no sample); rank 0's value is used on every rank; the HIP engine, or a stand-in with the
 *                                 optional precheck callbacks.  Costs one more kernel over a sender's probe rows and 1 byte
 *                                 per build key to every rank, saves (1 - survivors) of the probe exchange and of the owner's work.
 */
typedef struct fj_dist_comm fj_dist_comm;
typedef struct fj_dist_timings {
    size_t struct_size;                                /* IN: sizeof(fj_dist_timings) as the caller compiled it (the library writes no byte beyond it; 0 is refused) */
    double total_ms, split_ms, exchange_ms, join_ms;   /* host wall clock of this rank: whole step; waiting for the packing passes'
                                                          counts; the rest up to the local join's finish; finish + all-reduce */
    uint64_t local_count;                              /* matches this rank found in what it owns                 */
    uint64_t local_build_chunks, local_probe_chunks;   /* 256-key wire chunks this rank received (own share included); broadcast form: ROWS it joined (all ranks' build rows, its own probe rows) */
    uint64_t sent_chunks;                              /* ... and put on the links (own share excluded)            */
    int pieces, nranks, fan_log0, wire_chunk_bytes;    /* wire_chunk_bytes: 1792 (7 bytes per key) or 2048         */
    int prefilter;                                     /* 1: the sender-side precheck ran                          */
    double prefilter_sampled;                          /* share of the sampled probe rows that passed (-1: no sample) */
    uint64_t probe_rows_kept;                          /* probe rows of this rank that went into wire chunks       */
    uint64_t filter_bytes;                             /* bytes of partition filters this rank received            */
    int form;                                          /* FJ_DIST_FORM_SHUFFLE or FJ_DIST_FORM_BROADCAST: what the step ran as */
    int form_reserved;
    uint64_t wire_bytes_sent;                          /* bytes this rank put on the links (all peers)             */
    fj_timings local;                                  /* device timings of this rank's local join (fj_stream_finish) */
    /* the CU reserve (below): CUs this step's passes and joins left to the transport's own kernels; how that number came about
     * (0 = no reserve applies: one rank / a transport without kernels of its own, 1 = pinned by FJ_DIST_RESERVE_CUS, 2 = a measuring
     * step, 3 = chosen from the measurements); the measurements: this communicator's probe-side pass time per step, summed over the
     * ranks, with 32 CUs reserved and with none (ms; 0 until measured) */
    int reserve_cus, reserve_how;
    double reserve_with_ms, reserve_without_ms;
} fj_dist_timings;
typedef struct fj_dist_transport {
    void* user;
    int nranks, rank;
    int (*all_gather_u64)(void* user, const uint64_t* v, int n, uint64_t* out /* [nranks][n], rank-major */);
    int (*all_reduce_sum_u64)(void* user, uint64_t* v, int n);
    /* all-to-all of byte ranges that live in device memory (the driver has finished writing them): for part p < nparts and
     * peer r, send_bytes[p * nranks + r] bytes at send_ptr[...] go to r, recv_bytes[...] bytes from r land at recv_ptr[...];
     * returns once everything has landed.  Zero-byte entries (null pointers) are skipped on both sides. */
    int (*all_to_all_bytes)(void* user, int nparts, const void* const* send_ptr, const uint64_t* send_bytes,
                            void* const* recv_ptr, const uint64_t* recv_bytes);
} fj_dist_transport;
typedef struct fj_dist_engine_ops {   /* a stand-in for the rank's own work (the CPU test-suite); 0 = success everywhere */
    size_t struct_size;                                /* sizeof(fj_dist_engine_ops) as the caller compiled it: callbacks beyond it read as NULL (0 is refused) */
    void* user;
    size_t chunk_bytes;                                /* bytes per wire chunk the stand-in produces and consumes */
    const char* (*error)(void* user);
    int (*plan)(void* user, uint64_t nb_total, int nranks);
    void* (*alloc)(void* user, size_t bytes);
    void (*release)(void* user, void* p);
    int (*pack_begin)(void* user, const void* rows, uint64_t n, uint64_t nb_total, int nranks);
    int (*pack_counts)(void* user, uint64_t* used);
    int (*pack_finish)(void* user, void* const* dst_chunks, uint32_t* const* dst_dir);
    int (*open)(void* user, uint64_t nb_total, int nranks, int rank, uint64_t nb_bound, uint64_t np_bound, int pieces);
    int (*append)(void* user, int side, const void* chunks, uint32_t* dir, uint64_t nchunks);
    int (*finish)(void* user, uint64_t* count);
    void (*abort)(void* user);
    /* optional, all four or none (NULL: the stand-in has no sender-side precheck and fj_dist_join refuses prefilter_below > 0):
     * what fj_shuffle_part_filter_range / fj_stream_export_part_filters / fj_shuffle_pack_filter (called right after pack_begin
     * for a prechecked piece; *kept = rows it kept) / fj_part_filter_sample do */
    int (*filter_range)(void* user, uint64_t nb_total, int nranks, int rank, uint64_t* first, uint64_t* count, uint64_t* total, uint64_t* bytes_each);
    int (*export_filters)(void* user, void* dst);
    int (*pack_filter)(void* user, const void* filters, uint64_t* kept);
    int (*sample)(void* user, const void* rows, uint64_t n, uint64_t stride, const void* filters, uint64_t nb_total, int nranks, uint64_t* kept);
    /* optional, all seven or none (NULL: the stand-in has no build-broadcast form): what fj_bcast_region_bytes / fj_bcast_piece_span /
     * fj_bcast_plan (final partitions) / fj_bcast_pack + fj_bcast_pack_bounds (synchronous: bounds[pieces + 1]) / fj_bcast_probe /
     * fj_bcast_join / fj_bcast_finish do */
    uint64_t (*bc_region_bytes)(void* user, uint64_t nb_total, uint64_t nkeys);
    int (*bc_span)(void* user, uint64_t nb_total, uint64_t nkeys, uint64_t k_lo, uint64_t k_hi, int part, uint64_t* offset, uint64_t* bytes);
    int (*bc_nparts)(void* user, uint64_t nb_total, uint32_t* nparts);
    int (*bc_pack)(void* user, const void* rows, uint64_t n, uint64_t nb_total, void* region, int pieces, uint64_t* bounds);
    int (*bc_probe)(void* user, const void* rows, uint64_t n, uint64_t nb_total);
    int (*bc_join)(void* user, const void* base, int nsrc, const uint64_t* region_off, const uint64_t* nkeys, uint32_t part_lo, uint32_t part_hi);
    int (*bc_finish)(void* user, uint64_t* count);
} fj_dist_engine_ops;
/* The form a counting step takes (rank 0's setting is used on every rank):
 *   FJ_DIST_FORM_SHUFFLE   - the owner shuffle in chunk form: every row of both relations travels to the owner of its first radix
 *                            digit, 7 bytes per key;
 *   FJ_DIST_FORM_BROADCAST - the build broadcast (csrc/fj_bcast.hip): the probe rows stay, every rank's build rows travel to every
 *                            peer, 6 bytes per key; up to 16 ranks.  Materialising joins too (round 6): the values travel as a
 *                            fourth part (14 bytes per build row) and the pairs stay with the rank that holds the PROBE row
 *                            (fj_emit_pairs after the step).  In every form a build key that occurs more than once yields ONE pair
 *                            per matching probe row, with the value of one of its copies;
 *   FJ_DIST_FORM_AUTO      - (default) whichever a per-link / per-rank cost model puts ahead for the step's sizes: bytes per link
 *                            over link_bytes_per_s (<= 0: 55e9) against the kernel time per rank measured on one MI355X
 *                            (profiles/r05_scale_model.txt).  Probe-heavy joins (BASELINE configs[4]: 10 probe rows per build row)
 *                            broadcast at every N <= 16; build-heavy ones shuffle. */
#define FJ_DIST_FORM_AUTO 0
#define FJ_DIST_FORM_SHUFFLE 1
#define FJ_DIST_FORM_BROADCAST 2
int fj_dist_comm_set_form(fj_dist_comm* comm, int form, double link_bytes_per_s);
/* the model behind FJ_DIST_FORM_AUTO: modelled seconds of one counting step in either form (NULL: not wanted); returns the form it picks */
int fj_dist_model(int nranks, uint64_t nb_max, uint64_t np_max, uint64_t nb_total, uint64_t np_global, uint64_t region_max, double link_bytes_per_s,
                  double* t_shuffle, double* t_broadcast);
int fj_dist_unique_id(char* out128);
fj_dist_comm* fj_dist_comm_create(fj_ctx* ctx, const char* unique_id128, int nranks, int rank);
fj_dist_comm* fj_dist_comm_from_nccl(fj_ctx* ctx, void* nccl_comm);
fj_dist_comm* fj_dist_comm_from_transport(fj_ctx* ctx, const fj_dist_transport* transport, const fj_dist_engine_ops* engine);
void fj_dist_comm_destroy(fj_dist_comm* comm);
int fj_dist_comm_rank(const fj_dist_comm* comm);
int fj_dist_comm_size(const fj_dist_comm* comm);
int fj_dist_join_count(fj_dist_comm* comm, const uint64_t* d_build_keys, size_t nb, const uint64_t* d_probe_keys, size_t np, int pieces,
                       void* stream, uint64_t* out_global_count, fj_dist_timings* timings);
int fj_dist_join(fj_dist_comm* comm, const uint64_t* d_build_keys, const uint64_t* d_build_vals, size_t nb, const uint64_t* d_probe_keys, size_t np,
                 int pieces, int materialize, double prefilter_below, void* stream, uint64_t* out_global_count, uint64_t* out_local_count,
                 fj_dist_timings* timings);

/*
 * Deterministic synthetic relations, generated in HBM:
 *   build_keys[i] = (first+i+1)*M, build_vals[i] = first+i, M = 0x9E3779B97F4A7C15;
 *   probe j = first+i: r = 1 + mix(seed,j) % B, hit = mix(seed^1,j) % 10000 < hit_bp,
 *   key = (r + (hit ? 0 : B)) * M.   *h_expected_hits = number of hits generated (closed-form count).
 */
int fj_generate_build(fj_ctx* ctx, uint64_t* d_keys, uint64_t* d_vals, uint64_t first, size_t n, void* stream);
int fj_generate_probe(fj_ctx* ctx, uint64_t* d_keys, uint64_t first, size_t n, uint64_t build_total,
                      uint64_t seed, uint32_t hit_bp, uint64_t* h_expected_hits, void* stream);

/* plain device memory helpers for hosts without another allocator */
int fj_device_malloc(void** p, size_t bytes);
int fj_device_free(void* p);
int fj_memcpy_h2d(void* d, const void* h, size_t bytes);
int fj_memcpy_d2h(void* h, const void* d, size_t bytes);
int fj_memcpy_d2d(void* d_dst, const void* d_src, size_t bytes);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* FLASHJOIN_H */