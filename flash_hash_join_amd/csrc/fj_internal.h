// fj_internal.h -- host-side declarations shared by the .hip translation units.
#pragma once
#include "fj_common.h"

// device error word bits
#define FJ_ERR_POOL 1u       // chunk pool exhausted (sizing bug) -> call fails
#define FJ_ERR_LDS_FULL 2u   // a final partition does not fit its LDS table -> global-table fallback
#define FJ_STAT_DUPS 4u      // (status, not an error) the build side holds duplicate keys
#define FJ_STAT_RETRY 8u     // (status) some counting-join items overflowed the cuckoo table: part_count[item] == FJ_ITEM_RETRY marks them
#define FJ_STAT_EMIT_RETRY 16u // (status) some items of the emitting pass overflowed the cuckoo table: marked the same way, redone on the tagged table
#define FJ_ERR_VARIANT 32u   // filters handed to the filter kernel were built with another bloom_variant (sender-side precheck across ranks)
#define FJ_BLOOM_HDR_MAGIC 0xB100F000u   // exported filter sets end with 4 header words: [0] = magic | variant
#define FJ_STAT_TOOBIG 64u   // (status) some items' partitions hold more distinct build keys than even the tagged LDS table takes: part_count[item] == FJ_ITEM_TOOBIG marks them; the host re-partitions just those
#define FJ_ERR_OUTCAP 128u   // single-pass materialising join: more pairs than the caller's output buffers hold
#define FJ_ITEM_RETRY 0xFFFFFFFFu
#define FJ_ITEM_TOOBIG 0xFFFFFFFEu

// ---- partition pass ---------------------------------------------------------------------------
struct FjPartArgs {
    // input: flat arrays (in_list == nullptr) or a bucket-grouped chunk list over a chunk pool
    const u64* in_keys;
    const u64* in_vals;
    const u32* in_list;
    const u32* in_dir;
    const uint4* in_tiles;   // list input: tile table {first list index, chunks, bucket, -}
    const u32* in_ntiles;    // device scalar: number of tiles
    u64 n_flat;
    u32 parent0;             // bucket id of every key of a flat input
    // output chunk pool
    u64* out_keys;
    u64* out_vals;
    u32* out_dir;            // pre-set to FJ_DIR_INVALID
    u64* out_rel;            // per chunk: (producing segment << 32) | rank inside (segment, bucket)
    u32* seg_off;            // [max_segs][F] span offset of a segment inside a bucket's chunk list
    u32* bchunks;            // [nb_out] chunks per bucket (zeroed before the pass)
    u32* alloc;              // device scalar: next unallocated chunk id (zeroed before the pass)
    u32* seg_counter;        // device scalar: next segment id (zeroed before the pass)
    u32 cap_chunks;
    u32 max_segs;
    u32* err;
    // radix digit: bucket = (hash >> shift) & (2^fan_log - 1)
    u32 shift;
    u32 fan_log;
    u32 side;                // 0 = build relation, 1 = probe relation (selects the kernel's name only)
    u32 slab;                // chunk ids a workgroup takes per allocator hit (fj_slab_for; a multiple of the run length)
    u32 run_log;             // chunk ids per (segment, bucket) in aligned runs of 2^run_log (0 or FJ_RUN_LOG; FjChunkSet::run_log of the output)
    // chunk-list input in the owner shuffle's 7-byte wire format (chunks received from other GPUs, csrc/fj_pack.hip): chunk id i
    // lives at byte i * FJ_WIRE7_BYTES of in_keys; bits 56..63 of its keys = (in_b0 + the tile's parent bucket) >> in_top_shift
    u32 in_pk7, in_b0, in_top_shift;
};

// a chunk pool plus its per-bucket chunk lists (output of one pass, input of the next)
struct FjChunkSet {
    u64* keys;               // chunk pool, or (list == nullptr) a flat array of n_flat keys
    u64* vals;               // nullptr for the probe side
    u64 n_flat;
    u32* dir;                // [cap]
    u64* rel;                // [cap]
    u32* seg_off;            // [max_segs][fan]
    u32* alloc;              // device scalar
    u32 cap;
    u32 nb;                  // number of buckets at this level
    u32 fan_mask;            // fan-out of the producing pass - 1
    u32 max_segs;
    u32* bchunks;            // [nb]   chunks per bucket
    u32* boff;               // [nb+1] chunk-list offsets; boff[nb] = list length
    u32* list;               // [cap]  chunk ids grouped by bucket
    u32 run_log;             // chunk ids were handed out in aligned runs of 2^run_log ids per (segment, bucket), used in order, the unused rest
                             // marked FJ_DIR_INVALID (the partition pass: FJ_RUN_LOG); 0: every id stands alone (bloom stage, owner shuffle)
};

u32 fj_partition_lds_bytes(u32 fan_log, bool vals, int line_log);
// chunks (of 256 rows) per input tile of the pass kernel chosen for this fan-out and payload
u32 fj_partition_tile_chunks(u32 fan_log, bool vals);
hipError_t fj_launch_partition(const FjPartArgs& a, bool vals, int line_log, u32 grid, hipStream_t s);
// level bookkeeping after a pass: chunk-list offsets (clears cs.bchunks for the next join), chunk lists, and the tile table
// of the level's consumer (tc chunks per tile; 0 = none); zero_tail: optional [max_tiles] array whose entries past the
// number of tiles are cleared
hipError_t fj_launch_group(const FjChunkSet& cs, u32 tc, u32* toff, uint4* tiles, u32 max_tiles, u32* zero_tail, hipStream_t s);
hipError_t fj_launch_scan_u32_to_u64(const u32* in, u64* out, u32 n, hipStream_t s);
// received chunks (directory words only) -> (segment, rank, span offset) for fj_launch_group; fan = power of two >= nbk
hipError_t fj_launch_dir_rank(u32* dir, u32 n, u32 b_lo, u32 nbk, u32 fan, u64* rel, u32* seg_off, u32* bchunks, u32* nalloc, hipStream_t s);

// ---- bloom precheck between two probe-side passes (csrc/fj_bloom.hip) ----------------------------
struct FjBloomArgs {
    // hot: read for every tile - these stay in scalar registers across the kernel's loop
    const u64* pkeys;            // probe relation at the filtered level: chunk pool,
    const u32* plist;            // chunk lists,
    const uint4* tiles;          // and the tile table over them (tiles of fj_bloom_tile_chunks() chunks)
    u64* out_keys; u32* out_dir; u64* out_rel;   // output chunk pool: same bucket structure, survivors only (a level with fan-out 1)
    u32 cap_chunks;
    u32 pnb;                     // buckets of the level
    // cold: read at kernel start, at a bucket change, or once per slab of output chunks.  The kernel reads them through
    // bf_late_args() at the point of use, so that they do not occupy scalar registers across the loop (round 2's form of this
    // kernel spilled ~95 scalar registers into vector lanes: a v_readlane per use)
    const u32* toff;             // [pnb+1] first tile of every bucket; toff[pnb] = number of tiles
    const u64* bkeys; const u32* blist; const u32* bboff;   // build relation at the filtered level (filter source)
    u32* seg_off; u32* bchunks; u32* alloc; u32* seg_counter;
    u32 max_segs;
    u32* err;
    unsigned long long* survivors;   // device scalar: probe keys that passed
    const u32* prebuilt;             // non-null: filters come from HBM (FJ_BLOOM_WORDS words per bucket) instead of being built from the build keys
    unsigned long long* bucket_keys; // non-null: [pnb] survivors per bucket are accumulated here (zeroed by the caller)
};
u32 fj_bloom_tile_chunks();
u32 fj_bloom_waves_per_group();
u32 fj_bloom_slab_chunks();
hipError_t fj_launch_bloom_filter(const FjBloomArgs& a, u32 grid, int variant, hipStream_t s);
// all bucket filters of a build-side level -> HBM; chunk set -> dense array (base[nb+1] receives the buckets' offsets, base[nb] the total)
hipError_t fj_launch_bloom_export(const FjChunkSet& build, u32* out, u32 grid, int variant, hipStream_t s);
hipError_t fj_launch_flatten(const FjChunkSet& cs, const unsigned long long* bucket_keys, unsigned long long* base, u64* out, hipStream_t s);

// ---- owner shuffle, sender side (csrc/fj_pack.hip): level-1 chunk set -> dense wire-format chunks grouped by owner GPU ----------
struct FjPackArgs {
    const u64* keys; const u64* vals;            // the packing pass's chunk pool (vals: nullptr for keys-only relations)
    const u32* list; const u32* boff;            // its chunk lists
    u32 nb, fan_log, nranks, wire7;              // nb = 2^fan_log first-pass buckets; wire7: 7-byte chunks (fan_log >= 8), else 8-byte
    uint4* fi;                                   // [output chunks] descriptor: {list entry of the input chunk holding the chunk's key 0, the next entry,
                                                 //  position of key 0 inside that input chunk, index of that entry in the chunk list}
    u32* fb;                                     // [output chunks] the chunk's directory word (bucket << 9 | keys)
    u32* bkeys;                                  // [nb] keys per bucket
    u32* obase;                                  // [nb + 1] output chunks before bucket b; [nb + 1 + r]: first output chunk of owner r
    unsigned long long* used;                    // [nranks] output chunks per owner
    unsigned char* dst_k[64]; u64* dst_v[64]; u32* dst_d[64];   // per owner: where its chunks, values and directory words go
};
hipError_t fj_launch_pack_plan(const FjPackArgs& a, hipStream_t s);       // fills fi, bkeys, obase, used
// sender-side precheck in chunk form (csrc/fj_pack.hip): FJ_PFILT_BYTES of Bloom filter per final partition of the global plan
#define FJ_PF_COUNTERS 4u              // work counters per XCD of fj_part_filter_inplace
#ifndef FJ_PFILT_BYTES
#define FJ_PFILT_BYTES 4096u
#endif
hipError_t fj_launch_part_filter_export(const FjChunkSet& build, u64* out, u32 grid, hipStream_t s);            // owner: [build.nb][FJ_PFILT_BYTES / 8]
hipError_t fj_launch_part_filter_inplace(const FjChunkSet& cs, const u64* filters, u32 part_shift, unsigned long long* kept, u32* next_of_xcd /* [8 * FJ_PF_COUNTERS], zeroed */, u32 grid, hipStream_t s);
hipError_t fj_launch_part_filter_sample(const u64* raw, u64 n, u64 stride, const u64* filters, u32 part_shift, unsigned long long* kept, hipStream_t s);
hipError_t fj_launch_pack_squeeze(const FjPackArgs& a, u32 grid, hipStream_t s);

// A partition pass is ONE persistent workgroup per CU with a static share of the tiles: a workgroup that finds no free CU (RCCL's
// send / receive kernels hold some for the length of an exchange) starts when another one has finished, and the pass takes twice as
// long.  While a multi-GPU step runs, the passes therefore launch num_cus - n workgroups (csrc/fj_dist.hip; internal, not part of the
// public header).  n = 0 restores the full grid.
// (fj_ctx_reserve_cus: include/flashjoin.h)

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel, device) instead of on every launch
hipError_t fj_set_max_lds_once(const void* fn, u32 bytes);

// ---- joins ------------------------------------------------------------------------------------
struct FjLdsJoinArgs {
    FjChunkSet build, probe;     // final-level chunk sets (same nb), or flat arrays (list == nullptr)
    u32 nparts;                  // number of final partitions
    u32 nsplit;                  // flat probe side only: work items per partition (equal slices of the probe side)
    // chunk-list probe side: work items = tiles of the probe chunk lists, {first list index, chunks, partition, -}; a
    // partition with many probe chunks (skew) becomes many items, each rebuilding the partition's small table
    const uint4* items;          // nullptr for a flat probe side
    const u32* nitems_dev;       // device scalar: number of valid entries of items[]
    u32 items_cap;               // allocated entries (grid of the one-workgroup-per-item kernels)
    u32* part_count;             // [items] matches per work item
    unsigned long long* total;   // device scalar
    u32* err;
    // materialise
    const u64* out_off;          // [items+1] exclusive scan of part_count
    u64* out_keys;
    u64* out_vals;
    // single-pass materialising join (fj_emit_join_persistent<SINGLE>): no counting pass, no per-item offsets - every probe round
    // reserves its pairs' output range on this device cursor (one atomic per workgroup and round); out_capacity = pairs the
    // output buffers hold.  The cursor ends as the join's match count.
    unsigned long long* out_cursor;
    u64 out_capacity;
    u32 retry_only;              // tagged-table counting kernel: process only the items the cuckoo kernel marked FJ_ITEM_RETRY
    u32 mark_toobig;             // tagged-table counting kernel: a partition beyond the table marks its item FJ_ITEM_TOOBIG (FJ_STAT_TOOBIG) instead of raising FJ_ERR_LDS_FULL
    u32 want_dups;               // counting pass of a materialising join: report duplicate build keys (FJ_STAT_DUPS)
    u32 dedup;                   // materialising pass: build 'values' are row indices, the smallest wins, then orig_vals[idx]
    const u64* orig_vals;        // the caller's build_values (dedup only)
    u32 avg_build_keys;          // build rows per final partition on average (host hint: selects the 16384-slot counting kernel, fj_join_wide.hip)
    unsigned long long* dbg;     // diagnostic: per-item phase stamps (s_memrealtime), nullptr in production
    u32 dbg_flags;               // diagnostic ablations: 1 = skip lookups, 2 = skip inserts, 4 = no output stores (results wrong on purpose); 8 = test hook: the cuckoo emit kernel sends every 7th item down its retry path (results exact)
};
// next_item: device word for the persistent counting kernel's work counter (nullptr: one workgroup per item)
hipError_t fj_launch_lds_join(const FjLdsJoinArgs& a, bool materialize, hipStream_t s, u32* next_item = nullptr,
                              u32 persistent_min_items = 8192);
// the single-pass materialising join over chunk lists (a.out_cursor != nullptr; unique build keys: duplicates are reported, FJ_STAT_DUPS)
hipError_t fj_launch_emit_single(const FjLdsJoinArgs& a, hipStream_t s, u32* next_item);
// ---- counting join with a 16384-slot table, one 1024-thread workgroup per CU (csrc/fj_join_wide.hip) ----
// for joins whose build side is not small against the probe side (wide_join_planned, fj_plan.hip), and for the multi-GPU
// build-broadcast form, whose build side arrives as dense per-partition runs from every rank (DENSE)
#define FJ_WIDE_MAXSRC 16u
struct FjWideArgs {
    const u32* toff; u32 part_lo, part_hi;       // toff != nullptr: only the items of partitions [part_lo, part_hi) (toff = first item of every partition)
    // DENSE build side: source s keeps, at byte offsets into base, an offset table u32[nparts + 1] (keys before partition p), the keys'
    // low words u32[n_s] and the low (32 - bits) bits of their high words as u16[n_s] (mid_bytes == 2) or u32[n_s]; partition p supplies the top `bits` bits
    const unsigned char* base; u32 nsrc, bits, mid_bytes;
    u32 group_log;                               // items are dealt to the workgroups in runs of 2^group_log consecutive ones (0: one by one); > 0 where partitions are cut into several items: a run's items of one partition share one table build
    u32 max_units;                               // DENSE: a partition of more 256-slot load units than this is marked for the retry ladder (0: the kernel's own limit, 32 = 8192 key slots)
    u32 pmask;                                   // the bits of a mixed key's HIGH word that name its final partition (all radix digits come from hash word 1): what tells a
                                                 // table entry of the partition in place from one that an earlier partition left behind (fj_wide_pmask)
    u64 offs_off[FJ_WIDE_MAXSRC], lo_off[FJ_WIDE_MAXSRC], mid_off[FJ_WIDE_MAXSRC];
};
hipError_t fj_launch_count_join_wide(const FjLdsJoinArgs& a, const FjWideArgs& w, bool dense, u32 grid, hipStream_t s);
// FjWideArgs::pmask of a plan of `bits` radix bits taken from bit `top_bits` of the mixed key downwards (top_bits = 64 or 48)
static inline u32 fj_wide_pmask(int bits, int top_bits) { return bits <= 0 ? 0u : (bits >= 32 ? 0xFFFFFFFFu : ((1u << bits) - 1u) << (top_bits - 32 - bits)); }

// second chance for the items whose partition overflowed the cuckoo table (load > ~0.45): the tagged 2x4-slot table
// with linear-probing overflow holds up to 8128 keys; only a partition beyond that raises FJ_ERR_LDS_FULL
hipError_t fj_launch_lds_join_retry(const FjLdsJoinArgs& a, hipStream_t s);
hipError_t fj_launch_lds_emit_retry(const FjLdsJoinArgs& a, hipStream_t s, bool only_marked = true);   // same for the emitting pass (FJ_STAT_EMIT_RETRY); only_marked = false: every item

// how many of `nsamples` evenly spaced probe rows have their key in the build side (final chunk set `build`, partition id =
// (hash word 1 >> shift32) & pmask): one wave per sample scans the sample's build partition
hipError_t fj_launch_sample_hits(const FjChunkSet& build, const u64* pk, u64 np, u32 nsamples, u32 shift32, u32 pmask,
                                 unsigned long long* hits, hipStream_t s);

// many-to-many join of one work item's partition (csrc/fj_many.hip): counting (part_count / total) or emitting at out_off
hipError_t fj_launch_mm_join(const FjLdsJoinArgs& a, bool materialize, hipStream_t s);

struct FjGtArgs {                // global (non-partitioned) table
    u64* tkeys; u64* tvals; u32* bloom;    // bloom == nullptr: no precheck
    u64 cap_mask;                           // capacity - 1 (capacity = power of two, multiple of 8)
    u32* flags;                             // [0] = build saw FJ_EMPTY_KEY, value in empty_val
    u64* empty_val;
    const u64* bk; const u64* bv; u64 nb;
    const u64* pk; u64 np;
    u32* wg_count;                          // [grid] matches per workgroup
    unsigned long long* total;
    const u64* out_off;                     // [grid+1]
    u64* out_keys; u64* out_vals;
};
hipError_t fj_launch_gt_build(const FjGtArgs& a, hipStream_t s);
hipError_t fj_launch_gt_probe(const FjGtArgs& a, bool materialize, u32 grid, hipStream_t s);

// ---- multi-GPU owner split --------------------------------------------------------------------
hipError_t fj_launch_owner_hist(const u64* keys, u64 n, u32 nranks, unsigned long long* counts, hipStream_t s);
hipError_t fj_launch_owner_scatter(const u64* keys, const u64* vals, u64 n, u32 nranks,
                                   const unsigned long long* offsets, unsigned long long* cursors,
                                   u64* out_keys, u64* out_vals, hipStream_t s);

// ---- synthetic data ---------------------------------------------------------------------------
hipError_t fj_launch_iota(u64* out, u64 n, hipStream_t s);
hipError_t fj_launch_gen_build(u64* keys, u64* vals, u64 first, u64 n, hipStream_t s);
hipError_t fj_launch_gen_probe(u64* keys, u64 first, u64 n, u64 build_total, u64 seed, u32 hit_bp,
                               unsigned long long* expected_hits, hipStream_t s);

// owner GPU of a key: range reduction of the top 16 bits of hash word 1
__host__ __device__ inline u32 fj_owner_of_w1(u32 w1, u32 nranks) { return ((w1 >> 16) * nranks) >> 16; }
