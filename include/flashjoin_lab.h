/*
 * flashjoin_lab.h -- the building blocks BEHIND the C ABI of include/flashjoin.h: the entry points the multi-GPU driver
 * (csrc/fj_dist.hip) is made of, the stream joins, the sender-side prechecks, the partition diagnostic.  They are NOT part of the
 * drop-in boundary: libflashjoin_hip.so does not export them (csrc/exports.map).  The same objects linked without that export
 * list are libflashjoin_hip_lab.so, which the test-suite and the measurement tools load (flash_hash_join_amd/_lib.py,
 * FJ_LIB_VARIANT=lab; flash_hash_join_amd/lab.py) to exercise these pieces one by one.
 */
#ifndef FLASHJOIN_LAB_H
#define FLASHJOIN_LAB_H
#include "flashjoin.h"

#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__)
#pragma GCC visibility push(default)       /* visible in the objects; the product library's export list hides them again */
#endif

/* The two halves of fj_owner_split: fj_owner_hist counts rows per owner (synchronous: h_counts is valid on return),
 * fj_owner_scatter writes the owner-contiguous segments given those counts (asynchronous, ordered on `stream`). */
int fj_owner_hist(fj_ctx* ctx, const uint64_t* d_keys, size_t n, int nranks, uint64_t* h_counts, void* stream);
int fj_owner_scatter(fj_ctx* ctx, const uint64_t* d_keys, const uint64_t* d_vals, size_t n, int nranks, const uint64_t* h_counts,
                     uint64_t* d_out_keys, uint64_t* d_out_vals, void* stream);

/*
 * Owner shuffle in the shape of SURVEY.md 8(e) (no reference counterpart; radix partitions are independent join units,
 * hash_join.cpp:340-356, :515-525): the FIRST radix pass of the plan for the TOTAL build side is the owner split.
 * Every rank plans for nb_total = all ranks' build rows; pass 1 of that plan has 2^fan_log0 buckets and bucket b belongs to
 * rank (b * nranks) >> fan_log0.  A sender runs that pass over its local rows and rewrites the result for the wire: dense
 * 256-key chunks, bucket after bucket - so an owner's share is one contiguous range - of fj_shuffle_chunk_bytes() bytes each:
 * 1792 = 7 bytes per key when the first pass has >= 256 buckets (chunk pools hold a bijective mix of the key, fj_key_mix64,
 * and a chunk need not carry the 8 top bits its bucket implies: three planes - low words, bits 32..47, bits 48..55), else
 * 2048; plus one directory word (bucket << 9 | keys) per chunk.
 *   fj_shuffle_plan        - 0 if the chunk form applies (a plan of two or more passes, at least nranks first-pass buckets:
 *                            build sides above ~2M rows in all); else an error (use fj_owner_split + fj_stream_begin).
 *   fj_shuffle_chunk_bytes - bytes per wire chunk under that plan (0: the chunk form does not apply).
 *   fj_shuffle_pack_begin  - asynchronous on `stream`: the first pass over n local rows (values too when d_vals != NULL), its
 *                            bookkeeping, and the per-owner chunk counts on their way to the host.  May run while a stream
 *                            join is open on the context, on another stream.  One piece at a time per context.
 *   fj_shuffle_pack_counts - waits for those counts: h_used[r] = wire chunks for rank r (exact: skewed keys just make a share
 *                            larger).  The caller sizes its buffers and tells the receivers.
 *   fj_shuffle_pack_finish - asynchronous on `stream` (the same stream, or one ordered behind it): writes rank r's chunks to
 *                            d_dst_chunks[r] (h_used[r] * chunk_bytes bytes, 16-byte aligned), their directory words to
 *                            d_dst_dir[r] and - pieces with values - 256 values per chunk to d_dst_vals[r].  The pointers may
 *                            point anywhere: a rank's own share can go straight into its receive buffer.
 * The owner appends what it received - every sender's share concatenated, in any order - as level-1 chunk sets:
 *   fj_stream_open_shuffled        - as fj_stream_open, for the rows this rank will OWN (bounds; appends = pieces per side)
 *   fj_stream_append_build_chunks /
 *   fj_stream_append_probe_chunks  - chunk lists from the directory words (rewritten in place), then the plan's second pass
 *                                    over the piece (it unpacks the wire format in registers); asynchronous on `stream`, the
 *                                    piece stays allocated until the finish
 *   fj_stream_finish               - remaining passes, join, count.
 * Sender-side precheck in this form (role of the reference's bloom directory, hash_join.cpp:60-74, :122, :183-189, moved in front
 * of the exchange): an owner whose build side is complete writes one fj_shuffle_part_filter_bytes() (4 KiB) Bloom filter per FINAL
 * partition it owns (fj_stream_export_part_filters: fj_shuffle_part_filter_range tells every rank where an owner's filters sit in
 * the array of all partitions' filters - 1 byte per build key in all - which the caller assembles: an all-gather).  A sender then
 * starts a probe piece with defer_plan = 1 - only the first pass is queued - and continues it with fj_shuffle_pack_filter once
 * the filters have arrived: the piece's level-1 chunks are compacted in place to the keys some filter admits (the filters of one
 * level-1 bucket are 2 MiB and stay in L2 while its chunks stream by), then counts / copy / exchange as before over fewer keys.
 * fj_shuffle_pack_kept (after fj_shuffle_pack_counts) = the keys the piece kept.  fj_part_filter_sample tests every stride-th of n
 * RAW probe keys (synchronous): the share that would travel.  No key of the build side is ever dropped.
 * Materialising joins (_hash_join_radix_materialize, hash_join.cpp:315-381, across GPUs): open with with_vals = 1 and append the
 * build side with its values (256 per chunk, as fj_shuffle_pack_finish wrote them; d_vals == NULL otherwise); fj_stream_finish
 * then returns the count and leaves the partitions resident, fj_emit_pairs writes this owner's (probe_key, build_value) pairs -
 * they stay with the owner.  Duplicate build keys are refused there (first-occurrence semantics need the flat
 * build arrays): the owner-scatter form serves them.
 */
size_t fj_shuffle_chunk_bytes(size_t nb_total, int nranks);
int fj_shuffle_pack_begin(fj_ctx* ctx, const uint64_t* d_keys, const uint64_t* d_vals, size_t n, size_t nb_total, int nranks, int defer_plan, void* stream);
int fj_shuffle_pack_counts(fj_ctx* ctx, uint64_t* h_used);
int fj_shuffle_pack_finish(fj_ctx* ctx, void* const* d_dst_chunks, uint64_t* const* d_dst_vals, uint32_t* const* d_dst_dir, void* stream);
int fj_stream_open_shuffled(fj_ctx* ctx, size_t nb_total, int nranks, int rank, size_t nb_bound, int build_appends,
                            size_t np_bound, int probe_appends, int with_vals, void* stream);
int fj_stream_append_build_chunks(fj_ctx* ctx, const void* d_chunks, const uint64_t* d_vals, uint32_t* d_dir, size_t nchunks, void* stream);
int fj_stream_append_probe_chunks(fj_ctx* ctx, const void* d_chunks, uint32_t* d_dir, size_t nchunks, void* stream);
size_t fj_shuffle_part_filter_bytes(void);
int fj_shuffle_part_filter_range(size_t nb_total, int nranks, int rank, size_t* first_part, size_t* n_parts, size_t* total_parts);
int fj_stream_export_part_filters(fj_ctx* ctx, void* d_out, void* stream);
int fj_shuffle_pack_filter(fj_ctx* ctx, const void* d_part_filters, void* stream);
uint64_t fj_shuffle_pack_kept(fj_ctx* ctx);
int fj_part_filter_sample(fj_ctx* ctx, const uint64_t* d_raw_keys, size_t n, size_t stride, const void* d_part_filters, size_t nb_total, int nranks,
                          void* stream, uint64_t* kept);

/*
 * Multi-GPU counting join, BUILD-BROADCAST form (csrc/fj_bcast.hip; no reference counterpart: the reference is one process,
 * hash_join.cpp:318 - what is exploited is that radix partitions are independent join units, :340-356, :515-525): the probe
 * rows never move.  Every rank plans for the TOTAL build side, runs both passes of that plan over its own build rows and packs
 * them densely, final partition after final partition, without the bits a partition implies (6 bytes per key from 134M build rows
 * in all): region = offset table u32[nparts + 1] | low words u32[n] | rest of the high words u16[n] or u32[n].  The regions are
 * exchanged (fj_dist_join does that; any transport can: fj_bcast_piece_span says which bytes form a piece of consecutive
 * partitions), and every rank joins its own probe rows, partitioned by the same plan, against the runs of all ranks where they lie.
 *   fj_bcast_plan          - bits / final partitions / bytes per key of the high-word plane for a total build side (error: the
 *                            plan has no pass - such joins take the owner-scatter form)
 *   fj_bcast_region_bytes  - bytes of a rank's region holding nkeys keys (0: no such plan)
 *   fj_bcast_pack          - asynchronous: local build rows -> d_region (16-byte aligned, fj_bcast_region_bytes(nb_total, nb) bytes);
 *                            starts the step on this context
 *   fj_bcast_pack_bounds   - blocks until the pack has run: h_bounds[q] = first key index of piece q, h_bounds[pieces] = nb
 *   fj_bcast_probe         - asynchronous: local probe rows through the plan's passes
 *   fj_bcast_join          - asynchronous: partitions [part_lo, part_hi) of the local probe rows against nsrc (<= 16) regions inside
 *                            d_base (region i of nkeys[i] keys at byte region_off[i]); call once per landed piece
 *   fj_bcast_finish        - blocks; the local match count.  A final partition beyond the LDS table (skewed build keys) is an error:
 *                            the caller takes another form (fj_dist_join does).
 * MATERIALISING joins (_hash_join_radix_materialize, hash_join.cpp:315-381, across GPUs; round 6): fj_bcast_pack with with_vals = 1
 * writes the build values as a fourth part of the region (u64[n]: 14 bytes per build row; fj_bcast_region_bytes / _piece_span with
 * with_vals = 1 / part 3); fj_bcast_join counts with the counting step's kernel (a probe row counts once however many copies of its
 * key the build side holds; partitions of more than ~6000 keys are refused: they must fit the pair writer's table); fj_bcast_finish
 * leaves the regions and the probe partitions in place and fj_emit_pairs (fj_bcast_emit) writes this rank's pairs - (probe key, value
 * of one copy of the build key) of its OWN probe rows: in this form the pairs stay with the probe rows.
 */
size_t fj_bcast_region_bytes(size_t nb_total, size_t nkeys, int with_vals);
int fj_bcast_piece_span(size_t nb_total, size_t nkeys, size_t k_lo, size_t k_hi, int part, size_t* offset, size_t* bytes);
int fj_bcast_pack(fj_ctx* ctx, const uint64_t* d_build_keys, const uint64_t* d_build_vals, size_t nb, size_t nb_total, void* d_region, int pieces, int with_vals, void* stream);
int fj_bcast_pack_bounds(fj_ctx* ctx, uint64_t* h_bounds);
int fj_bcast_probe(fj_ctx* ctx, const uint64_t* d_probe_keys, size_t np, size_t nb_total, void* stream);
int fj_bcast_join(fj_ctx* ctx, const void* d_base, int nsrc, const uint64_t* region_off, const uint64_t* nkeys, uint32_t part_lo, uint32_t part_hi, void* stream);
int fj_bcast_finish(fj_ctx* ctx, void* stream, uint64_t* out_count, fj_timings* timings);
void fj_bcast_abort(fj_ctx* ctx);
int fj_bcast_emit(fj_ctx* ctx, uint64_t* d_out_keys, uint64_t* d_out_vals, size_t out_capacity, void* stream);

/*
 * Sender-side bloom precheck of the owner shuffle (no reference counterpart).  fj_bloom_export: an owner partitions the
 * nb build keys it owns by 9 radix bits (at the hash_top_bits it will join with) and writes one Bloom filter per bucket,
 * fj_bloom_filter_words() 32-bit words each, 512 buckets, into d_filters (the caller all-gathers them).  fj_bloom_prefilter:
 * a peer tests the n probe keys it is about to send to that owner against the owner's filters and writes the keys that may
 * match, densely, into d_out_keys (capacity >= n); *out_n = how many (synchronous).  No key that is in the owner's build
 * side is ever dropped.
 */
size_t fj_bloom_filter_words(void);        /* words per owner = 512 * words per bucket + 4 header words (the bloom_variant
                                              the filters were built with: fj_bloom_prefilter refuses another one)      */
int fj_bloom_export(fj_ctx* ctx, const uint64_t* d_build_keys, size_t nb, int hash_top_bits, uint32_t* d_filters, void* stream);
int fj_bloom_prefilter(fj_ctx* ctx, const uint64_t* d_probe_keys, size_t n, int hash_top_bits, const uint32_t* d_filters,
                       uint64_t* d_out_keys, size_t out_capacity, uint64_t* out_n, void* stream);

/*
 * A counting radix join whose relations arrive in pieces (multi-GPU: the pieces of an exchange).  No reference
 * counterpart; same result as fj_join_device(FJ_ALGO_RADIX, 0, 0, ...) on the concatenations.
 *
 * fj_stream_open sizes the plan for at most nb_bound build rows (in <= build_appends pieces) and np_bound probe rows
 * (in <= probe_appends pieces).  Every fj_stream_append_build / fj_stream_append_probe runs the first partition pass
 * over one piece (asynchronous on `stream`; a piece must stay allocated until fj_stream_finish returns); the two
 * sides may be appended in any order.  fj_stream_advance_probe closes the probe side and runs its remaining passes at
 * once, so that they overlap an exchange of the build side.  fj_stream_finish runs whatever remains, then the join,
 * and returns the match count.  A build side of <= 4096 rows (zero-pass plan) must arrive in one piece.
 *
 * fj_stream_begin = fj_stream_open + one fj_stream_append_build of the whole build side + its remaining passes
 * (the build side is complete before the probe pieces arrive: the owner-shuffle exchange).
 */
int fj_stream_open(fj_ctx* ctx, size_t nb_bound, int build_appends, size_t np_bound, int probe_appends, void* stream,
                   int hash_top_bits);
int fj_stream_append_build(fj_ctx* ctx, const uint64_t* d_build_keys, size_t n, void* stream);
int fj_stream_advance_probe(fj_ctx* ctx, void* stream);
int fj_stream_begin(fj_ctx* ctx, const uint64_t* d_build_keys, const uint64_t* d_build_vals, size_t nb, size_t np_bound,
                    int max_appends, void* stream, int hash_top_bits);
int fj_stream_append_probe(fj_ctx* ctx, const uint64_t* d_probe_keys, size_t n, void* stream);
int fj_stream_finish(fj_ctx* ctx, void* stream, uint64_t* out_count, fj_timings* timings);

/* While another library's kernels are resident on this GPU for the length of a step (RCCL's send / receive kernels during an exchange), the
 * partition passes and the wide join - one workgroup per CU, each wanting a whole CU - launch num_cus - n workgroups.  fj_dist_join does this
 * itself over RCCL (n = 32, FJ_DIST_RESERVE_CUS); a host that drives fj_bcast_* / fj_shuffle_* over its own GPU-side transport calls it.  n = 0
 * restores the full grid. */
void fj_ctx_reserve_cus(fj_ctx* ctx, unsigned n);

/*
 * Diagnostic for the test-suite: runs total_bits (2..24) of radix partitioning over a flat
 * device relation and writes the final per-bucket chunk lists, linearised bucket by bucket, into
 * host arrays of n rows (h_out_vals may be NULL when d_vals is NULL).  *h_nvalid = rows written.
 */
int fj_debug_partition(fj_ctx* ctx, const uint64_t* d_keys, const uint64_t* d_vals, size_t n, int total_bits,
                       int hash_top_bits, void* stream, uint64_t* h_out_keys, uint64_t* h_out_vals,
                       uint32_t* h_bucket_of, uint64_t* h_nvalid);


#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* FLASHJOIN_LAB_H */
