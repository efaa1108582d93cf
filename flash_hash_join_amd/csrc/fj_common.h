// fj_common.h -- shared device/host definitions for the MI355X (gfx950) hash-join kernels.
//
// Everything here is integer / indexing work bounded by HBM bandwidth: no MFMA.
// Wavefront = 64 lanes, LDS = 160 KiB per CU, 256 CUs in 8 XCDs.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint64_t u64;
typedef uint32_t u32;
typedef uint16_t u16;

// ---- chunked-bucket geometry -------------------------------------------------------------
// Partition passes write block-private *chunks*: 256 keys = 2 KiB, always 2-KiB aligned, so a
// consumer reads whole 128-B lines.  A directory word per chunk says which bucket it belongs
// to and how many keys it holds:  dir = (bucket << 9) | count   (count 1..256).
#define FJ_CHUNK_LOG 8
#define FJ_CHUNK (1u << FJ_CHUNK_LOG)          // keys per chunk
#define FJ_DIR_INVALID 0xFFFFFFFFu
#define FJ_DIR_CNT_BITS 9
#define FJ_DIR_CNT_MASK 0x1FFu
// chunk-list entry = ((count - 1) << 24) | chunk id   (ids < 2^24), so a consumer needs no directory lookup
#define FJ_LIST_ID(e) ((e) & 0xFFFFFFu)
#define FJ_LIST_CNT(e) (((e) >> 24) + 1u)
#define FJ_MAX_FAN_LOG 9                         // widest pass: 512 buckets
// Wire format of the owner shuffle (multi-GPU; csrc/fj_pack.hip): a chunk of a first-pass bucket need not carry the hash bits
// the bucket implies.  First passes of >= 256 buckets ship 7 bytes per key - three planes per 256-key chunk: the low words
// (1024 B), bits 32..47 (512 B), bits 48..55 (256 B); bits 56..63 = bucket >> (fan_log - 8) - and narrower first passes whole
// 8-byte words.
#define FJ_WIRE7_BYTES 1792u
#define FJ_WIRE7_MID 1024u
#define FJ_WIRE7_HI 1536u
// A bucket takes its chunk ids in aligned runs of 2^FJ_RUN_LOG consecutive ids (fj_partition_kernel), so that the level
// bookkeeping places a run's chunk-list entries with one set of gathers (fj_level_lists, FjChunkSet::run_log).
#ifndef FJ_RUN_LOG
#define FJ_RUN_LOG 2
#endif
// chunk ids a workgroup takes from the pool's allocator at a time (it takes several slabs in one piece when a tile opens more
// chunks than that: the first tile of a segment opens a run per bucket); what is left of a workgroup's last slab stays unused,
// so a level that is appended to by many launches (streamed pieces) takes small slabs.  1024 against 256 at c3: the passes
// ~0.5 % faster (longer contiguous stretches per producer), 64: the chunk-list pass 2 % slower.
static inline unsigned fj_slab_for(unsigned appends) { return appends > 1u ? 256u : 1024u; }
#define FJ_MAX_FANOUT 256u                      // buckets per partition pass (8 bits, as RADIX_BITS)

// LDS-resident join table (per final partition): 8192 slots, 8-B keys (+ 8-B values when
// materialising).  Linear probing, home slot aligned to a 4-slot group so that the probe
// compares 4 keys per LDS access (two ds_read_b128).
#define FJ_LDS_SLOTS_LOG 13
#define FJ_LDS_SLOTS (1u << FJ_LDS_SLOTS_LOG)
#define FJ_LDS_GROUP 4u
#define FJ_PART_TARGET_KEYS 4096u               // average build keys per final partition (load <= 0.5)
#define FJ_PLAN_BUMP_KEYS 3950u                 // ... above this average the plan takes one more radix bit when that is free (c2's 3906 stays: 0.79 ms with 256 partitions, 0.81 with 512)

// Bloom precheck of the partitioned join (csrc/fj_bloom.hip, fj_bloom_dev.h): an LDS-resident blocked Bloom filter over one
// bucket of an intermediate partition level: 35840 32-bit words = 140 KiB of the CU's 160 KiB (the rest stages survivors).
#define FJ_BLOOM_WORDS 35840u
#define FJ_BLOOM_BITS (FJ_BLOOM_WORDS * 32u)
#define FJ_BLOOM_MAX_KEYS 400000u                // build keys per filtered bucket above which the filter is not worth running (< 3 bits per key)
#define FJ_BLOOM_GOOD_KEYS 215000u               // ... below which it is strong (>= 5.5 bits per key): the plan widens its first pass to get here

#define FJ_PREFILTER_BITS 9                      // sender-side precheck of the owner shuffle: radix bits = 512 filters per owner
#define FJ_SAMPLE_KEYS 4096u                     // probe rows sampled by the adaptive joins to decide on the precheck

// Global (HBM / Infinity-Cache resident) table for the non-partitioned path: groups of 8 keys
// = one 64-B sector.
#define FJ_GT_GROUP 8u

// All 2^64 key values are legal, so the empty marker is handled out of band: a build key equal
// to FJ_EMPTY_KEY is never stored in a table; a per-table flag + value records it instead.
#define FJ_EMPTY_KEY 0xFFFFFFFFFFFFFFFFull

// Device hash = a BIJECTION of the 64-bit key.  The reference hashes with CRC32C*const (hash_join.cpp:40-44), 32 bits of
// entropy; join results are hash-independent, so the device uses its own mixer - and because equality is all a join needs,
// every chunk pool holds the MIXED key H = fj_key_mix(k) instead of k (flat caller arrays hold raw keys: a kernel mixes a key
// once, when it reads it from a flat array, and un-mixes it where a materialising join writes a probe key out):
//   a = lo*K1, b = hi*K2                      (32-bit multiplies only: a 64-bit multiply costs 4+ VALU issues on gfx950)
//   w1 = H >> 32  = fmix32(a ^ b)             -> owner GPU (top 16 bits) and radix digits, taken from the top
//   w2 = (u32)H   = fmix32(a + w1)            -> slot inside a partition's table, tag and bloom bits (low bits)
// (a, b) <-> (w1, w2) is one-to-one: fmix32 is a bijection of 32-bit words, x = a ^ b gives b from a, and a = fmix32^-1(w2) - w1.
// What that buys: later passes, the join and the filter kernel take digits, slots and bits straight from the stored word (no
// multiplies after the first read), and a chunk of a radix bucket need not store the digits the bucket implies - the owner
// shuffle ships 7 bytes per key (csrc/fj_pack.hip).
__host__ __device__ __forceinline__ u32 fj_fmix32(u32 x) {
    x ^= x >> 16; x *= 0x85ebca6bu;
    x ^= x >> 13; x *= 0xc2b2ae35u;
    x ^= x >> 16;
    return x;
}
__host__ __device__ __forceinline__ u32 fj_fmix32_inv(u32 x) {
    x ^= x >> 16; x *= 0x7ed1b41du;                 // 0xc2b2ae35^-1 mod 2^32
    x ^= (x >> 13) ^ (x >> 26); x *= 0xa5cb9243u;   // 0x85ebca6b^-1
    x ^= x >> 16;
    return x;
}
__host__ __device__ __forceinline__ u64 fj_key_mix(u64 k) {
    const u32 a = (u32)k * 0x9E3779B1u, b = (u32)(k >> 32) * 0x85EBCA77u;
    const u32 w1 = fj_fmix32(a ^ b);
    const u32 w2 = fj_fmix32(a + w1);
    return ((u64)w1 << 32) | w2;
}
__host__ __device__ __forceinline__ u64 fj_key_unmix(u64 h) {
    const u32 w1 = (u32)(h >> 32), w2 = (u32)h;
    const u32 a = fj_fmix32_inv(w2) - w1, b = fj_fmix32_inv(w1) ^ a;
    return ((u64)(b * 0xb6c92f47u) << 32) | (a * 0x0e8b2f51u);      // K2^-1, K1^-1
}
// the two hash words of a MIXED key (what chunk pools hold)
#define FJ_HW1(h) ((u32)((h) >> 32))
#define FJ_HW2(h) ((u32)(h))
// ... and of a raw key (flat arrays: the owner split, the HBM-table path); the unused word is dead-code-eliminated
__host__ __device__ __forceinline__ u32 fj_hash_w1(u64 k) {
    const u32 a = (u32)k * 0x9E3779B1u, b = (u32)(k >> 32) * 0x85EBCA77u;
    return fj_fmix32(a ^ b);
}
__host__ __device__ __forceinline__ u64 fj_hash64(u64 k) { return fj_key_mix(k); }

// splitmix64-style counter hash used by the synthetic generators; identical in
// NumPy (flash_hash_join_amd/datagen.py), C and HIP.
__host__ __device__ __forceinline__ u64 fj_mix(u64 seed, u64 j) {
    u64 z = seed * 0xD6E8FEB86659FD93ull + j + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
#define FJ_GOLDEN 0x9E3779B97F4A7C15ull

struct __attribute__((aligned(16))) u64x2 { u64 x, y; };

__device__ __forceinline__ u32 fj_lane() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// wave64 sum of a u32
__device__ __forceinline__ u32 fj_wave_sum(u32 v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}
