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

// Device hash.  The reference hashes with CRC32C*const (hash_join.cpp:40-44), 32 bits of entropy;
// join results are hash-independent, so the device uses its own mixer.  Both hot kernels are
// instruction-issue bound on gfx950 (PMC: profiles/r01_v2_c3_pmc_summary.txt), and a 64-bit
// multiply costs 4+ VALU issues, so the hash is built from 32-bit multiplies only:
//   a = lo*K1, b = hi*K2
//   w1 = fmix32(a ^ b)                  -> owner GPU (top 16 bits) and radix digits, taken from the top
//   w2 = fmix32(a + rotl(b,16) + K3)    -> slot inside a partition's table (low bits)
// A kernel that needs only one word gets the other one dead-code-eliminated.
__host__ __device__ __forceinline__ u32 fj_fmix32(u32 x) {
    x ^= x >> 16; x *= 0x85ebca6bu;
    x ^= x >> 13; x *= 0xc2b2ae35u;
    x ^= x >> 16;
    return x;
}
__host__ __device__ __forceinline__ u32 fj_hash_w1(u64 k) {
    const u32 a = (u32)k * 0x9E3779B1u, b = (u32)(k >> 32) * 0x85EBCA77u;
    return fj_fmix32(a ^ b);
}
__host__ __device__ __forceinline__ u32 fj_hash_w2(u64 k) {
    const u32 a = (u32)k * 0x9E3779B1u, b = (u32)(k >> 32) * 0x85EBCA77u;
    return fj_fmix32(a + ((b << 16) | (b >> 16)) + 0x27D4EB2Fu);
}
__host__ __device__ __forceinline__ u64 fj_hash64(u64 k) { return ((u64)fj_hash_w1(k) << 32) | fj_hash_w2(k); }

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
