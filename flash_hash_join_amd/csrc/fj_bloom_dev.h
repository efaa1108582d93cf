// fj_bloom_dev.h -- device-side primitives of the LDS-resident blocked Bloom filter, shared by the stand-alone filter kernel
// (csrc/fj_bloom.hip: sender-side precheck of the owner shuffle, wide last passes) and by the filtering form of the partition
// pass (csrc/fj_partition.hip: filter test fused into the probe side's last radix pass).
//
// Role of the reference's bloom directory (hash_join.cpp:60-74 tag table, :122 / :142 insert side, :165 precheck, :183-189
// get_bloom_tag / check_bloom_filter).  No false negatives by construction: insert and test use the same bits.
#pragma once
#include "fj_internal.h"

namespace {

constexpr u32 BF_NT = 1024;                 // threads per workgroup of every kernel that holds a filter (one workgroup per CU)

__device__ __forceinline__ u32 bf_uni(u32 v) { return (u32)__builtin_amdgcn_readfirstlane((int)v); }

// Filter position of a key: byte offset of its LDS word / block and the mask to test or set.
//   VAR 0 (default): the join's hash word 2 = the low word of the mixed key as stored (round 4: no multiply of its own), 4 bits in one 64-bit block.
//   VAR 1: two-multiply mixer over a fold of the key, 2 bits in one 32-bit word.
//   VAR 2: the two-multiply mixer, 4 bits in one 64-bit block (2 per half): three multiplies fewer per key than VAR 0 and
//          the same false-positive rate on the generator's keys - and the same kernel time (round 3, 2.26 ms either way at
//          config 4: the filter kernel is not bound by its hash), so the better-mixed VAR 0 stays the default.
//   (A mixer from 24-bit multiplies - full rate - spread (i+1)*M keys so badly that 32 % of the probe keys passed at 5 % hits.)
// All are independent of the radix digits (hash word 1); a weak spot costs false positives, never a wrong result.
template <int VAR>
__device__ __forceinline__ void bf_bits(u64 key, u32& byte_off, u32& mlo, u32& mhi) {
    if (VAR == 0) {
        // hash word 2 of the MIXED key as it is stored (no multiply besides the block index): block from the top ~14 bits,
        // bit positions from bits 0..17 (positions taken from bits the block half-determines cost false positives: VAR 2's note)
        const u32 w = FJ_HW2(key);
        byte_off = __umulhi(w, FJ_BLOOM_WORDS / 2u) * 8u;
        mlo = (1u << (w & 31u)) | (1u << ((w >> 5) & 31u));
        mhi = (1u << ((w >> 9) & 31u)) | (1u << ((w >> 13) & 31u));
        return;
    }
    u32 x = (u32)key ^ __builtin_rotateleft32((u32)(key >> 32), 15);
    x *= 0x9E3779B1u; x ^= x >> 15;
    x *= 0x85EBCA77u; x ^= x >> 13;
    if (VAR == 1) {
        byte_off = __umulhi(x, FJ_BLOOM_WORDS) * 4u;
        mlo = (1u << (x & 31u)) | (1u << ((x >> 5) & 31u));
        mhi = 0;
        return;
    }
    // the block comes from the top ~14 bits of x, so the bit positions stay below bit 18: with positions from bits 10..19 the
    // last one was half-determined by the block and 11.3 % of the misses passed instead of 8.9 % (VAR 0: 8.7 %; simulation
    // on the generator's keys, 196K build keys per filter)
    byte_off = __umulhi(x, FJ_BLOOM_WORDS / 2u) * 8u;
    mlo = (1u << (x & 31u)) | (1u << ((x >> 5) & 31u));
    mhi = (1u << ((x >> 9) & 31u)) | (1u << ((x >> 13) & 31u));
}
template <int VAR>
__device__ __forceinline__ void bf_insert(unsigned char* filt, u64 key) {
    u32 o, mlo, mhi;
    bf_bits<VAR>(key, o, mlo, mhi);
    if (VAR != 1) atomicOr(reinterpret_cast<unsigned long long*>(filt + o), ((unsigned long long)mhi << 32) | mlo);
    else atomicOr(reinterpret_cast<u32*>(filt + o), mlo);
}
// the filter word(s) of a key, and the test against them (split so that a caller can keep several LDS reads in flight)
template <int VAR>
__device__ __forceinline__ void bf_fetch(const unsigned char* filt, u32 byte_off, u32& wlo, u32& whi) {
    if (VAR != 1) { const u64 w = *reinterpret_cast<const u64*>(filt + byte_off); wlo = (u32)w; whi = (u32)(w >> 32); }
    else { wlo = *reinterpret_cast<const u32*>(filt + byte_off); whi = 0; }
}
template <int VAR>
__device__ __forceinline__ bool bf_pass(u32 wlo, u32 whi, u32 mlo, u32 mhi) {
    return (wlo & mlo) == mlo && (VAR == 1 || (whi & mhi) == mhi);
}

// Build the LDS filter of one bucket from the build relation's chunks of that bucket (all BF_NT threads of the workgroup; ends
// with a barrier).  32 chunks = 8192 build keys per step: the step's four list entries per thread, then its four 16-B key
// loads; the entries of the NEXT step are requested before this step's keys are used.  A wave's four loads of a step each
// cover half of ONE chunk (chunk i*8 + tid/128, byte offset (tid%128)*16), so chunk ids and counts are wave-uniform.
template <int VAR>
__device__ __forceinline__ void bf_build_filter(unsigned char* filt, const u64* __restrict__ bkeys, const u32* __restrict__ blist,
                                                const u32* __restrict__ bboff, u32 bucket, u32 tid) {
    const u32 jb = bf_uni(tid >> 7), off = (tid & 127u) * 2u, off16 = (tid & 127u) * 16u;
    for (u32 i = tid; i < FJ_BLOOM_WORDS / 2; i += BF_NT) reinterpret_cast<u64*>(filt)[i] = 0;
    __syncthreads();
    const u32 b0 = bboff[bucket], nbc = bboff[bucket + 1] - b0;
    auto bentries = [&](u32 c0, u32 (&e)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const u32 j = c0 + (u32)i * 8u + jb;
            e[i] = blist[b0 + (j < nbc ? j : (nbc ? nbc - 1 : 0u))];
        }
    };
    // software pipeline: the keys of step s+1 and the entries of step s+2 are requested before the keys of step s are inserted
    // (a filter of ~195K keys is 24 steps: with the key loads issued only after the previous step's inserts every step paid
    // a full memory latency while the CU's only workgroup did nothing else)
    u32 be[4];
    u64x2 q[4]; u32 cn[4];
    auto bkeys_of = [&](u32 c0, const u32 (&e)[4], u64x2 (&qq)[4], u32 (&cc)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const u32 eu = bf_uni(e[i]);
            cc[i] = (c0 + (u32)i * 8u + jb) < nbc ? FJ_LIST_CNT(eu) : 0u;
            const unsigned char* base = reinterpret_cast<const unsigned char*>(bkeys + (u64)FJ_LIST_ID(eu) * FJ_CHUNK);
            qq[i] = *reinterpret_cast<const u64x2*>(base + off16);
        }
    };
    if (nbc) {
        bentries(0, be);
        bkeys_of(0, be, q, cn);
        bentries(32 < nbc ? 32 : 0, be);
    }
    for (u32 c0 = 0; c0 < nbc; c0 += 32) {
        u64x2 qn[4]; u32 cnn[4] = {0u, 0u, 0u, 0u};
        const bool more = c0 + 32 < nbc;
        if (more) bkeys_of(c0 + 32, be, qn, cnn);
        u32 bn[4];
        bentries(c0 + 64 < nbc ? c0 + 64 : c0, bn);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (off < cn[i]) bf_insert<VAR>(filt, q[i].x);
            if (off + 1 < cn[i]) bf_insert<VAR>(filt, q[i].y);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) { be[i] = bn[i]; cn[i] = cnn[i]; if (more) q[i] = qn[i]; }
    }
    __syncthreads();
}

}  // namespace
