// fj_bloom.hip -- bloom precheck of the partitioned join for MI355X (gfx950).
//
// Role of the reference's bloom directory (hash_join.cpp:60-74 tag table, :122 / :142 insert side, :165 precheck,
// :183-189 get_bloom_tag / check_bloom_filter): reject probe keys that cannot be in the build side before the
// expensive part of the lookup.  In the reference that is one extra random 2-byte read per probe; here the expensive
// part is not the lookup (an LDS table) but MOVING the probe key through the remaining partition pass and the join,
// so the precheck sits between the probe side's partition passes and drops rejected keys from the data flow.
//
// Where the filter lives was measured (tools/ubench_bloom_probe.hip, profiles/r02_ubench_bloom_probe.csv): streaming
// keys and testing each against a filter in LDS runs at the read ceiling (757 G keys/s); against a filter in L2 at
// 205-270 G keys/s (one divergent L1/L2 access per key), in the Infinity Cache / HBM at 51 G keys/s.  So the filter is
// an LDS-resident blocked Bloom filter over ONE bucket of an intermediate partition level, built on the fly:
//
//   * one 1024-thread workgroup per CU owns ~144 KiB of LDS = 18432 blocks of 64 bits;
//   * work = tiles of the probe side's level-L chunk lists (tile table, bucket by bucket); a workgroup takes a
//     contiguous run of tiles (snapped to bucket boundaries when they are close);
//   * when its bucket changes the workgroup rebuilds the filter from the BUILD side's chunks of that bucket
//     (<= ~400K keys, read once more from HBM/L2: 8 B per build key per workgroup that visits the bucket);
//   * probe keys stream through registers two tiles ahead (no barrier in the steady state), each key tests 4 bits of
//     one 64-bit block; survivors are compacted per wave (ballot + mbcnt) straight into wave-private 2-KiB chunks of
//     an output chunk pool with the same bucket structure, so the next partition pass reads it like any other level.
//
// No false negatives by construction (insert and test use the same bits); false positives only cost the work the
// filter would have saved.  Algorithmic HBM bytes: 8 B per probe key read + 8 B per survivor written (+ the build
// keys of the level once or twice).
#include "fj_internal.h"

namespace {

constexpr u32 BF_NT = 1024;                 // threads per workgroup (one workgroup per CU)
constexpr u32 BF_KPT = 8;                   // probe keys per thread and tile
constexpr u32 BF_T = BF_NT * BF_KPT;        // 8192 keys = 32 chunks per tile
constexpr u32 BF_SLAB = 16;                 // output chunks a wave takes per allocator hit

// block index + the two 32-bit halves of the key's 4-bit mask.  Hash word 2 is independent of the radix digits (word 1).
__device__ __forceinline__ void bf_bits(u64 key, u32& idx, u32& mlo, u32& mhi) {
    const u32 w = fj_hash_w2(key);
    const u32 h = w * 0x9E3779B1u;
    idx = __umulhi(w, FJ_BLOOM_BLOCKS);
    mlo = (1u << (h & 31u)) | (1u << ((h >> 5) & 31u));
    mhi = (1u << ((h >> 10) & 31u)) | (1u << ((h >> 15) & 31u));
}

__global__ __launch_bounds__(BF_NT, 1) void fj_bloom_filter_kernel(FjBloomArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u64* filt = reinterpret_cast<u64*>(smem);                // [FJ_BLOOM_BLOCKS]
    const u32 tid = threadIdx.x, lane = tid & 63;
    const u32 ntiles = *a.ntiles;
    const u32 G = gridDim.x, g = blockIdx.x;
    const u32 nb = a.probe.nb;

    // contiguous run of tiles [t0, t1); a boundary that falls close to a bucket boundary is moved onto it, so that with
    // balanced buckets no filter is built twice (both neighbours compute the same snapped value)
    auto boundary = [&](u32 gg) -> u32 {
        u32 t = (u32)(((u64)gg * ntiles) / G);
        if (gg == 0 || gg >= G || ntiles == 0) return gg >= G ? ntiles : t;
        u32 lo = 0, hi = nb;                                 // last bucket p with toff[p] <= t
        while (hi - lo > 1) { const u32 mid = (lo + hi) >> 1; if (a.toff[mid] <= t) lo = mid; else hi = mid; }
        const u32 b0 = a.toff[lo], b1 = a.toff[lo + 1], slack = (b1 - b0) >> 4;
        if (t - b0 <= slack) t = b0; else if (b1 - t <= slack) t = b1;
        return t;
    };
    const u32 t0 = boundary(g), t1 = boundary(g + 1);
    if (t0 >= t1) return;
    const u32 nmine = t1 - t0;

    // ---- input side: descriptors -> chunk-list entries -> keys, each a tile earlier than its consumer ------------
    // thread tid reads key pairs (i*NT + tid)*2, i = 0..3: chunk j = i*8 + tid/128 of the tile, offset (tid%128)*2
    const u32 jbase = tid >> 7, off = (tid & 127u) * 2u;
    struct Desc { u32 pos, len, bucket; };
    auto get_desc = [&](u32 tt) -> Desc {
        const uint4 d = a.tiles[t0 + (tt < nmine ? tt : nmine - 1)];
        Desc r; r.pos = d.x; r.len = tt < nmine ? d.y : 0u; r.bucket = d.z; return r;
    };
    auto get_entries = [&](const Desc& d, u32 (&e)[4], u32& vm) {      // unconditional loads (clamped), validity in vm
        vm = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const u32 j = (u32)i * 8u + jbase;
            const u32 jj = j < d.len ? j : (d.len ? d.len - 1 : 0u);
            e[i] = a.probe.list[d.pos + jj];
            vm |= (j < d.len ? 1u : 0u) << i;
        }
    };
    auto get_keys = [&](const u32 (&e)[4], u32 vm, u64 (&kk)[BF_KPT], u32& okm) {
        okm = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const u32 cnt = (vm >> i) & 1u ? FJ_LIST_CNT(e[i]) : 0u;
            const u64x2 q = *reinterpret_cast<const u64x2*>(a.probe.keys + (u64)FJ_LIST_ID(e[i]) * FJ_CHUNK + off);
            kk[2 * i] = q.x; kk[2 * i + 1] = q.y;
            okm |= ((off < cnt ? 1u : 0u) | (off + 1 < cnt ? 2u : 0u)) << (2 * i);
        }
    };

    // ---- output side: wave-private chunks ------------------------------------------------------------------------
    u32 cur = FJ_DIR_INVALID, fill = FJ_CHUNK;      // current chunk and its fill (wave-uniform)
    u32 nch = 0, seg = 0;                           // chunks of this (wave, bucket run) = one segment of the bucket's chunk list
    u32 slab_cur = 0, slab_rem = 0;
    unsigned long long survivors = 0;
    const u32 cap = a.cap_chunks;
    auto new_chunk = [&](u32 bucket) -> u32 {       // wave-uniform
        if (slab_rem == 0) {
            u32 base = 0;
            if (lane == 0) { base = atomicAdd(a.alloc, BF_SLAB); if (base + BF_SLAB > cap) atomicOr(a.err, FJ_ERR_POOL); }
            slab_cur = (u32)__builtin_amdgcn_readfirstlane((int)base); slab_rem = BF_SLAB;
        }
        const u32 c = slab_cur; ++slab_cur; --slab_rem;
        if (lane == 0 && c < cap) a.out_rel[c] = ((u64)seg << 32) | nch;
        ++nch;
        return c;
    };
    auto end_segment = [&](u32 bucket) {            // close this wave's run inside `bucket`
        if (nch > 0 && lane == 0) {
            if (cur < cap && fill > 0) a.out_dir[cur] = (bucket << FJ_DIR_CNT_BITS) | fill;
            const u32 o = atomicAdd(&a.bchunks[bucket], nch);
            if (seg < a.max_segs) a.seg_off[seg] = o;
        }
        cur = FJ_DIR_INVALID; fill = FJ_CHUNK; nch = 0;
    };

    // ---- prologue ----------------------------------------------------------------------------------------------------
    Desc dC = get_desc(2), dD = get_desc(3);
    u64 kA[BF_KPT], kB[BF_KPT];
    u32 okA, okB, eC[4], vmC;
    u32 bktA, bktB;
    {
        const Desc d0 = get_desc(0), d1 = get_desc(1);
        u32 e0[4], e1[4], v0, v1;
        get_entries(d0, e0, v0); get_entries(d1, e1, v1);
        get_entries(dC, eC, vmC);
        get_keys(e0, v0, kA, okA); get_keys(e1, v1, kB, okB);
        bktA = d0.bucket; bktB = d1.bucket;
    }
    u32 bktC = dC.bucket;

    u32 cur_bucket = 0xFFFFFFFFu;
    for (u32 t = 0; t < nmine; ++t) {
        // tile t's keys are in kA; request tile t+2's keys, tile t+3's entries, tile t+4's descriptor
        u64 kC[BF_KPT]; u32 okC;
        get_keys(eC, vmC, kC, okC);
        u32 eD[4], vmD;
        get_entries(dD, eD, vmD);
        const Desc dE = get_desc(t + 4);

        const u32 bucket = bktA;
        if (bucket != cur_bucket) {                 // workgroup-uniform: rebuild the filter for this bucket
            if (cur_bucket != 0xFFFFFFFFu) end_segment(cur_bucket);
            __syncthreads();                        // every wave is done testing against the old filter
            for (u32 i = tid; i < FJ_BLOOM_BLOCKS; i += BF_NT) filt[i] = 0;
            __syncthreads();
            const u32 b0 = a.build.boff[bucket], nbc = a.build.boff[bucket + 1] - b0;
            for (u32 c0 = 0; c0 < nbc; c0 += 32) {  // 32 chunks = 8192 build keys per step, four 16-B loads per thread in flight
                u64x2 q[4]; u32 cn[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const u32 j = c0 + (u32)i * 8u + jbase;
                    const u32 e = a.build.list[b0 + (j < nbc ? j : nbc - 1)];
                    cn[i] = j < nbc ? FJ_LIST_CNT(e) : 0u;
                    q[i] = *reinterpret_cast<const u64x2*>(a.build.keys + (u64)FJ_LIST_ID(e) * FJ_CHUNK + off);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    u32 idx, mlo, mhi;
                    if (off < cn[i]) { bf_bits(q[i].x, idx, mlo, mhi); atomicOr((unsigned long long*)&filt[idx], ((unsigned long long)mhi << 32) | mlo); }
                    if (off + 1 < cn[i]) { bf_bits(q[i].y, idx, mlo, mhi); atomicOr((unsigned long long*)&filt[idx], ((unsigned long long)mhi << 32) | mlo); }
                }
            }
            __syncthreads();
            if (lane == 0) { seg = atomicAdd(a.seg_counter, 1u); if (seg >= a.max_segs) atomicOr(a.err, FJ_ERR_POOL); }
            seg = (u32)__builtin_amdgcn_readfirstlane((int)seg);
            cur_bucket = bucket;
        }

        // ---- test the 8 keys of this lane, compact the survivors of the wave into its chunk ----------------------
        u64 w[BF_KPT]; u32 mlo[BF_KPT], mhi[BF_KPT];
#pragma unroll
        for (int i = 0; i < (int)BF_KPT; ++i) { u32 idx; bf_bits(kA[i], idx, mlo[i], mhi[i]); w[i] = filt[idx]; }
#pragma unroll
        for (int i = 0; i < (int)BF_KPT; ++i) {
            const bool pass = ((okA >> i) & 1u) && ((u32)w[i] & mlo[i]) == mlo[i] && ((u32)(w[i] >> 32) & mhi[i]) == mhi[i];
            const u64 m = __ballot(pass);
            if (m) {
                const u32 n = (u32)__popcll(m);
                const u32 dst = fill + __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u));
                u32 nxt = cur;
                if (fill + n > FJ_CHUNK) nxt = new_chunk(bucket);
                if (pass) {
                    const u32 c = dst < FJ_CHUNK ? cur : nxt;
                    if (c < cap) a.out_keys[(u64)c * FJ_CHUNK + (dst & (FJ_CHUNK - 1))] = kA[i];
                }
                fill += n;
                if (fill >= FJ_CHUNK) {             // the current chunk is complete (fill == 256 exactly, or it overflowed into nxt)
                    if (cur != FJ_DIR_INVALID && lane == 0 && cur < cap) a.out_dir[cur] = (bucket << FJ_DIR_CNT_BITS) | FJ_CHUNK;
                    if (nxt != cur) { cur = nxt; fill -= FJ_CHUNK; }
                }
                survivors += n;
            }
        }

        // rotate the pipeline
#pragma unroll
        for (int i = 0; i < (int)BF_KPT; ++i) { kA[i] = kB[i]; kB[i] = kC[i]; }
        okA = okB; okB = okC; bktA = bktB; bktB = bktC; bktC = dD.bucket;
#pragma unroll
        for (int i = 0; i < 4; ++i) eC[i] = eD[i];
        vmC = vmD; dD = dE;
    }
    if (cur_bucket != 0xFFFFFFFFu) end_segment(cur_bucket);
    if (lane == 0 && survivors) atomicAdd(a.survivors, survivors);
}

}  // namespace

u32 fj_bloom_tile_chunks() { return BF_T / FJ_CHUNK; }
u32 fj_bloom_waves_per_group() { return BF_NT / 64; }

hipError_t fj_launch_bloom_filter(const FjBloomArgs& a, u32 grid, hipStream_t s) {
    const u32 lds = FJ_BLOOM_BLOCKS * 8;
    hipError_t e = fj_set_max_lds_once(reinterpret_cast<const void*>(fj_bloom_filter_kernel), lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(fj_bloom_filter_kernel, dim3(grid), dim3(BF_NT), lds, s, a);
    return hipGetLastError();
}
