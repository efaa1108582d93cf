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
//   * one 1024-thread workgroup per CU owns 140 KiB of LDS = 35840 words of filter (+ 16 KiB of per-wave staging);
//   * work = tiles of the probe side's level-L chunk lists (tile table, bucket by bucket); a workgroup takes a
//     contiguous run of tiles (snapped to bucket boundaries when they are close);
//   * when its bucket changes the workgroup rebuilds the filter from the BUILD side's chunks of that bucket
//     (<= ~400K keys, read once more from HBM/L2: 8 B per build key per workgroup that visits the bucket);
//   * probe keys stream through registers two tiles ahead (no barrier in the steady state), each key tests 2 bits of
//     one 32-bit word; survivors are compacted per wave (ballot + mbcnt) into an LDS staging row and leave in whole
//     512-B pieces for wave-private 2-KiB chunks of an output chunk pool with the same bucket structure, so the next
//     partition pass reads it like any other level.
//
// Measured at config 4 (1B probe keys, 5 % hits, 512 level-1 buckets of ~195K build keys): 2.26 ms (round 2: 2.4) for 8 GB of
// probe keys + 1.5 GB of build keys (~770 filter builds) read and 0.96 GB of survivors written.  Reading and counting alone
// takes 1.73 ms (5.5 TB/s; this box reads 6.1-6.4 TB/s at best), hashing and the LDS lookup add 0.03, compacting the survivors
// through LDS 0.35, storing them 0.17 (round 3, -DFJ_BLOOM_DIAG_COUNT_ONLY and store-less builds).  A wave's four loads of a
// tile each cover half of ONE chunk, so chunk ids and counts live in SGPRs (scalar address arithmetic); survivors go through
// LDS so that the global store is one 512-B instruction per 64 survivors.  The kernel has ~45 kernel arguments and three
// register-resident key buffers: round 2's form spilled 94 scalar registers into vector lanes (2187 v_readlane in the code);
// the cold arguments are now read from the kernel-argument segment where they are used (bf_late_args: 17 spills left).
// What was tried instead of this kernel + a plain pass over its survivors (round 3, profiles/r03_*): the filter test fused
// into the last radix pass (tile-synchronous, bucket sort of the survivors only: 4.2 ms against 2.26 + 0.43 - the pass's
// per-tile machinery, carry / scan / line descriptors behind six barriers, costs ~13K clocks per tile whether it sorts 8192
// keys or 980), and survivors stored one by one into 64 wave-private sub-bucket streams without LDS write-combining
// (tools/ubench_sparse_scatter.hip: +1.4 ms per 1B keys at 13 % survivors against +0.33 for the compacted 512-B stores).
//
// No false negatives by construction (insert and test use the same bits); false positives only cost the work the
// filter would have saved.  Algorithmic HBM bytes: 8 B per probe key read + 8 B per survivor written (+ the build
// keys of the level once or twice).
#include "fj_bloom_dev.h"

namespace {

constexpr u32 BF_NW = BF_NT / 64;
constexpr u32 BF_KPT = 8;                   // probe keys per thread and tile
constexpr u32 BF_T = BF_NT * BF_KPT;        // 8192 keys = 32 chunks per tile
constexpr u32 BF_STG = 128;                 // staging row of a wave: keys
#ifndef FJ_BLOOM_SNAP16
#define FJ_BLOOM_SNAP16 3
#endif
// a workgroup boundary this close to a bucket boundary (in 16ths of the bucket's tiles) moves onto it: one filter build less on
// each side (a build costs what ~5-10 % of a bucket's probe tiles cost); with equal buckets the tile-balanced boundaries drift
// from the bucket boundaries like a random walk (c4: up to ~9 % of a bucket in the middle of the range)
constexpr u32 BF_SNAP16 = FJ_BLOOM_SNAP16;
// output chunks a wave takes per allocator hit.  The hit is a returning global atomic: its wait drains the wave's whole load
// pipeline, so it must be rare - 128 chunks = 32768 survivors cover a wave's share of a 1B-row probe side at 13 % survivors
constexpr u32 BF_SLAB = 128;

// The kernel's cold arguments, read from the kernel-argument segment AT THE POINT OF USE: the empty asm makes the pointer
// opaque, so the compiler can neither preload these fields at kernel entry nor hoist their loads out of the loop - where they
// would occupy (and, as in round 2, overflow) the scalar register file.  FjBloomArgs is the kernel's only parameter: offset 0.
typedef const __attribute__((address_space(4))) FjBloomArgs* BfLateArgs;
__device__ __forceinline__ BfLateArgs bf_late_args() {
    BfLateArgs p = (BfLateArgs)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return p;
}

template <int VAR>
__global__ __launch_bounds__(BF_NT, 1) void fj_bloom_filter_kernel(FjBloomArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* filt = smem;                                          // FJ_BLOOM_WORDS * 4 bytes
    const u32 tid = threadIdx.x, lane = tid & 63, wave = bf_uni(tid >> 6);
    u64* stg = reinterpret_cast<u64*>(smem + FJ_BLOOM_WORDS * 4) + wave * BF_STG;   // this wave's staging row
    const u32 G = gridDim.x, g = blockIdx.x;
    // hot arguments (see FjBloomArgs)
    const u64* __restrict__ pkeys = a.pkeys; const u32* __restrict__ plist = a.plist; const uint4* __restrict__ tiles = a.tiles;
    u64* __restrict__ out_keys = a.out_keys; u32* __restrict__ out_dir = a.out_dir; u64* __restrict__ out_rel = a.out_rel;
    const u32 cap = a.cap_chunks;

    // contiguous run of tiles [t0, t1); a boundary that falls close to a bucket boundary is moved onto it, so that with
    // balanced buckets no filter is built twice (both neighbours compute the same snapped value)
    u32 t0, nmine;
    {
        BfLateArgs la = bf_late_args();
        const u32* toff = la->toff;
        const u32 nb = la->pnb, ntiles = toff[nb];
        auto boundary = [&](u32 gg) -> u32 {
            u32 t = (u32)(((u64)gg * ntiles) / G);
            if (gg == 0 || gg >= G || ntiles == 0) return gg >= G ? ntiles : t;
            u32 lo = 0, hi = nb;                                 // last bucket p with toff[p] <= t
            while (hi - lo > 1) { const u32 mid = (lo + hi) >> 1; if (toff[mid] <= t) lo = mid; else hi = mid; }
            const u32 b0 = toff[lo], b1 = toff[lo + 1], slack = ((b1 - b0) * BF_SNAP16) >> 4;
            if (t - b0 <= slack) t = b0; else if (b1 - t <= slack) t = b1;
            return t;
        };
        t0 = boundary(g);
        const u32 t1 = boundary(g + 1);
        if (t0 >= t1) return;
        nmine = t1 - t0;
    }

    // ---- input side: descriptors -> chunk-list entries -> keys, each a step earlier than its consumer ------------
    // thread tid reads key pairs (i*NT + tid)*2, i = 0..3: chunk j = i*8 + tid/128 of the tile, byte offset (tid%128)*16.
    // j is the same for all lanes of a wave: entries, ids and counts are wave-uniform.
    const u32 jb = bf_uni(tid >> 7), off = (tid & 127u) * 2u, off16 = (tid & 127u) * 16u;
    struct Desc { u32 pos, len, bucket; };
    auto get_desc = [&](u32 tt) -> Desc {                    // (vector load of a uniform address; made uniform at use)
        const uint4 d = tiles[t0 + (tt < nmine ? tt : nmine - 1)];
        Desc r; r.pos = d.x; r.len = tt < nmine ? d.y : 0u; r.bucket = d.z; return r;
    };
    auto get_entries = [&](const Desc& d, u32 (&e)[4]) {     // unconditional loads (index clamped): straight-line code
        const u32 pos = bf_uni(d.pos), len = bf_uni(d.len);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const u32 j = (u32)i * 8u + jb;
            e[i] = plist[pos + (j < len ? j : (len ? len - 1 : 0u))];
        }
    };
    // request the keys of a tile: scalar base per chunk + constant lane offset; the chunks' key counts stay in SGPRs
    // (validity of a lane's keys is one compare against them at test time)
    auto get_keys = [&](const u32 (&e)[4], u32 len, u64 (&kk)[BF_KPT], u32 (&cn)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const u32 eu = bf_uni(e[i]);
            const u32 cnt = ((u32)i * 8u + jb) < len ? FJ_LIST_CNT(eu) : 0u;
            const unsigned char* base = reinterpret_cast<const unsigned char*>(pkeys + (u64)FJ_LIST_ID(eu) * FJ_CHUNK);
            const u64x2 q = *reinterpret_cast<const u64x2*>(base + off16);
            kk[2 * i] = q.x; kk[2 * i + 1] = q.y;
            cn[i] = cnt;
        }
    };

    // ---- output side: wave-private chunks, filled 64 keys at a time from the wave's staging row --------------------
    u32 cur = FJ_DIR_INVALID, fill = FJ_CHUNK;      // current chunk and its fill (wave-uniform; a multiple of 64 until the segment ends)
    u32 ns = 0;                                     // staged survivors of this wave (< 64 between half steps)
    u32 nch = 0, seg = 0;                           // chunks of this (wave, bucket run) = one segment of the bucket's chunk list
    u32 seg_keys = 0;                               // survivors of this wave inside the current bucket run
    u32 slab_cur = 0, slab_rem = 0;
    unsigned long long survivors = 0;
    auto next_chunk = [&](u32 bucket) {             // wave-uniform: close `cur` (full), open a fresh chunk
        if (cur != FJ_DIR_INVALID && lane == 0 && cur < cap) out_dir[cur] = (bucket << FJ_DIR_CNT_BITS) | FJ_CHUNK;
        if (slab_rem == 0) {
            BfLateArgs la = bf_late_args();
            u32 base = 0;
            if (lane == 0) { base = atomicAdd(la->alloc, BF_SLAB); if (base + BF_SLAB > cap) atomicOr(la->err, FJ_ERR_POOL); }
            slab_cur = bf_uni(base); slab_rem = BF_SLAB;
        }
        cur = slab_cur; ++slab_cur; --slab_rem;
        if (lane == 0 && cur < cap) out_rel[cur] = ((u64)seg << 32) | nch;
        ++nch; fill = 0;
    };
    // The staging row is a RING of BF_STG keys: survivors are appended behind `head + ns`, whole 512-B pieces leave from `head`
    // (nothing is ever moved inside the row).
    u32 head = 0;
    auto flush = [&](u32 bucket, u32 n) {           // write staged keys [head, head + n), n <= 64, behind the chunk's fill
        if (fill == FJ_CHUNK) next_chunk(bucket);
        if (lane < n && cur < cap) {
            unsigned char* base = reinterpret_cast<unsigned char*>(out_keys + (u64)cur * FJ_CHUNK + fill);
            *reinterpret_cast<u64*>(base + lane * 8u) = stg[(head + lane) & (BF_STG - 1)];
        }
        fill += n; head = (head + n) & (BF_STG - 1);
    };
    auto end_segment = [&](u32 bucket) {            // close this wave's run inside `bucket`
        if (ns) { flush(bucket, ns); ns = 0; }
        head = 0;
        if (nch > 0 && lane == 0) {
            BfLateArgs la = bf_late_args();
            if (cur < cap) out_dir[cur] = (bucket << FJ_DIR_CNT_BITS) | fill;
            const u32 o = atomicAdd(&la->bchunks[bucket], nch);
            if (seg < la->max_segs) la->seg_off[seg] = o;
            unsigned long long* bkc = la->bucket_keys;
            if (bkc && seg_keys) atomicAdd(&bkc[bucket], (unsigned long long)seg_keys);
        }
        cur = FJ_DIR_INVALID; fill = FJ_CHUNK; nch = 0; seg_keys = 0;
    };

    // ---- prologue: keys of tiles 0-1, entries of tiles 2-3, descriptors of tiles 4-5 in flight ----------------------
    // Every buffer rotates by NAME (the loop body is written three times): a register copy of a buffer whose load is
    // still in flight would make the wave wait for that load.  And every dependent request is issued TWO steps before
    // its consumer and BEFORE the keys of its step: vector memory operations retire in issue order, so by the time a
    // step's keys (requested two steps ago) are there, the entries and the descriptor it needs are there too - a step
    // has one wait, for its keys.  (With a one-step distance the wait for the entries sat right behind the previous
    // step's survivor stores, which count in the same vmcnt: 2.29 ms per 1B keys against 1.88 ms with the stores
    // ablated.)
    u64 K0[BF_KPT], K1[BF_KPT], K2[BF_KPT];
    u32 C0[4], C1[4], C2[4];                           // key counts of the four chunks a wave reads per tile (wave-uniform)
    u32 E0[4], E1[4], E2[4];                           // chunk-list entries
    Desc D0, D1, D2;
    u32 bq0, bq1, bq2, bq3, lq0, lq1;                  // buckets of tiles t..t+3, lengths of tiles t+2, t+3 (SGPRs)
    {
        const Desc d0 = get_desc(0), d1 = get_desc(1), d2 = get_desc(2), d3 = get_desc(3);
        D1 = get_desc(4); D2 = get_desc(5);
        u32 e0[4], e1[4];
        get_entries(d0, e0); get_entries(d1, e1);
        get_entries(d2, E2); get_entries(d3, E0);
        get_keys(e0, bf_uni(d0.len), K0, C0); get_keys(e1, bf_uni(d1.len), K1, C1);
        bq0 = bf_uni(d0.bucket); bq1 = bf_uni(d1.bucket); bq2 = bf_uni(d2.bucket); bq3 = bf_uni(d3.bucket);
        lq0 = bf_uni(d2.len); lq1 = bf_uni(d3.len);
    }

    u32 cur_bucket = 0xFFFFFFFFu;
    u32 t = 0;
    // one tile: kCur holds tile t; requests go out for tile t+2's keys (entries eUse), tile t+4's entries (descriptor
    // dUse) and tile t+6's descriptor
    auto step = [&](const u64 (&kCur)[BF_KPT], const u32 (&cnCur)[4], u64 (&kNew)[BF_KPT], u32 (&cnNew)[4],
                    const u32 (&eUse)[4], u32 (&eNew)[4], const Desc& dUse, Desc& dNew) {
        const u32 bucket = bq0;
        const Desc du = dUse;                          // tile t+4 (read before dNew, which may alias another name's register, is written)
        dNew = get_desc(t + 6);
        get_entries(du, eNew);
        get_keys(eUse, lq0, kNew, cnNew);
        bq0 = bq1; bq1 = bq2; bq2 = bq3; bq3 = bf_uni(du.bucket);
        lq0 = lq1; lq1 = bf_uni(du.len);

        if (bucket != cur_bucket) {                 // workgroup-uniform: rebuild the filter for this bucket
            if (cur_bucket != 0xFFFFFFFFu) end_segment(cur_bucket);
            __syncthreads();                        // every wave is done testing against the old filter
            BfLateArgs la = bf_late_args();
            const u32* prebuilt = la->prebuilt;
            if (prebuilt) {                         // filters built elsewhere (another GPU's build keys: the sender-side precheck of the owner shuffle)
                // (they end with a header word naming the variant they were built with: insert and test must use the same bits)
                if (tid == 0 && prebuilt[(u64)la->pnb * FJ_BLOOM_WORDS] != (FJ_BLOOM_HDR_MAGIC | (u32)VAR)) atomicOr(la->err, FJ_ERR_VARIANT);
                const uint4* src = reinterpret_cast<const uint4*>(prebuilt + (u64)bucket * FJ_BLOOM_WORDS);
                for (u32 i = tid; i < FJ_BLOOM_WORDS / 4; i += BF_NT) reinterpret_cast<uint4*>(filt)[i] = src[i];
                __syncthreads();
            } else {
                bf_build_filter<VAR>(filt, la->bkeys, la->blist, la->bboff, bucket, tid);
            }
            if (lane == 0) { seg = atomicAdd(la->seg_counter, 1u); if (seg >= la->max_segs) atomicOr(la->err, FJ_ERR_POOL); }
            seg = bf_uni(seg);
            cur_bucket = bucket;
        }

        // ---- test the 8 keys of this lane, compact the survivors of the wave into its staging row -------------------
        // All eight tests first (two batches of four independent LDS reads), their ballots stay in scalar registers; then
        // the survivors are appended half a tile-step at a time: four masked LDS writes at scalar prefix offsets and ONE
        // check whether a 512-B piece can leave.
        u64 m[BF_KPT];
#pragma unroll
        for (int h = 0; h < (int)BF_KPT; h += 4) {
            u32 wlo[4], whi[4], mlo[4], mhi[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { u32 o; bf_bits<VAR>(kCur[h + i], o, mlo[i], mhi[i]); bf_fetch<VAR>(filt, o, wlo[i], whi[i]); }
#pragma unroll
            for (int i = 0; i < 4; ++i)
                m[h + i] = __ballot(bf_pass<VAR>(wlo[i], whi[i], mlo[i], mhi[i]) && (off + (u32)((h + i) & 1)) < cnCur[(h + i) >> 1]);
        }
#pragma unroll
        for (int h = 0; h < (int)BF_KPT; h += 4) {
            const u32 n0 = (u32)__popcll(m[h]), n1 = (u32)__popcll(m[h + 1]), n2 = (u32)__popcll(m[h + 2]), n3 = (u32)__popcll(m[h + 3]);
            const u32 tot = n0 + n1 + n2 + n3;
            survivors += tot;
#ifdef FJ_BLOOM_DIAG_COUNT_ONLY
            continue;                                  // diagnostic build: test and count, do not compact or write
#endif
            seg_keys += tot;
            if (ns + tot <= BF_STG) {                  // (always, unless more than half of the keys survive: ns < 64)
                const u32 pre[4] = {0u, n0, n0 + n1, n0 + n1 + n2};
                const u32 tail = head + ns;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if ((m[h + i] >> lane) & 1ull)
                        stg[(tail + pre[i] + __builtin_amdgcn_mbcnt_hi((u32)(m[h + i] >> 32), __builtin_amdgcn_mbcnt_lo((u32)m[h + i], 0u))) & (BF_STG - 1)] = kCur[h + i];
                ns += tot;
                if (ns >= 64) { flush(bucket, 64); ns -= 64; }      // a whole 512-B piece leaves
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) {          // key slot by key slot (each adds <= 64 to fewer than 64 staged keys)
                    const u64 mm = m[h + i];
                    if ((mm >> lane) & 1ull) stg[(head + ns + __builtin_amdgcn_mbcnt_hi((u32)(mm >> 32), __builtin_amdgcn_mbcnt_lo((u32)mm, 0u))) & (BF_STG - 1)] = kCur[h + i];
                    ns += (u32)__popcll(mm);
                    if (ns >= 64) { flush(bucket, 64); ns -= 64; }
                }
            }
        }
    };
    for (;;) {
        step(K0, C0, K2, C2, E2, E1, D1, D0); if (++t >= nmine) break;
        step(K1, C1, K0, C0, E0, E2, D2, D1); if (++t >= nmine) break;
        step(K2, C2, K1, C1, E1, E0, D0, D2); if (++t >= nmine) break;
    }
    if (cur_bucket != 0xFFFFFFFFu) end_segment(cur_bucket);
    // unused chunk ids of this wave's slab stay unlisted (the directory needs no memset: every id is defined by its owner)
    for (u32 j = lane; j < slab_rem; j += 64) { const u32 id = slab_cur + j; if (id < cap) out_dir[id] = FJ_DIR_INVALID; }
    if (lane == 0 && survivors) atomicAdd(bf_late_args()->survivors, survivors);
}

// Write the filters of ALL buckets of a build-side level to HBM (FJ_BLOOM_WORDS words each): what an owner GPU ships to its
// peers so that they can run the precheck before sending it their probe rows.
template <int VAR>
__global__ __launch_bounds__(BF_NT, 1) void fj_bloom_export_kernel(FjChunkSet build, u32* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const u32 tid = threadIdx.x;
    for (u32 b = blockIdx.x; b < build.nb; b += gridDim.x) {
        bf_build_filter<VAR>(smem, build.keys, build.list, build.boff, b, tid);
        uint4* dst = reinterpret_cast<uint4*>(out + (u64)b * FJ_BLOOM_WORDS);
        for (u32 i = tid; i < FJ_BLOOM_WORDS / 4; i += BF_NT) dst[i] = reinterpret_cast<const uint4*>(smem)[i];
        __syncthreads();
    }
    if (blockIdx.x == 0 && tid < 4) out[(u64)build.nb * FJ_BLOOM_WORDS + tid] = tid == 0 ? (FJ_BLOOM_HDR_MAGIC | (u32)VAR) : 0u;
}

// exclusive scan of per-bucket key counts (nb <= 1024: one workgroup) -> base[0..nb]
__global__ __launch_bounds__(1024) void fj_bucket_base_kernel(const unsigned long long* __restrict__ bkeys, unsigned long long* __restrict__ base, u32 nb) {
    __shared__ unsigned long long wtot[16];
    const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned long long x = tid < nb ? bkeys[tid] : 0ull;
    unsigned long long inc = x;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const unsigned long long y = __shfl_up(inc, d, 64); if ((int)lane >= d) inc += y; }
    if (lane == 63) wtot[wave] = inc;
    __syncthreads();
    unsigned long long run = inc - x, total = 0;
    for (u32 w = 0; w < 16; ++w) { if (w < wave) run += wtot[w]; total += wtot[w]; }
    if (tid < nb) base[tid] = run;
    if (tid == 0) base[nb] = total;
}

// chunk set -> dense array: BF_FLAT_SPLIT workgroups per bucket walk the bucket's chunk list in batches of 256 chunks (block
// scan of the chunks' key counts -> offsets; every workgroup of the bucket computes the same offsets), and share a batch's
// chunks among their waves: each wave copies whole chunks (256 keys = 4 per lane)
constexpr u32 BF_FLAT_SPLIT = 4;
__global__ __launch_bounds__(256) void fj_flatten_kernel(FjChunkSet cs, const unsigned long long* __restrict__ base, u64* __restrict__ out) {
    __shared__ u32 s_off[257];
    __shared__ u32 s_ent[256];
    __shared__ u32 wsum[4];
    const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const u32 part = blockIdx.x % BF_FLAT_SPLIT;
    for (u32 b = blockIdx.x / BF_FLAT_SPLIT; b < cs.nb; b += gridDim.x / BF_FLAT_SPLIT) {
        const u32 l0 = cs.boff[b], n = cs.boff[b + 1] - l0;
        unsigned long long run = base[b];
        for (u32 c0 = 0; c0 < n; c0 += 256) {
            const u32 e = c0 + tid < n ? cs.list[l0 + c0 + tid] : 0u;
            const u32 cnt = c0 + tid < n ? FJ_LIST_CNT(e) : 0u;
            u32 inc = cnt;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const u32 y = __shfl_up(inc, d, 64); if ((int)lane >= d) inc += y; }
            if (lane == 63) wsum[wave] = inc;
            __syncthreads();
            u32 pre = inc - cnt;
            for (u32 w = 0; w < wave; ++w) pre += wsum[w];
            s_off[tid] = pre; s_ent[tid] = e;
            if (tid == 255) s_off[256] = pre + cnt;
            __syncthreads();
            const u32 m = n - c0 < 256 ? n - c0 : 256;
            for (u32 j = wave * BF_FLAT_SPLIT + part; j < m; j += 4 * BF_FLAT_SPLIT) {
                const u32 ee = s_ent[j], cc = FJ_LIST_CNT(ee);
                const u64* src = cs.keys + (u64)FJ_LIST_ID(ee) * FJ_CHUNK;
                u64* dst = out + run + s_off[j];
                for (u32 k = lane; k < cc; k += 64) dst[k] = fj_key_unmix(src[k]);      // chunk pools hold mixed keys, the dense array goes to a peer as raw keys
            }
            run += s_off[256];
            __syncthreads();
        }
    }
}

}  // namespace

hipError_t fj_launch_bloom_export(const FjChunkSet& build, u32* out, u32 grid, int variant, hipStream_t s) {
    const u32 lds = FJ_BLOOM_WORDS * 4;
    auto kern = variant == 0 ? fj_bloom_export_kernel<0> : (variant == 1 ? fj_bloom_export_kernel<1> : fj_bloom_export_kernel<2>);
    hipError_t e = fj_set_max_lds_once(reinterpret_cast<const void*>(kern), lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(grid < build.nb ? grid : build.nb), dim3(BF_NT), lds, s, build, out);
    return hipGetLastError();
}

hipError_t fj_launch_flatten(const FjChunkSet& cs, const unsigned long long* bucket_keys, unsigned long long* base, u64* out, hipStream_t s) {
    if (cs.nb > 1024) return hipErrorInvalidValue;
    hipLaunchKernelGGL(fj_bucket_base_kernel, dim3(1), dim3(1024), 0, s, bucket_keys, base, cs.nb);
    hipLaunchKernelGGL(fj_flatten_kernel, dim3(cs.nb * BF_FLAT_SPLIT), dim3(256), 0, s, cs, base, out);
    return hipGetLastError();
}

u32 fj_bloom_tile_chunks() { return BF_T / FJ_CHUNK; }
u32 fj_bloom_waves_per_group() { return BF_NW; }
u32 fj_bloom_slab_chunks() { return BF_SLAB; }

hipError_t fj_launch_bloom_filter(const FjBloomArgs& a, u32 grid, int variant, hipStream_t s) {
    const u32 lds = FJ_BLOOM_WORDS * 4 + BF_NW * BF_STG * 8;
    auto kern = variant == 0 ? fj_bloom_filter_kernel<0> : (variant == 1 ? fj_bloom_filter_kernel<1> : fj_bloom_filter_kernel<2>);
    hipError_t e = fj_set_max_lds_once(reinterpret_cast<const void*>(kern), lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(BF_NT), lds, s, a);
    return hipGetLastError();
}
