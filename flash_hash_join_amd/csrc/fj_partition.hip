// fj_partition.hip -- radix partition pass for MI355X (gfx950), keys (+values).
//
// Mirrors the *function* of parallel_radix_partition_kv / _k (hash_join.cpp:209-292) but not
// its structure.  The reference does histogram -> prefix -> scatter with 8-byte random
// stores.  Here one pass is a single kernel with NO histogram read:
//
//   * persistent workgroups each take a contiguous run of input tiles; the keys of tile t+1 are
//     prefetched into registers (and the chunk metadata of tile t+2) while tile t is processed,
//     so HBM reads stay in flight across the barriers of the tile;
//   * a tile of keys is bucket-sorted inside LDS (LDS atomics give the rank);
//   * per bucket only WHOLE lines (LINE keys, 64 or 128 B, line-aligned in HBM) are written,
//     the < LINE remainder is carried in LDS to the next tile  (software write-combining);
//   * lines go into block-private 2-KiB *chunks* handed out by a slab allocator, so no two
//     workgroups ever share a line and no global cursor is contended;
//   * chunk-list bookkeeping is done by the producer: per chunk a directory word (bucket|count)
//     and its rank inside the (segment, bucket) it was produced in; per (segment, bucket) one
//     global atomic reserves a span of the bucket's chunk list.  After the pass a scan of the
//     per-bucket chunk counts plus an atomic-free kernel turn that into per-bucket chunk lists.
//
// Algorithmic HBM bytes per key and pass: 8 read + 8 written (keys only), 16 + 16 with values.
#include "fj_internal.h"
#include <cstdlib>
#include <mutex>
#include <vector>

namespace {

// u64 of padding per carry row (0: round 3's layout; an A/B knob: -DFJ_LO_PAD=0)
#ifndef FJ_LO_PAD
#define FJ_LO_PAD 1u
#endif
// dynamic-LDS layout, shared by kernel and host-side size computation
struct PartLds {
    u32 sorted_k, sorted_v, lo_k, lo_v, hist, toff, line_desc, t_chunk, t_cnt, wsum, misc, total;
};
__host__ __device__ inline PartLds part_lds_layout(u32 T, u32 F, u32 line, bool vals, u32 nwaves) {
    PartLds L;
    const u32 nsorted = T + 2;                             // bucket-sorted tile (remainders stay in lo_*)
    const u32 maxl = (T + (line - 1) * F) / line + 2;
    u32 o = 0;
    L.sorted_k = o; o += nsorted * 8;
    L.sorted_v = o; if (vals) o += nsorted * 8;
    L.lo_k = o; o += F * (line + FJ_LO_PAD) * 8;              // per-bucket carry rows, padded: a row stride of `line` u64 (128 B) puts
    L.lo_v = o; if (vals) o += F * (line + FJ_LO_PAD) * 8;   // row j of every bucket on ONE bank pair (the carry copy: 64-way conflicts)
    L.line_desc = o; o += maxl * 8;
    L.wsum = o; o += nwaves * 8;
    L.hist = o; o += (F + 4) * 4;
    L.toff = o; o += (F + 4) * 4;
    L.t_chunk = o; o += (T / FJ_CHUNK) * 4;
    L.t_cnt = o; o += (T / FJ_CHUNK) * 4;
    L.misc = o; o += 16 * 4;
    L.total = (o + 15) & ~15u;
    return L;
}

enum { M_SLAB_CUR = 0, M_SLAB_REM, M_NEW_BASE, M_NEED, M_NLINES, M_FLUSH, M_SEG, M_NEW_UNITS };

// The kernel is instruction-issue bound (not HBM bound) on gfx950, so the inner phases are written
// to minimise issued instructions per key: 32-bit-multiply hash, invalid lanes routed to a dummy
// bucket instead of branches, per-bucket state in the registers of thread b, the bucket scan done
// only by the waves that own buckets, whole contiguous lines per store instruction in the write-out.
// PROBE_SIDE only names the instantiation (identical code): a counting join runs the keys-only kernel over both
// relations, and per-kernel profiler statistics should not average 100M-row and 1B-row launches together.
// (Round 3's owner-grouped form of this kernel - one open slab per owner GPU - is gone: the multi-GPU sender runs this pass as
// it is and csrc/fj_pack.hip rewrites its output for the wire.)
// PK7: the input chunks are in the owner shuffle's 7-byte wire format (FJ_WIRE7_BYTES per chunk: the low words, the middle 16
// bits and bits 48..55 of the mixed keys as three planes; bits 56..63 are the top bits of the chunk's first-pass bucket) - the
// pass that reads what other GPUs sent unpacks it in registers (csrc/fj_pack.hip writes the format).
template <int NT, int KPT, int LINE_LOG, bool HAS_VALS, bool FLAT, bool PROBE_SIDE, int RLOG = FJ_RUN_LOG, bool PK7 = false>
__global__ __launch_bounds__(NT, 4) void fj_partition_kernel(FjPartArgs a) {
    constexpr u32 T = NT * KPT, LINE = 1u << LINE_LOG, TC = T / FJ_CHUNK, NW = NT / 64, LOS = LINE + FJ_LO_PAD;     // LOS: stride of a carry row
    static_assert(T % FJ_CHUNK == 0 && TC <= NT && LINE >= 4 && T + 64 < (1u << 17), "tile geometry");
    static_assert(!PK7 || (!FLAT && KPT % 4 == 0), "wire-format chunks come through chunk lists, four keys per load group");
    // RUNS (RLOG > 0): a bucket takes its chunk ids in aligned runs of RU = 2^RLOG ids (used in a rotated order; what a segment
    // leaves unused of its last run is marked FJ_DIR_INVALID), so that fj_level_lists places RU list entries per step
    // (FjChunkSet::run_log).  The allocator then counts in units of one run.  RLOG == 0: ids one by one in the order the tiles
    // open chunks (flat passes of <= 256 buckets, see fj_plan.hip).
    // (A compile-time choice: with the run length as a kernel argument every pass ran 6 % slower.)
    constexpr bool RUNS = RLOG > 0;
    constexpr u32 RL = (u32)RLOG, RU = 1u << RL;
    const u32 F = 1u << a.fan_log, FM = F - 1;
    const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const u32 sh32 = a.shift - 32;                       // digit comes from hash word 1 only = the high word of the mixed key

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const PartLds Lo = part_lds_layout(T, F, LINE, HAS_VALS, NW);
    u64* tile_k = (u64*)(smem + Lo.sorted_k);
    u64* tile_v = (u64*)(smem + Lo.sorted_v);
    u64* lo_k = (u64*)(smem + Lo.lo_k);
    u64* lo_v = (u64*)(smem + Lo.lo_v);
    u64* line_desc = (u64*)(smem + Lo.line_desc);
    u64* wsum = (u64*)(smem + Lo.wsum);
    u32* hist = (u32*)(smem + Lo.hist);
    u32* toff = (u32*)(smem + Lo.toff);
    u32* t_chunk = (u32*)(smem + Lo.t_chunk);
    u32* t_cnt = (u32*)(smem + Lo.t_cnt);
    u32* misc = (u32*)(smem + Lo.misc);

    const u32 Lc = FLAT ? (u32)((a.n_flat + FJ_CHUNK - 1) >> FJ_CHUNK_LOG) : 0u;
    const u32 ntiles = FLAT ? (Lc + TC - 1) / TC : *a.in_ntiles;
    const u32 G = gridDim.x, g = blockIdx.x;
    // this workgroup's contiguous run of tiles: t0 .. t0 + nmine - 1
    const u32 t0 = (u32)(((u64)g * ntiles) / G);
    const u32 nmine = (u32)(((u64)(g + 1) * ntiles) / G) - t0;
    u32 t = 0;
    const u32 thi = nmine;
    const u32 cap = a.cap_chunks;
    if (t >= thi) return;

    // ---- input side ---------------------------------------------------------------------------
    auto get_desc = [&](u32 tt, u32& pos, u32& len, u32& parent) {      // list input: tile table row
        const uint4 d = a.in_tiles[t0 + tt]; pos = d.x; len = d.y; parent = d.z;
    };
    auto meta_fetch = [&](u32 pos, u32 len, u32& id, u32& cnt) {        // chunk id + key count of chunk tid
        id = 0; cnt = 0;
        if (tid < len) { const u32 e = a.in_list[pos + tid]; id = FJ_LIST_ID(e); cnt = FJ_LIST_CNT(e); }
    };
    // request a tile's keys: 16 B per lane and load.  The loads are UNCONDITIONAL (address clamped to
    // readable memory, validity kept in vmask) so that they form straight-line code: the compiler can
    // then wait for the older chunk-list load with a counted vmcnt instead of draining these too.
    const bool tiny = FLAT && a.n_flat < 2;
    const u64 last_pair = FLAT ? ((a.n_flat - 2) & ~1ull) : 0;
    // PK7: the raw planes of a tile (7 words per 4 keys) stay in registers until the keys are needed - assembling them right
    // after the loads would make the prefetch wait for them
    struct Pk7Raw { uint4 lo; uint2 mid; u32 hi; };
    auto key_load7 = [&](u32 tt, Pk7Raw (&rr)[KPT / 4 ? KPT / 4 : 1], u64 (&vv)[KPT], u32& vmask) {
        (void)tt;
        vmask = 0;
#pragma unroll
        for (int i = 0; i < KPT / 4; ++i) {
            const u32 kidx = ((u32)i * NT + tid) * 4;
            const u32 j = kidx >> FJ_CHUNK_LOG, off = kidx & (FJ_CHUNK - 1);
            const u32 cnt = t_cnt[j];
            const unsigned char* base = reinterpret_cast<const unsigned char*>(a.in_keys) + (u64)t_chunk[j] * FJ_WIRE7_BYTES;   // whole chunks were received: every offset is readable
            rr[i].lo = *reinterpret_cast<const uint4*>(base + off * 4u);
            rr[i].mid = *reinterpret_cast<const uint2*>(base + FJ_WIRE7_MID + off * 2u);
            rr[i].hi = *reinterpret_cast<const u32*>(base + FJ_WIRE7_HI + off);
            if (HAS_VALS) {
                const u64x2 w0 = *reinterpret_cast<const u64x2*>(a.in_vals + (u64)t_chunk[j] * FJ_CHUNK + off);
                const u64x2 w1 = *reinterpret_cast<const u64x2*>(a.in_vals + (u64)t_chunk[j] * FJ_CHUNK + off + 2);
                vv[4 * i] = w0.x; vv[4 * i + 1] = w0.y; vv[4 * i + 2] = w1.x; vv[4 * i + 3] = w1.y;
            }
            const u32 nv = off < cnt ? (cnt - off < 4u ? cnt - off : 4u) : 0u;
            vmask |= ((1u << nv) - 1u) << (4 * i);
        }
    };
    auto key_load = [&](u32 tt, u64 (&kk)[KPT], u64 (&vv)[KPT], u32& vmask) {
        vmask = 0;
#pragma unroll
        for (int i = 0; i < KPT / 2; ++i) {
            u64 base; u32 nv;                               // nv = valid keys of this pair (0..2)
            if (FLAT) {
                base = (u64)(t0 + tt) * T + ((u32)i * NT + tid) * 2;
                nv = base + 1 < a.n_flat ? 2u : (base < a.n_flat ? 1u : 0u);
                if (base > last_pair) base = last_pair;     // stay inside the array (tail tile only)
            } else {
                const u32 kidx = ((u32)i * NT + tid) * 2;
                const u32 j = kidx >> FJ_CHUNK_LOG, off = kidx & (FJ_CHUNK - 1);
                const u32 cnt = t_cnt[j];
                base = (u64)t_chunk[j] * FJ_CHUNK + off;    // chunks are fully allocated: any offset is readable
                nv = off + 1 < cnt ? 2u : (off < cnt ? 1u : 0u);
            }
            if (tiny) {                                      // a 1-key relation: no 16-B load possible
                kk[2 * i] = nv ? a.in_keys[0] : 0; kk[2 * i + 1] = 0;
                if (HAS_VALS) { vv[2 * i] = nv ? a.in_vals[0] : 0; vv[2 * i + 1] = 0; }
            } else {
                const u64x2 q = *reinterpret_cast<const u64x2*>(a.in_keys + base);
                kk[2 * i] = q.x; kk[2 * i + 1] = q.y;
                if (HAS_VALS) {
                    const u64x2 w = *reinterpret_cast<const u64x2*>(a.in_vals + base);
                    vv[2 * i] = w.x; vv[2 * i + 1] = w.y;
                }
            }
            vmask |= (nv == 2 ? 3u : (nv == 1 ? 1u : 0u)) << (2 * i);
        }
        if (FLAT && !tiny && (a.n_flat & 1ull)) {            // odd length: the very last key has no pair partner
#pragma unroll
            for (int i = 0; i < KPT / 2; ++i) {
                const u64 base = (u64)(t0 + tt) * T + ((u32)i * NT + tid) * 2;
                if (base + 1 == a.n_flat) { kk[2 * i] = a.in_keys[base]; if (HAS_VALS) vv[2 * i] = a.in_vals[base]; }
            }
        }
    };

    if (tid == 0) { misc[M_SLAB_CUR] = 0; misc[M_SLAB_REM] = 0; misc[M_NEW_BASE] = 0; misc[M_SEG] = 0; misc[M_NEW_UNITS] = 0; }
    // per-bucket state lives in the registers of thread b (b < F)
    u32 st_left = 0, st_fill = FJ_CHUNK, st_cur = FJ_DIR_INVALID, st_nch = 0;
    u32 pend_cnt = 0, pend_nf = 0, pend_tb = 0;

    // carry step of the tile that was just written: the keys of bucket b that did not fill a whole
    // line move (or stay) in lo_*.  Runs after the tile's last barrier, before the tile region is reused.
    auto carry = [&]() {
        if (tid < F) {
            const u32 b = tid, tot = st_left + pend_cnt;
            if (pend_nf == 0) {                                   // nothing was written: append the new keys
                for (u32 j = 0; j < pend_cnt; ++j) {
                    lo_k[b * LOS + st_left + j] = tile_k[pend_tb + j];
                    if (HAS_VALS) lo_v[b * LOS + st_left + j] = tile_v[pend_tb + j];
                }
            } else {                                              // the remainder is the tail of the new keys
                const u32 nl2 = tot - pend_nf, src = pend_tb + (pend_nf - st_left);
                for (u32 j = 0; j < nl2; ++j) {
                    lo_k[b * LOS + j] = tile_k[src + j];
                    if (HAS_VALS) lo_v[b * LOS + j] = tile_v[src + j];
                }
            }
            st_left = tot - pend_nf;
            pend_cnt = 0; pend_nf = 0;
        }
    };

    // id of the j-th chunk this workgroup allocates in the current tile (j counts over all buckets in bucket order)
    auto alloc_id = [&](u32 j, u32) -> u32 {
        const u32 rem = misc[M_SLAB_REM];                   // (in allocation units: runs)
        return j < rem ? misc[M_SLAB_CUR] + (j << RL) : misc[M_NEW_BASE] + ((j - rem) << RL);
    };
    // A bucket uses the ids of a run in ROTATED order, starting at rot(b): all workgroups start together and fill their chunks
    // at the same rate, so with every bucket on its run's first id at the same time the lines in flight would sit in one
    // quarter of every 8-KiB stretch of the pool (measured: the c3 passes 4-10 % slower).
    auto rot = [&](u32 b) -> u32 { return ((b * 2654435761u + blockIdx.x * 40503u) >> 13) & (RU - 1u); };   // (not a function of b's low bits alone: those also place the run)
    // ids left in the run that chunk c0 (bucket b's open chunk) belongs to
    auto run_left = [&](u32 c0, u32 b) -> u32 { return (RUNS && c0 != FJ_DIR_INVALID) ? (RU - 1u) - ((c0 - rot(b)) & (RU - 1u)) : 0u; };
    auto run_step = [&](u32 c0, u32 k) -> u32 { return (c0 & ~(RU - 1u)) | ((c0 + k) & (RU - 1u)); };    // k ids further in c0's run
    // id of the kk-th (1-based) chunk a bucket opens in the current tile: first the rest of its run, then the runs it was
    // given (ukb = the bucket's prefix over the tile's new allocation units)
    auto chunk_id = [&](u32 kk, u32 c0, u32 ukb, u32 b) -> u32 {
        if constexpr (!RUNS) return alloc_id(ukb + kk - 1, b);
        const u32 rl = run_left(c0, b);
        if (kk <= rl) return run_step(c0, kk);
        const u32 q = kk - rl - 1u;
        return alloc_id(ukb + (q >> RL), b) + ((q + rot(b)) & (RU - 1u));
    };
    // (one thread) the open slab cannot cover `need` allocation units: take as many fresh slabs as the rest needs, in
    // one piece - alloc_id() hands out the open slab's remainder first, so nothing is abandoned
    auto take_slabs = [&](u32 need) {
        const u32 su = a.slab >> RL, k = (need - misc[M_SLAB_REM] + su - 1) / su;
        const u32 nb = atomicAdd(a.alloc, k * a.slab);
        if (nb + k * a.slab > cap) atomicOr(a.err, FJ_ERR_POOL);
        misc[M_NEW_BASE] = nb; misc[M_NEW_UNITS] = k * su;
    };
    auto commit_units = [&](u32 need) {                       // (one thread, after the ids were used)
        const u32 rem = misc[M_SLAB_REM];
        if (need <= rem) { misc[M_SLAB_CUR] += need << RL; misc[M_SLAB_REM] = rem - need; }
        else { const u32 used = need - rem; misc[M_SLAB_CUR] = misc[M_NEW_BASE] + (used << RL); misc[M_SLAB_REM] = misc[M_NEW_UNITS] - used; }
    };

    // end of a segment (= this workgroup's share of one parent bucket): write the carried
    // remainders, fix the last chunk's count, reserve the chunk-list spans, reset the state
    auto flush = [&](u32 parent) {
        if (tid == 0) misc[M_FLUSH] = 0;
        __syncthreads();
        u32 fresh = 0xFFFFFFFFu;                              // index of the fresh allocation unit this bucket's remainder needs
        if (tid < F && st_left > 0 && st_fill == FJ_CHUNK && !run_left(st_cur, tid)) fresh = atomicAdd(&misc[M_FLUSH], 1u);
        __syncthreads();
        if (tid == 0 && misc[M_FLUSH] > misc[M_SLAB_REM]) take_slabs(misc[M_FLUSH]);
        __syncthreads();
        if (tid < F) {
            const u32 b = tid, l = st_left, seg = misc[M_SEG];
            u32 f0 = st_fill, c = st_cur, n = st_nch;
            const u32 outb = parent * F + b;
            if (l > 0 && f0 == FJ_CHUNK) {
                if (fresh == 0xFFFFFFFFu) c = run_step(c, 1);
                else c = alloc_id(fresh, b) + rot(b);
                f0 = 0;
                if (c < cap) a.out_rel[c] = ((u64)seg << 32) | n;
                ++n;
            }
            if (c != FJ_DIR_INVALID && c < cap) {
                const u64 base = (u64)c * FJ_CHUNK + f0;
                for (u32 j = 0; j < l; ++j) {
                    a.out_keys[base + j] = lo_k[b * LOS + j];
                    if (HAS_VALS) a.out_vals[base + j] = lo_v[b * LOS + j];
                }
                a.out_dir[c] = (outb << FJ_DIR_CNT_BITS) | (f0 + l);
                if constexpr (RUNS) { for (u32 k = 1; k <= run_left(c, b); ++k) { const u32 id = run_step(c, k); if (id < cap) a.out_dir[id] = FJ_DIR_INVALID; } }   // rest of the run
            }
            if (n > 0) {
                const u32 off = atomicAdd(&a.bchunks[outb], n);
                if (seg < a.max_segs) a.seg_off[(u64)seg * F + b] = off;
            }
            st_left = 0; st_fill = FJ_CHUNK; st_cur = FJ_DIR_INVALID; st_nch = 0;
        }
        __syncthreads();
        if (tid == 0) commit_units(misc[M_FLUSH]);
        __syncthreads();
    };

    // ---- prologue: keys of the first tile into registers, metadata of the second into LDS,
    //      descriptor of the third into registers (every dependent load is issued a tile early) ----
    u64 kn[KPT], vn[KPT];
    Pk7Raw rn[KPT / 4 ? KPT / 4 : 1];
    u32 validn = 0, par_next = a.parent0, par_n2 = a.parent0, mid = 0, mcnt = 0;
    u32 d3_pos = 0, d3_len = 0, d3_par = 0;              // descriptor of tile t+2 at the top of iteration t
    if (FLAT) {
        key_load(t, kn, vn, validn);
    } else {
        u32 p, l;
        get_desc(t, p, l, par_next);
        meta_fetch(p, l, mid, mcnt);
        if (tid < TC) { t_chunk[tid] = mid; t_cnt[tid] = mcnt; }
        __syncthreads();
        if constexpr (PK7) key_load7(t, rn, vn, validn); else key_load(t, kn, vn, validn);
        mid = 0; mcnt = 0;
        if (t + 1 < thi) { get_desc(t + 1, p, l, par_n2); meta_fetch(p, l, mid, mcnt); }
        if (t + 2 < thi) get_desc(t + 2, d3_pos, d3_len, d3_par);
        __syncthreads();
        if (tid < TC) { t_chunk[tid] = mid; t_cnt[tid] = mcnt; }
    }
    __syncthreads();

    u32 cur_parent = 0xFFFFFFFFu;
    u32 hotb = 0xFFFFFFFFu;                          // see the ranking step
    for (; t < thi; ++t) {
        u64 k[KPT], v[KPT];
#pragma unroll
        for (int i = 0; i < KPT; ++i) { k[i] = FLAT ? fj_key_mix(kn[i]) : kn[i]; if (HAS_VALS) v[i] = vn[i]; }   // flat arrays hold raw keys, chunk pools mixed ones (fj_common.h)
        const u32 valid = validn, parent = par_next;
        if constexpr (PK7) {
            // bits 56..63 of every key of this tile: the top bits of the first-pass bucket its chunks belong to
            const u64 top = (u64)((a.in_b0 + parent) >> a.in_top_shift) << 56;
#pragma unroll
            for (int i = 0; i < KPT / 4; ++i) {
                const u32 lo[4] = {rn[i].lo.x, rn[i].lo.y, rn[i].lo.z, rn[i].lo.w};
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const u32 md = ((m & 2) ? rn[i].mid.y : rn[i].mid.x) >> ((m & 1) * 16) & 0xFFFFu;
                    const u32 hb = (rn[i].hi >> (8 * m)) & 0xFFu;
                    k[4 * i + m] = top | ((u64)((hb << 16) | md) << 32) | lo[m];
                }
            }
        }
        // keep HBM busy: tile t+2's chunk-list entries first (so that waiting for them later does not
        // wait for the younger key loads), then tile t+1's keys
        const u32 par_t1 = par_n2;
        if (!FLAT) {
            mid = 0; mcnt = 0;
            if (t + 2 < thi) { par_n2 = d3_par; meta_fetch(d3_pos, d3_len, mid, mcnt); }
            if (t + 3 < thi) get_desc(t + 3, d3_pos, d3_len, d3_par);
        }
        if (t + 1 < thi) { if constexpr (PK7) key_load7(t + 1, rn, vn, validn); else key_load(t + 1, kn, vn, validn); par_next = par_t1; }

        carry();
        if (parent != cur_parent) {
            if (cur_parent != 0xFFFFFFFFu) flush(cur_parent);
            if (tid == 0) {
                const u32 seg = atomicAdd(a.seg_counter, 1u);
                if (seg >= a.max_segs) atomicOr(a.err, FJ_ERR_POOL);
                misc[M_SEG] = seg;
            }
            cur_parent = parent;
        }
        for (u32 b = tid; b <= F; b += NT) hist[b] = 0;     // bucket F = dummy bucket for invalid lanes
        __syncthreads();

        // ---- hash, rank inside the tile with LDS atomics ------------------------------------------
        // Lanes of one instruction that hit the same counter are served one after the other, so a hot key (many equal
        // digits per wave) would make this the slowest step of the tile.  Once per tile a wave tests whether lane 0's
        // last digit is shared by >= 8 lanes; such a bucket is remembered (hotb, wave-uniform, survives across tiles
        // until it cools down) and its lanes are then ranked with ONE atomic + a ballot prefix per instruction.  With
        // uniform keys the test never fires and the loop below is the plain one.
        u32 br[KPT];
        if (hotb == 0xFFFFFFFFu) {
#pragma unroll
            for (int i = 0; i < KPT; ++i) {
                const u32 b = (valid & (1u << i)) ? ((FJ_HW1(k[i]) >> sh32) & FM) : F;
                br[i] = (b << 16) | atomicAdd(&hist[b], 1u);
            }
            // detection, for the tiles that follow: is lane 0's last digit shared by >= 8 lanes of this wave?
            const u32 bl = br[KPT - 1] >> 16;
            const u32 cand = (u32)__builtin_amdgcn_readfirstlane((int)bl);
            if (__popcll(__ballot(bl == cand)) >= 8) hotb = cand;
        } else {
            u32 seen = 0;
#pragma unroll
            for (int i = 0; i < KPT; ++i) {
                const u32 b = (valid & (1u << i)) ? ((FJ_HW1(k[i]) >> sh32) & FM) : F;
                const u64 m = __ballot(b == hotb);
                const u32 n = (u32)__popcll(m);
                seen += n;
                u32 r;
                if (n >= 2) {
                    const int leader = __builtin_ctzll(m);
                    u32 base = 0;
                    if ((int)lane == leader) base = atomicAdd(&hist[hotb], n);
                    base = (u32)__builtin_amdgcn_readlane((int)base, leader);
                    if (b == hotb) r = base + (u32)__popcll(m & ((1ull << lane) - 1ull));
                    else r = atomicAdd(&hist[b], 1u);
                } else {
                    r = atomicAdd(&hist[b], 1u);
                }
                br[i] = (b << 16) | r;
            }
            if (seen < 8) hotb = 0xFFFFFFFFu;                // cooled down
        }
        __syncthreads();

        // ---- packed exclusive scan over the buckets (only the waves that own buckets) -----------
        //      fields: tile offset | line offset | new chunks.   A bucket's *virtual run* is its
        //      carried remainder (lo_*, st_left keys) followed by its new keys (tile region).
        u32 cnt = 0, tot = 0, nf = 0, km = 0, kr = 0, kb = 0, l0 = 0;       // km new chunks, kr new allocation units, kb = prefix of kr
        if (wave * 64 < F) {
            u64 x = 0;
            if (tid < F) {
                cnt = hist[tid];
                tot = st_left + cnt;
                nf = tot & ~(LINE - 1);
                km = nf ? ((st_fill + nf - 1) >> FJ_CHUNK_LOG) : 0;
                const u32 rl = run_left(st_cur, tid);
                kr = RUNS ? (km > rl ? (km - rl + RU - 1u) >> RL : 0u) : km;
                x = (u64)cnt | ((u64)(nf >> LINE_LOG) << 20) | ((u64)kr << 40);
            }
            u64 inc = x;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const u64 y = __shfl_up(inc, d, 64);
                if ((int)lane >= d) inc += y;
            }
            if (lane == 63) wsum[wave] = inc;
            kb = (u32)((inc - x) >> 40); l0 = (u32)(((inc - x) >> 20) & 0xFFFFFu);
            if (tid < F) toff[tid] = (u32)((inc - x) & 0xFFFFFu);      // in-wave part, completed below
        }
        __syncthreads();
        if (wave * 64 < F) {
            u64 woff = 0;
            for (u32 w = 0; w < wave; ++w) woff += wsum[w];
            if (tid < F) {
                const u32 to = toff[tid] + (u32)(woff & 0xFFFFFu);
                toff[tid] = to;
                l0 += (u32)((woff >> 20) & 0xFFFFFu);
                kb += (u32)(woff >> 40);
                if (tid == F - 1) {
                    toff[F] = to + cnt;                    // dummy bucket goes behind everything
                    const u32 need = kb + kr;
                    misc[M_NLINES] = l0 + (nf >> LINE_LOG);
                    misc[M_NEED] = need;
                    if (need > misc[M_SLAB_REM]) take_slabs(need);
                }
            }
        }
        __syncthreads();

        // ---- bucket-sort the tile's new keys in LDS ---------------------------------------------
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            const u32 d = toff[br[i] >> 16] + (br[i] & 0xFFFFu);
            tile_k[d] = k[i];
            if (HAS_VALS) tile_v[d] = v[i];
        }
        if (wave * 64 < F) {
            // line descriptors: dst element (32) | bucket (10) | keys of this line that still sit in lo_* (5)
            //                   | tile index of the line's virtual key 0, biased by 32 (17)
            // Thread b writes the descriptors of bucket b -- up to `own_lines` of them (twice the average); a bucket
            // with more (a hot key: up to T/LINE lines) is finished by its whole wave, 64 descriptors per step.
            const bool own = tid < F;
            const u32 f0 = st_fill, c0 = st_cur, tb = own ? toff[tid] : 0u;
            const u32 nl = own ? (nf >> LINE_LOG) : 0u;
            auto put = [&](u32 pb, u32 pf0, u32 pc0, u32 pkb, u32 ptb, u32 pleft, u32 pl0, u32 j) {
                const u32 q = j << LINE_LOG, pq = pf0 + q, kk = pq >> FJ_CHUNK_LOG, off = pq & (FJ_CHUNK - 1);
                const u32 id = kk == 0 ? pc0 : chunk_id(kk, pc0, pkb, pb);
                const u32 dst = id < cap ? id * FJ_CHUNK + off : FJ_DIR_INVALID;
                const u32 lc = j == 0 ? pleft : 0u;
                const u32 sidx = ptb + q + 32u - pleft;             // tile index of virtual key q (may precede the run for line 0)
                line_desc[pl0 + j] = ((u64)dst << 32) | ((u64)pb << 22) | ((u64)lc << 17) | sidx;
            };
            const u32 avg2 = (2u * T / LINE) >> a.fan_log;          // twice the average number of lines per bucket and tile
            const u32 own_lines = avg2 > 4u ? avg2 : 4u;
            for (u32 j = 0; j < nl && j < own_lines; ++j) put(tid, f0, c0, kb, tb, st_left, l0, j);
            u64 big = __ballot(nl > own_lines);
            while (big) {                                           // wave-uniform
                const int src = __builtin_ctzll(big);
                big &= big - 1;
                const u32 pf0 = (u32)__builtin_amdgcn_readlane((int)f0, src), pc0 = (u32)__builtin_amdgcn_readlane((int)c0, src);
                const u32 pkb = (u32)__builtin_amdgcn_readlane((int)kb, src), ptb = (u32)__builtin_amdgcn_readlane((int)tb, src);
                const u32 pleft = (u32)__builtin_amdgcn_readlane((int)st_left, src), pl0 = (u32)__builtin_amdgcn_readlane((int)l0, src);
                const u32 pnl = (u32)__builtin_amdgcn_readlane((int)nl, src);
                for (u32 j = own_lines + lane; j < pnl; j += 64) put(wave * 64 + (u32)src, pf0, pc0, pkb, ptb, pleft, pl0, j);
            }
        }
        if (tid < F) {
            const u32 b = tid, f0 = st_fill, c0 = st_cur, n0 = st_nch, tb = toff[b];
            const u32 outb = (parent * F + b) << FJ_DIR_CNT_BITS;
            const u64 segw = (u64)misc[M_SEG] << 32;
            for (u32 kk = 1; kk <= km; ++kk) {
                const u32 id = chunk_id(kk, c0, kb, b);
                if (id < cap) { a.out_dir[id] = outb | FJ_CHUNK; a.out_rel[id] = segw | (n0 + kk - 1); }
            }
            // remember what the carry step needs (it runs at the top of the next iteration)
            pend_cnt = cnt; pend_nf = nf; pend_tb = tb;
            st_cur = km ? chunk_id(km, c0, kb, b) : c0;
            st_fill = f0 + nf - (km << FJ_CHUNK_LOG);
            st_nch = n0 + km;
        }
        __syncthreads();

        // metadata of tile t+2 (fetched at the top of this iteration) -> LDS, BEFORE this tile's stores are
        // issued: the wait for those loads must not also wait for the stores
        if (!FLAT && tid < TC) { t_chunk[tid] = mid; t_cnt[tid] = mcnt; }
        // ---- write whole lines --------------------------------------------------------------------
        const u32 nl = misc[M_NLINES];
        // 16 B (2 keys) per lane, LINE/2 consecutive lanes per line: one store instruction covers whole, contiguous lines
        // (1 % faster end to end than 32 B per lane in two instructions, A/B on one box)
        constexpr u32 WK = 2, WLPL = LINE / WK;
        for (u32 e = tid; e < nl * WLPL; e += NT) {
            const u32 l = e / WLPL, q = (e % WLPL) * WK;
            const u64 d = line_desc[l];
            const u32 dst = (u32)(d >> 32);
            if (dst != FJ_DIR_INVALID) {
                const u32 w = (u32)d, b = w >> 22, lc = (w >> 17) & 31u, sidx = (w & 0x1FFFFu) - 32u;
                u64 r[WK], rv[WK];
#pragma unroll
                for (int i = 0; i < (int)WK; ++i) {
                    const u32 vi = q + i;
                    const bool in_lo = vi < lc;
                    const u64* sk = in_lo ? (lo_k + b * LOS + vi) : (tile_k + (sidx + vi));
                    r[i] = *sk;
                    if (HAS_VALS) { const u64* sv = in_lo ? (lo_v + b * LOS + vi) : (tile_v + (sidx + vi)); rv[i] = *sv; }
                }
                u64x2 r0; r0.x = r[0]; r0.y = r[1];
                *reinterpret_cast<u64x2*>(a.out_keys + (u64)dst + q) = r0;
                if (HAS_VALS) { u64x2 w0; w0.x = rv[0]; w0.y = rv[1]; *reinterpret_cast<u64x2*>(a.out_vals + (u64)dst + q) = w0; }
            }
        }
        if (tid == 0) commit_units(misc[M_NEED]);
        __syncthreads();
    }
    carry();
    flush(cur_parent);
    // The chunk ids this workgroup took from the allocator but never used stay unlisted: their directory words say so.
    // (Every id below the allocator's high-water mark is thus defined by its owner - the directory needs no memset.)
    for (u32 j = tid; j < (misc[M_SLAB_REM] << RL); j += NT) { const u32 id = misc[M_SLAB_CUR] + j; if (id < cap) a.out_dir[id] = FJ_DIR_INVALID; }
}

// Single-workgroup exclusive scan (bucket counts -> offsets): sweeps of 16384 elements.  A thread takes four groups of four
// consecutive elements, 4096 elements apart (every load and store instruction covers 1 KiB of consecutive addresses: the
// scan runs on ONE CU, whose address unit handles one cache line per clock - 64-B-strided accesses were 5x slower);
// wave shuffles + 64 LDS words, two barriers per sweep.  Returns the total.
template <typename T, typename In, typename Out>
__device__ __forceinline__ T fj_block_scan(In value_at, Out put, u32 n) {
    __shared__ T wtot[4][16];
    const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    T carry = 0;
    for (u32 base = 0; base < n; base += 16384) {
        T x[4][4], sum[4], inc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const u32 e0 = base + 4u * ((u32)j * 1024u + tid);
#pragma unroll
            for (int k = 0; k < 4; ++k) x[j][k] = e0 + k < n ? (T)value_at(e0 + k) : (T)0;
            sum[j] = x[j][0] + x[j][1] + x[j][2] + x[j][3];
            T v = sum[j];
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const T y = __shfl_up(v, d, 64); if ((int)lane >= d) v += y; }
            inc[j] = v;
            if (lane == 63) wtot[j][wave] = v;
        }
        __syncthreads();
        T before = carry, total = carry;               // sum of the (group, wave) cells in front of this thread's / of the whole sweep
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            T mine = 0, all = 0;
            for (u32 w = 0; w < 16; ++w) { const T c = wtot[j][w]; all += c; if (w < wave) mine += c; }
            T run = total + mine + inc[j] - sum[j];
            const u32 e0 = base + 4u * ((u32)j * 1024u + tid);
#pragma unroll
            for (int k = 0; k < 4; ++k) { if (e0 + k < n) put(e0 + k, run); run += x[j][k]; }
            total += all;
        }
        (void)before;
        carry = total;
        __syncthreads();                               // wtot is rewritten by the next sweep / a following scan
    }
    return carry;
}

// After a pass, one workgroup: per-bucket chunk counts -> chunk-list offsets boff[0..n] and - for a consumer that reads
// `tc` chunks per tile - the tile offsets toff[0..n], both from one read of the counts (a packed 64-bit scan: chunks in the
// low word, tiles in the high word); the counts are cleared for the next join (nobody else reads them).
// n % 4 == 0 (every level has a power-of-two bucket count >= 32): 16-B loads and stores throughout.
__global__ __launch_bounds__(1024) void fj_level_scan(u32* __restrict__ bchunks, u32* __restrict__ boff, u32 n, u32 tc,
                                                      u32* __restrict__ toff) {
    __shared__ u64 wtot[4][16];
    const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float rtc = tc ? 1.0f / (float)tc : 0.f;
    auto tiles_of = [&](u32 c) -> u32 {          // ceil(c / tc) without an integer division (c < 2^24: exact after one correction)
        if (!tc) return 0u;
        u32 q = (u32)((float)c * rtc);
        while (q * tc < c) ++q;
        while (q && (q - 1) * tc >= c) --q;
        return q;
    };
    u64 carry = 0;
    for (u32 base = 0; base < n; base += 16384) {
        uint4 x[4]; u64 sum[4], inc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const u32 e0 = base + 4u * ((u32)j * 1024u + tid);
            x[j] = make_uint4(0, 0, 0, 0);
            if (e0 < n) { x[j] = *reinterpret_cast<const uint4*>(bchunks + e0); *reinterpret_cast<uint4*>(bchunks + e0) = make_uint4(0, 0, 0, 0); }
            sum[j] = ((u64)x[j].x + x[j].y + x[j].z + x[j].w) | ((u64)(tiles_of(x[j].x) + tiles_of(x[j].y) + tiles_of(x[j].z) + tiles_of(x[j].w)) << 32);
            u64 v = sum[j];
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const u64 y = __shfl_up(v, d, 64); if ((int)lane >= d) v += y; }
            inc[j] = v;
            if (lane == 63) wtot[j][wave] = v;
        }
        __syncthreads();
        u64 total = carry;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            u64 mine = 0, all = 0;
            for (u32 w = 0; w < 16; ++w) { const u64 c = wtot[j][w]; all += c; if (w < wave) mine += c; }
            u64 run = total + mine + inc[j] - sum[j];
            const u32 e0 = base + 4u * ((u32)j * 1024u + tid);
            if (e0 < n) {
                uint4 ob, ot;
                ob.x = (u32)run; ot.x = (u32)(run >> 32); run += (u64)x[j].x | ((u64)tiles_of(x[j].x) << 32);
                ob.y = (u32)run; ot.y = (u32)(run >> 32); run += (u64)x[j].y | ((u64)tiles_of(x[j].y) << 32);
                ob.z = (u32)run; ot.z = (u32)(run >> 32); run += (u64)x[j].z | ((u64)tiles_of(x[j].z) << 32);
                ob.w = (u32)run; ot.w = (u32)(run >> 32);
                *reinterpret_cast<uint4*>(boff + e0) = ob;
                if (tc) *reinterpret_cast<uint4*>(toff + e0) = ot;
            }
            total += all;
        }
        carry = total;
        __syncthreads();
    }
    if (tid == 0) { boff[n] = (u32)carry; if (tc) toff[n] = (u32)(carry >> 32); }
}

// The same scan for a level of many buckets (8192 .. 2^19: second passes of plans of 14+ bits), one 1024-thread workgroup per
// 4096 counts instead of one workgroup for all of them (262144 buckets = 16 sweeps on one CU took 120 us; two of them per
// 18-bit join).  No workgroup waits for another: each first sums the counts in front of its own piece - the array is at most
// 2 MiB and sits in L2 - then scans its piece.  Nobody may clear a count that a later workgroup still reads, so the counts are
// cleared by the launch that follows (fj_level_lists, `clr`).
__global__ __launch_bounds__(1024) void fj_level_scan_wide(const u32* __restrict__ bchunks, u32* __restrict__ boff, u32 n, u32 tc,
                                                           u32* __restrict__ toff) {
    __shared__ u64 wtot[2][16];
    const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = blockIdx.x;
    // ceil(c / tc): umulhi(c + tc - 1, ceil(2^32 / tc)) overshoots by at most one (c < 2^24 chunks: the multiplier's rounding adds
    // < (c + tc) / 2^32 to the quotient), one conditional step back makes it exact for every tc.  This kernel evaluates it for every
    // count in FRONT of its piece too: with the one-workgroup scan's float estimate + correction loops a 262144-bucket scan with a
    // tile table took 68 us against 14 without.
    const u32 magic = tc > 1 ? (u32)((0x100000000ull + tc - 1) / tc) : 0u;
    auto tiles_of = [&](u32 c) -> u32 {
        if (tc <= 1) return tc ? c : 0u;
        const u32 q = __umulhi(c + tc - 1u, magic);
        return (q && (q - 1u) * tc >= c) ? q - 1u : q;
    };
    auto packed = [&](const uint4& x) -> u64 {
        return ((u64)x.x + x.y + x.z + x.w) | ((u64)(tiles_of(x.x) + tiles_of(x.y) + tiles_of(x.z) + tiles_of(x.w)) << 32);
    };
    const u32 e0 = g * 4096u + 4u * tid;
    uint4 x = make_uint4(0, 0, 0, 0);
    if (e0 < n) x = *reinterpret_cast<const uint4*>(bchunks + e0);
    u64 pre = 0;
#pragma unroll 4
    for (u32 j = 0; j < g; ++j) pre += packed(*reinterpret_cast<const uint4*>(bchunks + 4u * (j * 1024u + tid)));
    const u64 own = packed(x);
    u64 inc = own;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const u64 y = __shfl_up(inc, d, 64), z = __shfl_up(pre, d, 64);
        if ((int)lane >= d) { inc += y; pre += z; }
    }
    if (lane == 63) { wtot[0][wave] = inc; wtot[1][wave] = pre; }
    __syncthreads();
    u64 run = inc - own;
    for (u32 w = 0; w < 16; ++w) { run += wtot[1][w]; if (w < wave) run += wtot[0][w]; }
    if (e0 < n) {
        uint4 ob, ot;
        ob.x = (u32)run; ot.x = (u32)(run >> 32); run += (u64)x.x | ((u64)tiles_of(x.x) << 32);
        ob.y = (u32)run; ot.y = (u32)(run >> 32); run += (u64)x.y | ((u64)tiles_of(x.y) << 32);
        ob.z = (u32)run; ot.z = (u32)(run >> 32); run += (u64)x.z | ((u64)tiles_of(x.z) << 32);
        ob.w = (u32)run; ot.w = (u32)(run >> 32); run += (u64)x.w | ((u64)tiles_of(x.w) << 32);
        *reinterpret_cast<uint4*>(boff + e0) = ob;
        if (tc) *reinterpret_cast<uint4*>(toff + e0) = ot;
        if (e0 + 4u == n) { boff[n] = (u32)run; if (tc) toff[n] = (u32)(run >> 32); }       // (n % 4 == 0)
    }
}

// tile t -> (first list index, chunks, bucket)
// BALANCE (the join's items): a bucket's k tiles share its chunks evenly, ceil(chunks / k) each, instead of k-1 full tiles and
// a short one - items of one size keep the workgroup slots level (k <= tc keeps every tile non-empty).
template <bool BALANCE>
__device__ __forceinline__ void fj_tile_expand_one(const u32* __restrict__ boff, const u32* __restrict__ toff, u32 n, u32 tc,
                                                   uint4* __restrict__ tiles, u32 t) {
    u32 lo = 0, hi = n;                       // last p with toff[p] <= t
    while (hi - lo > 1) { const u32 mid = (lo + hi) >> 1; if (toff[mid] <= t) lo = mid; else hi = mid; }
    u32 step = tc;
    if (BALANCE) {
        const u32 k = toff[lo + 1] - toff[lo], c = boff[lo + 1] - boff[lo];
        if (k > 1 && k <= tc) step = (c + k - 1) / k;
    }
    const u32 pos = boff[lo] + (t - toff[lo]) * step;
    const u32 rem = boff[lo + 1] - pos;
    tiles[t] = make_uint4(pos, rem < step ? rem : step, lo, 0);
}

// same for u64 outputs (result offsets can exceed 2^32)
__global__ __launch_bounds__(1024) void fj_scan_u32_to_u64(const u32* __restrict__ in, u64* __restrict__ out, u32 n) {
    const u64 total = fj_block_scan<u64>([&](u32 i) { return in[i]; }, [&](u32 i, u64 v) { out[i] = v; }, n);
    if (threadIdx.x == 0) out[n] = total;
}

// chunk lists without atomics: list[boff[bucket] + span offset of the producing segment + rank] = chunk.  The same launch
// expands the consumer's tile table (independent work on the scan's outputs) and clears the tail of an optional per-tile
// array (the join's per-item counts: entries past the device-side item count must read 0).
template <u32 RL>
__global__ __launch_bounds__(256) void fj_level_lists(const u32* __restrict__ dir, const u64* __restrict__ rel, const u32* __restrict__ nalloc,
                               u32 cap, const u32* __restrict__ boff, const u32* __restrict__ seg_off, u32 fan_mask,
                               u32 max_segs, u32* __restrict__ list,
                               u32 nb, u32 tc, const u32* __restrict__ toff, uint4* __restrict__ tiles, u32 max_tiles,
                               u32* __restrict__ zero_tail, u32* __restrict__ clr) {
    // a chain of dependent loads per chunk (dir/rel -> boff/seg_off -> store): four chains per thread and step are kept in
    // flight (the kernel is latency-bound: ~4M chunks at c3), and the grid fills the chip's thread slots.
    // RL > 0: the ids came in aligned runs of 2^RL per (segment, bucket), used in order with consecutive ranks - one chain
    // places a whole run (a 16-B directory load, one rel word, one pair of gathers, neighbouring list entries).
    u32 n = *nalloc; if (n > cap) n = cap;
    const u32 stride = gridDim.x * blockDim.x, gtid = blockIdx.x * blockDim.x + threadIdx.x;
    if (clr) for (u32 i = gtid; i < nb / 4u; i += stride) reinterpret_cast<uint4*>(clr)[i] = make_uint4(0, 0, 0, 0);    // the bucket counts fj_level_scan_wide read (nb % 4 == 0)
    if (tc) {
        u32 total = toff[nb];
        if (total > max_tiles) total = max_tiles;
        for (u32 t = gtid; t < max_tiles; t += stride) {
            if (t < total) { if (zero_tail) fj_tile_expand_one<true>(boff, toff, nb, tc, tiles, t); else fj_tile_expand_one<false>(boff, toff, nb, tc, tiles, t); }
            else if (zero_tail) zero_tail[t] = 0;
        }
    }
    if constexpr (RL > 0) {
        constexpr u32 R = 1u << RL;
        const u32 nr = n >> RL;                               // (the allocator moves in whole slabs: n % R == 0)
        for (u32 i0 = gtid; i0 < nr; i0 += 2 * stride) {
            u32 ev[2][R]; u64 r[2][R];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const u32 i = i0 + u * stride;
#pragma unroll
                for (u32 j = 0; j < R; ++j) { ev[u][j] = FJ_DIR_INVALID; r[u][j] = 0; }
                if (i < nr) {
                    if constexpr (RL == 2) {
                        const uint4 e = reinterpret_cast<const uint4*>(dir)[i];
                        ev[u][0] = e.x; ev[u][1] = e.y; ev[u][2] = e.z; ev[u][3] = e.w;
                    } else {
                        const uint2 e = reinterpret_cast<const uint2*>(dir)[i];
                        ev[u][0] = e.x; ev[u][1] = e.y;
                    }
#pragma unroll
                    for (u32 j = 0; j < R; j += 2) {
                        const ulonglong2 q = reinterpret_cast<const ulonglong2*>(rel)[((u64)i * R + j) >> 1];
                        r[u][j] = q.x; r[u][j + 1] = q.y;
                    }
                }
            }
            u32 base[2]; bool ok[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                // the (bucket, segment) of the run from any used id (the ids of a run are used in a rotated order)
                u32 e0 = FJ_DIR_INVALID; u64 r0 = 0;
#pragma unroll
                for (int j = (int)R - 1; j >= 0; --j) if (ev[u][j] != FJ_DIR_INVALID) { e0 = ev[u][j]; r0 = r[u][j]; }
                const u32 b = e0 >> FJ_DIR_CNT_BITS, seg = (u32)(r0 >> 32);
                ok[u] = e0 != FJ_DIR_INVALID && seg < max_segs;
                base[u] = ok[u] ? boff[b] + seg_off[(u64)seg * (fan_mask + 1) + (b & fan_mask)] : 0;
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (!ok[u]) continue;
                const u32 id = R * (i0 + u * stride);
#pragma unroll
                for (u32 j = 0; j < R; ++j)
                    if (ev[u][j] != FJ_DIR_INVALID) list[base[u] + (u32)r[u][j]] = (((ev[u][j] & FJ_DIR_CNT_MASK) - 1u) << 24) | (id + j);
            }
        }
    } else {
        for (u32 i0 = gtid; i0 < n; i0 += 4 * stride) {
            u32 e[4]; u64 r[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const u32 i = i0 + u * stride;
                e[u] = i < n ? dir[i] : FJ_DIR_INVALID;
                r[u] = i < n ? rel[i] : 0;
            }
            u32 pos[4]; bool ok[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const u32 b = e[u] >> FJ_DIR_CNT_BITS, seg = (u32)(r[u] >> 32);
                ok[u] = e[u] != FJ_DIR_INVALID && seg < max_segs;
                pos[u] = ok[u] ? boff[b] + seg_off[(u64)seg * (fan_mask + 1) + (b & fan_mask)] + (u32)r[u] : 0;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (ok[u]) list[pos[u]] = (((e[u] & FJ_DIR_CNT_MASK) - 1u) << 24) | (i0 + u * stride);
        }
    }
}

// The same lists for a level of few buckets (<= 512) whose ids stand alone (flat passes of <= 256 buckets, the bloom stage,
// chunks received from other GPUs).  There the scattered 4-byte stores are what fj_level_lists<0> spends its time on (68 us per
// 1B-row level; 19 us without the stores, 21 us with linear ones), and a block of consecutive ids holds many chunks per
// bucket whose list positions are neighbours (consecutive ranks of one producer segment).  A 1024-thread workgroup takes 4096
// consecutive ids, bins their (position, entry) pairs by bucket in LDS (counting sort: LDS atomics, one scan) and writes the
// bins out in order: a wave's 64 stores then fall into a few 16-64-byte pieces instead of 64 sectors.  Positions and entries are
// exactly those of fj_level_lists<0>.
//
// Round 6: the level's SCAN rides along (fj_level_scan's job: chunk counts -> list offsets boff and the consumer's tile offsets
// toff, counts cleared) - one launch per level instead of two.  Every workgroup scans the <= 512 counts itself (2 KiB from L2) into
// LDS and works from there; workgroup 0 also writes boff / toff out for the level's consumers; the counts may be cleared once
// everybody has read them: each workgroup ticks a counter behind the counts (bchunks[nb]: zero at rest like the counts) when it
// has, and the one that ticks last clears counts and counter.  FUSED = false: the offsets come from fj_level_scan (levels of more than
// 512 workgroups - a 1B-row level has 1200 - where 1200 little scans cost more than the launch they save: 80 us against 8 + 35).
template <bool FUSED>
__global__ __launch_bounds__(1024) void fj_level_lists_binned(const u32* __restrict__ dir, const u64* __restrict__ rel, const u32* __restrict__ nalloc,
                               u32 cap, u32* __restrict__ boff_g, const u32* __restrict__ seg_off, u32 fan_mask,
                               u32 max_segs, u32* __restrict__ list,
                               u32 nb, u32 tc, u32* __restrict__ toff_g, uint4* __restrict__ tiles, u32 max_tiles,
                               u32* __restrict__ zero_tail, u32* __restrict__ bchunks) {
    constexpr u32 NT = 1024, BLK = 4096, NBMAX = 512;
    __shared__ u32 hist[NBMAX + 1];
    __shared__ u32 wsum[NBMAX / 64];
    __shared__ uint2 ent[BLK];
    __shared__ u32 boff_s[FUSED ? NBMAX + 1 : 1], toff_s[FUSED ? NBMAX + 1 : 1];
    __shared__ u64 wtot[NBMAX / 64];
    __shared__ u32 s_last;
    const u32* boff = FUSED ? boff_s : boff_g;
    const u32* toff = FUSED ? toff_s : toff_g;
    u32 n = *nalloc; if (n > cap) n = cap;
    const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const u32 stride = gridDim.x * NT, gtid = blockIdx.x * NT + tid;
    if constexpr (FUSED) {   // ---- the scan (chunks in the low word, tiles in the high word of one 64-bit scan, as fj_level_scan does it) ----
        const float rtc = tc ? 1.0f / (float)tc : 0.f;
        auto tiles_of = [&](u32 c) -> u32 {          // ceil(c / tc) without an integer division (c < 2^24: exact after one correction)
            if (!tc) return 0u;
            u32 q = (u32)((float)c * rtc);
            while (q * tc < c) ++q;
            while (q && (q - 1) * tc >= c) --q;
            return q;
        };
        u64 mine = 0, v = 0;
        if (tid < NBMAX) {
            const u32 c = tid < nb ? bchunks[tid] : 0u;
            mine = (u64)c | ((u64)tiles_of(c) << 32);
            v = mine;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const u64 y = __shfl_up(v, d, 64); if ((int)lane >= d) v += y; }
            if (lane == 63) wtot[wave] = v;
        }
        __syncthreads();                               // this workgroup has read every count
        if (tid == 0) { __threadfence(); s_last = atomicAdd(&bchunks[nb], 1u) == gridDim.x - 1 ? 1u : 0u; }
        if (tid < NBMAX) {
            u64 woff = 0;
            for (u32 w = 0; w < wave; ++w) woff += wtot[w];
            const u64 ex = woff + v - mine;
            if (tid < nb) { boff_s[tid] = (u32)ex; toff_s[tid] = (u32)(ex >> 32); }
            if (tid == nb - 1) { boff_s[nb] = (u32)(ex + mine); toff_s[nb] = (u32)((ex + mine) >> 32); }
        }
        __syncthreads();
        if (s_last) { if (tid < nb) bchunks[tid] = 0; if (tid == 0) bchunks[nb] = 0; }       // every workgroup has read them: cleared for the next join
        if (blockIdx.x == 0 && tid <= nb) { boff_g[tid] = boff_s[tid]; if (tc) toff_g[tid] = toff_s[tid]; }
    }
    if (tc) {
        u32 total = toff[nb];
        if (total > max_tiles) total = max_tiles;
        for (u32 t = gtid; t < max_tiles; t += stride) {
            if (t < total) { if (zero_tail) fj_tile_expand_one<true>(boff, toff, nb, tc, tiles, t); else fj_tile_expand_one<false>(boff, toff, nb, tc, tiles, t); }
            else if (zero_tail) zero_tail[t] = 0;
        }
    }
    for (u32 base = blockIdx.x * BLK; base < n; base += gridDim.x * BLK) {
        if (tid < NBMAX) hist[tid] = 0;
        __syncthreads();
        u32 e[4]; u64 r[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const u32 i = base + (u32)u * NT + tid;
            e[u] = i < n ? dir[i] : FJ_DIR_INVALID;
            r[u] = i < n ? rel[i] : 0;
        }
        u32 pos[4], slot[4], bk[4]; bool ok[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const u32 b = e[u] >> FJ_DIR_CNT_BITS, seg = (u32)(r[u] >> 32);
            ok[u] = e[u] != FJ_DIR_INVALID && seg < max_segs && b < nb;
            bk[u] = ok[u] ? b : 0u;
            pos[u] = ok[u] ? boff[b] + seg_off[(u64)seg * (fan_mask + 1) + (b & fan_mask)] + (u32)r[u] : 0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) slot[u] = ok[u] ? atomicAdd(&hist[bk[u]], 1u) : 0u;
        __syncthreads();
        // exclusive scan of the bin sizes (the first NBMAX threads: whole waves)
        u32 v = 0, inc = 0;
        if (tid < NBMAX) {
            v = hist[tid]; inc = v;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const u32 y = __shfl_up(inc, d, 64); if ((int)lane >= d) inc += y; }
            if (lane == 63) wsum[wave] = inc;
        }
        __syncthreads();
        if (tid < NBMAX) {
            u32 woff = 0;
            for (u32 w = 0; w < wave; ++w) woff += wsum[w];
            hist[tid] = woff + inc - v;
            if (tid == NBMAX - 1) hist[NBMAX] = woff + inc;
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (ok[u]) ent[hist[bk[u]] + slot[u]] = make_uint2(pos[u], (((e[u] & FJ_DIR_CNT_MASK) - 1u) << 24) | (base + (u32)u * NT + tid));
        __syncthreads();
        const u32 total = hist[NBMAX];
        for (u32 j = tid; j < total; j += NT) { const uint2 x = ent[j]; list[x.x] = x.y; }
        __syncthreads();
    }
}

// Chunks that arrived from other GPUs (owner shuffle: every sender's region for this owner, concatenated) -> a chunk set the
// level bookkeeping understands.  Per chunk only its directory word came over the wire; this kernel gives every chunk its
// rank inside (block of 4096 chunks, bucket) with LDS atomics and every (block, bucket) a span of the bucket's chunk list with
// one global atomic - the same (segment, rank, span offset) scheme the partition pass produces for its own output, so that
// fj_level_scan / fj_level_lists run unchanged.  Directory words are rewritten with bucket ids relative to b_lo (this owner's
// first bucket); foreign or unused ids become FJ_DIR_INVALID.
__global__ __launch_bounds__(1024) void fj_dir_rank_kernel(u32* __restrict__ dir, u32 n, u32 b_lo, u32 nbk, u32 fan, u64* __restrict__ rel,
                                                           u32* __restrict__ seg_off, u32* __restrict__ bchunks, u32* __restrict__ nalloc) {
    __shared__ u32 h[1u << FJ_MAX_FAN_LOG];
    const u32 tid = threadIdx.x, base = blockIdx.x * 4096u;
    if (tid < (1u << FJ_MAX_FAN_LOG)) h[tid] = 0;
    __syncthreads();
    u32 bb[4], rr[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const u32 i = base + (u32)u * 1024u + tid;
        bb[u] = FJ_DIR_INVALID; rr[u] = 0;
        if (i < n) {
            const u32 e = dir[i], b = (e >> FJ_DIR_CNT_BITS) - b_lo, cnt = e & FJ_DIR_CNT_MASK;
            const bool ok = e != FJ_DIR_INVALID && b < nbk && cnt >= 1u && cnt <= FJ_CHUNK;
            if (ok) bb[u] = b;
            dir[i] = ok ? ((b << FJ_DIR_CNT_BITS) | cnt) : FJ_DIR_INVALID;
        }
        // rank inside (block, bucket).  What a sender packs is dense and bucket-ordered (fj_pack.hip): the lanes of a wave nearly
        // always share ONE bucket, and 64 LDS atomics on one word are served one after the other (0.6 ms per 1.2M-chunk piece) -
        // such a wave takes its ranks with one atomic and a ballot prefix.
        const bool ok = bb[u] != FJ_DIR_INVALID;
        const u64 m = __ballot(ok);
        if (m) {
            const u32 b0 = (u32)__builtin_amdgcn_readlane((int)bb[u], __builtin_ctzll(m));
            if (__ballot(ok && bb[u] != b0) == 0) {
                u32 wb = 0;
                if ((threadIdx.x & 63u) == (u32)__builtin_ctzll(m)) wb = atomicAdd(&h[b0], (u32)__popcll(m));
                wb = (u32)__builtin_amdgcn_readlane((int)wb, __builtin_ctzll(m));
                if (ok) rr[u] = wb + (u32)__popcll(m & ((1ull << (threadIdx.x & 63u)) - 1ull));
            } else if (ok) rr[u] = atomicAdd(&h[bb[u]], 1u);
        }
    }
    __syncthreads();
    if (tid < nbk) { const u32 c = h[tid]; if (c) seg_off[(u64)blockIdx.x * fan + tid] = atomicAdd(&bchunks[tid], c); }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const u32 i = base + (u32)u * 1024u + tid;
        if (bb[u] != FJ_DIR_INVALID) rel[i] = ((u64)blockIdx.x << 32) | rr[u];
    }
    if (blockIdx.x == 0 && tid == 0) *nalloc = n;
}

template <int NT, int KPT, int LINE_LOG, bool HAS_VALS, bool FLAT, bool PROBE_SIDE, int RLOG = FJ_RUN_LOG, bool PK7 = false>
hipError_t launch_part0(const FjPartArgs& a, u32 grid, hipStream_t s) {
    const u32 F = 1u << a.fan_log;
    const PartLds L = part_lds_layout(NT * KPT, F, 1u << LINE_LOG, HAS_VALS, NT / 64);
    auto kern = fj_partition_kernel<NT, KPT, LINE_LOG, HAS_VALS, FLAT, PROBE_SIDE, RLOG, PK7>;
    hipError_t e = fj_set_max_lds_once(reinterpret_cast<const void*>(kern), L.total);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), L.total, s, a);
    return hipGetLastError();
}

// run length of the output's chunk ids (a.run_log): single ids exist for flat inputs only
template <int NT, int KPT, int LINE_LOG, bool HAS_VALS, bool FLAT, bool PROBE_SIDE>
hipError_t launch_part1(const FjPartArgs& a, u32 grid, hipStream_t s) {
    if constexpr (FLAT) { if (a.run_log == 0) return launch_part0<NT, KPT, LINE_LOG, HAS_VALS, FLAT, PROBE_SIDE, 0>(a, grid, s); }
    if (a.run_log != FJ_RUN_LOG) return hipErrorInvalidValue;
    return launch_part0<NT, KPT, LINE_LOG, HAS_VALS, FLAT, PROBE_SIDE, FJ_RUN_LOG>(a, grid, s);
}

template <int NT, int KPT, int LINE_LOG, bool HAS_VALS, bool FLAT>
hipError_t launch_part(const FjPartArgs& a, u32 grid, hipStream_t s) {
    if constexpr (!HAS_VALS) {
        if (a.side == 0) return launch_part1<NT, KPT, LINE_LOG, HAS_VALS, FLAT, false>(a, grid, s);
    }
    return launch_part1<NT, KPT, LINE_LOG, HAS_VALS, FLAT, true>(a, grid, s);
}

template <int NT, int KPT, bool HAS_VALS>
hipError_t launch_part2(const FjPartArgs& a, int line_log, u32 grid, hipStream_t s) {
    const bool flat = a.in_list == nullptr;
    if (line_log == 3) return flat ? launch_part<NT, KPT, 3, HAS_VALS, true>(a, grid, s) : launch_part<NT, KPT, 3, HAS_VALS, false>(a, grid, s);
    return flat ? launch_part<NT, KPT, 4, HAS_VALS, true>(a, grid, s) : launch_part<NT, KPT, 4, HAS_VALS, false>(a, grid, s);
}

}  // namespace

hipError_t fj_set_max_lds_once(const void* fn, u32 bytes) {
    struct Seen { const void* fn; int dev; u32 bytes; };
    static std::mutex mu;
    static std::vector<Seen> seen;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lk(mu);
    for (Seen& x : seen)
        if (x.fn == fn && x.dev == dev) {
            if (x.bytes >= bytes) return hipSuccess;
            e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
            if (e == hipSuccess) x.bytes = bytes;
            return e;
        }
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) seen.push_back({fn, dev, bytes});
    return e;
}

u32 fj_partition_lds_bytes(u32 fan_log, bool vals, int line_log) {
    const u32 F = 1u << fan_log;
    if (vals) return part_lds_layout(1024 * 4, F, 1u << line_log, true, 16).total;
    return part_lds_layout(1024 * 8, F, 1u << line_log, false, 16).total;
}

u32 fj_partition_tile_chunks(u32 fan_log, bool vals) { (void)fan_log; return vals ? 16u : 32u; }

// One partition pass, one 1024-thread workgroup per CU: keys only 8192-key tiles (8 keys per thread; half as many barriers
// and bucket scans per key as the 4096-key tiles of two 512-thread workgroups), with values 4096-row tiles (4 rows per thread).

hipError_t fj_launch_partition(const FjPartArgs& a, bool vals, int line_log, u32 grid, hipStream_t s) {
    if (a.shift < 32) return hipErrorInvalidValue;       // radix digits must come from hash word 1
    if (a.fan_log > FJ_MAX_FAN_LOG || a.slab < (1u << FJ_RUN_LOG) || (a.slab & ((1u << FJ_RUN_LOG) - 1u))) return hipErrorInvalidValue;
    if (a.in_pk7) {
        // input chunks in the owner shuffle's 7-byte wire format (what other GPUs sent): same pass, keys unpacked in registers
        if (!a.in_list || a.run_log != FJ_RUN_LOG) return hipErrorInvalidValue;
        if (vals) {
            if (a.fan_log == 9 && line_log != 3) return hipErrorInvalidValue;
            return line_log == 3 ? launch_part0<1024, 4, 3, true, false, true, FJ_RUN_LOG, true>(a, grid, s)
                                 : launch_part0<1024, 4, 4, true, false, true, FJ_RUN_LOG, true>(a, grid, s);
        }
        if (line_log != 4) return hipErrorInvalidValue;
        const u32 g = a.fan_log == 9 ? grid : (grid < 256 ? grid : 256);
        return a.side == 0 ? launch_part0<1024, 8, 4, false, false, false, FJ_RUN_LOG, true>(a, g, s)
                           : launch_part0<1024, 8, 4, false, false, true, FJ_RUN_LOG, true>(a, g, s);
    }
    if (a.fan_log == 9) {
        // 512 buckets: one bucket per thread needs >= 512 threads and the open lines take 64 KiB (keys) -- one
        // 1024-thread workgroup per CU; with values the lines shrink to 64 B so that both payloads still fit
        if (vals) return line_log == 3 ? launch_part2<1024, 4, true>(a, 3, grid, s) : hipErrorInvalidValue;
        return launch_part2<1024, 8, false>(a, line_log, grid, s);     // (all 512 workgroups: capping them at 256 cost 2 % at c4)
    }
    if (vals) {
        // 1024 threads x 4 rows: one workgroup per CU (LDS), but 16 waves of it: 1.63 -> 1.42 ms build phase at c3
        return launch_part2<1024, 4, true>(a, line_log, grid, s);
    }
    // keys only: ONE 1024-thread workgroup per CU over 8192-key tiles (late round 2: 3.31 -> 3.22 ms per 1B-key pass against
    // two 512-thread workgroups over 4096-key tiles, the shape of round 1; the per-tile costs - scan, descriptors, barriers -
    // are paid half as often and 16 waves share one set of open lines)
    return launch_part2<1024, 8, false>(a, line_log, grid < 256 ? grid : 256, s);
}

// After a pass, two launches: per-bucket chunk counts -> offsets (+ the consumer's tile offsets), then chunk lists
// (no atomics) + the consumer's tile table.  tc == 0: no consumer tile table.
hipError_t fj_launch_group(const FjChunkSet& cs, u32 tc, u32* toff, uint4* tiles, u32 max_tiles, u32* zero_tail, hipStream_t s) {
    if (cs.nb & 3u) return hipErrorInvalidValue;           // fj_level_scan works in 16-B pieces
    static_assert(FJ_RUN_LOG == 1 || FJ_RUN_LOG == 2, "fj_level_lists reads a run's directory words with one 8-B or 16-B load");
    if (cs.run_log != 0 && cs.run_log != FJ_RUN_LOG) return hipErrorInvalidValue;
    if (cs.run_log == 0 && cs.nb <= 512 && cs.cap >= 4096) {       // few buckets, single ids: binned list entries + tile table - and, up to 512 workgroups, the scan in the same launch
        const u32 blocks = (cs.cap + 4095u) / 4096u;
        if (blocks <= 512u) {
            hipLaunchKernelGGL(fj_level_lists_binned<true>, dim3(blocks), dim3(1024), 0, s, cs.dir, cs.rel, cs.alloc, cs.cap, cs.boff,
                               cs.seg_off, cs.fan_mask, cs.max_segs, cs.list, cs.nb, tc, toff, tiles, max_tiles, zero_tail, cs.bchunks);
        } else {
            hipLaunchKernelGGL(fj_level_scan, dim3(1), dim3(1024), 0, s, cs.bchunks, cs.boff, cs.nb, tc, toff);
            hipLaunchKernelGGL(fj_level_lists_binned<false>, dim3(blocks < 4096u ? blocks : 4096u), dim3(1024), 0, s, cs.dir, cs.rel, cs.alloc, cs.cap, cs.boff,
                               cs.seg_off, cs.fan_mask, cs.max_segs, cs.list, cs.nb, tc, toff, tiles, max_tiles, zero_tail, cs.bchunks);
        }
        return hipGetLastError();
    }
    // many buckets: one workgroup per 4096 counts, and the chunk-list launch clears the counts (fj_level_scan_wide)
    const bool wide = cs.nb >= 8192u && cs.nb <= (1u << 19);
    if (wide) hipLaunchKernelGGL(fj_level_scan_wide, dim3((cs.nb + 4095u) / 4096u), dim3(1024), 0, s, cs.bchunks, cs.boff, cs.nb, tc, toff);
    else hipLaunchKernelGGL(fj_level_scan, dim3(1), dim3(1024), 0, s, cs.bchunks, cs.boff, cs.nb, tc, toff);
    auto kern = cs.run_log ? fj_level_lists<FJ_RUN_LOG> : fj_level_lists<0>;
    hipLaunchKernelGGL(kern, dim3(2048), dim3(256), 0, s, cs.dir, cs.rel, cs.alloc, cs.cap, cs.boff, cs.seg_off,
                       cs.fan_mask, cs.max_segs, cs.list, cs.nb, tc, toff, tiles, max_tiles, zero_tail, wide ? cs.bchunks : (u32*)nullptr);
    return hipGetLastError();
}

hipError_t fj_launch_dir_rank(u32* dir, u32 n, u32 b_lo, u32 nbk, u32 fan, u64* rel, u32* seg_off, u32* bchunks, u32* nalloc, hipStream_t s) {
    if (nbk > (1u << FJ_MAX_FAN_LOG) || fan < nbk || n == 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(fj_dir_rank_kernel, dim3((n + 4095u) / 4096u), dim3(1024), 0, s, dir, n, b_lo, nbk, fan, rel, seg_off, bchunks, nalloc);
    return hipGetLastError();
}

hipError_t fj_launch_scan_u32_to_u64(const u32* in, u64* out, u32 n, hipStream_t s) {
    hipLaunchKernelGGL(fj_scan_u32_to_u64, dim3(1), dim3(1024), 0, s, in, out, n);
    return hipGetLastError();
}
