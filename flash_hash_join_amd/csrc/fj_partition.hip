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

namespace {

// dynamic-LDS layout, shared by kernel and host-side size computation
struct PartLds {
    u32 sorted_k, sorted_v, lo_k, lo_v, hist, left, fill, cur, toff, kbase, lineoff, nfull, nch;
    u32 line_src, line_dst, t_chunk, t_cnt, wsum, misc, total;
};
__host__ __device__ inline PartLds part_lds_layout(u32 T, u32 F, u32 line, bool vals, u32 nwaves) {
    PartLds L;
    u32 nsorted = T + (line - 1) * F;
    u32 maxl = nsorted / line + 2;
    u32 o = 0;
    L.sorted_k = o; o += nsorted * 8;
    L.sorted_v = o; if (vals) o += nsorted * 8;
    L.lo_k = o; o += F * line * 8;
    L.lo_v = o; if (vals) o += F * line * 8;
    L.wsum = o; o += nwaves * 8;
    L.hist = o; o += F * 4;
    L.left = o; o += F * 4;
    L.fill = o; o += F * 4;
    L.cur = o; o += F * 4;
    L.toff = o; o += F * 4;
    L.kbase = o; o += F * 4;
    L.lineoff = o; o += F * 4;
    L.nfull = o; o += F * 4;
    L.nch = o; o += F * 4;
    L.line_src = o; o += maxl * 4;
    L.line_dst = o; o += maxl * 4;
    L.t_chunk = o; o += (T / FJ_CHUNK) * 4;
    L.t_cnt = o; o += (T / FJ_CHUNK) * 4;
    L.misc = o; o += 16 * 4;
    L.total = (o + 15) & ~15u;
    return L;
}

enum { M_SLAB_CUR = 0, M_SLAB_REM, M_NEW_BASE, M_NEED, M_NLINES, M_FLUSH, M_SEG };

template <int NT, int KPT, int LINE_LOG, bool HAS_VALS>
__global__ __launch_bounds__(NT, NT / 128) void fj_partition_kernel(FjPartArgs a) {
    constexpr u32 T = NT * KPT, LINE = 1u << LINE_LOG, TC = T / FJ_CHUNK, NW = NT / 64, LPL = LINE / 2;
    static_assert(T % FJ_CHUNK == 0 && TC <= NT, "tile geometry");
    const u32 F = 1u << a.fan_log, FM = F - 1;
    const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const PartLds Lo = part_lds_layout(T, F, LINE, HAS_VALS, NW);
    u64* sorted_k = (u64*)(smem + Lo.sorted_k);
    u64* sorted_v = (u64*)(smem + Lo.sorted_v);
    u64* lo_k = (u64*)(smem + Lo.lo_k);
    u64* lo_v = (u64*)(smem + Lo.lo_v);
    u64* wsum = (u64*)(smem + Lo.wsum);
    u32* hist = (u32*)(smem + Lo.hist);
    u32* left = (u32*)(smem + Lo.left);
    u32* fill = (u32*)(smem + Lo.fill);
    u32* cur = (u32*)(smem + Lo.cur);
    u32* toff = (u32*)(smem + Lo.toff);
    u32* kbase = (u32*)(smem + Lo.kbase);
    u32* lineoff = (u32*)(smem + Lo.lineoff);
    u32* nfull = (u32*)(smem + Lo.nfull);
    u32* nch = (u32*)(smem + Lo.nch);
    u32* line_src = (u32*)(smem + Lo.line_src);
    u32* line_dst = (u32*)(smem + Lo.line_dst);
    u32* t_chunk = (u32*)(smem + Lo.t_chunk);
    u32* t_cnt = (u32*)(smem + Lo.t_cnt);
    u32* misc = (u32*)(smem + Lo.misc);

    const bool flat = (a.in_list == nullptr);
    const u32 Lc = flat ? (u32)((a.n_flat + FJ_CHUNK - 1) >> FJ_CHUNK_LOG) : 0u;
    const u32 ntiles = flat ? (Lc + TC - 1) / TC : *a.in_ntiles;
    const u32 G = gridDim.x, g = blockIdx.x;
    u32 t = (u32)(((u64)g * ntiles) / G);
    const u32 thi = (u32)(((u64)(g + 1) * ntiles) / G);
    const u32 cap = a.cap_chunks;
    if (t >= thi) return;

    // tile descriptor: first chunk (list index), number of chunks, parent bucket
    auto get_desc = [&](u32 tt, u32& pos, u32& len, u32& parent) {
        if (flat) { pos = tt * TC; len = (Lc - pos) < TC ? (Lc - pos) : TC; parent = a.parent0; }
        else { const uint4 d = a.in_tiles[tt]; pos = d.x; len = d.y; parent = d.z; }
    };
    // chunk id + key count of the tile's tid-th chunk (count 0 beyond the tile)
    auto meta_fetch = [&](u32 pos, u32 len, u32& id, u32& cnt) {
        id = 0; cnt = 0;
        if (tid < len) {
            if (flat) {
                id = pos + tid;
                const u64 rem = a.n_flat - (u64)id * FJ_CHUNK;
                cnt = rem >= FJ_CHUNK ? FJ_CHUNK : (u32)rem;
            } else {
                id = a.in_list[pos + tid];
                cnt = a.in_dir[id] & FJ_DIR_CNT_MASK;
            }
        }
    };
    // issue the tile's loads (16 B per lane) using the chunk metadata currently in LDS
    auto key_load = [&](u64 (&kk)[KPT], u64 (&vv)[KPT], u32& vmask) {
        vmask = 0;
#pragma unroll
        for (int i = 0; i < KPT / 2; ++i) {
            const u32 kidx = ((u32)i * NT + tid) * 2;
            const u32 j = kidx >> FJ_CHUNK_LOG, off = kidx & (FJ_CHUNK - 1);
            const u32 cnt = t_cnt[j];
            kk[2 * i] = 0; kk[2 * i + 1] = 0;
            if (HAS_VALS) { vv[2 * i] = 0; vv[2 * i + 1] = 0; }
            if (off < cnt) {
                const u64 base = (u64)t_chunk[j] * FJ_CHUNK + off;
                if (off + 1 < cnt) {
                    const u64x2 q = *reinterpret_cast<const u64x2*>(a.in_keys + base);
                    kk[2 * i] = q.x; kk[2 * i + 1] = q.y;
                    if (HAS_VALS) {
                        const u64x2 w = *reinterpret_cast<const u64x2*>(a.in_vals + base);
                        vv[2 * i] = w.x; vv[2 * i + 1] = w.y;
                    }
                    vmask |= 3u << (2 * i);
                } else {
                    kk[2 * i] = a.in_keys[base];
                    if (HAS_VALS) vv[2 * i] = a.in_vals[base];
                    vmask |= 1u << (2 * i);
                }
            }
        }
    };

    for (u32 b = tid; b < F; b += NT) { left[b] = 0; fill[b] = FJ_CHUNK; cur[b] = FJ_DIR_INVALID; nch[b] = 0; }
    if (tid == 0) { misc[M_SLAB_CUR] = 0; misc[M_SLAB_REM] = 0; misc[M_NEW_BASE] = 0; misc[M_SEG] = 0; }

    // id of the j-th chunk this workgroup allocates in the current tile
    auto alloc_id = [&](u32 j) -> u32 {
        const u32 rem = misc[M_SLAB_REM];
        return j < rem ? misc[M_SLAB_CUR] + j : misc[M_NEW_BASE] + (j - rem);
    };

    // end of a segment (= this workgroup's share of one parent bucket): write the carried
    // remainders, fix the last chunk's count, reserve the chunk-list spans, reset the state
    auto flush = [&](u32 parent) {
        if (tid == 0) {
            if (misc[M_SLAB_REM] < F) {
                const u32 nb = atomicAdd(a.alloc, FJ_SLAB);
                if (nb + FJ_SLAB > cap) atomicOr(a.err, FJ_ERR_POOL);
                misc[M_SLAB_CUR] = nb; misc[M_SLAB_REM] = FJ_SLAB;
            }
            misc[M_FLUSH] = 0;
        }
        __syncthreads();
        if (tid < F) {
            const u32 b = tid, l = left[b], seg = misc[M_SEG];
            u32 f0 = fill[b], c = cur[b], n = nch[b];
            const u32 outb = parent * F + b;
            if (l > 0 && f0 == FJ_CHUNK) {
                c = misc[M_SLAB_CUR] + atomicAdd(&misc[M_FLUSH], 1u); f0 = 0;
                if (c < cap) a.out_rel[c] = ((u64)seg << 32) | n;
                ++n;
            }
            if (c != FJ_DIR_INVALID && c < cap) {
                const u64 base = (u64)c * FJ_CHUNK + f0;
                for (u32 j = 0; j < l; ++j) {
                    a.out_keys[base + j] = lo_k[b * LINE + j];
                    if (HAS_VALS) a.out_vals[base + j] = lo_v[b * LINE + j];
                }
                a.out_dir[c] = (outb << FJ_DIR_CNT_BITS) | (f0 + l);
            }
            if (n > 0) {
                const u32 off = atomicAdd(&a.bchunks[outb], n);
                if (seg < a.max_segs) a.seg_off[(u64)seg * F + b] = off;
            }
            left[b] = 0; fill[b] = FJ_CHUNK; cur[b] = FJ_DIR_INVALID; nch[b] = 0;
        }
        __syncthreads();
        if (tid == 0) { const u32 n = misc[M_FLUSH]; misc[M_SLAB_CUR] += n; misc[M_SLAB_REM] -= n; }
        __syncthreads();
    };

    // ---- prologue: keys of the first tile into registers, metadata of the second into LDS ----
    u64 kn[KPT], vn[KPT];
    u32 validn = 0, par_next = 0, par_n2 = 0, mid = 0, mcnt = 0;
    {
        u32 p, l;
        get_desc(t, p, l, par_next);
        meta_fetch(p, l, mid, mcnt);
        if (tid < TC) { t_chunk[tid] = mid; t_cnt[tid] = mcnt; }
        __syncthreads();
        key_load(kn, vn, validn);
        mid = 0; mcnt = 0;
        if (t + 1 < thi) { get_desc(t + 1, p, l, par_n2); meta_fetch(p, l, mid, mcnt); }
        __syncthreads();
        if (tid < TC) { t_chunk[tid] = mid; t_cnt[tid] = mcnt; }
        __syncthreads();
    }

    u32 cur_parent = 0xFFFFFFFFu;
    for (; t < thi; ++t) {
        u64 k[KPT], v[KPT];
#pragma unroll
        for (int i = 0; i < KPT; ++i) { k[i] = kn[i]; if (HAS_VALS) v[i] = vn[i]; }
        const u32 valid = validn, parent = par_next;
        // keep HBM busy: tile t+1's keys and tile t+2's chunk metadata are requested now
        if (t + 1 < thi) { key_load(kn, vn, validn); par_next = par_n2; }
        mid = 0; mcnt = 0;
        if (t + 2 < thi) { u32 p, l; get_desc(t + 2, p, l, par_n2); meta_fetch(p, l, mid, mcnt); }

        if (parent != cur_parent) {
            if (cur_parent != 0xFFFFFFFFu) flush(cur_parent);
            if (tid == 0) {
                const u32 seg = atomicAdd(a.seg_counter, 1u);
                if (seg >= a.max_segs) atomicOr(a.err, FJ_ERR_POOL);
                misc[M_SEG] = seg;
            }
            cur_parent = parent;
        }
        for (u32 b = tid; b < F; b += NT) hist[b] = left[b];
        __syncthreads();

        // ---- hash, rank inside the tile with LDS atomics ----------------------------------------
        u32 br[KPT];
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            br[i] = 0;
            if (valid & (1u << i)) {
                const u32 b = (u32)(fj_hash64(k[i]) >> a.shift) & FM;
                br[i] = (b << 16) | atomicAdd(&hist[b], 1u);
            }
        }
        __syncthreads();

        // ---- one packed exclusive scan over the buckets: tile offset | line offset | new chunks
        u32 tot = 0, nf = 0, km = 0;
        u64 x = 0;
        if (tid < F) {
            tot = hist[tid];
            nf = tot & ~(LINE - 1);
            km = nf ? ((fill[tid] + nf - 1) >> FJ_CHUNK_LOG) : 0;
            x = (u64)tot | ((u64)(nf >> LINE_LOG) << 20) | ((u64)km << 40);
        }
        u64 inc = x;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const u64 y = __shfl_up(inc, d, 64);
            if ((int)lane >= d) inc += y;
        }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        u64 woff = 0;
        for (u32 w = 0; w < wave; ++w) woff += wsum[w];
        const u64 exc = inc - x + woff;
        if (tid < F) {
            toff[tid] = (u32)(exc & 0xFFFFFu);
            lineoff[tid] = (u32)((exc >> 20) & 0xFFFFFu);
            kbase[tid] = (u32)(exc >> 40);
            nfull[tid] = nf;
            if (tid == F - 1) {
                const u32 need = (u32)(exc >> 40) + km;
                misc[M_NLINES] = (u32)((exc >> 20) & 0xFFFFFu) + (nf >> LINE_LOG);
                misc[M_NEED] = need;
                if (need > misc[M_SLAB_REM]) {
                    const u32 nb = atomicAdd(a.alloc, FJ_SLAB);
                    if (nb + FJ_SLAB > cap) atomicOr(a.err, FJ_ERR_POOL);
                    misc[M_NEW_BASE] = nb;
                }
            }
        }
        __syncthreads();

        // ---- bucket-sort the tile in LDS; carried remainders go in front of their bucket ------
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            if (valid & (1u << i)) {
                const u32 d = toff[br[i] >> 16] + (br[i] & 0xFFFFu);
                sorted_k[d] = k[i];
                if (HAS_VALS) sorted_v[d] = v[i];
            }
        }
        for (u32 e = tid; e < F * LINE; e += NT) {
            const u32 b = e >> LINE_LOG, j = e & (LINE - 1);
            if (j < left[b]) {
                sorted_k[toff[b] + j] = lo_k[e];
                if (HAS_VALS) sorted_v[toff[b] + j] = lo_v[e];
            }
        }
        u32 new_cur = 0, new_fill = 0, new_left = 0, new_nch = 0;
        if (tid < F) {
            const u32 b = tid, f0 = fill[b], l0 = lineoff[b], s0 = toff[b], kb = kbase[b];
            const u32 c0 = cur[b], n0 = nch[b];
            for (u32 q = 0; q < nf; q += LINE) {
                const u32 pq = f0 + q, kk = pq >> FJ_CHUNK_LOG, off = pq & (FJ_CHUNK - 1);
                const u32 id = kk == 0 ? c0 : alloc_id(kb + kk - 1);
                line_src[l0 + (q >> LINE_LOG)] = s0 + q;
                line_dst[l0 + (q >> LINE_LOG)] = id < cap ? id * FJ_CHUNK + off : FJ_DIR_INVALID;
            }
            const u32 outb = (parent * F + b) << FJ_DIR_CNT_BITS;
            const u64 segw = (u64)misc[M_SEG] << 32;
            for (u32 kk = 1; kk <= km; ++kk) {
                const u32 id = alloc_id(kb + kk - 1);
                if (id < cap) { a.out_dir[id] = outb | FJ_CHUNK; a.out_rel[id] = segw | (n0 + kk - 1); }
            }
            new_cur = km ? alloc_id(kb + km - 1) : c0;
            new_fill = f0 + nf - (km << FJ_CHUNK_LOG);
            new_left = tot & (LINE - 1);
            new_nch = n0 + km;
        }
        __syncthreads();

        // ---- write whole lines: LINE/2 lanes x 16 B per line -------------------------------
        const u32 nl = misc[M_NLINES];
        for (u32 e = tid; e < nl * LPL; e += NT) {
            const u32 l = e / LPL, j2 = (e % LPL) * 2;
            const u32 dst = line_dst[l];
            if (dst != FJ_DIR_INVALID) {
                const u32 src = line_src[l] + j2;
                u64x2 q; q.x = sorted_k[src]; q.y = sorted_k[src + 1];
                *reinterpret_cast<u64x2*>(a.out_keys + (u64)dst + j2) = q;
                if (HAS_VALS) {
                    u64x2 w; w.x = sorted_v[src]; w.y = sorted_v[src + 1];
                    *reinterpret_cast<u64x2*>(a.out_vals + (u64)dst + j2) = w;
                }
            }
        }
        for (u32 e = tid; e < F * LINE; e += NT) {
            const u32 b = e >> LINE_LOG, j = e & (LINE - 1);
            if (j < (hist[b] & (LINE - 1))) {
                const u32 s = toff[b] + nfull[b] + j;
                lo_k[e] = sorted_k[s];
                if (HAS_VALS) lo_v[e] = sorted_v[s];
            }
        }
        if (tid < F) { cur[tid] = new_cur; fill[tid] = new_fill; left[tid] = new_left; nch[tid] = new_nch; }
        if (tid < TC) { t_chunk[tid] = mid; t_cnt[tid] = mcnt; }      // metadata of tile t+2
        if (tid == 0) {
            const u32 need = misc[M_NEED], rem = misc[M_SLAB_REM];
            if (need <= rem) { misc[M_SLAB_CUR] += need; misc[M_SLAB_REM] = rem - need; }
            else { const u32 used = need - rem; misc[M_SLAB_CUR] = misc[M_NEW_BASE] + used; misc[M_SLAB_REM] = FJ_SLAB - used; }
        }
        __syncthreads();
    }
    flush(cur_parent);
}

// single-workgroup exclusive scan of u32 counts -> u32 offsets[n+1]
__global__ __launch_bounds__(1024) void fj_scan_u32(const u32* in, u32* __restrict__ out, u32 n) {
    __shared__ u32 wtot[16];
    __shared__ u32 carry;
    const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (u32 base = 0; base < n; base += 1024) {
        const u32 i = base + tid;
        const u32 x = i < n ? in[i] : 0;
        u32 inc = x;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const u32 y = __shfl_up(inc, d, 64); if ((int)lane >= d) inc += y; }
        if (lane == 63) wtot[wave] = inc;
        __syncthreads();
        u32 woff = carry;
        for (u32 w = 0; w < wave; ++w) woff += wtot[w];
        if (i < n) out[i] = inc - x + woff;
        __syncthreads();
        if (tid == 1023) carry = woff + inc;
        __syncthreads();
    }
    if (tid == 0) out[n] = carry;
}

// number of tiles of `tc` chunks per bucket (tiles never span buckets), exclusive-scanned
__global__ __launch_bounds__(1024) void fj_tile_scan(const u32* __restrict__ boff, u32* __restrict__ toff, u32 n, u32 tc) {
    __shared__ u32 wtot[16];
    __shared__ u32 carry;
    const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (u32 base = 0; base < n; base += 1024) {
        const u32 i = base + tid;
        const u32 x = i < n ? (boff[i + 1] - boff[i] + tc - 1) / tc : 0;
        u32 inc = x;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const u32 y = __shfl_up(inc, d, 64); if ((int)lane >= d) inc += y; }
        if (lane == 63) wtot[wave] = inc;
        __syncthreads();
        u32 woff = carry;
        for (u32 w = 0; w < wave; ++w) woff += wtot[w];
        if (i < n) toff[i] = inc - x + woff;
        __syncthreads();
        if (tid == 1023) carry = woff + inc;
        __syncthreads();
    }
    if (tid == 0) toff[n] = carry;
}

// tile t -> (first list index, chunks, bucket)
__global__ void fj_tile_expand(const u32* __restrict__ boff, const u32* __restrict__ toff, u32 n, u32 tc,
                               uint4* __restrict__ tiles, u32 max_tiles) {
    u32 total = toff[n];
    if (total > max_tiles) total = max_tiles;
    for (u32 t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
        u32 lo = 0, hi = n;                       // last p with toff[p] <= t
        while (hi - lo > 1) { const u32 mid = (lo + hi) >> 1; if (toff[mid] <= t) lo = mid; else hi = mid; }
        const u32 pos = boff[lo] + (t - toff[lo]) * tc;
        const u32 rem = boff[lo + 1] - pos;
        tiles[t] = make_uint4(pos, rem < tc ? rem : tc, lo, 0);
    }
}

// same for u64 outputs (result offsets can exceed 2^32)
__global__ __launch_bounds__(1024) void fj_scan_u32_to_u64(const u32* __restrict__ in, u64* __restrict__ out, u32 n) {
    __shared__ u64 wtot[16];
    __shared__ u64 carry;
    const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (u32 base = 0; base < n; base += 1024) {
        const u32 i = base + tid;
        const u64 x = i < n ? in[i] : 0;
        u64 inc = x;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const u64 y = __shfl_up(inc, d, 64); if ((int)lane >= d) inc += y; }
        if (lane == 63) wtot[wave] = inc;
        __syncthreads();
        u64 woff = carry;
        for (u32 w = 0; w < wave; ++w) woff += wtot[w];
        if (i < n) out[i] = inc - x + woff;
        __syncthreads();
        if (tid == 1023) carry = woff + inc;
        __syncthreads();
    }
    if (tid == 0) out[n] = carry;
}

// chunk lists without atomics: list[boff[bucket] + span offset of the producing segment + rank] = chunk
__global__ void fj_list_build(const u32* __restrict__ dir, const u64* __restrict__ rel, const u32* __restrict__ nalloc,
                              u32 cap, const u32* __restrict__ boff, const u32* __restrict__ seg_off, u32 fan_mask,
                              u32 max_segs, u32* __restrict__ list) {
    u32 n = *nalloc; if (n > cap) n = cap;
    for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const u32 e = dir[i];
        if (e != FJ_DIR_INVALID) {
            const u32 b = e >> FJ_DIR_CNT_BITS;
            const u64 r = rel[i];
            const u32 seg = (u32)(r >> 32);
            if (seg < max_segs) list[boff[b] + seg_off[(u64)seg * (fan_mask + 1) + (b & fan_mask)] + (u32)r] = i;
        }
    }
}

template <int NT, int KPT, int LINE_LOG, bool HAS_VALS>
hipError_t launch_part(const FjPartArgs& a, u32 grid, hipStream_t s) {
    const u32 F = 1u << a.fan_log;
    const PartLds L = part_lds_layout(NT * KPT, F, 1u << LINE_LOG, HAS_VALS, NT / 64);
    auto kern = fj_partition_kernel<NT, KPT, LINE_LOG, HAS_VALS>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)L.total);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), L.total, s, a);
    return hipGetLastError();
}

}  // namespace

u32 fj_partition_lds_bytes(u32 fan_log, bool vals, int line_log) {
    const u32 F = 1u << fan_log;
    if (vals) return part_lds_layout(512 * 4, F, 1u << line_log, true, 8).total;
    return part_lds_layout(512 * 8, F, 1u << line_log, false, 8).total;
}

// One partition pass.  Keys-only tiles are 4096 keys (512 threads x 8), key+value tiles 2048.
hipError_t fj_launch_partition(const FjPartArgs& a, bool vals, int line_log, u32 grid, hipStream_t s) {
    if (vals) {
        if (line_log == 3) return launch_part<512, 4, 3, true>(a, grid, s);
        return launch_part<512, 4, 4, true>(a, grid, s);
    }
    if (line_log == 3) return launch_part<512, 8, 3, false>(a, grid, s);
    return launch_part<512, 8, 4, false>(a, grid, s);
}

// After a pass: per-bucket chunk counts -> offsets -> chunk lists (no atomics).
hipError_t fj_launch_group(const FjChunkSet& cs, hipStream_t s) {
    hipLaunchKernelGGL(fj_scan_u32, dim3(1), dim3(1024), 0, s, cs.bchunks, cs.boff, cs.nb);
    hipLaunchKernelGGL(fj_list_build, dim3(512), dim3(256), 0, s, cs.dir, cs.rel, cs.alloc, cs.cap, cs.boff, cs.seg_off,
                       cs.fan_mask, cs.max_segs, cs.list);
    return hipGetLastError();
}

// Tile table of a chunk set for a consumer pass with `tc` chunks per tile.
hipError_t fj_launch_tile_table(const FjChunkSet& cs, u32 tc, u32* toff, uint4* tiles, u32 max_tiles, hipStream_t s) {
    hipLaunchKernelGGL(fj_tile_scan, dim3(1), dim3(1024), 0, s, cs.boff, toff, cs.nb, tc);
    hipLaunchKernelGGL(fj_tile_expand, dim3(256), dim3(256), 0, s, cs.boff, toff, cs.nb, tc, tiles, max_tiles);
    return hipGetLastError();
}

hipError_t fj_launch_scan_u32_to_u64(const u32* in, u64* out, u32 n, hipStream_t s) {
    hipLaunchKernelGGL(fj_scan_u32_to_u64, dim3(1), dim3(1024), 0, s, in, out, n);
    return hipGetLastError();
}
