// fj_partition.hip -- radix partition pass for MI355X (gfx950), keys (+values).
//
// Mirrors the *function* of parallel_radix_partition_kv / _k (hash_join.cpp:209-292) but not
// its structure.  The reference does histogram -> prefix -> scatter with 8-byte random
// stores.  Here one pass is a single kernel with NO histogram read:
//
//   * persistent workgroups each take a contiguous slice of the input;
//   * a tile of keys is bucket-sorted inside LDS (LDS atomics give the rank);
//   * per bucket only WHOLE lines (LINE keys, 64 or 128 B, line-aligned in HBM) are written,
//     the < LINE remainder is carried in LDS to the next tile  (software write-combining);
//   * lines go into block-private 2-KiB *chunks* handed out by a slab allocator, so no two
//     workgroups ever share a line and no global cursor is contended;
//   * a directory word per chunk (bucket | count) is grouped by bucket afterwards
//     (fj_group_* kernels) into per-bucket chunk lists, which is what the next pass / the
//     join kernel consume.
//
// Algorithmic HBM bytes per key and pass: 8 read + 8 written (keys only), 16 + 16 with values.
#include "fj_internal.h"

namespace {

// dynamic-LDS layout, shared by kernel and host-side size computation
struct PartLds {
    u32 sorted_k, sorted_v, lo_k, lo_v, hist, left, fill, cur, toff, kbase, lineoff, nfull;
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
    L.line_src = o; o += maxl * 4;
    L.line_dst = o; o += maxl * 4;
    L.t_chunk = o; o += (T / FJ_CHUNK) * 4;
    L.t_cnt = o; o += (T / FJ_CHUNK) * 4;
    L.misc = o; o += 16 * 4;
    L.total = (o + 15) & ~15u;
    return L;
}

enum { M_SLAB_CUR = 0, M_SLAB_REM, M_NEW_BASE, M_NEED, M_TLEN, M_NLINES, M_FLUSH };

template <int NT, int KPT, int LINE_LOG, bool HAS_VALS>
__global__ __launch_bounds__(NT) void fj_partition_kernel(FjPartArgs a) {
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
    u32* line_src = (u32*)(smem + Lo.line_src);
    u32* line_dst = (u32*)(smem + Lo.line_dst);
    u32* t_chunk = (u32*)(smem + Lo.t_chunk);
    u32* t_cnt = (u32*)(smem + Lo.t_cnt);
    u32* misc = (u32*)(smem + Lo.misc);

    const bool flat = (a.in_list == nullptr);
    const u32 Lc = flat ? (u32)((a.n_flat + FJ_CHUNK - 1) >> FJ_CHUNK_LOG) : *a.in_nlist;
    const u32 G = gridDim.x, g = blockIdx.x;
    u32 pos = (u32)(((u64)g * Lc) / G);
    const u32 hi = (u32)(((u64)(g + 1) * Lc) / G);
    const u32 cap = a.cap_chunks;

    for (u32 b = tid; b < F; b += NT) { left[b] = 0; fill[b] = FJ_CHUNK; cur[b] = FJ_DIR_INVALID; }
    if (tid == 0) { misc[M_SLAB_CUR] = 0; misc[M_SLAB_REM] = 0; misc[M_NEW_BASE] = 0; }
    u32 cur_parent = 0xFFFFFFFFu;
    __syncthreads();

    // id of the j-th chunk this workgroup allocates in the current tile
    auto alloc_id = [&](u32 j) -> u32 {
        u32 rem = misc[M_SLAB_REM];
        return j < rem ? misc[M_SLAB_CUR] + j : misc[M_NEW_BASE] + (j - rem);
    };

    // write out the carried remainders of the segment that just ended and reset the state
    auto flush = [&](u32 parent) {
        if (tid == 0) {
            if (misc[M_SLAB_REM] < F) {
                u32 nb = atomicAdd(a.alloc, FJ_SLAB);
                if (nb + FJ_SLAB > cap) atomicOr(a.err, FJ_ERR_POOL);
                misc[M_SLAB_CUR] = nb; misc[M_SLAB_REM] = FJ_SLAB;
            }
            misc[M_FLUSH] = 0;
        }
        __syncthreads();
        if (tid < F) {
            const u32 b = tid, l = left[b];
            u32 f0 = fill[b], c = cur[b];
            if (l > 0 && f0 == FJ_CHUNK) { c = misc[M_SLAB_CUR] + atomicAdd(&misc[M_FLUSH], 1u); f0 = 0; }
            if (c != FJ_DIR_INVALID && c < cap) {
                const u64 base = (u64)c * FJ_CHUNK + f0;
                for (u32 j = 0; j < l; ++j) {
                    a.out_keys[base + j] = lo_k[b * LINE + j];
                    if (HAS_VALS) a.out_vals[base + j] = lo_v[b * LINE + j];
                }
                a.out_dir[c] = ((parent * F + b) << FJ_DIR_CNT_BITS) | (f0 + l);
            }
            left[b] = 0; fill[b] = FJ_CHUNK; cur[b] = FJ_DIR_INVALID;
        }
        __syncthreads();
        if (tid == 0) { u32 n = misc[M_FLUSH]; misc[M_SLAB_CUR] += n; misc[M_SLAB_REM] -= n; }
        __syncthreads();
    };

    while (pos < hi) {
        // ---- tile = up to TC consecutive chunks of one parent bucket ------------------------
        if (tid == 0) misc[M_TLEN] = (hi - pos) < TC ? (hi - pos) : TC;
        __syncthreads();
        u32 parent = a.parent0;
        if (!flat) parent = a.in_dir[a.in_list[pos]] >> FJ_DIR_CNT_BITS;
        if (tid < TC && pos + tid < hi) {
            const u32 c = pos + tid;
            u32 id, cnt;
            if (flat) {
                id = c;
                const u64 rem = a.n_flat - (u64)c * FJ_CHUNK;
                cnt = rem >= FJ_CHUNK ? FJ_CHUNK : (u32)rem;
            } else {
                id = a.in_list[c];
                const u32 e = a.in_dir[id];
                cnt = e & FJ_DIR_CNT_MASK;
                if ((e >> FJ_DIR_CNT_BITS) != parent) atomicMin(&misc[M_TLEN], tid);
            }
            t_chunk[tid] = id; t_cnt[tid] = cnt;
        }
        if (parent != cur_parent) {
            if (cur_parent != 0xFFFFFFFFu) flush(cur_parent);
            cur_parent = parent;
        }
        for (u32 b = tid; b < F; b += NT) hist[b] = left[b];
        __syncthreads();
        const u32 tc = misc[M_TLEN];

        // ---- load (16 B per lane), hash, rank inside the tile with LDS atomics --------------
        u64 k[KPT], v[KPT];
        u32 br[KPT];
        u32 valid = 0;
#pragma unroll
        for (int i = 0; i < KPT / 2; ++i) {
            const u32 kidx = ((u32)i * NT + tid) * 2;
            const u32 j = kidx >> FJ_CHUNK_LOG, off = kidx & (FJ_CHUNK - 1);
            u32 cnt = 0;
            if (j < tc) cnt = t_cnt[j];
            k[2 * i] = 0; k[2 * i + 1] = 0;
            if (HAS_VALS) { v[2 * i] = 0; v[2 * i + 1] = 0; }
            if (off < cnt) {
                const u64 base = (u64)t_chunk[j] * FJ_CHUNK + off;
                if (off + 1 < cnt) {
                    const u64x2 kk = *reinterpret_cast<const u64x2*>(a.in_keys + base);
                    k[2 * i] = kk.x; k[2 * i + 1] = kk.y;
                    if (HAS_VALS) {
                        const u64x2 vv = *reinterpret_cast<const u64x2*>(a.in_vals + base);
                        v[2 * i] = vv.x; v[2 * i + 1] = vv.y;
                    }
                    valid |= 3u << (2 * i);
                } else {
                    k[2 * i] = a.in_keys[base];
                    if (HAS_VALS) v[2 * i] = a.in_vals[base];
                    valid |= 1u << (2 * i);
                }
            }
        }
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
        u32 new_cur = 0, new_fill = 0, new_left = 0;
        if (tid < F) {
            const u32 b = tid, f0 = fill[b], l0 = lineoff[b], s0 = toff[b], kb = kbase[b];
            const u32 c0 = cur[b];
            for (u32 q = 0; q < nf; q += LINE) {
                const u32 pq = f0 + q, kk = pq >> FJ_CHUNK_LOG, off = pq & (FJ_CHUNK - 1);
                const u32 id = kk == 0 ? c0 : alloc_id(kb + kk - 1);
                line_src[l0 + (q >> LINE_LOG)] = s0 + q;
                line_dst[l0 + (q >> LINE_LOG)] = id < cap ? id * FJ_CHUNK + off : FJ_DIR_INVALID;
            }
            const u32 outb = (parent * F + b) << FJ_DIR_CNT_BITS;
            for (u32 kk = 1; kk <= km; ++kk) {
                const u32 id = alloc_id(kb + kk - 1);
                if (id < cap) a.out_dir[id] = outb | FJ_CHUNK;
            }
            new_cur = km ? alloc_id(kb + km - 1) : c0;
            new_fill = f0 + nf - (km << FJ_CHUNK_LOG);
            new_left = tot & (LINE - 1);
        }
        __syncthreads();

        // ---- write whole lines: LINE/2 lanes x 16 B per line -------------------------------
        const u32 nl = misc[M_NLINES];
        for (u32 e = tid; e < nl * LPL; e += NT) {
            const u32 l = e / LPL, j2 = (e % LPL) * 2;
            const u32 dst = line_dst[l];
            if (dst != FJ_DIR_INVALID) {
                const u32 src = line_src[l] + j2;
                u64x2 kk; kk.x = sorted_k[src]; kk.y = sorted_k[src + 1];
                *reinterpret_cast<u64x2*>(a.out_keys + (u64)dst + j2) = kk;
                if (HAS_VALS) {
                    u64x2 vv; vv.x = sorted_v[src]; vv.y = sorted_v[src + 1];
                    *reinterpret_cast<u64x2*>(a.out_vals + (u64)dst + j2) = vv;
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
        if (tid < F) { cur[tid] = new_cur; fill[tid] = new_fill; left[tid] = new_left; }
        if (tid == 0) {
            const u32 need = misc[M_NEED], rem = misc[M_SLAB_REM];
            if (need <= rem) { misc[M_SLAB_CUR] += need; misc[M_SLAB_REM] = rem - need; }
            else { const u32 used = need - rem; misc[M_SLAB_CUR] = misc[M_NEW_BASE] + used; misc[M_SLAB_REM] = FJ_SLAB - used; }
        }
        pos += tc;
        __syncthreads();
    }
    if (cur_parent != 0xFFFFFFFFu) flush(cur_parent);
}

// ---- directory grouping: (bucket|count) words -> per-bucket chunk lists ----------------------
__global__ void fj_group_count(const u32* __restrict__ dir, const u32* __restrict__ nalloc, u32 cap,
                               u32* __restrict__ bchunks, u64* __restrict__ bkeys) {
    u32 n = *nalloc; if (n > cap) n = cap;
    for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const u32 e = dir[i];
        if (e != FJ_DIR_INVALID) {
            atomicAdd(&bchunks[e >> FJ_DIR_CNT_BITS], 1u);
            atomicAdd((unsigned long long*)&bkeys[e >> FJ_DIR_CNT_BITS], (unsigned long long)(e & FJ_DIR_CNT_MASK));
        }
    }
}

// single-workgroup exclusive scan of u32 counts -> u32 offsets[n+1]; also zeroes a cursor array
__global__ __launch_bounds__(1024) void fj_scan_u32(const u32* in, u32* __restrict__ out, u32 n, u32* zero_me) {
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
        if (i < n) { out[i] = inc - x + woff; if (zero_me) zero_me[i] = 0; }
        __syncthreads();
        if (tid == 1023) carry = woff + inc;
        __syncthreads();
    }
    if (tid == 0) out[n] = carry;
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

__global__ void fj_group_scatter(const u32* __restrict__ dir, const u32* __restrict__ nalloc, u32 cap,
                                 const u32* __restrict__ boff, u32* __restrict__ bcur, u32* __restrict__ list) {
    u32 n = *nalloc; if (n > cap) n = cap;
    for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const u32 e = dir[i];
        if (e != FJ_DIR_INVALID) {
            const u32 b = e >> FJ_DIR_CNT_BITS;
            list[boff[b] + atomicAdd(&bcur[b], 1u)] = i;
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

// Group a pass's chunk directory by bucket.  cs.boff/bkeys/list are filled; cs.bchunks is scratch.
hipError_t fj_launch_group(const FjChunkSet& cs, hipStream_t s) {
    hipError_t e;
    if ((e = hipMemsetAsync(cs.bchunks, 0, sizeof(u32) * cs.nb, s)) != hipSuccess) return e;
    if ((e = hipMemsetAsync(cs.bkeys, 0, sizeof(u64) * cs.nb, s)) != hipSuccess) return e;
    const u32 blocks = 512;
    hipLaunchKernelGGL(fj_group_count, dim3(blocks), dim3(256), 0, s, cs.dir, cs.alloc, cs.cap, cs.bchunks, cs.bkeys);
    hipLaunchKernelGGL(fj_scan_u32, dim3(1), dim3(1024), 0, s, cs.bchunks, cs.boff, cs.nb, cs.bchunks);
    hipLaunchKernelGGL(fj_group_scatter, dim3(blocks), dim3(256), 0, s, cs.dir, cs.alloc, cs.cap, cs.boff, cs.bchunks, cs.list);
    return hipGetLastError();
}

hipError_t fj_launch_scan_u32_to_u64(const u32* in, u64* out, u32 n, hipStream_t s) {
    hipLaunchKernelGGL(fj_scan_u32_to_u64, dim3(1), dim3(1024), 0, s, in, out, n);
    return hipGetLastError();
}
