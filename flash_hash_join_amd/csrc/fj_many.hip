// fj_many.hip -- many-to-many equi-join on the partitioned plan (an EXTENSION: SURVEY 8(f) rank 4).
//
// The reference deduplicates build keys at insert (hash_join.cpp:125, :147) and stops a probe at the first match
// (:172-176): an N:1 join.  inner_join / inner_join_count keep every build row: a probe row yields one output pair per
// build row with its key (SQL inner join semantics), count = sum over probe rows of the key's multiplicity.
//
// Same partitioning and work items as the other joins; per final partition (<= 4096 build ROWS, the plan aims at 2048)
// one 1024-thread workgroup keeps in LDS
//   tkeys[8192]  distinct keys, linear probing, slot claimed by a 64-bit compare-and-swap (find-or-insert is exact:
//                no racing copies of one key),
//   head[8192]   first build row of the key's chain,   rnext[4096]  next row of the chain,   rvals[4096]  the rows' values;
// a probe walks its key's chain: counting adds the chain's length, materialising writes (probe key, value) per link at a
// position from a wave-wide exclusive scan + one LDS cursor bump per wave.  Two passes like the other materialising
// joins (count per item -> scan -> emit at exact offsets).  Written for correctness and reasonable speed, not tuned like
// the N:1 kernels (64-bit LDS CAS inserts, one key per lane and step).
#include "fj_internal.h"

namespace {

constexpr u32 MM_S = 8192, MM_ROWS = 4096, MM_NT = 1024, MM_NONE = 0xFFFFFFFFu;
struct MmHdr { u32 nrows, full, empty_head, cursor; unsigned long long cnt; u64 pad; };

__device__ __forceinline__ u32 mm_entry(const FjChunkSet& cs, u32 idx) {       // ((count-1) << 24) | chunk id; flat arrays as virtual chunks
    if (cs.list) return cs.list[idx];
    const u64 rem = cs.n_flat - (u64)idx * FJ_CHUNK;
    const u32 cnt = rem >= FJ_CHUNK ? FJ_CHUNK : (u32)rem;
    return ((cnt - 1u) << 24) | idx;
}

template <bool MAT>
__global__ __launch_bounds__(MM_NT, 1) void fj_mm_join_kernel(FjLdsJoinArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    MmHdr* hdr = reinterpret_cast<MmHdr*>(smem);
    u64* tkeys = reinterpret_cast<u64*>(smem + sizeof(MmHdr));
    u64* rvals = tkeys + MM_S;
    u32* head = reinterpret_cast<u32*>(rvals + MM_ROWS);
    u32* rnext = head + MM_S;
    const u32 tid = threadIdx.x, lane = tid & 63;
    const u32 item = blockIdx.x;
    u32 p, b0 = 0, nbc, s_lo, s_hi;
    if (a.items) {
        if (item >= *a.nitems_dev) return;
        const uint4 it = a.items[item];
        p = it.z; s_lo = it.x; s_hi = it.x + it.y;
    } else {
        const u32 slice = item % a.nsplit;
        p = item / a.nsplit;
        const u32 npc = (u32)((a.probe.n_flat + FJ_CHUNK - 1) >> FJ_CHUNK_LOG);
        s_lo = (u32)(((u64)slice * npc) / a.nsplit); s_hi = (u32)(((u64)(slice + 1) * npc) / a.nsplit);
    }
    if (a.build.list) { b0 = a.build.boff[p]; nbc = a.build.boff[p + 1] - b0; }
    else nbc = (u32)((a.build.n_flat + FJ_CHUNK - 1) >> FJ_CHUNK_LOG);
    if (nbc == 0 || s_lo >= s_hi) { if (!MAT && tid == 0) a.part_count[item] = 0; return; }
    if (MAT && a.part_count[item] == 0) return;

    for (u32 i = tid; i < MM_S; i += MM_NT) { tkeys[i] = FJ_EMPTY_KEY; head[i] = MM_NONE; }
    if (tid == 0) { hdr->nrows = 0; hdr->full = 0; hdr->empty_head = MM_NONE; hdr->cursor = 0; hdr->cnt = 0; }
    __syncthreads();

    // ---- build: every row is kept; a key's rows form a chain ----
    for (u32 c0 = 0; c0 < nbc; c0 += MM_NT / FJ_CHUNK) {
        const u32 c = c0 + tid / FJ_CHUNK, off = tid % FJ_CHUNK;
        if (c < nbc) {
            const u32 e = mm_entry(a.build, b0 + c);
            if (off < FJ_LIST_CNT(e)) {
                const u64 src = (u64)FJ_LIST_ID(e) * FJ_CHUNK + off;
                const u64 key = a.build.list ? a.build.keys[src] : fj_key_mix(a.build.keys[src]);     // chunk pools hold mixed keys, flat arrays raw ones (fj_common.h)
                const u32 r = atomicAdd(&hdr->nrows, 1u);
                if (r >= MM_ROWS) hdr->full = 1;
                else {
                    if (MAT) rvals[r] = a.build.vals[src];
                    u32* h;
                    if (key == FJ_EMPTY_KEY) h = &hdr->empty_head;        // the empty marker is never stored in the table
                    else {
                        u32 pos = FJ_HW2(key) & (MM_S - 1);
                        for (;;) {                                          // <= 4096 distinct keys in 8192 slots: always terminates
                            const u64 old = atomicCAS((unsigned long long*)&tkeys[pos], (unsigned long long)FJ_EMPTY_KEY, (unsigned long long)key);
                            if (old == FJ_EMPTY_KEY || old == key) break;
                            pos = (pos + 1) & (MM_S - 1);
                        }
                        h = &head[pos];
                    }
                    rnext[r] = atomicExch(h, r);
                }
            }
        }
    }
    __syncthreads();
    if (hdr->full) {                                   // more rows than the LDS tables hold: the host reports it (no fallback for this extension)
        if (tid == 0) { atomicOr(a.err, FJ_ERR_LDS_FULL); if (!MAT) a.part_count[item] = 0; }
        return;
    }

    // ---- probe: one key per lane and step ----
    const u64 obase = MAT ? a.out_off[item] : 0;
    unsigned long long local = 0;
    for (u32 pc = s_lo; pc < s_hi; pc += MM_NT / FJ_CHUNK) {
        const u32 c = pc + tid / FJ_CHUNK, off = tid % FJ_CHUNK;
        u64 key = 0; bool ok = false;
        if (c < s_hi) {
            const u32 e = mm_entry(a.probe, c);
            if (off < FJ_LIST_CNT(e)) { key = a.probe.keys[(u64)FJ_LIST_ID(e) * FJ_CHUNK + off]; if (!a.probe.list) key = fj_key_mix(key); ok = true; }
        }
        u32 h = MM_NONE;
        if (ok) {
            if (key == FJ_EMPTY_KEY) h = hdr->empty_head;
            else {
                u32 pos = FJ_HW2(key) & (MM_S - 1);
                for (;;) {
                    const u64 t = tkeys[pos];
                    if (t == key) { h = head[pos]; break; }
                    if (t == FJ_EMPTY_KEY) break;
                    pos = (pos + 1) & (MM_S - 1);
                }
            }
        }
        u32 cnt = 0;
        for (u32 r = h; r != MM_NONE; r = rnext[r]) ++cnt;
        if (!MAT) { local += cnt; continue; }
        // exclusive scan of cnt over the wave, one LDS cursor bump per wave
        u32 inc = cnt;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const u32 y = __shfl_up(inc, d, 64); if ((int)lane >= d) inc += y; }
        const u32 wave_total = __shfl(inc, 63, 64);
        u32 wb = 0;
        if (wave_total) {
            if (lane == 63) wb = atomicAdd(&hdr->cursor, wave_total);
            wb = __shfl(wb, 63, 64);
            u64 o = obase + wb + (inc - cnt);
            const u64 raw = fj_key_unmix(key);
            for (u32 r = h; r != MM_NONE; r = rnext[r]) { a.out_keys[o] = raw; a.out_vals[o] = rvals[r]; ++o; }
        }
    }
    if (!MAT) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) local += __shfl_xor(local, d, 64);
        if (lane == 0 && local) atomicAdd(&hdr->cnt, local);
        __syncthreads();
        if (tid == 0) {
            const unsigned long long n = hdr->cnt;
            if (n > 0xFFFFFFFFull) atomicOr(a.err, FJ_ERR_POOL);       // (cannot happen: <= 131072 probe rows x 4096 build rows per item)
            a.part_count[item] = (u32)n;
            if (n) atomicAdd(a.total, n);
        }
    }
}

}  // namespace

hipError_t fj_launch_mm_join(const FjLdsJoinArgs& a, bool materialize, hipStream_t s) {
    const u32 nb = a.items ? a.items_cap : a.nparts * a.nsplit;
    const u32 lds = sizeof(MmHdr) + MM_S * 8 + MM_ROWS * 8 + MM_S * 4 + MM_ROWS * 4;
    auto kern = materialize ? fj_mm_join_kernel<true> : fj_mm_join_kernel<false>;
    hipError_t e = fj_set_max_lds_once(reinterpret_cast<const void*>(kern), lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(nb), dim3(MM_NT), lds, s, a);
    return hipGetLastError();
}
