// fj_bcast.hip -- the multi-GPU counting join in its BUILD-BROADCAST form: what one rank does to its own rows.
//
// No reference counterpart: the reference is one process (hash_join.cpp:318).  What is exploited is that radix partitions are
// independent join units (hash_join.cpp:340-356, :515-525) and that a final partition's build keys share their top radix bits.
// Every rank plans for the TOTAL build side (the same 2^bits final partitions everywhere), runs BOTH passes of that plan over
// its own build rows and writes them out densely, partition after partition, in a wire format that drops the bits the
// partition implies (fj_bcast_pack):
//     region = offset table u32[nparts + 1] (keys before partition p) | low words u32[n] | the low (32 - bits) bits of the high
//              words as u16[n] (bits >= 16) or u32[n]
// - 6 bytes per key for build sides of 134M rows and more, exact (no chunk padding).  The regions of all ranks are exchanged
// (every rank sends its region to every peer: csrc/fj_dist.hip, in pieces of consecutive partitions), the probe rows NEVER
// move: every rank partitions its own probe rows with the same plan (fj_bcast_probe) and joins them against the runs of all
// ranks, partition range by partition range as the ranges land (fj_bcast_join -> fj_count_join_wide<DENSE>, which reads the
// runs where they lie: the receiver neither re-partitions nor copies anything).
// Against the owner shuffle (7 bytes per probe AND build key across the links, a pack copy and a pass over wire chunks on the
// receiving side) a rank puts 6 * nb_local bytes on every link however many probe rows there are: at BASELINE configs[4]
// (125M x 1.25B rows per GPU, 8 GPUs) 0.75 GB per link instead of 1.2 GB, and no kernel beyond the plain join's but the pack.
#include "fj_host.h"
#include <type_traits>
using namespace fjh;

namespace {

constexpr u32 DP_NT = 256;

// keys per final partition, from the final level's chunk lists (entry = (count - 1) << 24 | id)
__global__ __launch_bounds__(DP_NT) void fj_dense_count(const u32* __restrict__ boff, const u32* __restrict__ list, u32 nparts, u32* __restrict__ cnt) {
    const u32 p = blockIdx.x * DP_NT + threadIdx.x;
    if (p >= nparts) return;
    u32 n = 0;
    for (u32 i = boff[p]; i < boff[p + 1]; ++i) n += FJ_LIST_CNT(list[i]);
    cnt[p] = n;
}

// in-place exclusive scan of cnt[0 .. nparts) (one 1024-thread workgroup; nparts <= 2^22), cnt[nparts] = total; the key index at
// which piece q of `pieces` pieces of consecutive partitions starts goes to bounds[q] (bounds[pieces] = total).  16384 entries per
// sweep (four 16-byte loads per thread in flight: 262144 entries in 16 sweeps; 4096 per sweep took 150 us)
__global__ __launch_bounds__(1024) void fj_dense_scan(u32* __restrict__ cnt, u32 nparts, u32 pieces, u32* __restrict__ bounds) {
    __shared__ u32 wsum[4][16];
    const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    u32 carry = 0;
    for (u32 base = 0; base < nparts; base += 16384) {
        uint4 x[4]; u32 sum[4], inc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const u32 e0 = base + 4u * ((u32)j * 1024u + tid);
            x[j] = make_uint4(0, 0, 0, 0);
            if (e0 + 3 < nparts) x[j] = *reinterpret_cast<const uint4*>(cnt + e0);
            else { if (e0 < nparts) x[j].x = cnt[e0]; if (e0 + 1 < nparts) x[j].y = cnt[e0 + 1]; if (e0 + 2 < nparts) x[j].z = cnt[e0 + 2]; }
            sum[j] = x[j].x + x[j].y + x[j].z + x[j].w;
            u32 v = sum[j];
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const u32 y = __shfl_up(v, d, 64); if ((int)lane >= d) v += y; }
            inc[j] = v;
            if (lane == 63) wsum[j][wave] = v;
        }
        __syncthreads();
        u32 total = carry;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            u32 mine = 0, all = 0;
            for (u32 v = 0; v < 16; ++v) { const u32 c = wsum[j][v]; all += c; if (v < wave) mine += c; }
            u32 run = total + mine + inc[j] - sum[j];
            const u32 e0 = base + 4u * ((u32)j * 1024u + tid);
            if (e0 < nparts) cnt[e0] = run; run += x[j].x;
            if (e0 + 1 < nparts) cnt[e0 + 1] = run; run += x[j].y;
            if (e0 + 2 < nparts) cnt[e0 + 2] = run; run += x[j].z;
            if (e0 + 3 < nparts) cnt[e0 + 3] = run;
            total += all;
        }
        carry = total;
        __syncthreads();
    }
    if (tid == 0) cnt[nparts] = carry;
    __threadfence_block();
    __syncthreads();
    if (tid <= pieces) bounds[tid] = tid == pieces ? carry : cnt[(u32)(((u64)nparts * tid) / pieces)];
}

// the same scan for 8192 .. 2^19 partitions (a power of two), out of place, one workgroup per 4096 counts: each sums the counts in
// front of its piece itself (<= 2 MiB, in L2) and scans its piece - 262144 partitions in ~10 us instead of 150 on one CU
__global__ __launch_bounds__(1024) void fj_dense_scan_wide(const u32* __restrict__ cnt, u32* __restrict__ offs, u32 nparts, u32 pieces, u32* __restrict__ bounds) {
    __shared__ u32 wtot[2][16];
    const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = blockIdx.x;
    const u32 e0 = g * 4096u + 4u * tid;
    const uint4 x = *reinterpret_cast<const uint4*>(cnt + e0);
    u32 pre = 0;
#pragma unroll 4
    for (u32 j = 0; j < g; ++j) { const uint4 y = *reinterpret_cast<const uint4*>(cnt + 4u * (j * 1024u + tid)); pre += y.x + y.y + y.z + y.w; }
    const u32 own = x.x + x.y + x.z + x.w;
    u32 inc = own;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const u32 y = __shfl_up(inc, d, 64), z = __shfl_up(pre, d, 64);
        if ((int)lane >= d) { inc += y; pre += z; }
    }
    if (lane == 63) { wtot[0][wave] = inc; wtot[1][wave] = pre; }
    __syncthreads();
    u32 run = inc - own;
    for (u32 w = 0; w < 16; ++w) { run += wtot[1][w]; if (w < wave) run += wtot[0][w]; }
    const uint4 o = make_uint4(run, run + x.x, run + x.x + x.y, run + x.x + x.y + x.z);
    *reinterpret_cast<uint4*>(offs + e0) = o;
    const u32 ov[4] = {o.x, o.y, o.z, o.w};
    for (u32 q = 0; q < pieces; ++q) {
        const u32 idx = (u32)(((u64)nparts * q) / pieces);
        if (idx - e0 < 4u) bounds[q] = ov[idx - e0];
    }
    if (e0 + 4u == nparts) { offs[nparts] = run + own; bounds[pieces] = run + own; }
}

// one wave per final partition: its chunks' keys -> the two planes at the partition's offset.  The keys are staged in a per-wave
// LDS tile whose index mirrors the alignment of the output (tile_base is a multiple of 8 keys) and leave as whole 16-byte words
// wherever a word belongs to this partition alone; the partition's first and last few keys - words shared with the neighbouring
// partitions, other waves' business - go out one by one.  (One 4-byte and one 2-byte store per KEY made this copy 0.41 ms per 125M
// keys: the memory pipeline charges per store instruction.)  LDS operations of one wave execute in order: no barrier.
constexpr u32 DC_T = 512;                      // keys per tile (a multiple of 8)
template <int MIDB>
__global__ __launch_bounds__(DP_NT) void fj_dense_copy(const u64* __restrict__ keys, const u32* __restrict__ boff, const u32* __restrict__ list,
                                                      u32 nparts, const u32* __restrict__ offs, u32 midmask, u32* __restrict__ lo, void* __restrict__ mid) {
    typedef typename std::conditional<MIDB == 2, u16, u32>::type mid_t;
    __shared__ __attribute__((aligned(16))) u32 s_lo[DP_NT / 64][DC_T];
    __shared__ __attribute__((aligned(16))) mid_t s_mid[DP_NT / 64][DC_T];
    const u32 wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const u32 p = blockIdx.x * (DP_NT / 64) + wv;
    if (p >= nparts) return;
    u32* tl = s_lo[wv]; mid_t* tm = s_mid[wv];
    mid_t* gm = reinterpret_cast<mid_t*>(mid);
    const u32 o = offs[p], n = offs[p + 1] - o, end = o + n;
    if (n == 0) return;
    constexpr u32 MG = 16 / sizeof(mid_t);                           // high-word elements per 16-byte word
    // the tile [tile_base, tile_base + DC_T) of the output leaves the LDS stage: whole words inside [o, end), single elements at its edges
    auto flush = [&](u32 tile_base) {
        const u32 vlo = o > tile_base ? o : tile_base, vhi = end < tile_base + DC_T ? end : tile_base + DC_T;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        for (u32 g = (vlo - tile_base) / 4 + lane; g * 4 < vhi - tile_base; g += 64) {
            const u32 g0 = tile_base + 4 * g;
            const uint4 q = *reinterpret_cast<const uint4*>(tl + 4 * g);
            if (g0 >= vlo && g0 + 4 <= vhi) *reinterpret_cast<uint4*>(lo + g0) = q;
            else {
                const u32 w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                for (u32 e = 0; e < 4; ++e) if (g0 + e >= vlo && g0 + e < vhi) lo[g0 + e] = w[e];
            }
        }
        for (u32 g = (vlo - tile_base) / MG + lane; g * MG < vhi - tile_base; g += 64) {
            const u32 g0 = tile_base + MG * g;
            const uint4 q = *reinterpret_cast<const uint4*>(tm + MG * g);
            if (g0 >= vlo && g0 + MG <= vhi) *reinterpret_cast<uint4*>(gm + g0) = q;
            else {
#pragma unroll
                for (u32 e = 0; e < MG; ++e) if (g0 + e >= vlo && g0 + e < vhi) gm[g0 + e] = tm[MG * g + e];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    };
    u32 tile_base = o & ~7u, pos = o;                                // pos: output index of the next chunk's first key
    for (u32 i = boff[p]; i < boff[p + 1]; ++i) {
        const u32 e = list[i], cnt = FJ_LIST_CNT(e);
        const u64* ck = keys + (u64)FJ_LIST_ID(e) * FJ_CHUNK;
        u64x2 q[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {                                // (only the lines that hold keys: the chunks of a final partition are ~2/3 full)
            q[t].x = 0; q[t].y = 0;
            if (2 * lane + 128 * t < cnt) q[t] = *reinterpret_cast<const u64x2*>(ck + 2 * lane + 128 * t);     // (chunks are allocated whole: the pair is readable)
        }
        const u64 h[4] = {q[0].x, q[0].y, q[1].x, q[1].y};
        auto stage = [&]() {                                         // this chunk's keys that fall into the current tile
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const u32 k = 2 * lane + 128 * (t >> 1) + (t & 1), at = pos + k - tile_base;      // (pos + k < tile_base wraps to a huge index)
                if (k < cnt && at < DC_T) { tl[at] = FJ_HW2(h[t]); tm[at] = (mid_t)(FJ_HW1(h[t]) & midmask); }
            }
        };
        stage();
        if (pos + cnt >= tile_base + DC_T) {                         // the tile is full (cnt <= 256 < DC_T: at most once per chunk)
            flush(tile_base);
            tile_base += DC_T;
            stage();
        }
        pos += cnt;
    }
    if (pos > tile_base) flush(tile_base);
}

struct Layout { u32 bits, nparts, mid_bytes; size_t lo_off, mid_off, bytes; };
int layout_of(size_t nb_total, size_t nkeys, Layout* L) {
    const Plan p = make_plan(nb_total, 64);
    if (p.npass < 1 || p.bits < 5) return set_err("build broadcast: a build side of %zu rows in all needs no partitioning (use the owner-scatter form)", nb_total);
    if (nkeys >= (1ull << 32)) return set_err("build broadcast: %zu build rows on one rank (the offset tables are 32-bit)", nkeys);
    L->bits = (u32)p.bits; L->nparts = 1u << p.bits; L->mid_bytes = 32 - p.bits <= 16 ? 2u : 4u;
    L->lo_off = (((size_t)L->nparts + 1) * 4 + 15) & ~(size_t)15;
    L->mid_off = L->lo_off + ((nkeys * 4 + 15) & ~(size_t)15) + 16;
    L->bytes = L->mid_off + ((nkeys * L->mid_bytes + 15) & ~(size_t)15) + 16;
    return 0;
}

}  // namespace

extern "C" {

int fj_bcast_plan(size_t nb_total, int* bits, uint32_t* nparts, int* mid_bytes) {
    Layout L;
    if (layout_of(nb_total, 0, &L)) return 1;
    if (bits) *bits = (int)L.bits;
    if (nparts) *nparts = L.nparts;
    if (mid_bytes) *mid_bytes = (int)L.mid_bytes;
    return 0;
}

size_t fj_bcast_region_bytes(size_t nb_total, size_t nkeys) {
    Layout L;
    return layout_of(nb_total, nkeys, &L) ? 0 : L.bytes;
}

// where piece q's bytes of a region of `nkeys` keys lie: part 0 = the offset table (travels with piece 0), part 1 = low words,
// part 2 = high-word plane, of the keys [k_lo, k_hi)
int fj_bcast_piece_span(size_t nb_total, size_t nkeys, size_t k_lo, size_t k_hi, int part, size_t* offset, size_t* bytes) {
    Layout L;
    if (layout_of(nb_total, nkeys, &L)) return 1;
    if (k_lo > k_hi || k_hi > nkeys || part < 0 || part > 2) return set_err("fj_bcast_piece_span: bad range");
    if (part == 0) { *offset = 0; *bytes = ((size_t)L.nparts + 1) * 4; }
    else if (part == 1) { *offset = L.lo_off + k_lo * 4; *bytes = (k_hi - k_lo) * 4; }
    else { *offset = L.mid_off + k_lo * L.mid_bytes; *bytes = (k_hi - k_lo) * L.mid_bytes; }
    return 0;
}

// This rank's build rows -> its region (asynchronous on `stream`; starts the step: the plan's scalars are cleared here).
// pieces: the region will travel in that many pieces of consecutive partitions; fj_bcast_pack_bounds blocks until their key
// boundaries are known.
int fj_bcast_pack(fj_ctx* c, const uint64_t* d_keys, size_t nb, size_t nb_total, void* d_region, int pieces, void* stream) {
    if (!c) return set_err("fj_bcast_pack: null context");
    if ((nb && !d_keys) || !d_region || (((uintptr_t)d_keys | (uintptr_t)d_region) & 15)) return set_err("fj_bcast_pack: null or misaligned pointer");
    if (pieces < 1 || pieces > 16) return set_err("fj_bcast_pack: pieces must be 1..16");
    if (c->st.active) return set_err("fj_bcast_pack: a stream join is open on this context");
    Layout L;
    if (layout_of(nb_total, nb, &L)) return 1;
    FJ_ENTER(c);
    hipStream_t s = (hipStream_t)stream;
    BcastState& bc = c->bc;
    bc = BcastState();
    bc.nb_total = nb_total; bc.pieces = pieces; bc.plan = make_plan(nb_total, 64);
    begin_plan(c);
    HIPCHK(hipEventRecord(c->ev[E_START], s));
    if (clear_plan_scalars(c, s)) return 1;
    unsigned char* reg = (unsigned char*)d_region;
    u32* offs = (u32*)reg;
    if (nb == 0) {
        HIPCHK(hipMemsetAsync(offs, 0, ((size_t)L.nparts + 1) * 4, s));
        HIPCHK(hipMemsetAsync(c->d_sc->bc_bounds, 0, sizeof(c->d_sc->bc_bounds), s));
    } else {
        PassIter it;
        pass_init(it, 0, false, nb, bc.plan, 64);
        FjChunkSet cs{};
        if (run_passes(c, it, (const u64*)d_keys, nullptr, s, &cs, nullptr)) return 1;
        if (L.nparts >= 8192u && L.nparts <= (1u << 19)) {
            void* tmp = nullptr;                             // (a slot of the owner shuffle's packing pass: idle in this form)
            if (get_buf(c, W_PK_BKEYS, (size_t)L.nparts * 4, &tmp)) return 1;
            hipLaunchKernelGGL(fj_dense_count, dim3((L.nparts + DP_NT - 1) / DP_NT), dim3(DP_NT), 0, s, cs.boff, cs.list, L.nparts, (u32*)tmp);
            hipLaunchKernelGGL(fj_dense_scan_wide, dim3(L.nparts / 4096u), dim3(1024), 0, s, (const u32*)tmp, offs, L.nparts, (u32)pieces, c->d_sc->bc_bounds);
        } else {
            hipLaunchKernelGGL(fj_dense_count, dim3((L.nparts + DP_NT - 1) / DP_NT), dim3(DP_NT), 0, s, cs.boff, cs.list, L.nparts, offs);
            hipLaunchKernelGGL(fj_dense_scan, dim3(1), dim3(1024), 0, s, offs, L.nparts, (u32)pieces, c->d_sc->bc_bounds);
        }
        const u32 midmask = L.bits ? (L.bits >= 32 ? 0u : (0xFFFFFFFFu >> L.bits)) : 0xFFFFFFFFu;
        const u32 grid = (L.nparts + DP_NT / 64 - 1) / (DP_NT / 64);
        if (L.mid_bytes == 2) hipLaunchKernelGGL(fj_dense_copy<2>, dim3(grid), dim3(DP_NT), 0, s, cs.keys, cs.boff, cs.list, L.nparts, offs, midmask, (u32*)(reg + L.lo_off), (void*)(reg + L.mid_off));
        else hipLaunchKernelGGL(fj_dense_copy<4>, dim3(grid), dim3(DP_NT), 0, s, cs.keys, cs.boff, cs.list, L.nparts, offs, midmask, (u32*)(reg + L.lo_off), (void*)(reg + L.mid_off));
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipMemcpyAsync(c->pk_h, c->d_sc->bc_bounds, (size_t)(pieces + 1) * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(c->pk_h + 32, &c->d_sc->err, 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipEventRecord(c->pk_ev, s));
    HIPCHK(hipEventRecord(c->ev[E_BUILD], s));
    bc.packed = true; bc.nb = nb;
    return 0;
}

// blocks until the pack has run: h_bounds[q] = index of the first key of piece q in this rank's region, h_bounds[pieces] = nb
int fj_bcast_pack_bounds(fj_ctx* c, uint64_t* h_bounds) {
    if (!c || !h_bounds) return set_err("fj_bcast_pack_bounds: null argument");
    if (!c->bc.packed) return set_err("fj_bcast_pack_bounds: no pack in flight");
    FJ_ENTER(c);
    HIPCHK(hipEventSynchronize(c->pk_ev));
    const u32* b = reinterpret_cast<const u32*>(c->pk_h);
    const u32 err = *reinterpret_cast<const u32*>(c->pk_h + 32);
    if (err & FJ_ERR_POOL) return set_err("internal error: chunk pool exhausted while partitioning the build side for the broadcast");
    for (int q = 0; q <= c->bc.pieces; ++q) h_bounds[q] = b[q];
    if (h_bounds[c->bc.pieces] != c->bc.nb) return set_err("internal error: the broadcast region holds %llu of %zu build rows", (unsigned long long)h_bounds[c->bc.pieces], c->bc.nb);
    return 0;
}

// This rank's probe rows through both passes of the global plan (asynchronous); they stay where they are.
int fj_bcast_probe(fj_ctx* c, const uint64_t* d_pk, size_t np, size_t nb_total, void* stream) {
    if (!c) return set_err("fj_bcast_probe: null context");
    if ((np && !d_pk) || ((uintptr_t)d_pk & 15)) return set_err("fj_bcast_probe: null or misaligned pointer");
    BcastState& bc = c->bc;
    if (!bc.packed || bc.nb_total != nb_total) return set_err("fj_bcast_probe: fj_bcast_pack of the same step comes first");
    FJ_ENTER(c);
    hipStream_t s = (hipStream_t)stream;
    bc.np = np;
    if (np) {
        pass_init(bc.pit, 1, false, np, bc.plan, 64);
        bc.pit.want_items = true; bc.pit.item_tc_max = 32;
        bc.ja = FjLdsJoinArgs();
        if (run_passes(c, bc.pit, (const u64*)d_pk, nullptr, s, &bc.ja.probe, &bc.evc)) return 1;
        bc.ja.items = bc.pit.tiles; bc.ja.nitems_dev = bc.pit.ntiles; bc.ja.items_cap = bc.pit.items_cap; bc.ja.part_count = bc.pit.part_count;
        bc.ja.nparts = bc.ja.probe.nb; bc.ja.nsplit = 1;
        bc.ja.total = &c->d_sc->total; bc.ja.err = &c->d_sc->err;
    }
    HIPCHK(hipEventRecord(c->ev[E_PPART], s));
    bc.probed = true;
    return 0;
}

// join the probe rows of partitions [part_lo, part_hi) against the runs of nsrc sources: source i's region (of nkeys[i] keys,
// laid out by fj_bcast_pack on its rank) starts region_off[i] bytes into d_base
int fj_bcast_join(fj_ctx* c, const void* d_base, int nsrc, const uint64_t* region_off, const uint64_t* nkeys, uint32_t part_lo, uint32_t part_hi, void* stream) {
    if (!c) return set_err("fj_bcast_join: null context");
    BcastState& bc = c->bc;
    if (!bc.probed) return set_err("fj_bcast_join: fj_bcast_probe of the same step comes first");
    if (!d_base || nsrc < 1 || nsrc > (int)FJ_WIDE_MAXSRC || !region_off || !nkeys) return set_err("fj_bcast_join: 1..%u sources", FJ_WIDE_MAXSRC);
    Layout L0;
    if (layout_of(bc.nb_total, 0, &L0)) return 1;
    if (part_lo > part_hi || part_hi > L0.nparts) return set_err("fj_bcast_join: partitions [%u, %u) of %u", part_lo, part_hi, L0.nparts);
    if (bc.np == 0 || part_lo == part_hi) return 0;
    FJ_ENTER(c);
    FjWideArgs w{};
    w.toff = bc.pit.toff; w.part_lo = part_lo; w.part_hi = part_hi;
    w.base = (const unsigned char*)d_base; w.nsrc = (u32)nsrc; w.bits = L0.bits; w.mid_bytes = L0.mid_bytes; w.pmask = fj_wide_pmask((int)L0.bits, 64);
    // few ranks = few, fat partitions: > ~28 probe chunks each means two or three items per partition (items hold <= 32 probe chunks),
    // dealt in runs of 8 so that a partition's items find their table built (2 and 4 ranks of config 5: 3 and 2 items per partition)
    w.group_log = bc.np / L0.nparts > 7000 ? 3u : 0u;
    for (int i = 0; i < nsrc; ++i) {
        Layout L;
        if (layout_of(bc.nb_total, (size_t)nkeys[i], &L)) return 1;
        if (region_off[i] & 15) return set_err("fj_bcast_join: region offsets must be multiples of 16");
        w.offs_off[i] = region_off[i]; w.lo_off[i] = region_off[i] + L.lo_off; w.mid_off[i] = region_off[i] + L.mid_off;
    }
    const u32 cus = c->reserve_cus < c->num_cus ? c->num_cus - c->reserve_cus : 1u;
    HIPCHK(fj_launch_count_join_wide(bc.ja, w, true, cus, (hipStream_t)stream));
    return 0;
}

int fj_bcast_finish(fj_ctx* c, void* stream, uint64_t* out_count, fj_timings* timings) {
    if (!c) return set_err("fj_bcast_finish: null context");
    BcastState& bc = c->bc;
    if (!bc.probed) return set_err("fj_bcast_finish: no broadcast join in flight");
    FJ_ENTER(c);
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(hipEventRecord(c->ev[E_JOIN], s));
    if (read_scalars(c, s)) return 1;
    const u32 err = c->h_sc->err;
    const int npass = bc.plan.npass, bits = bc.plan.bits, evc = bc.evc;
    bc = BcastState();
    if (err & FJ_ERR_POOL) return set_err("internal error: chunk pool exhausted during a partition pass");
    end_plan(c);
    // a final partition beyond the LDS table (> ~8000 build keys in all: build-side skew): the caller takes another form
    if (err & (FJ_STAT_RETRY | FJ_ERR_LDS_FULL)) return set_err("build broadcast: a final partition does not fit the LDS table (skewed build keys)");
    if (out_count) *out_count = c->h_sc->total;
    fj_timings t; memset(&t, 0, sizeof t);
    t.path = 0; t.passes = npass; t.radix_bits = bits; t.partitions = 1ull << bits; t.sampled_hit_bp = -1;
    for (int i = 0; i < evc && i < 4; ++i) t.probe_part_kernel_ms[i] = ev_ms(c, E_PK0 + 2 * i, E_PK0 + 2 * i + 1);
    t.build_phase_ms = ev_ms(c, E_START, E_BUILD); t.join_ms = ev_ms(c, E_PPART, E_JOIN);
    t.probe_phase_ms = ev_ms(c, E_BUILD, E_JOIN); t.total_ms = ev_ms(c, E_START, E_JOIN);
    if (timings) *timings = t;
    last_timings() = t;
    return 0;
}

void fj_bcast_abort(fj_ctx* c) { if (c) { c->bc = BcastState(); } }

}  // extern "C"
