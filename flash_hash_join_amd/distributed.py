"""Multi-GPU radix join: one process per GPU over RCCL/xGMI.

The reference is single-process; this is new design.  Two exchange strategies, chosen per call
by FJ_DIST_STRATEGY = shuffle | replicate | auto (a per-link byte + local-work cost model, choose_strategy); unset it
means the shuffle north_star names:

replicate -- every rank all-gathers the build KEYS (and values when materialising) and joins its own probe rows
  against all of them; probe rows never move, their partition passes run while the build keys are on the wire
  (fj_stream_open / append_probe / advance_probe, then append_build per arrived piece / finish; FJ_REPLICATE_PIECES,
  default 4 asynchronous all-gathers: the first pass over piece c runs while piece c+1 is on the wire).  An xGMI mesh
  has one link per peer, so an exchange is bound by bytes per link: B*8 here against (P*8 + B*16)/N for the shuffle -- fewer up to
  N = 12 for the probe-heavy (P = 10 B) joins this path is built for, 6x fewer at N = 2.  Cost: every rank
  partitions all N*B build keys.  Global count = sum of local counts; pairs stay with their probe row.

shuffle -- radix partitions are independent join units (hash_join.cpp:340-356, :515-525), so the level-0 digit is
  the owner GPU:   owner(key) = (top 16 bits of hash(key) * world) >> 16

  Joins whose GLOBAL plan has two or more passes take the CHUNK form: the first radix pass of the plan for the
  total build side IS the owner split - bucket b of its 256 / 512 buckets belongs to rank (b * world) >> log2(buckets); a sender
  runs that pass and rewrites its output for the wire (fj_shuffle_pack_begin / _counts / _finish: dense 256-key chunks, 7 bytes
  per key, one directory word per chunk), and the owner starts at the plan's second pass.  The protocol lives in ONE place, the
  C++ driver csrc/fj_dist.hip (fj_dist_join / fj_dist_join_count); this module hands it RCCL (nccl backend), or three callbacks
  into torch.distributed (_CallbackTransport: gloo, a transport object), or - in the CPU test-suite - also a stand-in for the
  rank's own work.  Everything else (small build sides, the sender-side precheck, duplicate build keys in a materialising join,
  and any step the chunk form fails on) takes the OWNER-SCATTER form, driven from here:
  1. every rank splits its local rows of both relations by owner (fj_owner_split: LDS counting
     sort per tile, contiguous per-owner segments);
  2. ONE all-to-all per relation moves each segment to its owner (torch.distributed
     all_to_all_single, backend "nccl" == RCCL on ROCm; a fully connected xGMI mesh carries one
     peer per link); counting joins cut the probe exchange into pieces and overlap it with the split of the
     next piece and the first partition pass of the previous one;
  3. each rank joins what it owns with the single-GPU radix join (hash_top_bits = 48: the owner
     digit is already consumed);
  4. the global count is one all-reduce of a single int64.  Materialised pairs stay sharded by owner.

`engine` abstracts the per-rank primitives so the protocol can be exercised on CPU (gloo) in the
test-suite with a stand-in engine; the default engine is the HIP one and has no CPU fallback.
"""
from __future__ import annotations

import ctypes
import os
import time
from typing import List, Optional, Tuple


class HipEngine:
    """Per-rank primitives on an MI355X through the C ABI (include/flashjoin.h)."""

    def __init__(self, device=None):
        import torch
        from . import _lib, api
        self.torch = torch
        self.L = _lib.load()
        self._lib = _lib
        self.api = api
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device())
        self.device = torch.device(device)
        self.index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.ctx = api.context(self.index)

    def empty(self, n: int):
        return self.torch.empty(n, dtype=self.torch.int64, device=self.device)

    def normalize(self, bk, bv, pk):
        """The dtype / contiguity / alignment normalisation of api.join_device, for callers that hand raw data_ptr()s on."""
        bk, bv, pk = self.api._dev_tensor(bk, "build_keys"), self.api._dev_tensor(bv, "build_values"), self.api._dev_tensor(pk, "probe_keys")
        if bv.numel() < bk.numel():
            raise ValueError(f"build_values has {bv.numel()} elements, build_keys has {bk.numel()}")
        return bk, bv, pk

    def empty_like(self, t):
        return self.torch.empty_like(t)

    def cat(self, parts):
        return self.torch.cat(list(parts))

    def counts_tensor(self, counts: List[int]):
        return self.torch.tensor(counts, dtype=self.torch.int64, device=self.device)

    def owner_split(self, keys, vals, world: int):
        t = self.torch
        n = keys.numel()
        out_k = self.empty(n)
        out_v = self.empty(n) if vals is not None else None
        counts = (ctypes.c_uint64 * 64)()
        self._lib.check(self.L.fj_owner_split(
            self.ctx, keys.data_ptr(), vals.data_ptr() if vals is not None else None, n, world,
            out_k.data_ptr(), out_v.data_ptr() if out_v is not None else None, counts,
            t.cuda.current_stream(self.index).cuda_stream))
        return out_k, out_v, [int(counts[r]) for r in range(world)]

    def owner_hist(self, keys, world: int) -> List[int]:
        counts = (ctypes.c_uint64 * 64)()
        self._lib.check(self.L.fj_owner_hist(self.ctx, keys.data_ptr(), keys.numel(), world, counts,
                                             self.torch.cuda.current_stream(self.index).cuda_stream))
        return [int(counts[r]) for r in range(world)]

    def owner_scatter(self, keys, world: int, counts: List[int]):
        """Owner-contiguous copy of `keys` given its per-owner counts; asynchronous on the current stream."""
        out = self.empty(keys.numel())
        c = (ctypes.c_uint64 * 64)(*counts)
        self._lib.check(self.L.fj_owner_scatter(self.ctx, keys.data_ptr(), None, keys.numel(), world, c, out.data_ptr(), None,
                                                self.torch.cuda.current_stream(self.index).cuda_stream))
        return out

    def stream_begin(self, bk, bv, np_bound: int, max_appends: int, hash_top_bits: int):
        self._keep = [bk, bv]                                    # inputs must outlive the asynchronous kernels
        self._lib.check(self.L.fj_stream_begin(self.ctx, bk.data_ptr(), bv.data_ptr(), bk.numel(), np_bound, max_appends,
                                               self.torch.cuda.current_stream(self.index).cuda_stream, hash_top_bits))

    def stream_open(self, nb_bound: int, build_appends: int, np_bound: int, probe_appends: int, hash_top_bits: int):
        self._keep = []
        self._lib.check(self.L.fj_stream_open(self.ctx, nb_bound, build_appends, np_bound, probe_appends,
                                              self.torch.cuda.current_stream(self.index).cuda_stream, hash_top_bits))

    @staticmethod
    def _aligned(t):
        """The C ABI wants contiguous, 16-byte aligned pieces (a view into a larger tensor may be neither)."""
        if not t.is_contiguous() or t.data_ptr() % 16:
            t = t.contiguous().clone() if t.data_ptr() % 16 else t.contiguous()
        return t

    def stream_append_build(self, piece):
        piece = self._aligned(piece)
        self._keep.append(piece)
        self._lib.check(self.L.fj_stream_append_build(self.ctx, piece.data_ptr(), piece.numel(),
                                                      self.torch.cuda.current_stream(self.index).cuda_stream))

    def stream_advance_probe(self):
        self._lib.check(self.L.fj_stream_advance_probe(self.ctx, self.torch.cuda.current_stream(self.index).cuda_stream))

    def stream_append(self, piece):
        piece = self._aligned(piece)
        self._keep.append(piece)
        self._lib.check(self.L.fj_stream_append_probe(self.ctx, piece.data_ptr(), piece.numel(),
                                                      self.torch.cuda.current_stream(self.index).cuda_stream))

    def stream_finish(self) -> int:
        cnt = ctypes.c_uint64(0)
        t = self._lib.FjTimings()
        try:
            self._lib.check(self.L.fj_stream_finish(self.ctx, self.torch.cuda.current_stream(self.index).cuda_stream,
                                                    ctypes.byref(cnt), ctypes.byref(t)))
        finally:
            self._keep = []
        self.api._last = t
        return int(cnt.value)

    def bloom_export(self, build_keys, hash_top_bits: int):
        """Bloom filters of the build keys this rank owns: 512 radix buckets x fj_bloom_filter_words()/512 words (int32 tensor)."""
        t = self.torch
        build_keys = self._aligned(build_keys)
        out = t.empty(int(self.L.fj_bloom_filter_words()), dtype=t.int32, device=self.device)
        self._lib.check(self.L.fj_bloom_export(self.ctx, build_keys.data_ptr(), build_keys.numel(), hash_top_bits, out.data_ptr(),
                                               t.cuda.current_stream(self.index).cuda_stream))
        return out

    def bloom_prefilter(self, keys, filters, hash_top_bits: int):
        """The rows of `keys` that may match the owner whose filters these are (no row that matches is dropped; the order changes)."""
        t = self.torch
        keys = self._aligned(keys)
        out = self.empty(keys.numel())
        n = ctypes.c_uint64(0)
        self._lib.check(self.L.fj_bloom_prefilter(self.ctx, keys.data_ptr(), keys.numel(), hash_top_bits, filters.data_ptr(),
                                                  out.data_ptr(), out.numel(), ctypes.byref(n), t.cuda.current_stream(self.index).cuda_stream))
        return out[: int(n.value)]

    # ---- owner shuffle in chunk form (the first radix pass of the global plan is the owner split) ----
    def empty_i32(self, n: int):
        return self.torch.empty(n, dtype=self.torch.int32, device=self.device)

    def shuffle_plan(self, nb_total: int, world: int) -> Optional[int]:
        """log2 of the first-pass fan-out of the plan for a build side of nb_total rows in all, or None when the chunk form
        does not apply (a one-pass plan: build sides under ~2M rows)."""
        f0, npass = ctypes.c_int(0), ctypes.c_int(0)
        if self.L.fj_shuffle_plan(nb_total, world, ctypes.byref(f0), ctypes.byref(npass)):
            return None
        return int(f0.value)

    def shuffle_chunk_bytes(self, nb_total: int, world: int) -> int:
        return int(self.L.fj_shuffle_chunk_bytes(nb_total, world))

    def part_filter_range(self, nb_total: int, world: int, rank: int):
        """(first, count, total, bytes_each): where rank's per-partition Bloom filters sit among all final partitions' filters."""
        sz = ctypes.c_size_t
        first, count, total = sz(0), sz(0), sz(0)
        self._lib.check(self.L.fj_shuffle_part_filter_range(nb_total, world, rank, ctypes.byref(first), ctypes.byref(count), ctypes.byref(total)))
        return int(first.value), int(count.value), int(total.value), int(self.L.fj_shuffle_part_filter_bytes())

    def stream_export_part_filters(self, out):
        """Filters of the final partitions this owner holds (build side complete), into the uint8 tensor `out`."""
        self._lib.check(self.L.fj_stream_export_part_filters(self.ctx, out.data_ptr(), self.torch.cuda.current_stream(self.index).cuda_stream))

    def shuffle_pack(self, keys, vals, nb_total: int, world: int, filters=None):
        """First pass of the global plan over local rows, rewritten for the wire (fj_shuffle_pack_begin / _counts / _finish).
        Returns (chunks, dir, used): per owner r a uint8 tensor of used[r] * shuffle_chunk_bytes() bytes (dense 256-key chunks
        in the 7-byte wire format when the first pass has >= 256 buckets) and an int32 tensor of used[r] directory words.
        filters: all partitions' Bloom filters (uint8 tensor) - the piece is prechecked against them (fj_shuffle_pack_filter);
        self.last_pack_kept = the rows it kept."""
        t = self.torch
        keys = self._aligned(keys)
        stream = t.cuda.current_stream(self.index).cuda_stream
        cb = self.shuffle_chunk_bytes(nb_total, world)
        if cb == 0:
            raise RuntimeError(self._lib.last_error())
        if vals is not None:
            vals = self._aligned(vals)
        self._lib.check(self.L.fj_shuffle_pack_begin(self.ctx, keys.data_ptr(), vals.data_ptr() if vals is not None else None, keys.numel(), nb_total, world,
                                                     int(filters is not None), stream))
        if filters is not None:
            self._lib.check(self.L.fj_shuffle_pack_filter(self.ctx, filters.data_ptr(), stream))
        used = (ctypes.c_uint64 * 64)()
        self._lib.check(self.L.fj_shuffle_pack_counts(self.ctx, used))
        self.last_pack_kept = int(self.L.fj_shuffle_pack_kept(self.ctx)) if filters is not None else keys.numel()
        used = [int(used[r]) for r in range(world)]
        chunks = [t.empty(max(16, u * cb), dtype=t.uint8, device=self.device) for u in used]
        dirs = [self.empty_i32(max(4, u)) for u in used]
        vp = ctypes.c_void_p
        dk = (vp * 64)(*[c.data_ptr() for c in chunks])
        dd = (vp * 64)(*[d.data_ptr() for d in dirs])
        if vals is None:
            self._lib.check(self.L.fj_shuffle_pack_finish(self.ctx, dk, None, dd, stream))
            return [c[: u * cb] for c, u in zip(chunks, used)], [d[:u] for d, u in zip(dirs, used)], used
        vouts = [self.empty(max(2, u * 256)) for u in used]
        dv = (vp * 64)(*[v.data_ptr() for v in vouts])
        self._lib.check(self.L.fj_shuffle_pack_finish(self.ctx, dk, dv, dd, stream))
        return [c[: u * cb] for c, u in zip(chunks, used)], [d[:u] for d, u in zip(dirs, used)], used, [v[: u * 256] for v, u in zip(vouts, used)]

    def stream_open_shuffled(self, nb_total: int, world: int, rank: int, nb_bound: int, build_appends: int, np_bound: int, probe_appends: int,
                             with_vals: bool = False):
        self._keep = []
        self._lib.check(self.L.fj_stream_open_shuffled(self.ctx, nb_total, world, rank, nb_bound, build_appends, np_bound, probe_appends, int(with_vals),
                                                       self.torch.cuda.current_stream(self.index).cuda_stream))

    def stream_append_chunks(self, side: int, chunks, dirw, vals=None):
        """A received piece: wire-format chunks (uint8 tensor) + their directory words (rewritten in place) [+ 256 values per build chunk]."""
        self._keep += [chunks, dirw, vals]
        stream = self.torch.cuda.current_stream(self.index).cuda_stream
        if side:
            self._lib.check(self.L.fj_stream_append_probe_chunks(self.ctx, chunks.data_ptr(), dirw.data_ptr(), dirw.numel(), stream))
        else:
            self._lib.check(self.L.fj_stream_append_build_chunks(self.ctx, chunks.data_ptr(), vals.data_ptr() if vals is not None else None, dirw.data_ptr(),
                                                                 dirw.numel(), stream))

    # ---- build-broadcast form (csrc/fj_bcast.hip): probe rows never move, every rank's build rows travel as dense per-partition runs ----
    def bcast_plan(self, nb_total: int):
        """(radix bits, final partitions, bytes of the high-word plane per key) of the plan for a total build side, or None when
        that plan has no pass (such joins take the owner-scatter form)."""
        b, n, m = ctypes.c_int(0), ctypes.c_uint32(0), ctypes.c_int(0)
        if self.L.fj_bcast_plan(nb_total, ctypes.byref(b), ctypes.byref(n), ctypes.byref(m)):
            return None
        return b.value, n.value, m.value

    def bcast_region_bytes(self, nb_total: int, nkeys: int) -> int:
        return int(self.L.fj_bcast_region_bytes(nb_total, nkeys))

    def bcast_pack(self, keys, nb_total: int, region, pieces: int) -> None:
        """Asynchronous: this rank's build keys -> `region` (a uint8 tensor view of bcast_region_bytes(nb_total, keys.numel()) bytes)."""
        s = self.torch.cuda.current_stream(self.index).cuda_stream
        self._lib.check(self.L.fj_bcast_pack(self.ctx, keys.data_ptr(), keys.numel(), nb_total, region.data_ptr(), pieces, s))

    def bcast_pack_bounds(self, pieces: int) -> List[int]:
        b = (ctypes.c_uint64 * (pieces + 1))()
        self._lib.check(self.L.fj_bcast_pack_bounds(self.ctx, b))
        return [int(x) for x in b]

    def bcast_probe(self, probe_keys, nb_total: int) -> None:
        s = self.torch.cuda.current_stream(self.index).cuda_stream
        self._lib.check(self.L.fj_bcast_probe(self.ctx, probe_keys.data_ptr(), probe_keys.numel(), nb_total, s))

    def bcast_join(self, base, region_off: List[int], nkeys: List[int], part_lo: int, part_hi: int) -> None:
        n = len(region_off)
        ro, nk = (ctypes.c_uint64 * n)(*region_off), (ctypes.c_uint64 * n)(*nkeys)
        s = self.torch.cuda.current_stream(self.index).cuda_stream
        self._lib.check(self.L.fj_bcast_join(self.ctx, base.data_ptr(), n, ro, nk, part_lo, part_hi, s))

    def bcast_finish(self) -> int:
        cnt = ctypes.c_uint64(0)
        t = self._lib.FjTimings()
        s = self.torch.cuda.current_stream(self.index).cuda_stream
        self._lib.check(self.L.fj_bcast_finish(self.ctx, s, ctypes.byref(cnt), ctypes.byref(t)))
        self.last_bcast_timings = t.as_dict()
        return int(cnt.value)

    def emit_pairs(self, n: int):
        """The pairs of the materialising join that was just counted on this context (fj_emit_pairs): two int64 tensors of n rows."""
        ok, ov = self.empty(max(n, 2)), self.empty(max(n, 2))
        t = self._lib.FjTimings()
        self._lib.check(self.L.fj_emit_pairs(self.ctx, ok.data_ptr(), ov.data_ptr(), n, self.torch.cuda.current_stream(self.index).cuda_stream, ctypes.byref(t)))
        return ok[:n], ov[:n]

    def chunk_rows(self, dirw) -> int:
        """Rows in a set of chunks, from their directory words (bucket << 9 | count; unused ids are all ones)."""
        if dirw.numel() == 0:
            return 0
        cnt = dirw & 0x1FF
        return int(cnt[dirw != -1].sum().item())

    # ---- the multi-GPU driver (csrc/fj_dist.hip): fj_dist_join_count over an RCCL communicator of the library's own (torch does
    #      not hand out its ncclComm_t) ----
    _native_comms: dict = {}

    def native_comm(self, dist, group):
        """fj_dist_comm for (this device, the group's ranks): rank 0's unique id travels by a torch.distributed broadcast, every
        rank then joins ncclCommInitRank inside fj_dist_comm_create.  Cached for the life of the process (closed at exit)."""
        ranks = tuple(dist.get_process_group_ranks(group)) if group is not None else tuple(range(dist.get_world_size()))
        key = (self.index, ranks)
        comm = HipEngine._native_comms.get(key)
        if comm:
            return comm
        t = self.torch
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        idt = t.zeros(128, dtype=t.uint8, device=self.device)
        if rank == 0:
            raw = ctypes.create_string_buffer(128)
            self._lib.check(self.L.fj_dist_unique_id(raw))
            idt.copy_(t.frombuffer(bytearray(raw.raw), dtype=t.uint8))
        dist.broadcast(idt, 0 if group is None else dist.get_global_rank(group, 0), group=group)
        raw = bytes(idt.cpu().numpy().tobytes())
        with t.cuda.device(self.index):
            comm = self.L.fj_dist_comm_create(self.ctx, raw, world, rank)
        if not comm:
            raise RuntimeError(self._lib.last_error())
        if not HipEngine._native_comms:
            import atexit
            atexit.register(HipEngine.close_native_comms)
        HipEngine._native_comms[key] = comm
        return comm

    @staticmethod
    def close_native_comms():
        """Destroy the cached RCCL communicators (before torch.distributed.destroy_process_group, or at exit)."""
        from . import _lib
        comms, HipEngine._native_comms = HipEngine._native_comms, {}
        for comm in comms.values():
            try:
                _lib.load().fj_dist_comm_destroy(comm)
            except Exception:                    # noqa: BLE001  (interpreter shutdown)
                pass

    def stream_abort(self):
        """Error recovery: drop a stream join that will not be finished, so that the context serves other joins again."""
        self._keep = []
        self._lib.check(self.L.fj_stream_abort(self.ctx))

    def local_join(self, bk, bv, pk, materialize: bool, bloom: bool, hash_top_bits: int, return_arrays: bool):
        return self.api.join_device(self.api.ALGO_RADIX, int(bloom), int(materialize), bk, bv, pk,
                                    return_arrays=return_arrays, hash_top_bits=hash_top_bits)

    def synchronize(self):
        self.torch.cuda.synchronize(self.index)


# RCCL moves wrong data when one peer-to-peer message of an all-to-all exceeds 4 GiB (measured here: an int64
# all_to_all_single is exact at 0.8 GB per peer, wrong at 4.8 GB), so larger segments travel in several rounds.
_MAX_ELEMS_PER_MESSAGE = 1 << 27          # 1 GiB of int64 per (source, destination) and round


def _views_all_to_all(dist, group, ins, outs):
    """All-to-all over per-peer VIEWS (slices of larger tensors, not adjacent in memory): grouped point-to-point sends and
    receives - ncclSend / ncclRecv inside one group call under RCCL, isend / irecv under gloo (whose list-form all_to_all does
    not exist) - plus a local copy for this rank's own slice.  Empty slices are skipped on both sides (sender and receiver
    know the sizes).  Returns the list of outstanding works (wait on all of them)."""
    me = dist.get_rank(group)
    world = len(ins)
    if outs[me].numel():
        outs[me].copy_(ins[me])
    peer = (lambda r: r) if group is None else (lambda r: dist.get_global_rank(group, r))
    ops = []
    for r in range(world):
        if r != me and ins[r].numel():
            ops.append(dist.P2POp(dist.isend, ins[r], peer(r), group))
    for r in range(world):
        if r != me and outs[r].numel():
            ops.append(dist.P2POp(dist.irecv, outs[r], peer(r), group))
    return list(dist.batch_isend_irecv(ops)) if ops else []


def _exchange(dist, group, engine, send, send_counts: List[int], recv_counts: List[int], rounds: int):
    """One all-to-all of variable-size int64 segments. `rounds` (identical on every rank) > 1 splits every
    segment into `rounds` near-equal pieces; piece r of every segment moves in round r, straight into its
    final position of the receive buffer (list-based all_to_all on views: no staging copies)."""
    recv = engine.empty(sum(recv_counts))
    if rounds <= 1:
        dist.all_to_all_single(recv, send, output_split_sizes=recv_counts, input_split_sizes=send_counts, group=group)
        return recv
    soff = [0]
    for c in send_counts:
        soff.append(soff[-1] + c)
    roff = [0]
    for c in recv_counts:
        roff.append(roff[-1] + c)
    for r in range(rounds):
        ins = [send[soff[d] + send_counts[d] * r // rounds: soff[d] + send_counts[d] * (r + 1) // rounds]
               for d in range(len(send_counts))]
        outs = [recv[roff[q] + recv_counts[q] * r // rounds: roff[q] + recv_counts[q] * (r + 1) // rounds]
                for q in range(len(recv_counts))]
        for w in _views_all_to_all(dist, group, ins, outs):
            w.wait()
    return recv


def _abort_stream(engine) -> None:
    if hasattr(engine, "stream_abort"):
        try:
            engine.stream_abort()
        except Exception:                        # noqa: BLE001  (the original error is the one to report)
            pass


def _prefilter_mode(bloom: bool) -> str:
    """Sender-side bloom precheck of the probe exchange: "on" | "off" | "auto".  FJ_DIST_PREFILTER=1 / 0 / auto decides;
    unset, the *_bloom meaning (`bloom=True`) asks for "auto": the filters are exported and a sample of the probe rows is
    tested against them; the precheck runs when few enough rows survive to pay for the extra pass (_prefilter_break_even)."""
    env = os.environ.get("FJ_DIST_PREFILTER", "")
    if env in ("0", "1", "auto"):
        return {"0": "off", "1": "on", "auto": "auto"}[env]
    return "auto" if bloom else "off"


# fj_bloom_prefilter on one MI355X, 156M-row segments against 125M-key owners (profiles/r02_prefilter_probe.txt):
# 1.26 ms at 13 % survivors ... 1.91 ms at 100 %  =  7.4 ps + 4.9 ps x survivors, per row
_PREFILTER_S_PER_ROW = 7.4e-12
_PREFILTER_S_PER_SURVIVOR = 4.9e-12
_PREFILTER_SAMPLE_ROWS = 1 << 20


def _prefilter_break_even(world: int) -> float:
    """Survivor fraction below which the precheck pays: a sender filters `world` segments one after the other while its
    links carry one segment each in parallel, so per segment row it spends world * (a + b f) and saves (8 B / link) * (1 - f).
    A 0.8 margin covers what the model leaves out (the filters' own 73 MB per link, the lost scatter/exchange overlap)."""
    link = 8.0 / _LINK_BYTES_PER_S
    f = (link - _PREFILTER_S_PER_ROW * world) / (link + _PREFILTER_S_PER_SURVIVOR * world)
    return max(0.0, 0.8 * f)


# the precheck in chunk form (fj_dist_join(prefilter_below), csrc/fj_pack.hip: fj_part_filter_inplace): what a config-5 shard costs
# per local probe row on ONE MI355X through the driver (profiles/r04_prefilter_one_rank.txt: 15.0-16.3 ms without it; with it the
# same at 8 % survivors, +4.7-5.8 ms at 52 %), as  off = FIXED + REST,  on(f) = FIXED + FILTER + f x REST
_CHUNK_FIXED_S_PER_ROW = 6.0e-12        # first pass of the global plan over the probe rows + the build side's and the host's share of the step
_CHUNK_FILTER_S_PER_ROW = 7.4e-12       # the precheck at the 8-rank plan (512 partitions = 2 MiB of filters per first-pass bucket: 2.3 ms per
                                        # 312M rows, profiles/r04_precheck_probe.txt; 4.9-6.6 ps at the 1-rank plan's 128 partitions per bucket)
_CHUNK_REST_S_PER_ROW = 6.1e-12         # copy into the wire format + the owner's second pass, lists and join: scale with what survives
_CHUNK_WIRE_BYTES_PER_ROW = 7.02


def _chunk_prefilter_break_even(world: int, nb_total: int, np_local: int, link_bytes_per_s: Optional[float] = None) -> float:
    """Survivor fraction below which the precheck of the chunk form pays (FJ_DIST_PREFILTER_BELOW overrides the model).  Per local
    probe row a step costs max(wire, kernels) - the model of tools/scale_model.py: wire = 7.02 B x (f x probe rows + build rows) /
    (world x link rate) on each of the links that work in parallel, plus - with the precheck - the filters (1.07 bytes per build
    key to every rank: ~nb_total / world bytes per link); kernels as above; ~0.5 ms of latency before the first probe piece can be
    checked.  It never pays where the kernels bound the step (one rank; links faster than ~12 ps per row); on wire-bound steps it
    does: below ~75 % survivors at 8 GPUs and 45 GB/s per link (~55 % at 55 GB/s, ~15 % at 65), below ~85 % at 2-4 GPUs."""
    env = os.environ.get("FJ_DIST_PREFILTER_BELOW")
    if env:
        return float(env)
    if np_local <= 0 or world <= 1:
        return 0.0
    link = float(link_bytes_per_s or _LINK_BYTES_PER_S)                      # (a multi-rank caller passes rank 0's: one verdict everywhere)
    build_share = nb_total / world / np_local                                # build rows per probe row: they travel either way
    per_row = _CHUNK_WIRE_BYTES_PER_ROW / (world * link)                     # one row's share of one link
    filters = 1.07 * build_share / link                                      # the all-gathered filters, per probe row
    off = max(per_row * (1.0 + build_share), _CHUNK_FIXED_S_PER_ROW + _CHUNK_REST_S_PER_ROW)

    def on(f):
        return max(per_row * (f + build_share) + filters, _CHUNK_FIXED_S_PER_ROW + _CHUNK_FILTER_S_PER_ROW + f * _CHUNK_REST_S_PER_ROW) + 0.5e-3 / np_local
    if on(0.0) >= off:
        return 0.0
    lo, hi = 0.0, 1.0
    for _ in range(30):
        mid = 0.5 * (lo + hi)
        lo, hi = (mid, hi) if on(mid) < off else (lo, mid)
    return 0.9 * lo


# "auto" remembers what the last join of the same shape sampled: exporting and all-gathering the filters only to decline again would
# cost every step of a repeated join ~3 ms at 8 ranks (1 byte per build key to every rank).  (world, build rows, probe rows, join_id)
# -> [calls, sampled survivor share]; every 32nd call samples afresh.  Collective calls keep the ranks' memos identical: the
# threshold is computed from rank 0's link rate (it travels with the relation sizes), never from a per-rank value.  join_id: what
# the caller names the join (distributed_join(join_id=...)) - two joins of one shape but different hit rates (another probe column)
# must not inherit each other's verdict; unnamed joins of one shape share one entry.
_PRECHECK_MEMO: dict = {}
_PRECHECK_RESAMPLE_EVERY = 32


def _precheck_threshold(mode: str, world: int, nb_total: int, np_global: int, link_bytes_per_s: Optional[float] = None, join_id=None):
    """(prefilter_below for fj_dist_join, memo key or None, how it was decided)."""
    if mode != "auto":
        return {"off": 0.0, "on": 2.0}[mode], None, mode
    below = _chunk_prefilter_break_even(world, nb_total, max(1, np_global // world), link_bytes_per_s)
    key = (world, nb_total, np_global, join_id)
    memo = _PRECHECK_MEMO.get(key)
    if below <= 0.0:
        return 0.0, None, "model: cannot pay"
    if memo is not None and memo[0] % _PRECHECK_RESAMPLE_EVERY != 0:
        return (2.0, key, "memo: runs") if memo[1] < below else (0.0, key, "memo: declined")
    return below, key, "sampled"


def _precheck_remember(key, timings: dict, decision: str) -> None:
    timings["prefilter_decision"] = decision
    if key is None:
        return
    memo = _PRECHECK_MEMO.get(key)
    if decision == "sampled" and timings.get("prefilter_sampled_survivors") is not None:
        _PRECHECK_MEMO[key] = [1, float(timings["prefilter_sampled_survivors"])]
    elif memo is not None:
        memo[0] += 1


def _chunk_prefilter_mode(bloom: bool, world: int) -> str:
    """The chunk form's precheck: FJ_DIST_PREFILTER=1 / 0 / auto decides; unset it is "auto" for every join across more than one
    rank (the *_bloom functions and the plain ones alike: results are identical, and a step whose links are the bottleneck is
    shorter by what the owners' filters keep off them - the model above prices it with the measured link rate when bench.py or the
    host called set_link_rate) and for the *_bloom functions on one rank (where the model declines)."""
    env = os.environ.get("FJ_DIST_PREFILTER", "")
    if env in ("0", "1", "auto"):
        return {"0": "off", "1": "on", "auto": "auto"}[env]
    return "auto" if (bloom or world > 1) else "off"


def _sampled_survivors(dist, group, engine, world, probe_keys, filters) -> float:
    """Fraction of a strided sample of every rank's probe rows that passes the owners' filters (identical on all ranks)."""
    n = probe_keys.numel()
    kept = m = 0
    if n:
        m = min(n, _PREFILTER_SAMPLE_ROWS)
        sample = probe_keys[:: max(1, n // m)][:m].contiguous()
        m = sample.numel()
        s, _, counts = engine.owner_split(sample, None, world)
        off = 0
        for d in range(world):
            kept += int(engine.bloom_prefilter(s[off: off + counts[d]], filters[d], 48).numel())
            off += counts[d]
    v = engine.counts_tensor([kept, m])
    dist.all_reduce(v, op=dist.ReduceOp.SUM, group=group)
    kept, m = (int(x) for x in v.tolist())
    return kept / m if m else 1.0


def _pipelined_count(dist, group, engine, world, build_keys, build_values, probe_keys, pieces: int, timings: Optional[dict],
                     prefilter: str = "off"):
    """Counting join with the probe exchange cut into `pieces` rounds: the owner-scatter of piece c+1 and the first
    partition pass over piece c-1 run while piece c is on the wire (asynchronous all-to-all).

    prefilter ("on" / "auto"): every owner exports Bloom filters of the build keys it received (512 radix buckets, one
    LDS-sized filter each: fj_bloom_export), one all-gather hands them to every rank, and a rank sends an owner only the
    probe rows that pass that owner's filters (fj_bloom_prefilter).  Costs 73 MB per link for the filters + one more
    partition pass on the sender; saves (1 - survivors) of the probe exchange, which is what bounds the shuffle
    (DESIGN.md section 6)."""
    t0 = time.perf_counter()
    # build side: split, exchange, start the build-side passes
    bk_s, bv_s, b_counts = engine.owner_split(build_keys, build_values, world)
    n = probe_keys.numel()
    bounds = [(n * c // pieces) & ~1 for c in range(pieces)] + [n]      # even row offsets: every piece stays 16-byte aligned
    views = [probe_keys[bounds[c]: bounds[c + 1]] for c in range(pieces)]
    p_counts = [engine.owner_hist(v, world) for v in views]                  # [piece][owner]
    t1 = time.perf_counter()

    def exchange_counts(flat, per_rank):
        send_c = engine.counts_tensor(flat)
        recv_c = engine.counts_tensor([0] * len(flat))
        dist.all_to_all_single(recv_c, send_c, group=group)
        return recv_c.reshape(world, per_rank).tolist()

    def message_rounds(largest):
        mx = engine.counts_tensor([largest])
        dist.all_reduce(mx, op=dist.ReduceOp.MAX, group=group)
        return max(1, -(-int(mx.item()) // _MAX_ELEMS_PER_MESSAGE))

    filtered, sampled = False, None
    if prefilter == "off":
        # one all-to-all tells every rank what it will receive: build counts + per-piece probe counts
        flat = []
        for d in range(world):
            flat += [b_counts[d]] + [p_counts[c][d] for c in range(pieces)]
        rc = exchange_counts(flat, pieces + 1)
        b_recv = [int(r[0]) for r in rc]
        p_recv = [[int(rc[src][c + 1]) for src in range(world)] for c in range(pieces)]      # [piece][source]
        rounds = message_rounds(max([max(b_counts)] + [max(pc) for pc in p_counts]))
        bk_r = _exchange(dist, group, engine, bk_s, b_counts, b_recv, rounds)
        bv_r = _exchange(dist, group, engine, bv_s, b_counts, b_recv, rounds)
    else:
        # the build side travels first: its owners' filters decide what the probe side sends
        b_recv = [int(r[0]) for r in exchange_counts(list(b_counts), 1)]
        b_rounds = message_rounds(max(b_counts))
        bk_r = _exchange(dist, group, engine, bk_s, b_counts, b_recv, b_rounds)
        bv_r = _exchange(dist, group, engine, bv_s, b_counts, b_recv, b_rounds)
        mine = engine.bloom_export(bk_r, 48)
        filters = [engine.empty_like(mine) for _ in range(world)]
        dist.all_gather(filters, mine, group=group)
        filtered = True
        if prefilter == "auto":
            sampled = _sampled_survivors(dist, group, engine, world, probe_keys, filters)
            filtered = sampled < _prefilter_break_even(world)
        rounds = message_rounds(max(max(pc) for pc in p_counts))          # (survivors never outnumber the rows they come from)
        if not filtered:
            del filters, mine
            flat = []
            for d in range(world):
                flat += [p_counts[c][d] for c in range(pieces)]
            rc = exchange_counts(flat, pieces)
            p_recv = [[int(rc[src][c]) for src in range(world)] for c in range(pieces)]

    keep, works, recvs = [], [], []

    def put_on_the_wire(c, s_c):
        r_c = engine.empty(sum(p_recv[c]))
        if rounds <= 1:
            w = dist.all_to_all_single(r_c, s_c, output_split_sizes=p_recv[c], input_split_sizes=p_counts[c], group=group, async_op=True)
        else:                                    # very large pieces: fall back to blocking rounds for this piece
            r_c = _exchange(dist, group, engine, s_c, p_counts[c], p_recv[c], rounds)
            w = None
        keep.append(s_c); works.append(w); recvs.append(r_c)

    if filtered:
        # piece c is scattered and filtered while piece c-1 is on the wire; its survivor counts are exchanged just before it
        # leaves.  The owner's stream join opens once every piece was filtered (fj_bloom_prefilter and an open stream join
        # share the context's chunk pools); what it then appends is the small filtered remainder.
        p_recv = []
        for c in range(pieces):
            s_c = engine.owner_scatter(views[c], world, p_counts[c])
            off, kept = 0, []
            for d in range(world):
                kept.append(engine.bloom_prefilter(s_c[off: off + p_counts[c][d]], filters[d], 48))
                off += p_counts[c][d]
            p_counts[c] = [int(k.numel()) for k in kept]
            p_recv.append([int(r[0]) for r in exchange_counts(p_counts[c], 1)])
            put_on_the_wire(c, engine.cat(kept))
        del filters, mine
        np_total = sum(sum(p) for p in p_recv)
    try:
        if filtered:
            engine.stream_begin(bk_r, bv_r, np_total, pieces, 48)
            for c in range(pieces):
                if works[c] is not None:
                    works[c].wait()
                engine.stream_append(recvs[c])
        else:
            np_total = sum(sum(p) for p in p_recv)
            engine.stream_begin(bk_r, bv_r, np_total, pieces, 48)
            for c in range(pieces):
                put_on_the_wire(c, engine.owner_scatter(views[c], world, p_counts[c]))
                if c >= 1:
                    if works[c - 1] is not None:
                        works[c - 1].wait()
                    engine.stream_append(recvs[c - 1])
            if works[-1] is not None:
                works[-1].wait()
            engine.stream_append(recvs[-1])
        sent_rows = sum(sum(pc) for pc in p_counts)
        t2 = time.perf_counter()
        local_count = engine.stream_finish()     # (a skewed partition beyond the LDS tables: fj_stream_finish falls back by itself)
    except BaseException:
        _abort_stream(engine)                    # an error between begin and finish must not leave the context occupied
        raise
    tot = engine.counts_tensor([local_count])
    dist.all_reduce(tot, op=dist.ReduceOp.SUM, group=group)
    engine.synchronize()
    t3 = time.perf_counter()
    del keep
    if timings is not None:
        timings.update(split_s=t1 - t0, exchange_s=t2 - t1, join_s=t3 - t2, exchange_rounds=rounds, pieces=pieces,
                       local_build_rows=sum(b_recv), local_probe_rows=np_total, local_count=local_count,
                       prefilter=filtered, prefilter_mode=prefilter, prefilter_sampled_survivors=sampled, probe_rows_sent=sent_rows)
    return int(tot.item()), t3 - t0


class _CallbackTransport:
    """fj_dist_transport over torch.distributed (or an object with its functions): three blocking callbacks for the C++ driver
    (csrc/fj_dist.hip) - the transport of gloo jobs, of FJ_DIST_NATIVE=0 and of the test-suite; by default the driver talks to
    RCCL itself under the nccl backend.  memory: "device" - the driver's pointers are HIP device pointers, payload is staged
    through host tensors (gloo) or device tensors (nccl); "host" - plain host pointers (a stand-in engine)."""

    def __init__(self, dist, group, memory: str, device=None):
        import numpy as np
        import torch
        from . import _lib
        self.np, self.torch, self._lib, self.dist, self.group = np, torch, _lib, dist, group
        self.L = _lib.load() if memory == "device" else None
        self.memory = memory
        # tensors handed to the collectives: device tensors under nccl, host tensors otherwise
        self.tdev = device if (memory == "device" and device is not None and getattr(dist, "get_backend", lambda g: "gloo")(group) == "nccl") else None
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.error = None
        self.bytes_sent = 0
        # (the C side calls these for as long as the communicator lives: keep the ctypes trampolines referenced here, not only in the struct)
        self._cbs = (_lib.AllGatherFn(self._all_gather), _lib.AllReduceFn(self._all_reduce), _lib.AllToAllFn(self._all_to_all))
        self.struct = _lib.FjDistTransport(None, self.world, self.rank, *self._cbs)

    def _guard(self, fn):
        try:
            fn()
            return 0
        except BaseException as ex:              # noqa: BLE001  (an exception must not cross the C frames)
            self.error = ex
            return 1

    def _all_gather(self, _user, v, n, out):
        def run():
            t = self.torch
            mine = t.tensor([int(v[i]) for i in range(n)], dtype=t.int64, device=self.tdev)
            allv = t.zeros(n * self.world, dtype=t.int64, device=self.tdev)
            self.dist.all_gather_into_tensor(allv, mine, group=self.group)
            for i, x in enumerate(allv.tolist()):
                out[i] = x
        return self._guard(run)

    def _all_reduce(self, _user, v, n):
        def run():
            t = self.torch
            x = t.tensor([int(v[i]) for i in range(n)], dtype=t.int64, device=self.tdev)
            self.dist.all_reduce(x, op=self.dist.ReduceOp.SUM, group=self.group)
            for i, y in enumerate(x.tolist()):
                v[i] = y
        return self._guard(run)

    def _read(self, ptr, nbytes):
        t = self.torch
        if self.tdev is not None:
            buf = t.empty(nbytes, dtype=t.uint8, device=self.tdev)
            self._lib.check(self.L.fj_memcpy_d2d(buf.data_ptr(), ptr, nbytes))
            return buf
        buf = self.np.empty(nbytes, dtype=self.np.uint8)
        if self.memory == "device":
            self._lib.check(self.L.fj_memcpy_d2h(buf.ctypes.data, ptr, nbytes))
        else:
            ctypes.memmove(buf.ctypes.data, ptr, nbytes)
        return t.from_numpy(buf)

    def _write(self, ptr, t):
        if self.tdev is not None:
            self.torch.cuda.synchronize(self.tdev)
            self._lib.check(self.L.fj_memcpy_d2d(ptr, t.data_ptr(), t.numel()))
            return
        a = t.numpy()
        if self.memory == "device":
            self._lib.check(self.L.fj_memcpy_h2d(ptr, a.ctypes.data, a.size))
        else:
            ctypes.memmove(ptr, a.ctypes.data, a.size)

    def _all_to_all(self, _user, nparts, sp, sb, rp, rb):
        def run():
            dist, t = self.dist, self.torch
            peer = (lambda r: r) if self.group is None else (lambda r: dist.get_global_rank(self.group, r))
            ops, landed = [], []
            for p in range(nparts):
                for r in range(self.world):
                    i = p * self.world + r
                    ns, nr = int(sb[i]), int(rb[i])
                    if r == self.rank:           # (only the loop-back test hook routes a rank's own share through the transport)
                        if ns:
                            self._write(rp[i], self._read(sp[i], ns))
                        continue
                    if ns:
                        ops.append(dist.P2POp(dist.isend, self._read(sp[i], ns), peer(r), self.group))
                        self.bytes_sent += ns
                    if nr:
                        buf = t.empty(nr, dtype=t.uint8, device=self.tdev)
                        landed.append((rp[i], buf))
                        ops.append(dist.P2POp(dist.irecv, buf, peer(r), self.group))
            for w in (dist.batch_isend_irecv(ops) if ops else []):
                w.wait()
            for ptr, buf in landed:
                self._write(ptr, buf)
        return self._guard(run)


def _engine_ops_struct(ops, keep: list):
    """fj_dist_engine_ops for a stand-in engine object (tests): methods plan / alloc / release / pack_begin / pack_counts /
    pack_finish / open / append / finish / abort and the attribute chunk_bytes.  Exceptions become error returns."""
    from . import _lib
    state = {"err": b""}

    def guard(fn, fail=1):
        def wrapped(_user, *a):
            try:
                r = fn(*a)
                return 0 if r is None else r
            except BaseException as ex:          # noqa: BLE001
                state["err"] = repr(ex).encode()
                return fail
        return wrapped

    def pack_counts(used):
        for r, u in enumerate(ops.pack_counts()):
            used[r] = int(u)

    def finish(out):
        out[0] = int(ops.finish())

    world = ops.world
    cbs = dict(
        error=_lib.EngErrorFn(lambda _u: state["err"]),
        plan=_lib.EngPlanFn(guard(lambda nb_total, nranks: 0 if ops.plan(nb_total, nranks) else 1)),
        alloc=_lib.EngAllocFn(guard(ops.alloc, fail=None)),
        release=_lib.EngReleaseFn(guard(ops.release, fail=None)),
        pack_begin=_lib.EngPackBeginFn(guard(ops.pack_begin)),
        pack_counts=_lib.EngPackCountsFn(guard(pack_counts)),
        pack_finish=_lib.EngPackFinishFn(guard(lambda dk, dd: ops.pack_finish([dk[r] for r in range(world)], [dd[r] for r in range(world)]))),
        open=_lib.EngOpenFn(guard(ops.open)),
        append=_lib.EngAppendFn(guard(ops.append)),
        finish=_lib.EngFinishFn(guard(finish)),
        abort=_lib.EngAbortFn(guard(ops.abort, fail=None)),
    )
    # the optional sender-side precheck of a stand-in (all four methods or none)
    pre = [_lib.EngFilterRangeFn(), _lib.EngExportFn(), _lib.EngPackFilterFn(), _lib.EngSampleFn()]
    if all(hasattr(ops, m) for m in ("filter_range", "export_filters", "pack_filter", "sample")):
        def filter_range(nb_total, nranks, rank, first, count, total, each):
            first[0], count[0], total[0], each[0] = (int(x) for x in ops.filter_range(nb_total, nranks, rank))

        def pack_filter(filters, kept):
            kept[0] = int(ops.pack_filter(filters))

        def sample(rows, n, stride, filters, nb_total, nranks, kept):
            kept[0] = int(ops.sample(rows, n, stride, filters, nb_total, nranks))
        pre = [_lib.EngFilterRangeFn(guard(filter_range)), _lib.EngExportFn(guard(ops.export_filters)), _lib.EngPackFilterFn(guard(pack_filter)),
               _lib.EngSampleFn(guard(sample))]
    cbs["precheck"] = pre
    # the optional build-broadcast form of a stand-in (all seven methods or none)
    bc = [_lib.EngBcRegionFn(), _lib.EngBcSpanFn(), _lib.EngBcNpartsFn(), _lib.EngBcPackFn(), _lib.EngBcProbeFn(), _lib.EngBcJoinFn(), _lib.EngBcFinishFn()]
    if all(hasattr(ops, m) for m in ("bc_region_bytes", "bc_span", "bc_nparts", "bc_pack", "bc_probe", "bc_join", "bc_finish")):
        def bc_span(nb_total, nkeys, k_lo, k_hi, part, off, nbytes):
            off[0], nbytes[0] = (int(x) for x in ops.bc_span(nb_total, nkeys, k_lo, k_hi, part))

        def bc_nparts(nb_total, out):
            out[0] = int(ops.bc_nparts(nb_total))

        def bc_pack(rows, n, nb_total, region, pieces, bounds):
            for q, b in enumerate(ops.bc_pack(rows, n, nb_total, region, pieces)):
                bounds[q] = int(b)

        def bc_join(base, nsrc, off, nk, lo, hi):
            return ops.bc_join(base, [int(off[i]) for i in range(nsrc)], [int(nk[i]) for i in range(nsrc)], lo, hi)

        def bc_finish(out):
            out[0] = int(ops.bc_finish())
        bc = [_lib.EngBcRegionFn(guard(lambda nb_total, nkeys: int(ops.bc_region_bytes(nb_total, nkeys)), fail=0)), _lib.EngBcSpanFn(guard(bc_span)),
              _lib.EngBcNpartsFn(guard(bc_nparts)), _lib.EngBcPackFn(guard(bc_pack)), _lib.EngBcProbeFn(guard(ops.bc_probe)), _lib.EngBcJoinFn(guard(bc_join)),
              _lib.EngBcFinishFn(guard(bc_finish))]
    cbs["bcast"] = bc
    keep.append(cbs)
    return _lib.FjDistEngineOps(None, int(ops.chunk_bytes), cbs["error"], cbs["plan"], cbs["alloc"], cbs["release"], cbs["pack_begin"],
                                cbs["pack_counts"], cbs["pack_finish"], cbs["open"], cbs["append"], cbs["finish"], cbs["abort"], *pre, *bc)


FORM_AUTO, FORM_SHUFFLE, FORM_BROADCAST = 0, 1, 2      # include/flashjoin.h: FJ_DIST_FORM_*


def _driver_count(dist, group, engine, build_keys, probe_keys, pieces: int, timings: Optional[dict], transport, build_values=None, return_arrays=False,
                  prefilter_below: float = 0.0, prefilter_mode: str = "off", form: int = FORM_SHUFFLE):
    """The owner shuffle in chunk form through the ONE driver, csrc/fj_dist.hip (fj_dist_join): natively over RCCL under the nccl
    backend; over a callback transport (torch.distributed with host staging: gloo, a transport object) otherwise; with a stand-in
    engine's callbacks in the CPU test-suite.  build_values: a materialising join (the pairs stay with the owner; returned when
    return_arrays).  prefilter_below: the sender-side precheck in chunk form (include/flashjoin.h: fj_dist_join) - 0 never, >= 2
    always, else the survivor share of a sample below which it runs.  Collective; a failure on any rank raises on every rank."""
    from . import _lib
    L = _lib.load()
    t0 = time.perf_counter()
    cnt, local = ctypes.c_uint64(0), ctypes.c_uint64(0)
    dt = _lib.FjDistTimings()
    keep: list = []
    pairs = None
    standin = hasattr(engine, "dist_engine_ops")
    native = (not standin and transport is None and os.environ.get("FJ_DIST_NATIVE", "1") != "0" and dist.get_backend(group) == "nccl")
    if native:
        comm, own = engine.native_comm(dist, group), False
        form_label = "chunks (fj_dist_join_count over RCCL)"
    else:
        tr = _CallbackTransport(dist, group, "host" if standin else "device", None if standin else engine.device)
        ops = _engine_ops_struct(engine.dist_engine_ops(tr.world), keep) if standin else None
        comm = L.fj_dist_comm_from_transport(None if standin else engine.ctx, ctypes.byref(tr.struct), ctypes.byref(ops) if ops is not None else None)
        if not comm:
            raise RuntimeError(_lib.last_error())
        own = True
        form_label = "chunks (fj_dist_join_count over a callback transport)"
    try:
        _lib.check(L.fj_dist_comm_set_form(comm, int(form), float(_LINK_BYTES_PER_S)))
        if standin:
            rc = L.fj_dist_join(comm, build_keys.data_ptr(), None, build_keys.numel(), probe_keys.data_ptr(), probe_keys.numel(), pieces, 0,
                                float(prefilter_below), None, ctypes.byref(cnt), ctypes.byref(local), ctypes.byref(dt))
        else:
            tt = engine.torch
            bk, pk = engine._aligned(build_keys), engine._aligned(probe_keys)
            bv = engine._aligned(build_values) if build_values is not None else None
            with tt.cuda.device(engine.index):
                rc = L.fj_dist_join(comm, bk.data_ptr(), bv.data_ptr() if bv is not None else None, bk.numel(), pk.data_ptr(), pk.numel(), pieces,
                                    int(bv is not None), float(prefilter_below), tt.cuda.current_stream(engine.index).cuda_stream, ctypes.byref(cnt), ctypes.byref(local), ctypes.byref(dt))
                if rc == 0 and bv is not None and return_arrays:
                    pairs = engine.emit_pairs(int(local.value))
        if rc:
            msg = _lib.last_error()
            if not native and tr.error is not None:
                raise RuntimeError(f"{msg} ({tr.error!r})")
            raise RuntimeError(msg)
    finally:
        if own:
            L.fj_dist_comm_destroy(comm)
    sec = time.perf_counter() - t0
    if not standin:
        engine.api._last = dt.local
    if timings is not None and int(dt.form) == FORM_BROADCAST:
        timings.update(strategy="broadcast", shuffle_form="build broadcast (fj_dist_join_count: probe rows stay, dense 6-byte build runs to every peer)",
                       split_s=dt.split_ms * 1e-3, exchange_s=dt.exchange_ms * 1e-3, join_s=dt.join_ms * 1e-3, exchange_rounds=1, pieces=int(dt.pieces),
                       local_build_rows=int(dt.local_build_chunks), local_probe_rows=probe_keys.numel(), local_count=int(dt.local_count), prefilter=False, prefilter_mode="off",
                       prefilter_sampled_survivors=None, prefilter_below=0.0, probe_rows_sent=0, filter_bytes_received=0, wire_chunk_bytes=0,
                       wire_bytes_sent=int(dt.wire_bytes_sent))
    elif timings is not None:
        timings.update(strategy="shuffle", shuffle_form=form_label, split_s=dt.split_ms * 1e-3, exchange_s=dt.exchange_ms * 1e-3,
                       join_s=dt.join_ms * 1e-3, exchange_rounds=1, pieces=int(dt.pieces), local_build_rows=int(dt.local_build_chunks) * 256,
                       local_probe_rows=int(dt.local_probe_chunks) * 256, local_count=int(dt.local_count), prefilter=bool(dt.prefilter), prefilter_mode=prefilter_mode,
                       prefilter_sampled_survivors=(float(dt.prefilter_sampled) if dt.prefilter_sampled >= 0 else None), prefilter_below=float(prefilter_below),
                       probe_rows_sent=int(dt.probe_rows_kept) if (not standin or dt.prefilter) else probe_keys.numel(), filter_bytes_received=int(dt.filter_bytes),
                       rows_are_chunk_capacity=True,
                       wire_chunk_bytes=int(dt.wire_chunk_bytes), wire_bytes_sent=int(dt.sent_chunks) * (int(dt.wire_chunk_bytes) + 4))
    if pairs is not None:
        return int(cnt.value), sec, pairs[0], pairs[1]
    return int(cnt.value), sec


# ---- strategy 2: replicate the build side ------------------------------------------------------------------
# xGMI is a point-to-point mesh: a GPU has ONE link to each peer, so what bounds an exchange is the bytes per link.
# The owner shuffle puts (P*8 + B*16)/N bytes on every link (P, B = local probe / build rows); sending every rank's
# build KEYS to every peer puts B*8 there (B*16 with values).  For the probe-heavy joins this path is built for
# (P = 10 B) that is fewer bytes up to N = 12 -- 6x fewer at N = 2 -- and the probe side never moves: its partition
# passes run while the build keys are on the wire.  The price is local: every rank partitions all N*B build keys.
_LINK_BYTES_PER_S = 45e9          # one xGMI link, one direction, effective (153.6 GB/s bidirectional raw)
# local work, measured on one MI355X (profiles/r01_replicate_local.txt; c3 = 100M x 1B rows per GPU):
_PROBE_PASS_S_PER_ROW = 3.25e-12   # one probe-side partition pass (3.1-3.3 ms per 1B rows, profiles/r02_c3_kernel_stats.csv)
_BUILD_S_PER_ROW = 15e-12          # a build row's share of passes + table build, counting join (800M rows: 6.4 + 5.2 ms)
_SPLIT_S_PER_ROW = 2.7e-12         # shuffle: owner histogram + the un-overlapped first owner scatter
_JOIN_S_PER_ROW = 1.7e-12          # per-partition join, per probe row


def _npass(bits: int) -> int:
    return (1 if bits > 0 else 0) if bits <= 9 else (2 if bits <= 18 else -(-bits // 9))


def _plan_passes(nb: int) -> int:
    """Partition passes of the single-GPU plan for a build side of nb rows (csrc/fj_plan.hip make_plan)."""
    bits = 0
    if nb > 4096:
        bits = (-(-nb // 4096) - 1).bit_length()
    if (nb >> bits) > 3950:                       # one more radix bit where it costs no extra pass (FJ_PLAN_BUMP_KEYS)
        nb1 = 5 if bits == 0 else bits + 1
        if bits == 0 or _npass(nb1) == _npass(bits):
            bits = nb1
    if 0 < bits < 5:
        bits = 5
    return _npass(bits)


def strategy_costs(world: int, nb: int, np_: int, materialize: bool) -> dict:
    """Modelled seconds of one step for per-rank relation sizes nb x np_ under both strategies."""
    kb = 2 if materialize else 1
    # shuffle: exchange of 1/N of every relation per link; the probe exchange hides the splits and the first pass
    x_sh = (np_ * 8 + nb * 16) / world / _LINK_BYTES_PER_S
    t_shuffle = (_SPLIT_S_PER_ROW * np_ + x_sh + (_plan_passes(nb) - 1) * _PROBE_PASS_S_PER_ROW * np_
                 + _JOIN_S_PER_ROW * np_ + _BUILD_S_PER_ROW * nb * kb)
    # replicate: the build keys of one peer per link; the probe passes hide under it, the N-fold build work does not
    x_rep = nb * 8 * kb / _LINK_BYTES_PER_S
    t_replicate = (max(x_rep, _plan_passes(world * nb) * _PROBE_PASS_S_PER_ROW * np_)
                   + _BUILD_S_PER_ROW * world * nb * kb + _JOIN_S_PER_ROW * np_)
    return {"shuffle": t_shuffle, "replicate": t_replicate}


def form_model(world: int, nb: int, np_: int, link_bytes_per_s: Optional[float] = None) -> dict:
    """The C++ driver's cost model for a counting step of `world` ranks holding nb x np_ rows each (fj_dist_model: what
    FJ_DIST_FORM_AUTO decides with): modelled seconds in either form and the pick."""
    from . import _lib
    ts, tb = ctypes.c_double(0), ctypes.c_double(0)
    f = _lib.load().fj_dist_model(world, nb, np_, nb * world, np_ * world, 0, float(link_bytes_per_s or _LINK_BYTES_PER_S), ctypes.byref(ts), ctypes.byref(tb))
    return {"shuffle": ts.value, "broadcast": tb.value, "pick": "broadcast" if f == FORM_BROADCAST else "shuffle"}


_LINK_MEASURED = False


def set_link_rate(bytes_per_s: float, measured: bool = True) -> None:
    """Replace the built-in per-link rate of the cost model (FJ_DIST_STRATEGY=auto, the sender-side precheck's break-even)
    by a measured one (tools/xgmi_probe.py; bench.py does this once at N > 1)."""
    global _LINK_BYTES_PER_S, _LINK_MEASURED
    if bytes_per_s > 0:
        _LINK_BYTES_PER_S = float(bytes_per_s)
        _LINK_MEASURED = bool(measured)


def choose_strategy(world: int, nb: int, np_: int, materialize: bool) -> str:
    """What a multi-rank join runs as, for per-rank relation sizes nb x np_ (the maxima over the ranks):
      'auto'      (unset FJ_DIST_STRATEGY) counting joins: the C++ driver's per-link / per-rank cost model picks the owner shuffle or the
                  build broadcast for the step's sizes (csrc/fj_dist.hip: the same verdict on every rank); materialising joins shuffle;
      'shuffle'   always the owner shuffle (north_star's all-to-all of radix partitions);
      'broadcast' counting joins in the build-broadcast form (probe rows never move; csrc/fj_bcast.hip), materialising joins shuffle;
      'replicate' the build KEYS all-gathered unpartitioned (rounds 1-4's alternative; kept for comparison)."""
    forced = os.environ.get("FJ_DIST_STRATEGY", "auto")
    if forced in ("replicate", "shuffle", "broadcast", "auto"):
        return forced
    raise ValueError(f"FJ_DIST_STRATEGY={forced!r}: shuffle | broadcast | replicate | auto")


def self_check(dist, group, engine, small_inputs, expected_small: int, message_elems: int, transport=None) -> dict:
    """A few seconds before a multi-rank job's first timed step: (1) one exchange of per-peer VIEWS at the largest message size
    the step will use (`message_elems` int64 per peer, capped at the RCCL-safe size), every element a function of (source,
    destination, index), verified in full on arrival - the transport of the chunk-form shuffle, and the > 4 GiB defect the
    bounded rounds work around, on real ranks; (2) one small distributed_join against its closed-form count.  Every rank
    returns the same verdict: {"ok", "error", "failed_ranks", "message_bytes", "seconds"}."""
    import torch
    t0 = time.perf_counter()
    me, world = dist.get_rank(group), dist.get_world_size(group)
    n = int(max(1, min(message_elems, _MAX_ELEMS_PER_MESSAGE)))
    err = None
    try:
        send = engine.empty(n * world)
        dev = send.device
        idx = torch.arange(n, dtype=torch.int64, device=dev)
        for d in range(world):
            send[d * n: (d + 1) * n] = idx * 1000003 + (me * 64 + d) * 7919 + 12345
        pad = engine.empty(n * world + 1024)                        # receive through views with gaps between them, like the chunk regions
        stride = n + (1024 // max(1, world))
        stride -= stride % 2
        outs = [pad[q * stride: q * stride + n] for q in range(world)]
        ins = [send[d * n: (d + 1) * n] for d in range(world)]
        for w in _views_all_to_all(dist, group, ins, outs):
            w.wait()
        if os.environ.get("FJ_SELFCHECK_CORRUPT") and me == 0:      # test hook: what a transport that moves wrong data looks like
            outs[world - 1][n // 2] ^= 1
        for q in range(world):
            want = idx * 1000003 + (q * 64 + me) * 7919 + 12345
            bad = int((outs[q] != want).sum().item())
            if bad:
                err = f"exchange self-check: {bad} of {n} elements received from rank {q} are wrong ({n * 8} bytes per peer)"
                break
        del send, pad, outs, ins, idx
    except Exception as ex:                                        # noqa: BLE001
        err = f"exchange self-check raised {ex!r}"
    flag = engine.counts_tensor([1 if err else 0])
    dist.all_reduce(flag, op=dist.ReduceOp.SUM, group=group)
    nbad = int(flag.item())
    if nbad == 0:
        try:
            bk, bv, pk = small_inputs
            got = distributed_join(bk, bv, pk, group=group, engine=engine, transport=transport)[0]
            if int(got) != int(expected_small):
                err = f"join self-check: count {got} != closed form {expected_small}"
        except Exception as ex:                                    # noqa: BLE001  (raised on every rank, or agreed on inside)
            err = f"join self-check raised {ex!r}"
        flag = engine.counts_tensor([1 if err else 0])
        dist.all_reduce(flag, op=dist.ReduceOp.SUM, group=group)
        nbad = int(flag.item())
    # (3) the same join with the sender-side precheck forced on (what "auto" may choose in the timed steps: filters exported,
    #     all-gathered, probe pieces compacted).  A failure does not fail the check: the precheck is switched off for this process
    #     (FJ_DIST_PREFILTER=0, the ranks agree) and the verdict says so.
    precheck = None
    if nbad == 0 and os.environ.get("FJ_DIST_PREFILTER", "") != "0" and hasattr(engine, "shuffle_plan") and not hasattr(engine, "dist_engine_ops"):
        perr = None
        saved = os.environ.get("FJ_DIST_PREFILTER")
        os.environ["FJ_DIST_PREFILTER"] = "1"
        try:
            tt: dict = {}
            got = distributed_join(bk, bv, pk, group=group, engine=engine, transport=transport, timings=tt)[0]
            if int(got) != int(expected_small):
                perr = f"count {got} != closed form {expected_small}"
            precheck = {"ok": perr is None, "ran": bool(tt.get("prefilter")), "rows_sent": tt.get("probe_rows_sent"), "form": tt.get("shuffle_form")}
        except Exception as ex:                                    # noqa: BLE001
            perr = f"raised {ex!r}"
            precheck = {"ok": False}
        finally:
            if saved is None:
                os.environ.pop("FJ_DIST_PREFILTER", None)
            else:
                os.environ["FJ_DIST_PREFILTER"] = saved
        flag = engine.counts_tensor([1 if perr else 0])
        dist.all_reduce(flag, op=dist.ReduceOp.SUM, group=group)
        if int(flag.item()):
            os.environ["FJ_DIST_PREFILTER"] = "0"
            precheck = {"ok": False, "error": perr, "failed_ranks": int(flag.item()), "action": "FJ_DIST_PREFILTER=0 for the timed steps"}
    return {"ok": nbad == 0, "error": err, "failed_ranks": nbad, "message_bytes": n * 8, "seconds": round(time.perf_counter() - t0, 3), "precheck": precheck}


def _gather_rows(dist, group, engine, world, t, sizes: List[int], lo_frac=(0, 1), async_op=False):
    """All ranks' slice [n*a/b, n*(a+1)/b) of their tensor `t`, concatenated in rank order.  Returns (work, out, fix)
    where fix(out) compacts the result when the slices are not all the same length."""
    a, b = lo_frac
    lens = [n * (a + 1) // b - n * a // b for n in sizes]
    me = dist.get_rank(group)
    mine = t[sizes[me] * a // b: sizes[me] * (a + 1) // b]
    mx = max(lens)
    if min(lens) == mx:
        out = engine.empty(mx * world)
        w = dist.all_gather_into_tensor(out, mine.contiguous(), group=group, async_op=async_op)
        return w, out, None
    pad = engine.empty(mx)
    pad[: lens[me]] = mine
    out = engine.empty(mx * world)
    w = dist.all_gather_into_tensor(out, pad, group=group, async_op=async_op)

    def fix(o):
        import torch
        return torch.cat([o[r * mx: r * mx + lens[r]] for r in range(world)])
    return w, out, fix


def _replicated_join(dist, group, engine, world, build_keys, build_values, probe_keys, sizes_b: List[int], materialize: bool,
                     bloom: bool, return_arrays: bool, pieces: int, timings: Optional[dict]):
    """Every rank joins ITS probe rows against ALL build rows: build keys (and values when materialising) are
    all-gathered, probe rows never leave their GPU.  The global count is the sum of the local counts; materialised
    pairs stay on the rank that holds the probe row."""
    t0 = time.perf_counter()
    nb_total = sum(sizes_b)
    need = max(1, -(-max(sizes_b) // _MAX_ELEMS_PER_MESSAGE))        # keep every rank's contribution to one collective <= 1 GiB
    if materialize or not hasattr(engine, "stream_open"):
        import torch

        def gather_all(t):
            parts = []
            for c in range(need):
                _, o, fix = _gather_rows(dist, group, engine, world, t, sizes_b, (c, need))
                parts.append(fix(o) if fix else o)
            if need == 1:
                return parts[0]
            # piece-major -> rank-major, so that the first occurrence of a duplicate key is the one of the lowest rank
            lens = [[n * (c + 1) // need - n * c // need for n in sizes_b] for c in range(need)]
            offs = [[sum(l[:r]) for r in range(world)] for l in lens]
            return torch.cat([parts[c][offs[c][r]: offs[c][r] + lens[c][r]] for r in range(world) for c in range(need)])
        bk_all = gather_all(build_keys)
        bv_all = gather_all(build_values)
        engine.synchronize()
        t1 = time.perf_counter()
        res = engine.local_join(bk_all, bv_all, probe_keys, materialize, bloom, 64, return_arrays)
        local_count = int(res[0])
    else:
        # counting: keys only, in `pieces` asynchronous all-gathers; the probe side is partitioned meanwhile
        pieces = max(pieces, need)
        if nb_total <= 8192 or min(sizes_b) < pieces:        # (a zero-pass build side must arrive as one piece)
            pieces = 1
        gathers = [_gather_rows(dist, group, engine, world, build_keys, sizes_b, (c, pieces), async_op=True) for c in range(pieces)]
        try:
            engine.stream_open(nb_total, pieces, probe_keys.numel(), 1, 64)
            engine.stream_append(probe_keys)
            engine.stream_advance_probe()
            keep = []
            for w, out, fix in gathers:
                if w is not None:
                    w.wait()
                if fix: out = fix(out)
                keep.append(out)
                engine.stream_append_build(out)
            t1 = time.perf_counter()
            local_count = engine.stream_finish() # (a skewed partition beyond the LDS tables: fj_stream_finish falls back by itself)
        except BaseException:
            _abort_stream(engine)
            raise
        res = None
        del keep
    tot = engine.counts_tensor([local_count])
    dist.all_reduce(tot, op=dist.ReduceOp.SUM, group=group)
    engine.synchronize()
    t2 = time.perf_counter()
    if timings is not None:
        timings.update(strategy="replicate", split_s=0.0, exchange_s=t1 - t0, join_s=t2 - t1, exchange_rounds=1, pieces=pieces,
                       local_build_rows=nb_total, local_probe_rows=probe_keys.numel(), local_count=local_count)
    out = (int(tot.item()), t2 - t0)
    if materialize and return_arrays:
        return out + (res[2], res[3])
    return out


def distributed_join(build_keys, build_values, probe_keys, *, materialize: bool = False, bloom: bool = False,
                     group=None, engine=None, return_arrays: bool = False, timings: Optional[dict] = None, transport=None, join_id=None):
    """Join relations whose rows are block-distributed over the ranks of `group`.

    Every rank passes its LOCAL rows (int64 tensors on its GPU) and gets back
    `(global_match_count, seconds)`; with `materialize and return_arrays` also the pairs this
    rank owns.  `seconds` is this rank's wall time for the whole step (split + exchange + join).
    `transport`: an object with torch.distributed's collective functions (default: torch.distributed itself) - the
    self-tests that run several ranks on one GPU pass one that stages device tensors through the host for gloo.
    `join_id` (hashable, the same on every rank): names a repeated join for the sender-side precheck's memory of what it sampled.
    """
    import torch.distributed as _td
    dist = transport if transport is not None else _td
    if engine is None:
        engine = HipEngine()
    if hasattr(engine, "normalize"):             # int64/uint64, contiguous, 16-byte aligned: what the C ABI's pointers must be
        build_keys, build_values, probe_keys = engine.normalize(build_keys, build_values, probe_keys)
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1 and os.environ.get("FJ_FORCE_EXCHANGE") and not dist.is_initialized():
        raise RuntimeError("FJ_FORCE_EXCHANGE needs an initialised process group")
    t0 = time.perf_counter()
    if world == 1 and not os.environ.get("FJ_FORCE_EXCHANGE"):      # FJ_FORCE_EXCHANGE: run the full protocol on one rank (tests)
        res = engine.local_join(build_keys, build_values, probe_keys, materialize, bloom, 64, return_arrays)
        if timings is not None:
            timings.update(split_s=0.0, exchange_s=0.0, join_s=time.perf_counter() - t0)
        return res

    pieces = int(os.environ.get("FJ_DIST_PIECES", "4"))
    # relation sizes of every rank: one tiny all-gather decides the strategy identically everywhere
    # (third word: this rank's per-link rate in kB/s - every rank models with rank 0's, so that hosts that called set_link_rate with
    #  per-rank measurements still take the same decisions and the same fallback paths)
    mine = engine.counts_tensor([build_keys.numel(), probe_keys.numel(), int(_LINK_BYTES_PER_S / 1e3)])
    allsz = engine.counts_tensor([0] * (3 * world))
    dist.all_gather_into_tensor(allsz, mine, group=group)
    allsz = allsz.reshape(world, 3).tolist()
    link0 = float(allsz[0][2]) * 1e3
    sizes_b = [int(x[0]) for x in allsz]
    strategy = choose_strategy(world, max(sizes_b), max(int(x[1]) for x in allsz), materialize)
    if strategy == "replicate":
        return _replicated_join(dist, group, engine, world, build_keys, build_values, probe_keys, sizes_b, materialize, bloom,
                                return_arrays, int(os.environ.get("FJ_REPLICATE_PIECES", "4")), timings)
    if timings is not None:
        timings["strategy"] = "shuffle"
    form = {"auto": FORM_AUTO, "broadcast": FORM_BROADCAST}.get(strategy, FORM_SHUFFLE)
    if not materialize and form != FORM_SHUFFLE:
        # counting joins: the driver may take the build-broadcast form (a stand-in engine: when it has that form's callbacks)
        can = (hasattr(engine, "bcast_plan") and engine.bcast_plan(sum(sizes_b)) is not None) if not hasattr(engine, "dist_engine_ops") else getattr(engine, "has_bcast", False)
        if not can or world > 16:
            if strategy == "broadcast" and os.environ.get("FJ_DIST_NO_FALLBACK"):
                raise RuntimeError("FJ_DIST_STRATEGY=broadcast: this join cannot take the build-broadcast form (engine, > 16 ranks, or a total build side without a partitioned plan)")
            form = FORM_SHUFFLE
        elif form == FORM_BROADCAST or not hasattr(engine, "shuffle_plan") or engine.shuffle_plan(sum(sizes_b), world) is None or pieces <= 1:
            # (forced, or the chunk form of the shuffle is not available for these sizes: nothing for the driver to choose between)
            try:
                tt = timings if timings is not None else {}
                return _driver_count(dist, group, engine, build_keys, probe_keys, max(1, pieces), tt, transport, form=FORM_BROADCAST)
            except RuntimeError as ex:
                if os.environ.get("FJ_DIST_NO_FALLBACK"):
                    raise
                _abort_stream(engine)
                if timings is not None:
                    timings["broadcast_form_error"] = str(ex)
                form = FORM_SHUFFLE
    if not materialize and pieces > 1 and hasattr(engine, "stream_begin"):
        standin = hasattr(engine, "dist_engine_ops")
        mode = "off" if not hasattr(engine, "bloom_export") else _prefilter_mode(bloom) if standin else _chunk_prefilter_mode(bloom, world)
        nb_total, np_global = sum(sizes_b), sum(int(x[1]) for x in allsz)
        # the chunk form (the first radix pass of the global plan is the owner split) serves every counting shuffle whose
        # global plan has two or more passes, the sender-side precheck included (per-partition filters: fj_dist_join's
        # prefilter_below); the owner-scatter form below serves the small ones (and a stand-in engine's precheck)
        if ((mode == "off" or not standin or getattr(engine, "chunk_precheck", False)) and os.environ.get("FJ_DIST_CHUNK_SHUFFLE", "1") != "0" and hasattr(engine, "shuffle_plan")
                and engine.shuffle_plan(nb_total, world) is not None):      # (ranks with few or no rows: the driver sends everything as one piece)
            # ONE driver for every transport: csrc/fj_dist.hip (fj_dist_join_count) - over RCCL under the nccl backend, over
            # callbacks into torch.distributed under gloo / a transport object, with a stand-in engine in the CPU tests.  A step
            # that fails on any rank fails on every rank (the driver agrees on it), so all of them fall back together to the
            # owner-scatter form, whose segments are sized from an owner histogram: the answer to heavily skewed keys (an owner
            # that receives far more than 1.5x its share overflows the chunk form's pools).
            below, memo_key, decision = _precheck_threshold(mode, world, nb_total, np_global, link0, join_id)
            for attempt in ((below, mode), (0.0, "off")) if below > 0 else ((0.0, mode),):     # (a failed step with the precheck is retried without it)
                try:
                    tt = timings if timings is not None else {}
                    # (form AUTO: the driver weighs the build broadcast against the shuffle for these sizes; a precheck asks for the shuffle)
                    res = _driver_count(dist, group, engine, build_keys, probe_keys, pieces, tt, transport, prefilter_below=attempt[0], prefilter_mode=attempt[1],
                                        form=form if attempt[0] == 0.0 else FORM_SHUFFLE)
                    _precheck_remember(memo_key, tt, decision)
                    return res
                except RuntimeError as ex:
                    if os.environ.get("FJ_DIST_NO_FALLBACK"):
                        raise
                    _abort_stream(engine)
                    if timings is not None:
                        timings["chunk_form_error"] = str(ex)
                    form = FORM_SHUFFLE                      # (a broadcast step that failed - a skewed partition - is retried as the shuffle)
                    if attempt[0] > 0 and memo_key is not None:
                        # a step that failed WITH the precheck (an owner of hot probe keys overflows pools sized from the mean before any
                        # probe piece is seen) must not be attempted again on every later step of the same shape: remembered as declined,
                        # re-examined with the next resample like any other verdict (all ranks fail together: the memos stay identical)
                        _PRECHECK_MEMO[memo_key] = [1, 2.0]
        if timings is not None:
            timings["shuffle_form"] = "owner-scatter"
        if not standin and hasattr(engine, "bloom_export"):
            mode = _prefilter_mode(bloom)               # (the owner-scatter form's own rule and break-even)
        return _pipelined_count(dist, group, engine, world, build_keys, build_values, probe_keys, pieces, timings, prefilter=mode)

    # materialising joins: the chunk form too (the build rows travel with their values, the pairs stay with the owner: SURVEY 8(e)) -
    # through the same driver, sender-side precheck included; duplicate build keys and small build sides take the owner-scatter form below
    if (materialize and hasattr(engine, "emit_pairs") and os.environ.get("FJ_DIST_CHUNK_SHUFFLE", "1") != "0"
            and engine.shuffle_plan(sum(sizes_b), world) is not None):
        try:
            mode = _chunk_prefilter_mode(bloom, world)
            below, memo_key, decision = _precheck_threshold(mode, world, sum(sizes_b), sum(int(x[1]) for x in allsz), link0, join_id)
            tt = timings if timings is not None else {}
            res = _driver_count(dist, group, engine, build_keys, probe_keys, pieces, tt, transport, build_values=build_values, return_arrays=return_arrays,
                                prefilter_below=below, prefilter_mode=mode)
            _precheck_remember(memo_key, tt, decision)
            return res
        except RuntimeError as ex:
            if os.environ.get("FJ_DIST_NO_FALLBACK"):
                raise
            _abort_stream(engine)
            if timings is not None:
                timings["chunk_form_error"] = str(ex)
    if timings is not None:
        timings["shuffle_form"] = "owner-scatter"

    # 1. split by owner
    bk_s, bv_s, b_counts = engine.owner_split(build_keys, build_values, world)
    pk_s, _, p_counts = engine.owner_split(probe_keys, None, world)
    t1 = time.perf_counter()

    # 2. counts, then payload: one all-to-all per array
    def counts_to_owners(rows):
        send_c = engine.counts_tensor(rows)
        recv_c = engine.counts_tensor([0] * len(rows))
        dist.all_to_all_single(recv_c, send_c, group=group)
        return recv_c.reshape(world, len(rows) // world).tolist()

    def message_rounds(largest):                 # the largest single message anywhere in the group decides (same value on every rank)
        mx = engine.counts_tensor([largest])
        dist.all_reduce(mx, op=dist.ReduceOp.MAX, group=group)
        return max(1, -(-int(mx.item()) // _MAX_ELEMS_PER_MESSAGE))

    mode = _prefilter_mode(bloom) if hasattr(engine, "bloom_export") else "off"
    filtered, sampled, rows_before = False, None, sum(p_counts)
    if mode == "off":
        flat = []
        for d in range(world):
            flat += [b_counts[d], p_counts[d]]
        rc = counts_to_owners(flat)
        b_recv = [int(x[0]) for x in rc]
        p_recv = [int(x[1]) for x in rc]
        rounds = message_rounds(max(b_counts + p_counts))
        bk_r = _exchange(dist, group, engine, bk_s, b_counts, b_recv, rounds)
        bv_r = _exchange(dist, group, engine, bv_s, b_counts, b_recv, rounds)
    else:
        # sender-side precheck (see _pipelined_count): build side first, the owners' filters come back, survivors travel
        b_recv = [int(x[0]) for x in counts_to_owners(list(b_counts))]
        rounds = message_rounds(max(b_counts))
        bk_r = _exchange(dist, group, engine, bk_s, b_counts, b_recv, rounds)
        bv_r = _exchange(dist, group, engine, bv_s, b_counts, b_recv, rounds)
        mine = engine.bloom_export(bk_r, 48)
        filters = [engine.empty_like(mine) for _ in range(world)]
        dist.all_gather(filters, mine, group=group)
        filtered = True
        if mode == "auto":
            sampled = _sampled_survivors(dist, group, engine, world, probe_keys, filters)
            filtered = sampled < _prefilter_break_even(world)
        if filtered:
            off, kept = 0, []
            for d in range(world):
                kept.append(engine.bloom_prefilter(pk_s[off: off + p_counts[d]], filters[d], 48))
                off += p_counts[d]
            p_counts = [int(k.numel()) for k in kept]
            pk_s = engine.cat(kept)
        del filters, mine
        p_recv = [int(x[0]) for x in counts_to_owners(list(p_counts))]
        rounds = message_rounds(max(p_counts))
    pk_r = _exchange(dist, group, engine, pk_s, p_counts, p_recv, rounds)
    engine.synchronize()
    t2 = time.perf_counter()

    # 3. join what this rank owns
    res = engine.local_join(bk_r, bv_r, pk_r, materialize, bloom, 48, return_arrays)
    local_count = int(res[0])

    # 4. global count
    tot = engine.counts_tensor([local_count])
    dist.all_reduce(tot, op=dist.ReduceOp.SUM, group=group)
    engine.synchronize()
    t3 = time.perf_counter()
    if timings is not None:
        timings.update(split_s=t1 - t0, exchange_s=t2 - t1, join_s=t3 - t2, exchange_rounds=rounds,
                       local_build_rows=sum(b_recv), local_probe_rows=sum(p_recv), local_count=local_count,
                       prefilter=filtered, prefilter_mode=mode, prefilter_sampled_survivors=sampled,
                       probe_rows_sent=sum(p_counts) if filtered else rows_before)
    out = (int(tot.item()), t3 - t0)
    if materialize and return_arrays:
        return out + (res[2], res[3])
    return out
