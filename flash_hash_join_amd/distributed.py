"""Multi-GPU radix join: one process per GPU over RCCL/xGMI.

The reference is single-process (hash_join.cpp:318); this is new design.  What it exploits is that radix
partitions are independent join units (hash_join.cpp:340-356, :515-525).  A join of relations whose rows are block-distributed
over the ranks of a group takes the first of three FORMS that applies and succeeds - a ladder, the same on every rank:

1. BUILD BROADCAST (counting joins; csrc/fj_bcast.hip behind fj_dist_join): the probe rows never move.  Every rank partitions
   its build rows by the plan for the TOTAL build side, packs them densely (6 bytes per key) and sends them to every peer; every
   rank joins its own probe partitions against the runs of all ranks.  0.75 GB per link and step at BASELINE configs[4], whatever
   the number of ranks.  Taken when the C++ driver's cost model (fj_dist_model: bytes per link over the link rate against the
   kernel time per rank) puts it ahead of the shuffle - it does for probe-heavy joins such as configs[4] - and the replicas fit.
2. OWNER SHUFFLE in chunk form (north_star's all-to-all of radix partitions; counting and materialising joins whose global plan
   has two or more passes): the first radix pass of the plan for the total build side is the owner split - bucket b of its 256 /
   512 buckets belongs to rank (b * world) >> log2(buckets); a sender runs that pass and rewrites its output for the wire (dense
   256-key chunks, 7 bytes per key), the owner starts at the plan's second pass.  With the sender-side precheck (the owners'
   per-partition Bloom filters, all-gathered in front of the probe exchange) where a per-link model says it pays.
   Both forms run inside ONE C++ driver, csrc/fj_dist.hip (fj_dist_join): over RCCL under the nccl backend, over three callbacks
   into torch.distributed otherwise (_CallbackTransport: gloo, a transport object), with a stand-in for the rank's own work in the
   CPU test-suite.  A step that fails on one rank fails on every rank (the driver agrees on it), so all ranks move down together.
3. OWNER SCATTER (the plain last resort, driven from here: small build sides and any step the forms above failed on): every rank splits its rows by owner GPU = (top 16 hash bits * world) >> 16
   (fj_owner_split), ONE all-to-all per relation moves each segment to its owner (torch.distributed all_to_all_single), each rank
   joins what it owns with the single-GPU radix join (hash_top_bits = 48), one all-reduce adds the counts.  No precheck of its own.

A build key that occurs more than once - on one rank or on several - yields ONE pair per matching probe row in every form, with the
value of one of its copies (a single GPU emits the first occurrence's, hash_join.cpp:125; across GPUs there is no global row order
to define a first one).

FJ_DIST_STRATEGY = auto (default) | broadcast | shuffle | scatter pins the first rung; a rung that failed for a join shape is
remembered and skipped for the next 32 steps of that shape (_FORM_MEMO).  `engine` abstracts the per-rank primitives so that the
protocol can be exercised on CPU (gloo) with a stand-in engine; the default engine is the HIP one and has no CPU fallback.
Python reads six environment variables here: FJ_DIST_STRATEGY, FJ_DIST_PIECES, FJ_DIST_PREFILTER (0 | 1 | auto), FJ_DIST_NATIVE
(0: the driver over torch.distributed callbacks instead of RCCL directly), FJ_DIST_NO_FALLBACK, and FJ_LIB_VARIANT (_lib.py).
"""
from __future__ import annotations

import ctypes
import os
import time
from typing import List, Optional


class HipEngine:
    """Per-rank primitives on an MI355X through the C ABI (include/flashjoin.h).  (The building blocks behind the ABI - stream
    joins, the pieces of the driver - are reached through lab.LabEngine and the lab build of the library.)"""

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

    def counts_tensor(self, counts: List[int]):
        return self.torch.tensor(counts, dtype=self.torch.int64, device=self.device)

    def free_bytes(self) -> int:
        return int(self.torch.cuda.mem_get_info(self.index)[0])

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

    @staticmethod
    def _aligned(t):
        """The C ABI wants contiguous, 16-byte aligned pieces (a view into a larger tensor may be neither)."""
        if not t.is_contiguous() or t.data_ptr() % 16:
            t = t.contiguous().clone() if t.data_ptr() % 16 else t.contiguous()
        return t

    def shuffle_plan(self, nb_total: int, world: int) -> Optional[int]:
        """log2 of the first-pass fan-out of the plan for a build side of nb_total rows in all, or None when the chunk form
        does not apply (a one-pass plan: build sides under ~2M rows)."""
        f0, npass = ctypes.c_int(0), ctypes.c_int(0)
        if self.L.fj_shuffle_plan(nb_total, world, ctypes.byref(f0), ctypes.byref(npass)):
            return None
        return int(f0.value)

    def bcast_plan(self, nb_total: int):
        """(radix bits, final partitions, bytes of the high-word plane per key) of the plan for a total build side, or None when
        that plan has no pass (such joins take the owner-scatter form)."""
        b, n, m = ctypes.c_int(0), ctypes.c_uint32(0), ctypes.c_int(0)
        if self.L.fj_bcast_plan(nb_total, ctypes.byref(b), ctypes.byref(n), ctypes.byref(m)):
            return None
        return b.value, n.value, m.value

    def emit_pairs(self, n: int):
        """The pairs of the materialising join that was just counted on this context (fj_emit_pairs): two int64 tensors of n rows."""
        ok, ov = self.empty(max(n, 2)), self.empty(max(n, 2))
        t = self._lib.FjTimings()
        self._lib.check(self.L.fj_emit_pairs(self.ctx, ok.data_ptr(), ov.data_ptr(), n, self.torch.cuda.current_stream(self.index).cuda_stream, ctypes.byref(t)))
        return ok[:n], ov[:n]

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
        """Error recovery: drop a stream join that a failed step left open, so that the context serves other joins again."""
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



_LINK_BYTES_PER_S = 45e9          # one xGMI link, one direction, effective (153.6 GB/s bidirectional raw): the built-in guess; set_link_rate replaces it
_LINK_MEASURED = False


def set_link_rate(bytes_per_s: float, measured: bool = True) -> None:
    """Replace the built-in per-link rate of the cost models (the form choice, the sender-side precheck's break-even) by a measured
    one (tools/xgmi_probe.py; bench.py does this once at N > 1)."""
    global _LINK_BYTES_PER_S, _LINK_MEASURED
    if bytes_per_s > 0:
        _LINK_BYTES_PER_S = float(bytes_per_s)
        _LINK_MEASURED = bool(measured)


def form_model(world: int, nb: int, np_: int, link_bytes_per_s: Optional[float] = None, nb_total: Optional[int] = None, np_global: Optional[int] = None,
               materialize: bool = False) -> dict:
    """The C++ driver's cost model for a step of `world` ranks holding at most nb x np_ rows each (fj_dist_model): modelled seconds in
    either form and the pick.  materialize: the same arithmetic with the materialising kernels' constants, measured on one MI355X as
    one rank of 1 / 2 / 8 (profiles/r06_bcast_mat_one_rank.txt, tools/bcast_one_gpu.py ... 1): the regions carry the values (14 bytes
    per build row, 16 under 16-bit plans), the pack takes 22.9 ps per build row, the step is counted by the counting step's kernel
    (2.0 ps per build key of all ranks + 2.2 ps per local probe key), writing the pairs (behind the step: nothing overlaps it) costs
    5.5 ps + 7.45 ps; the shuffle ships 15 bytes per build row and its owner writes the pairs of what it received (~4 ms per 1.25B
    probe rows at 50 % hits)."""
    from . import _lib
    nb_total = nb_total if nb_total is not None else nb * world
    np_global = np_global if np_global is not None else np_ * world
    rate = float(link_bytes_per_s or _LINK_BYTES_PER_S)
    ts, tb = ctypes.c_double(0), ctypes.c_double(0)
    f = _lib.load().fj_dist_model(world, nb, np_, nb_total, np_global, 0, rate, ctypes.byref(ts), ctypes.byref(tb))
    t_s, t_b = ts.value, tb.value
    if materialize:
        bits = max(5, (max(1, -(-nb_total // 4096)) - 1).bit_length()) if nb_total > 4096 else 5
        region = nb * (14 if bits >= 16 else 16) + 4 * ((1 << bits) + 1) + 64
        pack, passes = nb * 22.9e-12, np_ * (6.34e-12 + 0.27e-12 * max(0, min(bits, 18) - 16))
        count, emit = nb_total * 2.0e-12 + np_ * 2.2e-12, nb_total * 5.5e-12 + np_ * 7.45e-12
        wire = region / rate if world > 1 else 0.0
        t_b = max(wire + pack + count / 4.0, pack + passes + count) + emit
        t_s += ((15.0 - 7.02) * nb_total / (world * world) / rate if world > 1 else 0.0) + np_ * 3.2e-12
        f = FORM_BROADCAST if t_b < t_s else FORM_SHUFFLE
    return {"shuffle": t_s, "broadcast": t_b, "pick": "broadcast" if f == FORM_BROADCAST else "shuffle"}


# the precheck in chunk form (fj_dist_join(prefilter_below), csrc/fj_pack.hip: fj_part_filter_inplace): what a config-5 shard costs
# per local probe row on ONE MI355X through the driver (profiles/r04_prefilter_one_rank.txt: 15.0-16.3 ms without it; with it the
# same at 8 % survivors, +4.7-5.8 ms at 52 %), as  off = FIXED + REST,  on(f) = FIXED + FILTER + f x REST
_CHUNK_FIXED_S_PER_ROW = 6.0e-12        # first pass of the global plan over the probe rows + the build side's and the host's share of the step
_CHUNK_FILTER_S_PER_ROW = 7.4e-12       # the precheck at the 8-rank plan (512 partitions = 2 MiB of filters per first-pass bucket: 2.3 ms per
                                        # 312M rows, profiles/r04_precheck_probe.txt; 4.9-6.6 ps at the 1-rank plan's 128 partitions per bucket)
_CHUNK_REST_S_PER_ROW = 6.1e-12         # copy into the wire format + the owner's second pass, lists and join: scale with what survives
_CHUNK_WIRE_BYTES_PER_ROW = 7.02
PREFILTER_BELOW_OVERRIDE: Optional[float] = None


def _chunk_prefilter_break_even(world: int, nb_total: int, np_local: int, link_bytes_per_s: Optional[float] = None) -> float:
    """Survivor fraction below which the precheck of the chunk form pays (PREFILTER_BELOW_OVERRIDE, a module variable, pins it).  Per local
    probe row a step costs max(wire, kernels) - the model of tools/scale_model.py: wire = 7.02 B x (f x probe rows + build rows) /
    (world x link rate) on each of the links that work in parallel, plus - with the precheck - the filters (1.07 bytes per build
    key to every rank: ~nb_total / world bytes per link); kernels as above; ~0.5 ms of latency before the first probe piece can be
    checked.  It never pays where the kernels bound the step (one rank; links faster than ~12 ps per row); on wire-bound steps it
    does: below ~75 % survivors at 8 GPUs and 45 GB/s per link (~55 % at 55 GB/s, ~15 % at 65), below ~85 % at 2-4 GPUs."""
    if PREFILTER_BELOW_OVERRIDE is not None:                                 # (tests and measurements pin the threshold)
        return float(PREFILTER_BELOW_OVERRIDE)
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


def _chunk_prefilter_mode(bloom: bool, world: int, override: Optional[str] = None) -> str:
    """The chunk form's precheck: FJ_DIST_PREFILTER=1 / 0 / auto decides; unset it is "auto" for every join across more than one
    rank (the *_bloom functions and the plain ones alike: results are identical, and a step whose links are the bottleneck is
    shorter by what the owners' filters keep off them - the model above prices it with the measured link rate when bench.py or the
    host called set_link_rate) and for the *_bloom functions on one rank (where the model declines)."""
    if override in ("off", "on", "auto"):
        return override
    env = os.environ.get("FJ_DIST_PREFILTER", "")
    if env in ("0", "1", "auto"):
        return {"0": "off", "1": "on", "auto": "auto"}[env]
    return "auto" if (bloom or world > 1) else "off"


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
    return _lib.FjDistEngineOps(ctypes.sizeof(_lib.FjDistEngineOps), None, int(ops.chunk_bytes), cbs["error"], cbs["plan"], cbs["alloc"], cbs["release"], cbs["pack_begin"],
                                cbs["pack_counts"], cbs["pack_finish"], cbs["open"], cbs["append"], cbs["finish"], cbs["abort"], *pre, *bc)



FORM_AUTO, FORM_SHUFFLE, FORM_BROADCAST = 0, 1, 2      # include/flashjoin.h: FJ_DIST_FORM_*


def _driver_join(dist, group, engine, build_keys, probe_keys, pieces: int, timings: Optional[dict], transport, build_values=None, return_arrays=False,
                 prefilter_below: float = 0.0, prefilter_mode: str = "off", form: int = FORM_SHUFFLE):
    """One step through the ONE driver, csrc/fj_dist.hip (fj_dist_join), in the given form: natively over RCCL under the nccl
    backend; over a callback transport (torch.distributed with host staging: gloo, a transport object) otherwise; with a stand-in
    engine's callbacks in the CPU test-suite.  build_values: a materialising join (the pairs stay with the owner; returned when
    return_arrays).  prefilter_below: the sender-side precheck of the shuffle - 0 never, >= 2 always, else the survivor share of a
    sample below which it runs.  Collective; a failure on any rank raises on every rank."""
    from . import _lib
    L = _lib.load()
    t0 = time.perf_counter()
    cnt, local = ctypes.c_uint64(0), ctypes.c_uint64(0)
    dt = _lib.FjDistTimings()
    dt.struct_size = ctypes.sizeof(_lib.FjDistTimings)
    keep: list = []
    pairs = None
    standin = hasattr(engine, "dist_engine_ops")
    native = (not standin and transport is None and os.environ.get("FJ_DIST_NATIVE", "1") != "0" and dist.get_backend(group) == "nccl")
    if native:
        comm, own = engine.native_comm(dist, group), False
        via = "RCCL"
    else:
        tr = _CallbackTransport(dist, group, "host" if standin else "device", None if standin else engine.device)
        ops = _engine_ops_struct(engine.dist_engine_ops(tr.world), keep) if standin else None
        comm = L.fj_dist_comm_from_transport(None if standin else engine.ctx, ctypes.byref(tr.struct), ctypes.byref(ops) if ops is not None else None)
        if not comm:
            raise RuntimeError(_lib.last_error())
        own = True
        via = "a callback transport"
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
    if timings is not None:
        # the CU reserve of the step (fj_dist_timings.reserve_*): how many CUs the passes left to the transport's kernels, how that came
        # about, and the measurements behind a choice (every rank's probe-side pass time of one step with and one without, added up)
        timings["cu_reserve"] = {"cus": int(dt.reserve_cus), "how": ("none applies", "pinned (FJ_DIST_RESERVE_CUS)", "measuring step", "chosen from the measurements")[int(dt.reserve_how) & 3],
                                 "pass_ms_with_32_reserved_all_ranks": round(float(dt.reserve_with_ms), 3), "pass_ms_with_none_all_ranks": round(float(dt.reserve_without_ms), 3)}
    if timings is not None and int(dt.form) == FORM_BROADCAST:
        timings.update(strategy="broadcast", shuffle_form=f"build broadcast (fj_dist_join over {via}: probe rows stay, dense 6-byte build runs to every peer)",
                       split_s=dt.split_ms * 1e-3, exchange_s=dt.exchange_ms * 1e-3, join_s=dt.join_ms * 1e-3, exchange_rounds=1, pieces=int(dt.pieces),
                       local_build_rows=int(dt.local_build_chunks), local_probe_rows=probe_keys.numel(), local_count=int(dt.local_count), prefilter=False, prefilter_mode="off",
                       prefilter_sampled_survivors=None, prefilter_below=0.0, probe_rows_sent=0, filter_bytes_received=0, wire_chunk_bytes=0,
                       wire_bytes_sent=int(dt.wire_bytes_sent))
    elif timings is not None:
        timings.update(strategy="shuffle", shuffle_form=f"chunks (fj_dist_join over {via})", split_s=dt.split_ms * 1e-3, exchange_s=dt.exchange_ms * 1e-3,
                       join_s=dt.join_ms * 1e-3, exchange_rounds=1, pieces=int(dt.pieces), local_build_rows=int(dt.local_build_chunks) * 256,
                       local_probe_rows=int(dt.local_probe_chunks) * 256, local_count=int(dt.local_count), prefilter=bool(dt.prefilter), prefilter_mode=prefilter_mode,
                       prefilter_sampled_survivors=(float(dt.prefilter_sampled) if dt.prefilter_sampled >= 0 else None), prefilter_below=float(prefilter_below),
                       probe_rows_sent=int(dt.probe_rows_kept) if (not standin or dt.prefilter) else probe_keys.numel(), filter_bytes_received=int(dt.filter_bytes),
                       rows_are_chunk_capacity=True,
                       wire_chunk_bytes=int(dt.wire_chunk_bytes), wire_bytes_sent=int(dt.sent_chunks) * (int(dt.wire_chunk_bytes) + 4))
    if pairs is not None:
        return int(cnt.value), sec, pairs[0], pairs[1]
    return int(cnt.value), sec


# A rung of the ladder that failed for a join shape - a skewed partition beyond the LDS table in the broadcast form, pools that an
# owner of hot keys overflows in the chunk form, replicas that do not fit - must not be attempted again on every later step of that
# shape: every failed attempt costs the whole step once more.  (world, build rows, probe rows, materialize, join_id) -> [calls since,
# first rung to try]; re-examined every 32nd call.  All ranks fail together (the driver agrees on failures), so the memos stay identical.
_FORM_MEMO: dict = {}
_FORM_RETRY_EVERY = 32
RUNGS = ("broadcast", "shuffle", "scatter")


def choose_strategy() -> str:
    """FJ_DIST_STRATEGY: 'auto' (default: the first rung that applies, by the driver's cost model), or the rung the ladder starts at -
    'broadcast' (counting joins; materialising joins start at the shuffle), 'shuffle', 'scatter'."""
    forced = os.environ.get("FJ_DIST_STRATEGY", "auto")
    if forced in ("auto",) + RUNGS:
        return forced
    raise ValueError(f"FJ_DIST_STRATEGY={forced!r}: auto | broadcast | shuffle | scatter")


def self_check(dist, group, engine, small_inputs, expected_small: int, message_elems: int, transport=None, corrupt: bool = False) -> dict:
    """A few seconds before a multi-rank job's first timed step: (1) one exchange of per-peer VIEWS at the largest message size
    the step will use (`message_elems` int64 per peer, capped at the RCCL-safe size), every element a function of (source,
    destination, index), verified in full on arrival - the > 4 GiB defect the bounded rounds work around, on real ranks; (2) one
    small distributed_join against its closed-form count; (3) the same join with the sender-side precheck forced on.  Every rank
    returns the same verdict: {"ok", "error", "failed_ranks", "message_bytes", "seconds", "precheck"}.  corrupt: test hook - rank 0
    flips one received bit (what a transport that moves wrong data looks like)."""
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
        if corrupt and me == 0:
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

    def agree(e) -> int:
        flag = engine.counts_tensor([1 if e else 0])
        dist.all_reduce(flag, op=dist.ReduceOp.SUM, group=group)
        return int(flag.item())
    nbad = agree(err)
    bk, bv, pk = small_inputs
    if nbad == 0:
        try:
            got = distributed_join(bk, bv, pk, group=group, engine=engine, transport=transport)[0]
            if int(got) != int(expected_small):
                err = f"join self-check: count {got} != closed form {expected_small}"
        except Exception as ex:                                    # noqa: BLE001  (raised on every rank, or agreed on inside)
            err = f"join self-check raised {ex!r}"
        nbad = agree(err)
    # the precheck forced on: what "auto" may choose in the timed steps (filters exported, all-gathered, probe pieces compacted).  A
    # failure does not fail the check: the verdict says so and the caller switches the precheck off (bench.py: FJ_DIST_PREFILTER=0)
    precheck = None
    if nbad == 0 and os.environ.get("FJ_DIST_PREFILTER", "") != "0" and hasattr(engine, "shuffle_plan") and not hasattr(engine, "dist_engine_ops"):
        perr = None
        try:
            tt: dict = {}
            got = distributed_join(bk, bv, pk, group=group, engine=engine, transport=transport, timings=tt, prefilter="on", strategy="shuffle")[0]
            if int(got) != int(expected_small):
                perr = f"count {got} != closed form {expected_small}"
            precheck = {"ok": perr is None, "ran": bool(tt.get("prefilter")), "rows_sent": tt.get("probe_rows_sent"), "form": tt.get("shuffle_form")}
        except Exception as ex:                                    # noqa: BLE001
            perr = f"raised {ex!r}"
        pbad = agree(perr)
        if pbad:
            precheck = {"ok": False, "error": perr, "failed_ranks": pbad, "action": "switch the precheck off for the timed steps (FJ_DIST_PREFILTER=0)"}
    return {"ok": nbad == 0, "error": err, "failed_ranks": nbad, "message_bytes": n * 8, "seconds": round(time.perf_counter() - t0, 3), "precheck": precheck}


def _owner_scatter_join(dist, group, engine, world, build_keys, build_values, probe_keys, materialize, bloom, return_arrays, timings, t0):
    """The last rung: split by owner, one all-to-all per array, the single-GPU join of what this rank owns, one all-reduce."""
    bk_s, bv_s, b_counts = engine.owner_split(build_keys, build_values, world)
    pk_s, _, p_counts = engine.owner_split(probe_keys, None, world)
    t1 = time.perf_counter()
    flat = []
    for d in range(world):
        flat += [b_counts[d], p_counts[d]]
    send_c = engine.counts_tensor(flat)
    recv_c = engine.counts_tensor([0] * len(flat))
    dist.all_to_all_single(recv_c, send_c, group=group)
    rc = recv_c.reshape(world, 2).tolist()
    b_recv, p_recv = [int(x[0]) for x in rc], [int(x[1]) for x in rc]
    mx = engine.counts_tensor([max(b_counts + p_counts)])              # the largest single message anywhere in the group decides the rounds
    dist.all_reduce(mx, op=dist.ReduceOp.MAX, group=group)
    rounds = max(1, -(-int(mx.item()) // _MAX_ELEMS_PER_MESSAGE))
    bk_r = _exchange(dist, group, engine, bk_s, b_counts, b_recv, rounds)
    bv_r = _exchange(dist, group, engine, bv_s, b_counts, b_recv, rounds)
    pk_r = _exchange(dist, group, engine, pk_s, p_counts, p_recv, rounds)
    engine.synchronize()
    t2 = time.perf_counter()
    res = engine.local_join(bk_r, bv_r, pk_r, materialize, bloom, 48, return_arrays)
    local_count = int(res[0])
    tot = engine.counts_tensor([local_count])
    dist.all_reduce(tot, op=dist.ReduceOp.SUM, group=group)
    engine.synchronize()
    t3 = time.perf_counter()
    if timings is not None:
        timings.update(strategy="scatter", shuffle_form="owner-scatter", split_s=t1 - t0, exchange_s=t2 - t1, join_s=t3 - t2, exchange_rounds=rounds,
                       local_build_rows=sum(b_recv), local_probe_rows=sum(p_recv), local_count=local_count,
                       prefilter=False, prefilter_mode="off", prefilter_sampled_survivors=None, probe_rows_sent=sum(p_counts),
                       wire_bytes_sent=8 * (sum(p_counts) - p_counts[dist.get_rank(group)]) + 16 * (sum(b_counts) - b_counts[dist.get_rank(group)]))
    out = (int(tot.item()), t3 - t0)
    if materialize and return_arrays:
        return out + (res[2], res[3])
    return out


def distributed_join(build_keys, build_values, probe_keys, *, materialize: bool = False, bloom: bool = False,
                     group=None, engine=None, return_arrays: bool = False, timings: Optional[dict] = None, transport=None, join_id=None,
                     strategy: Optional[str] = None, prefilter: Optional[str] = None, force_exchange: bool = False):
    """Join relations whose rows are block-distributed over the ranks of `group`.

    Every rank passes its LOCAL rows (int64 tensors on its GPU) and gets back `(global_match_count, seconds)`; with
    `materialize and return_arrays` also the pairs this rank owns.  `seconds` is this rank's wall time for the whole step.
    `transport`: an object with torch.distributed's collective functions (default: torch.distributed itself) - the self-tests
    that run several ranks on one GPU pass one that stages device tensors through the host for gloo.
    `join_id` (hashable, the same on every rank): names a repeated join for what is remembered about its shape (the precheck's
    sampled verdict, a rung that failed).  `strategy`: overrides FJ_DIST_STRATEGY for this call; `prefilter`: "off" | "on" | "auto"
    overrides FJ_DIST_PREFILTER; force_exchange: run the full protocol on a one-rank group too (tests).
    """
    import torch.distributed as _td
    dist = transport if transport is not None else _td
    if engine is None:
        engine = HipEngine()
    if hasattr(engine, "normalize"):             # int64/uint64, contiguous, 16-byte aligned: what the C ABI's pointers must be
        build_keys, build_values, probe_keys = engine.normalize(build_keys, build_values, probe_keys)
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1 and force_exchange and not dist.is_initialized():
        raise RuntimeError("force_exchange needs an initialised process group")
    t0 = time.perf_counter()
    if world == 1 and not force_exchange:
        res = engine.local_join(build_keys, build_values, probe_keys, materialize, bloom, 64, return_arrays)
        if timings is not None:
            timings.update(split_s=0.0, exchange_s=0.0, join_s=time.perf_counter() - t0)
        return res

    pieces = int(os.environ.get("FJ_DIST_PIECES", "0"))         # 0: the driver decides (4; 8 where the wire bounds a broadcast step - fj_dist_join)
    no_fallback = bool(os.environ.get("FJ_DIST_NO_FALLBACK"))
    standin = hasattr(engine, "dist_engine_ops")
    # relation sizes of every rank, rank 0's link rate (every rank models with it: hosts that called set_link_rate with per-rank
    # measurements still take the same decisions) and the free device memory: one tiny all-gather, the same ladder everywhere
    free_mb = (engine.free_bytes() >> 20) if hasattr(engine, "free_bytes") else (1 << 40)
    mine = engine.counts_tensor([build_keys.numel(), probe_keys.numel(), int(_LINK_BYTES_PER_S / 1e3), free_mb])
    allsz = engine.counts_tensor([0] * (4 * world))
    dist.all_gather_into_tensor(allsz, mine, group=group)
    allsz = allsz.reshape(world, 4).tolist()
    link0 = float(allsz[0][2]) * 1e3
    sizes_b, sizes_p = [int(x[0]) for x in allsz], [int(x[1]) for x in allsz]
    nb_total, np_global = sum(sizes_b), sum(sizes_p)
    strategy = strategy or choose_strategy()

    # ---- which rungs apply, in order ----
    rungs = []
    if strategy in ("auto", "broadcast") and world <= 16 and (not materialize or (hasattr(engine, "emit_pairs") and not standin)):
        can = (hasattr(engine, "bcast_plan") and engine.bcast_plan(nb_total) is not None) if not standin else getattr(engine, "has_bcast", False)
        # every rank holds a replica of the whole build side (6 bytes per key; 14 with the values of a materialising join) beside its
        # own pools (~20 bytes per local row of both relations)
        need_mb = ((14.1 if materialize else 6.1) * nb_total + 20.0 * (max(sizes_b) + max(sizes_p))) / 2**20
        fits = min(int(x[3]) for x in allsz) > need_mb
        if can and fits and (strategy == "broadcast" or form_model(world, max(sizes_b), max(sizes_p), link0, nb_total, np_global, materialize=materialize)["pick"] == "broadcast"):
            rungs.append("broadcast")
        elif strategy == "broadcast" and no_fallback:
            raise RuntimeError("FJ_DIST_STRATEGY=broadcast: this join cannot take the build-broadcast form (engine, > 16 ranks, a total build side without "
                               "a partitioned plan, or replicas that do not fit the free device memory)")
    if (strategy != "scatter" and hasattr(engine, "shuffle_plan") and engine.shuffle_plan(nb_total, world) is not None
            and (not materialize or hasattr(engine, "emit_pairs"))):
        rungs.append("shuffle")
    rungs.append("scatter")
    memo_key = (world, nb_total, np_global, bool(materialize), join_id)
    memo = _FORM_MEMO.get(memo_key)
    if memo is not None and strategy == "auto":
        memo[0] += 1
        if memo[0] % _FORM_RETRY_EVERY:
            rungs = [r for r in rungs if RUNGS.index(r) >= memo[1]] or ["scatter"]

    # ---- down the ladder ----
    tt = timings if timings is not None else {}
    for i, rung in enumerate(rungs):
        try:
            if rung == "broadcast":
                res = _driver_join(dist, group, engine, build_keys, probe_keys, max(0, pieces), tt, transport, build_values=build_values if materialize else None,
                                   return_arrays=return_arrays, form=FORM_BROADCAST)
            elif rung == "shuffle":
                mode = "off" if (standin and not getattr(engine, "chunk_precheck", False)) else _chunk_prefilter_mode(bloom, world, prefilter)
                below, pf_key, decision = _precheck_threshold(mode, world, nb_total, np_global, link0, join_id)
                res = None
                for attempt in ((below, mode), (0.0, "off")) if below > 0 else ((0.0, mode),):     # (a failed step with the precheck is retried without it)
                    try:
                        res = _driver_join(dist, group, engine, build_keys, probe_keys, max(0, pieces), tt, transport, build_values=build_values if materialize else None,
                                           return_arrays=return_arrays, prefilter_below=attempt[0], prefilter_mode=attempt[1], form=FORM_SHUFFLE)
                        _precheck_remember(pf_key, tt, decision)
                        break
                    except RuntimeError as ex:
                        if no_fallback or attempt[0] == 0.0:
                            raise
                        _abort_stream(engine)
                        tt["chunk_form_error"] = str(ex)
                        if pf_key is not None:
                            # a step that failed WITH the precheck (an owner of hot probe keys overflows pools sized from the mean before any probe
                            # piece is seen) is remembered as declined, re-examined with the next resample like any other verdict
                            _PRECHECK_MEMO[pf_key] = [1, 2.0]
            else:
                res = _owner_scatter_join(dist, group, engine, world, build_keys, build_values, probe_keys, materialize, bloom, return_arrays, tt, t0)
            if i > 0 and strategy == "auto":
                _FORM_MEMO[memo_key] = [0, RUNGS.index(rung)]           # the rungs above failed for this shape: start here for a while
            elif i == 0 and memo is not None and memo[0] % _FORM_RETRY_EVERY == 0:
                _FORM_MEMO.pop(memo_key, None)                          # re-examined, and the first rung holds again
            return res
        except RuntimeError as ex:
            if no_fallback or rung == "scatter":
                raise
            _abort_stream(engine)
            tt["broadcast_form_error" if rung == "broadcast" else "chunk_form_error"] = str(ex)
    raise AssertionError("unreachable: the owner-scatter rung returns or raises")
