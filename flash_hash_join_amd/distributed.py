"""Multi-GPU radix join: one process per GPU over RCCL/xGMI.

The reference is single-process; this is new design.  Two exchange strategies, chosen per call
by FJ_DIST_STRATEGY = shuffle | replicate | auto (a per-link byte + local-work cost model, choose_strategy); unset it
means the shuffle north_star names, or the model's choice once a link rate was measured on the node (set_link_rate):

replicate -- every rank all-gathers the build KEYS (and values when materialising) and joins its own probe rows
  against all of them; probe rows never move, their partition passes run while the build keys are on the wire
  (fj_stream_open / append_probe / advance_probe, then append_build per arrived piece / finish; FJ_REPLICATE_PIECES,
  default 4 asynchronous all-gathers: the first pass over piece c runs while piece c+1 is on the wire).  An xGMI mesh
  has one link per peer, so an exchange is bound by bytes per link: B*8 here against (P*8 + B*16)/N for the shuffle -- fewer up to
  N = 12 for the probe-heavy (P = 10 B) joins this path is built for, 6x fewer at N = 2.  Cost: every rank
  partitions all N*B build keys.  Global count = sum of local counts; pairs stay with their probe row.

shuffle -- radix partitions are independent join units (hash_join.cpp:340-356, :515-525), so the level-0 digit is
  the owner GPU:   owner(key) = (top 16 bits of hash(key) * world) >> 16

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
        dist.all_to_all(outs, ins, group=group)
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
    """Partition passes of the single-GPU plan for a build side of nb rows (csrc/fj_api.hip make_plan)."""
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


_LINK_MEASURED = False


def set_link_rate(bytes_per_s: float, measured: bool = True) -> None:
    """Replace the built-in per-link rate of the strategy model by a measured one (tools/xgmi_probe.py; bench.py does this
    once at N > 1).  With a measured rate the default strategy becomes the model's choice."""
    global _LINK_BYTES_PER_S, _LINK_MEASURED
    if bytes_per_s > 0:
        _LINK_BYTES_PER_S = float(bytes_per_s)
        _LINK_MEASURED = bool(measured)


def choose_strategy(world: int, nb: int, np_: int, materialize: bool) -> str:
    """'shuffle' (the owner exchange north_star names) or 'replicate', for per-rank relation sizes nb x np_ (the maxima
    over the ranks).  FJ_DIST_STRATEGY=replicate|shuffle forces one, =auto lets the cost model decide.  Unset: the shuffle,
    unless a link rate was MEASURED on this node (set_link_rate; bench.py measures one all-to-all at N > 1) - then the model
    decides with that rate: on a point-to-point mesh the probe-heavy shuffle puts (P*8 + B*16)/N bytes on every link
    (6 GB per step at N = 2 for config 5's shards) where replicating the build side puts B*8 (1 GB)."""
    forced = os.environ.get("FJ_DIST_STRATEGY", "auto" if _LINK_MEASURED else "shuffle")
    if forced in ("replicate", "shuffle"):
        return forced
    if world * nb >= (1 << 31):               # replicated build side must stay inside one GPU's chunk directory
        return "shuffle"
    c = strategy_costs(world, nb, np_, materialize)
    return "replicate" if c["replicate"] <= c["shuffle"] else "shuffle"


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
                     group=None, engine=None, return_arrays: bool = False, timings: Optional[dict] = None):
    """Join relations whose rows are block-distributed over the ranks of `group`.

    Every rank passes its LOCAL rows (int64 tensors on its GPU) and gets back
    `(global_match_count, seconds)`; with `materialize and return_arrays` also the pairs this
    rank owns.  `seconds` is this rank's wall time for the whole step (split + exchange + join).
    """
    import torch.distributed as dist
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
    mine = engine.counts_tensor([build_keys.numel(), probe_keys.numel()])
    allsz = engine.counts_tensor([0] * (2 * world))
    dist.all_gather_into_tensor(allsz, mine, group=group)
    allsz = allsz.reshape(world, 2).tolist()
    sizes_b = [int(x[0]) for x in allsz]
    strategy = choose_strategy(world, max(sizes_b), max(int(x[1]) for x in allsz), materialize)
    if strategy == "replicate":
        return _replicated_join(dist, group, engine, world, build_keys, build_values, probe_keys, sizes_b, materialize, bloom,
                                return_arrays, int(os.environ.get("FJ_REPLICATE_PIECES", "4")), timings)
    if timings is not None:
        timings["strategy"] = "shuffle"
    if not materialize and pieces > 1 and hasattr(engine, "stream_begin"):
        return _pipelined_count(dist, group, engine, world, build_keys, build_values, probe_keys, pieces, timings,
                                prefilter=_prefilter_mode(bloom) if hasattr(engine, "bloom_export") else "off")

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
