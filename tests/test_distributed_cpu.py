"""CPU test of the multi-GPU protocol (flash_hash_join_amd/distributed.py + the C++ driver csrc/fj_dist.hip): world_size 2 and 3
over gloo.  The per-rank primitives are replaced by a stand-in engine built on the CPU oracle (test infrastructure); what is
under test is the host logic: the ladder of forms (build broadcast -> owner shuffle in chunk form -> owner scatter), the driver's
state machine over a callback transport, agreed failures and what is remembered about them, the sender-side precheck's protocol."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _fmix64(k):
    k = k.astype(np.uint64)
    with np.errstate(over="ignore"):
        k ^= k >> np.uint64(33); k *= np.uint64(0xff51afd7ed558ccd)
        k ^= k >> np.uint64(33); k *= np.uint64(0xc4ceb9fe1a85ec53)
        k ^= k >> np.uint64(33)
    return k


class OracleEngine:
    """Same interface as distributed.HipEngine, CPU tensors, joins through the oracle."""

    def __init__(self):
        from oracle import oracle as O
        self.O = O

    def empty(self, n):
        return torch.empty(n, dtype=torch.int64)

    def counts_tensor(self, counts):
        return torch.tensor(counts, dtype=torch.int64)

    def owner_split(self, keys, vals, world):
        k = keys.numpy().view(np.uint64)
        owner = (((_fmix64(k.copy()) >> np.uint64(48)) * np.uint64(world)) >> np.uint64(16)).astype(np.int64)
        order = np.argsort(owner, kind="stable")
        counts = np.bincount(owner, minlength=world).tolist()
        ok = torch.from_numpy(k[order].view(np.int64).copy())
        ov = torch.from_numpy(vals.numpy()[order].copy()) if vals is not None else None
        return ok, ov, counts

    # stand-in for the chunk form of the owner shuffle: the rank's own work behind the C++ driver (csrc/fj_dist.hip runs the
    # protocol; fj_dist_engine_ops callbacks do what fj_shuffle_pack_* / fj_stream_open_shuffled / fj_stream_append_*_chunks do
    # on a GPU).  32 first-pass buckets from the top hash bits, bucket b belongs to rank (b * world) >> 5; wire chunks = 2048
    # bytes = 256 raw int64 keys, deliberately ragged (200-key chunks) with a directory word (bucket << 9 | count) each.
    def dist_engine_ops(self, world):
        return _StandInOps(self, world)

    @property
    def chunk_precheck(self):                    # the stand-in's sender-side precheck in chunk form (its optional engine callbacks): opt-in per test
        return os.environ.get("FJ_TEST_CHUNK_PRECHECK") == "1"

    has_bcast = True                             # the stand-in's engine callbacks include the build-broadcast form

    def shuffle_plan(self, nb_total, world):
        return 5 if nb_total >= 10000 and world <= 32 else None

    def local_join(self, bk, bv, pk, materialize, bloom, hash_top_bits, return_arrays):
        res = self.O.c_join(bk.numpy(), bv.numpy(), pk.numpy(), algo="radix", bloom=bloom, materialize=materialize,
                            threads=2, return_arrays=return_arrays)
        if materialize and return_arrays:
            return res[0], res[1], torch.from_numpy(res[2].view(np.int64)), torch.from_numpy(res[3].view(np.int64))
        return res[0], res[1]

    def synchronize(self):
        pass


class _StandInOps:
    """fj_dist_engine_ops of the CPU tests: "device memory" is host memory, rows and chunks are NumPy views of raw pointers."""
    chunk_bytes = 2048
    F0LOG = 5

    def __init__(self, eng, world):
        import ctypes
        self.ct, self.O, self.world = ctypes, eng.O, world
        self.bufs = {}
        self.packed = None
        self.npacks = 0
        # test hook FJ_TEST_FAIL = "pack:<rank>" / "copy:<rank>" / "append:<rank>": that rank's third packing pass / the copy into the
        # wire format behind it / its second probe append fails
        what, _, who = os.environ.get("FJ_TEST_FAIL", ":").partition(":")
        self.fail = what if who != "" and int(who) == dist.get_rank() else ""

    def _view(self, ptr, n, dtype):
        if n == 0:
            return np.empty(0, dtype=dtype)
        return np.frombuffer((self.ct.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr), dtype=dtype)

    def plan(self, nb_total, nranks):
        return nb_total >= 10000 and nranks <= 32

    def alloc(self, nbytes):
        a = np.zeros(nbytes + 64, dtype=np.uint8)
        addr = (a.ctypes.data + 63) & ~63
        self.bufs[addr] = a
        return addr

    def release(self, ptr):
        self.bufs.pop(ptr, None)

    def pack_begin(self, rows, n, nb_total, nranks):
        assert nranks == self.world and self.packed is None
        self.npacks += 1
        if self.fail == "pack" and self.npacks == 3:
            raise RuntimeError("injected packing failure")
        k = self._view(rows, n, np.uint64).copy()
        b = (_fmix64(k.copy()) >> np.uint64(64 - self.F0LOG)).astype(np.int64)
        per = [[] for _ in range(nranks)]                      # per owner: (bucket, keys of one chunk)
        for bb in range(32):
            rws, o = k[b == bb].view(np.int64), (bb * nranks) >> self.F0LOG
            for i in range(0, rws.size, 200):
                per[o].append((bb, rws[i: i + 200]))
        self.packed = per

    def pack_counts(self):
        return [len(x) for x in self.packed]

    def pack_finish(self, dst_chunks, dst_dir):
        if self.fail == "copy" and self.npacks == 3:      # behind the piece's agreement points: the peers are already posting its receives
            self.packed = None                            # (the engine can start the next piece's pass: the failure is then agreed on in the final all-reduce)
            raise RuntimeError("injected copy failure")
        for o, chunks in enumerate(self.packed):
            if not chunks:
                continue
            ck = self._view(dst_chunks[o], len(chunks) * 256, np.int64).reshape(-1, 256)
            dw = self._view(dst_dir[o], len(chunks), np.int32)
            for i, (bb, rws) in enumerate(chunks):
                ck[i, : rws.size] = rws
                ck[i, rws.size:] = -1
                dw[i] = (bb << 9) | rws.size
        self.packed = None

    def open(self, nb_total, nranks, rank, nb_bound, np_bound, pieces):
        self.mine = [bb for bb in range(32) if (bb * nranks) >> self.F0LOG == rank]
        self.chunks = {0: [], 1: []}
        self.left = {0: 1, 1: pieces}
        self.bound = {0: nb_bound, 1: np_bound}

    def append(self, side, chunks, dirw, nchunks):
        assert self.left[side] > 0
        self.left[side] -= 1
        if self.fail == "append" and side == 1 and self.left[1] == 2:
            raise RuntimeError("injected append failure")
        d, c = self._view(dirw, nchunks, np.int32), self._view(chunks, nchunks * 256, np.int64).reshape(-1, 256)
        for i in range(nchunks):
            assert (int(d[i]) >> 9) in self.mine and 1 <= (int(d[i]) & 0x1FF) <= 256
            self.chunks[side].append(c[i, : int(d[i]) & 0x1FF].copy())
        assert sum(x.size for x in self.chunks[side]) <= self.bound[side]

    def finish(self):
        cat = lambda xs: np.concatenate(xs) if xs else np.empty(0, dtype=np.int64)
        bk, pk = cat(self.chunks[0]), cat(self.chunks[1])
        self.chunks = None
        if bk.size == 0 or pk.size == 0:
            return 0
        return self.O.c_join(bk, np.zeros_like(bk), pk, algo="radix", threads=2)[0]

    def abort(self):
        self.chunks = None

    # ---- the optional build-broadcast callbacks: 64 "final partitions" (6 bits of the mixed key); a region = offset table u32[65]
    # (272 bytes with padding) + the rank's keys sorted by partition as whole 8-byte words; no third part ----
    BC_PARTS, BC_HDR = 64, 272

    def _bc_part(self, k):
        return (_fmix64(k.copy()) >> np.uint64(58)).astype(np.int64)

    def bc_region_bytes(self, nb_total, nkeys):
        return self.BC_HDR + 8 * nkeys + 16 if nb_total >= 5000 else 0

    def bc_span(self, nb_total, nkeys, k_lo, k_hi, part):
        assert 0 <= k_lo <= k_hi <= nkeys
        return {0: (0, 4 * (self.BC_PARTS + 1)), 1: (self.BC_HDR + 8 * k_lo, 8 * (k_hi - k_lo)), 2: (0, 0)}[part]

    def bc_nparts(self, nb_total):
        return self.BC_PARTS

    def bc_pack(self, rows, n, nb_total, region, pieces):
        if self.fail == "bcpack":
            raise RuntimeError("injected packing failure")
        k = self._view(rows, n, np.uint64).copy()
        p = self._bc_part(k)
        order = np.argsort(p, kind="stable")
        offs = np.zeros(self.BC_PARTS + 1, dtype=np.uint32)
        offs[1:] = np.cumsum(np.bincount(p, minlength=self.BC_PARTS))
        self._view(region, self.BC_PARTS + 1, np.uint32)[:] = offs
        if n:
            self._view(region + self.BC_HDR, n, np.uint64)[:] = k[order]
        self.bc_probe_rows, self.bc_count = None, 0
        return [int(offs[self.BC_PARTS * q // pieces]) for q in range(pieces + 1)]

    def bc_probe(self, rows, n, nb_total):
        k = self._view(rows, n, np.uint64).copy()
        self.bc_probe_rows = (k, self._bc_part(k))

    def bc_join(self, base, region_off, nkeys, part_lo, part_hi):
        if self.fail == "bcjoin" and part_lo > 0:
            raise RuntimeError("injected join failure")
        assert len(region_off) == self.world and self.bc_probe_rows is not None
        build = []
        for off, n in zip(region_off, nkeys):
            offs = self._view(base + off, self.BC_PARTS + 1, np.uint32)
            assert int(offs[self.BC_PARTS]) == n
            lo, hi = int(offs[part_lo]), int(offs[part_hi])
            run = self._view(base + off + self.BC_HDR + 8 * lo, hi - lo, np.uint64)
            assert np.all((self._bc_part(run) >= part_lo) & (self._bc_part(run) < part_hi))      # the piece has landed, and in the right place
            build.append(run.copy())
        k, p = self.bc_probe_rows
        mine = k[(p >= part_lo) & (p < part_hi)]
        self.bc_count += int(np.isin(mine, np.concatenate(build)).sum()) if mine.size else 0

    def bc_finish(self):
        self.bc_probe_rows = None
        return self.bc_count

    # ---- the optional precheck callbacks: one 4096-byte bitmap per first-pass bucket ("partition" of this stand-in), bit = 15 bits of the mixed key ----
    FB = 4096

    @staticmethod
    def _bit(k):
        return ((_fmix64(k.copy()) >> np.uint64(20)) & np.uint64(32767)).astype(np.int64)

    def _bucket(self, k):
        return (_fmix64(k.copy()) >> np.uint64(64 - self.F0LOG)).astype(np.int64)

    def filter_range(self, nb_total, nranks, rank):
        mine = [bb for bb in range(32) if (bb * nranks) >> self.F0LOG == rank]
        return (mine[0] if mine else 0), len(mine), 32, self.FB

    def export_filters(self, dst):
        if self.fail == "export":
            raise RuntimeError("injected export failure")
        assert self.left[0] == 0                                   # the build side is complete
        out = self._view(dst, len(self.mine) * self.FB, np.uint8)
        out[:] = 0
        for c in self.chunks[0]:
            k = c.view(np.uint64)
            pos = (self._bucket(k) - self.mine[0]) * (self.FB * 8) + self._bit(k)
            np.bitwise_or.at(out, pos >> 3, (1 << (pos & 7)).astype(np.uint8))

    def _passes(self, k, filters):
        f = self._view(filters, 32 * self.FB, np.uint8)
        pos = self._bucket(k) * (self.FB * 8) + self._bit(k)
        return (f[pos >> 3] >> (pos & 7).astype(np.uint8)) & 1 != 0

    def pack_filter(self, filters):
        kept, per = 0, [[] for _ in range(self.world)]
        for o, chunks in enumerate(self.packed):
            by_bucket = {}
            for bb, rws in chunks:
                by_bucket.setdefault(bb, []).append(rws)
            for bb, parts in by_bucket.items():
                rws = np.concatenate(parts)
                rws = rws[self._passes(rws.view(np.uint64), filters)]
                kept += rws.size
                for i in range(0, rws.size, 200):
                    per[o].append((bb, rws[i: i + 200]))
        self.packed = per
        return kept

    def sample(self, rows, n, stride, filters, nb_total, nranks):
        k = self._view(rows, n * stride, np.uint64)[::stride][:n]
        return int(self._passes(k, filters).sum()) if n else 0


def _worker_chunk_precheck(rank, world, port, nb, npk, q):
    """The sender-side precheck of the chunk form through the C++ driver with the stand-in engine's optional callbacks: forced,
    by a sample on both sides of the threshold (and remembered), and with one rank's filter export failing."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), FJ_DIST_STRATEGY="shuffle", FJ_TEST_CHUNK_PRECHECK="1")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from flash_hash_join_amd import datagen
        import flash_hash_join_amd.distributed as D
        b0, b1 = rank * nb // world, (rank + 1) * nb // world
        p0, p1 = rank * npk // world, (rank + 1) * npk // world
        bk, bv = datagen.build_numpy(b1 - b0, first=b0)
        pk, exp_local = datagen.probe_numpy(p1 - p0, nb, seed=1, hit_bp=5000, first=p0)
        tb, tv, tp = (torch.from_numpy(x.view(np.int64)) for x in (bk, bv, pk))
        e = torch.tensor([exp_local]); dist.all_reduce(e); exp = int(e.item())
        out = {}
        for name, env, below in (("on", {"FJ_DIST_PREFILTER": "1"}, None), ("auto_runs", {"FJ_DIST_PREFILTER": "auto"}, 0.9),
                                 ("auto_again", {"FJ_DIST_PREFILTER": "auto"}, 0.9),
                                 ("auto_declines", {"FJ_DIST_PREFILTER": "auto"}, 0.05),
                                 ("export_fails", {"FJ_DIST_PREFILTER": "1", "FJ_TEST_FAIL": f"export:{world - 1}"}, None)):
            for k in ("FJ_DIST_PREFILTER", "FJ_TEST_FAIL"):
                os.environ.pop(k, None)
            os.environ.update(env)
            D.PREFILTER_BELOW_OVERRIDE = below
            D._FORM_MEMO.clear()
            if name in ("auto_runs", "auto_declines"):
                D._PRECHECK_MEMO.clear()
            t = {}
            cnt, _ = D.distributed_join(tb, tv, tp, engine=OracleEngine(), timings=t)
            assert cnt == exp, (name, cnt, exp)
            assert t["shuffle_form"].startswith("chunks (fj_dist_join over a callback transport"), (name, t)
            glob = torch.tensor([t["probe_rows_sent"]]); dist.all_reduce(glob)
            if name in ("on", "auto_runs", "auto_again"):
                assert t["prefilter"] is True and "chunk_form_error" not in t, (name, t)
                assert exp <= int(glob.item()) < 0.56 * npk                # 50 % hits; a 32768-bit filter per bucket lets few misses pass
                assert t["filter_bytes_received"] == (32 - len([bb for bb in range(32) if (bb * world) >> 5 == rank])) * 4096, t
            if name == "auto_runs":
                assert t["prefilter_decision"] == "sampled" and 0.45 < t["prefilter_sampled_survivors"] < 0.6, t
            if name == "auto_again":
                assert t["prefilter_decision"] == "memo: runs" and t["prefilter_sampled_survivors"] is None, t
            if name == "auto_declines":
                assert t["prefilter"] is False and t["prefilter_decision"] == "sampled" and int(glob.item()) == npk, t
            if name == "export_fails":       # agreed on by every rank before any filter is exchanged; the step is retried without the precheck
                assert t["prefilter"] is False and "partition filters could not be prepared on 1 rank" in t["chunk_form_error"], t
                assert ("injected export failure" in t["chunk_form_error"]) == (rank == world - 1), t
            out[name] = (t["prefilter"], int(glob.item()))
        q.put((rank, exp, out))
    finally:
        dist.destroy_process_group()


def _worker(rank, world, port, nb, npk, q, strategy):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    variant = strategy.split("_", 1)[1] if "_" in strategy else ""
    strategy = strategy.split("_")[0]
    os.environ["FJ_DIST_PREFILTER"] = "0"
    os.environ["FJ_DIST_STRATEGY"] = strategy
    failing = variant in ("packfail", "copyfail", "appendfail")
    if failing:    # one rank's packing pass / local join fails inside the C++ driver: EVERY rank sees the failure
        os.environ["FJ_TEST_FAIL"] = variant[:-4] + ":" + str(world - 1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from flash_hash_join_amd import datagen
        import flash_hash_join_amd.distributed as D
        from flash_hash_join_amd.distributed import distributed_join
        if variant == "smallmessages":          # every message of the owner-scatter exchange carries <= 3000 rows: several rounds per array
            D._MAX_ELEMS_PER_MESSAGE = 3000
        # block distribution of the global relation (GPU g holds rows [g*N/G, (g+1)*N/G))
        b0, b1 = rank * nb // world, (rank + 1) * nb // world
        p0, p1 = rank * npk // world, (rank + 1) * npk // world
        if variant == "uneven":                 # every probe row on rank 0, every build row on the last rank: empty shards elsewhere
            b0, b1 = (0, nb) if rank == world - 1 else (0, 0)
            p0, p1 = (0, npk) if rank == 0 else (0, 0)
        bk, bv = datagen.build_numpy(b1 - b0, first=b0)
        pk, exp_local = datagen.probe_numpy(p1 - p0, nb, seed=1, hit_bp=5000, first=p0)
        if variant in ("skew", "lateskew"):
            # hot probe key: 60 % of every rank's probe rows carry ONE build key, so its owner receives far more than 1.5x an even share.
            # "skew": spread over all pieces - the driver sizes the owner's pools from the first piece it sees, the chunk form succeeds;
            # "lateskew": only in the second half of the rows - what the first piece cannot announce
            hot = np.uint64(12345 * 0x9E3779B97F4A7C15 & 0xFFFFFFFFFFFFFFFF)
            pk = pk.copy()
            idx = np.arange(pk.size)
            sel = (idx % 5 < 3) if variant == "skew" else ((idx >= pk.size // 2) & (idx % 10 < 9))
            exp_local += int(sel.sum()) - int(np.isin(pk[sel], np.arange(1, nb + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)).sum())
            pk[sel] = hot
        t = {}
        tb, tv, tp = (torch.from_numpy(x.view(np.int64)) for x in (bk, bv, pk))
        if strategy == "shuffle" and variant == "":
            # the pre-flight check of a multi-rank job: the exchange of per-peer views + one small join, the same verdict on
            # every rank; a transport that moves wrong data (test hook on rank 0) is reported by everybody
            from flash_hash_join_amd.distributed import self_check
            e0 = torch.tensor([exp_local]); dist.all_reduce(e0)
            chk = self_check(dist, None, OracleEngine(), (tb, tv, tp), int(e0.item()), 5000)
            assert chk["ok"] and chk["failed_ranks"] == 0 and chk["message_bytes"] == 40000, chk
            chk = self_check(dist, None, OracleEngine(), (tb, tv, tp), int(e0.item()), 5000, corrupt=True)
            assert not chk["ok"] and chk["failed_ranks"] == 1 and (rank != 0 or "elements received from rank" in chk["error"]), chk
            chk = self_check(dist, None, OracleEngine(), (tb, tv, tp), int(e0.item()) + 1, 5000)        # a wrong expectation: the join check fires
            assert not chk["ok"] and chk["failed_ranks"] == world and "join self-check" in chk["error"], chk
        # materialising: a stand-in engine cannot emit an owner's pairs from the chunk form: the owner-scatter rung
        res = distributed_join(tb, tv, tp, materialize=True, return_arrays=True, engine=OracleEngine(), timings=t)
        assert t["strategy"] == "scatter" and t["shuffle_form"] == "owner-scatter", t
        exp = torch.tensor([exp_local]); dist.all_reduce(exp)
        tc = {}
        if failing:
            os.environ["FJ_DIST_NO_FALLBACK"] = "1"
            # ("copyfail": the failing rank stays in step - nobody waits for an exchange it never posted - up to the final all-reduce,
            #  where what it sent instead of the piece may have failed its receivers' joins too: "on N rank(s)")
            with pytest.raises(RuntimeError, match={"packfail": "packing a piece failed on 1 rank", "copyfail": "the local join failed on [1-3] rank",
                                                    "appendfail": "the local join failed on 1 rank"}[variant]) as ei:
                distributed_join(tb, tv, tp, engine=OracleEngine())
            if variant == "copyfail":
                assert rank != world - 1 or "injected copy failure" in str(ei.value)
            else:
                assert ("injected" in str(ei.value)) == (rank == world - 1)       # the failing rank says why, the others that somebody failed
            del os.environ["FJ_DIST_NO_FALLBACK"]
        cnt, _ = distributed_join(tb, tv, tp, engine=OracleEngine(), timings=tc)        # counting
        assert cnt == int(exp.item())
        if failing:    # without FJ_DIST_NO_FALLBACK all ranks move down the ladder together: the owner-scatter rung
            assert "failed on " in tc["chunk_form_error"] and tc["shuffle_form"] == "owner-scatter" and tc["strategy"] == "scatter", tc
            # ... and under "auto" the shape is remembered: the next step starts at the rung that worked (no second failed attempt)
            os.environ["FJ_DIST_STRATEGY"] = "auto"
            D._FORM_MEMO.clear()
            t1, t2 = {}, {}
            assert distributed_join(tb, tv, tp, engine=OracleEngine(), timings=t1, strategy="shuffle")[0] == cnt          # (pinned: nothing is remembered)
            assert not D._FORM_MEMO and "chunk_form_error" in t1
            os.environ["FJ_DIST_STRATEGY"] = strategy
        elif strategy == "scatter":
            assert tc["strategy"] == "scatter" and tc["shuffle_form"] == "owner-scatter" and tc["probe_rows_sent"] == p1 - p0, tc
            if variant == "smallmessages":
                assert tc["exchange_rounds"] >= 2, tc
        elif variant in ("skew", "uneven", ""):
            assert tc["strategy"] == "shuffle" and tc["shuffle_form"].startswith("chunks (fj_dist_join over a callback transport") and "chunk_form_error" not in tc, tc
        if variant == "uneven":
            assert tc["pieces"] == 1          # a rank without probe rows: the exchange is not cut into pieces
        if variant == "lateskew":                    # (at these sizes the pools' constant slack absorbs it; at scale the ranks would rerun together in the owner-scatter form)
            assert tc["shuffle_form"].startswith("chunks") or "failed on" in tc["chunk_form_error"], tc
        if strategy == "shuffle" and variant == "":
            assert tc["probe_rows_sent"] == p1 - p0 and tc["wire_chunk_bytes"] == 2048 and tc["pieces"] == 4
            # (the driver reports received CHUNKS x 256: the stand-in's ragged 200-key chunks count as whole ones)
            glob = torch.tensor([tc["local_probe_rows"], tc["local_build_rows"], tc["local_count"]]); dist.all_reduce(glob)
            assert npk <= glob[0] <= 2.5 * npk and nb <= glob[1] <= 2.5 * nb and int(glob[2]) == int(exp.item())
        # every pair this rank owns must hash to this rank (the owner-scatter rung: owner = top 16 hash bits * world >> 16)
        keys = res[2].numpy().view(np.uint64)
        owner = ((_fmix64(keys.copy()) >> np.uint64(48)) * np.uint64(world)) >> np.uint64(16)
        owned = bool(np.all(owner == rank))
        q.put((rank, int(res[0]), int(exp.item()), int(res[2].numel()), owned, t.get("local_count")))
    finally:
        dist.destroy_process_group()


def _worker_broadcast(rank, world, port, nb, npk, q, variant):
    """The build-broadcast form through the ONE C++ driver (csrc/fj_dist.hip: dist_join_bcast) over gloo, the rank's work done by the
    stand-in engine's bc_* callbacks: forced (FJ_DIST_STRATEGY=broadcast), chosen by the driver's cost model (unset strategy, a
    probe-heavy join), with empty blocks, and with one rank failing in the pack / in a range's join."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["FJ_DIST_PREFILTER"] = "0"
    os.environ.pop("FJ_DIST_STRATEGY", None)
    auto = variant.startswith("auto")
    if not auto:
        os.environ["FJ_DIST_STRATEGY"] = "broadcast"
    if variant == "autofail":
        variant = "bcjoinfail"
    if variant in ("bcpackfail", "bcjoinfail"):
        os.environ["FJ_TEST_FAIL"] = variant[:-4] + ":" + str(world - 1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from flash_hash_join_amd import datagen
        import flash_hash_join_amd.distributed as D
        from flash_hash_join_amd.distributed import distributed_join
        b0, b1 = rank * nb // world, (rank + 1) * nb // world
        p0, p1 = rank * npk // world, (rank + 1) * npk // world
        if variant == "uneven":                 # every probe row on rank 0, every build row on the last rank: empty blocks elsewhere
            b0, b1 = (0, nb) if rank == world - 1 else (0, 0)
            p0, p1 = (0, npk) if rank == 0 else (0, 0)
        bk, bv = datagen.build_numpy(b1 - b0, first=b0)
        pk, exp_local = datagen.probe_numpy(p1 - p0, nb, seed=1, hit_bp=5000, first=p0)
        tb, tv, tp = (torch.from_numpy(x.view(np.int64)) for x in (bk, bv, pk))
        exp = torch.tensor([exp_local]); dist.all_reduce(exp)
        if variant in ("bcpackfail", "bcjoinfail"):
            os.environ["FJ_DIST_NO_FALLBACK"] = "1"
            with pytest.raises(RuntimeError, match="packing the build side failed on 1 rank" if variant == "bcpackfail" else "the local join failed on 1 rank") as ei:
                distributed_join(tb, tv, tp, engine=OracleEngine())
            assert ("injected" in str(ei.value)) == (rank == world - 1)       # the failing rank says why, the others that somebody failed
            del os.environ["FJ_DIST_NO_FALLBACK"]
        tc = {}
        cnt, _ = distributed_join(tb, tv, tp, engine=OracleEngine(), timings=tc)
        assert cnt == int(exp.item()), (cnt, int(exp.item()))
        if variant in ("bcpackfail", "bcjoinfail"):     # without FJ_DIST_NO_FALLBACK all ranks move down the ladder together: the chunk shuffle comes next
            assert "failed on 1 rank" in tc["broadcast_form_error"] and tc["strategy"] == "shuffle" and "chunk_form_error" not in tc, tc
            if auto:
                # under "auto" a failed broadcast used to drop to the owner-scatter form - and to be attempted again on every
                # step.  Now the shuffle is the next rung, and the shape remembers where to start: no second failed attempt
                t2 = {}
                assert distributed_join(tb, tv, tp, engine=OracleEngine(), timings=t2)[0] == cnt
                assert t2["strategy"] == "shuffle" and "broadcast_form_error" not in t2, t2
                assert list(D._FORM_MEMO.values()) == [[1, 1]], D._FORM_MEMO
        else:
            assert tc["strategy"] == "broadcast" and tc["shuffle_form"].startswith("build broadcast") and tc["probe_rows_sent"] == 0, tc
            assert tc["pieces"] == 4 and tc["local_probe_rows"] == p1 - p0
            # every rank's region (272 + 8 bytes per key, 260 bytes of offset table on the wire) went to every peer, nothing else did
            assert tc["wire_bytes_sent"] == (world - 1) * (260 + 8 * (b1 - b0)), tc
            glob = torch.tensor([tc["local_count"]]); dist.all_reduce(glob)
            assert int(glob.item()) == int(exp.item())
        # a materialising join under the same setting shuffles (the pairs stay with the owner of the key)
        t = {}
        res = distributed_join(tb, tv, tp, materialize=True, return_arrays=True, engine=OracleEngine(), timings=t)
        assert res[0] == int(exp.item()) and t["strategy"] == "scatter"
        q.put((rank, int(cnt), int(exp.item())))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


@pytest.mark.parametrize("strategy", ["shuffle", "shuffle_packfail", "shuffle_copyfail", "shuffle_appendfail", "shuffle_skew", "shuffle_lateskew", "shuffle_uneven", "scatter", "scatter_smallmessages"])
@pytest.mark.parametrize("world", [2, 3])
def test_distributed_join_gloo(world, strategy, oracle):
    nb, npk = 20000, 90000
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, nb, npk, q, strategy)) for r in range(world)]
    for p in procs:
        p.start()
    rows = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    total_pairs = 0
    for rank, cnt, exp, npairs, owned, local in rows:
        assert cnt == exp                       # global count is the closed-form expectation on every rank
        assert owned and npairs == local        # pairs stay sharded by owner
        total_pairs += npairs
    assert total_pairs == rows[0][1]


@pytest.mark.parametrize("variant", ["", "auto", "autofail", "uneven", "bcpackfail", "bcjoinfail"])
@pytest.mark.parametrize("world", [2, 3])
def test_build_broadcast_form_through_the_driver_gloo(world, variant, oracle):
    nb, npk = 20000, (400000 if variant.startswith("auto") else 90000)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_broadcast, args=(r, world, port, nb, npk, q, variant)) for r in range(world)]
    for p in procs:
        p.start()
    rows = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(cnt == exp for _, cnt, exp in rows) and len({exp for _, _, exp in rows}) == 1


@pytest.mark.parametrize("world", [2, 3])
def test_chunk_form_precheck_through_the_driver_gloo(world, oracle):
    """fj_dist_join(prefilter_below) at world 2 and 3 without a GPU: the C++ driver runs the precheck's protocol - build side first,
    the owners' filters exported and exchanged over the transport (uneven ranges at world 3), probe pieces compacted, the sample's
    all-reduced verdict, an export failure agreed on by every rank - over the stand-in engine's optional callbacks."""
    nb, npk = 20000, 90000
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_chunk_precheck, args=(r, world, port, nb, npk, q)) for r in range(world)]
    for p in procs:
        p.start()
    rows = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert len({(exp, tuple(sorted(out.items()))) for _, exp, out in rows}) == 1        # every rank reports the same verdicts and totals


def test_chunk_form_precheck_break_even_model(monkeypatch):
    """distributed._chunk_prefilter_break_even: never on one rank or for joins too small for the filters' fixed cost; generous where
    one or three links carry the shuffle; at 8 GPUs ~0.75 with 45 GB/s links, ~0.55 with 55, ~0.15 with 65; zero when the links outrun the
    kernels.  _chunk_prefilter_mode: "auto" for every multi-rank join unless FJ_DIST_PREFILTER says otherwise."""
    from flash_hash_join_amd import distributed as D
    monkeypatch.delenv("FJ_DIST_PREFILTER", raising=False); monkeypatch.setattr(D, "PREFILTER_BELOW_OVERRIDE", None)
    monkeypatch.setattr(D, "_LINK_BYTES_PER_S", 45e9)
    f = D._chunk_prefilter_break_even
    assert f(1, 125_000_000, 1_250_000_000) == 0.0 and f(8, 8_000_000, 10_000_000) == 0.0 and f(4, 10**9, 0) == 0.0
    assert 0.8 < f(2, 250_000_000, 1_250_000_000) < 0.95 and 0.75 < f(4, 500_000_000, 1_250_000_000) < 0.95
    assert 0.6 < f(8, 10**9, 1_250_000_000) < 0.9          # 45 GB/s: the links bound the step by far
    monkeypatch.setattr(D, "_LINK_BYTES_PER_S", 55e9)
    assert 0.4 < f(8, 10**9, 1_250_000_000) < 0.65         # 55 GB/s: around one half
    monkeypatch.setattr(D, "_LINK_BYTES_PER_S", 65e9)
    assert f(8, 10**9, 1_250_000_000) < 0.2
    monkeypatch.setattr(D, "_LINK_BYTES_PER_S", 200e9)
    assert f(8, 10**9, 1_250_000_000) == 0.0
    monkeypatch.setattr(D, "PREFILTER_BELOW_OVERRIDE", 0.25)
    assert f(1, 1, 1) == 0.25
    assert D._chunk_prefilter_mode(False, 1) == "off" and D._chunk_prefilter_mode(True, 1) == "auto" and D._chunk_prefilter_mode(False, 8) == "auto"
    monkeypatch.setenv("FJ_DIST_PREFILTER", "0")
    assert D._chunk_prefilter_mode(True, 8) == "off"
    monkeypatch.setenv("FJ_DIST_PREFILTER", "1")
    assert D._chunk_prefilter_mode(False, 1) == "on"


def test_a_failed_precheck_attempt_is_remembered_as_declined(monkeypatch):
    """a chunk-form step that fails WITH the sender-side precheck (an owner of hot probe keys overflows pools sized from
    the mean) reran without it on EVERY later step of the same shape, roughly doubling the step.  The failed attempt is now memoised
    as "declined" (survivor share 2.0: above any break-even) until the next resample."""
    from flash_hash_join_amd import distributed as D
    monkeypatch.setattr(D, "_LINK_BYTES_PER_S", 10e9)
    D._PRECHECK_MEMO.clear()
    calls = []

    def fake_driver(dist, group, engine, bk, pk, pieces, tt, transport, prefilter_below=0.0, prefilter_mode="off", form=0, **kw):
        calls.append(prefilter_below)
        if prefilter_below > 0:
            raise RuntimeError("fj_dist_join_count: the local join failed on 1 rank(s)")
        tt.update(strategy="shuffle", prefilter=False, prefilter_sampled_survivors=None)
        return 7, 0.0

    monkeypatch.setattr(D, "_driver_join", fake_driver)

    class Eng:
        def counts_tensor(self, c): return torch.tensor(c, dtype=torch.int64)
        def shuffle_plan(self, nb_total, world): return 9
        def stream_abort(self): pass
        has_bcast = False
        def dist_engine_ops(self, world): return None
        chunk_precheck = True

    class FakeDist:
        def is_initialized(self): return True
        def get_world_size(self, group=None): return 8
        def all_gather_into_tensor(self, out, t, group=None): out.copy_(t.repeat(8))

    monkeypatch.setenv("FJ_DIST_STRATEGY", "shuffle"); monkeypatch.setenv("FJ_DIST_PREFILTER", "auto"); monkeypatch.setattr(D, "PREFILTER_BELOW_OVERRIDE", 0.6)
    bk = torch.arange(125_000, dtype=torch.int64); pk = torch.arange(1_250_000, dtype=torch.int64)
    below = D._precheck_threshold("auto", 8, 8 * 125_000, 8 * 1_250_000)[0]
    assert below == 0.6                                    # a threshold in the model's place: sample, then decide
    for _ in range(3):
        t = {}
        assert D.distributed_join(bk, bk, pk, engine=Eng(), transport=FakeDist(), timings=t)[0] == 7
    # first step: the attempt with the precheck fails, the rerun without it succeeds; later steps go straight to the form that works
    assert calls == [below, 0.0, 0.0, 0.0], calls
    D._PRECHECK_MEMO.clear()


def test_precheck_verdict_is_remembered_per_join_shape(monkeypatch):
    """distributed._precheck_threshold / _precheck_remember: "auto" samples once per (world, build rows, probe rows), then runs or
    declines without exporting a filter, and samples afresh on every 32nd call; forced modes and joins the model rules out keep no memo."""
    from flash_hash_join_amd import distributed as D
    monkeypatch.delenv("FJ_DIST_PREFILTER", raising=False)
    monkeypatch.setattr(D, "PREFILTER_BELOW_OVERRIDE", 0.4)
    D._PRECHECK_MEMO.clear()
    shape = (8, 10**9, 10**10)
    assert D._precheck_threshold("on", *shape) == (2.0, None, "on") and D._precheck_threshold("off", *shape) == (0.0, None, "off")
    below, key, how = D._precheck_threshold("auto", *shape)
    assert (below, key, how) == (0.4, shape + (None,), "sampled")
    t = {"prefilter_sampled_survivors": 0.55}
    D._precheck_remember(key, t, how)
    assert t["prefilter_decision"] == "sampled" and D._PRECHECK_MEMO[shape + (None,)] == [1, 0.55]
    for call in range(2, 33):                                    # calls 2..32: remembered - declined (0.55 >= 0.4), nothing sampled
        below, key, how = D._precheck_threshold("auto", *shape)
        assert (below, how) == (0.0, "memo: declined"), call
        D._precheck_remember(key, {}, how)
    below, key, how = D._precheck_threshold("auto", *shape)        # call 33: a fresh sample
    assert (below, how) == (0.4, "sampled")
    D._precheck_remember(key, {"prefilter_sampled_survivors": 0.1}, how)
    assert D._precheck_threshold("auto", *shape)[::2] == (2.0, "memo: runs")      # few survivors now: runs, without sampling
    assert D._precheck_threshold("auto", 8, 10**9, 5 * 10**9)[2] == "sampled"     # another shape: its own verdict
    assert D._precheck_threshold("auto", *shape, None, "orders x lineitem")[2] == "sampled"   # the same shape under a name: its own verdict too
    # the threshold is a function of the link rate it is GIVEN (distributed_join passes rank 0's), not of this process's setting
    monkeypatch.setattr(D, "PREFILTER_BELOW_OVERRIDE", None)
    slow, fast = D._chunk_prefilter_break_even(8, 10**9, 1_250_000_000, 45e9), D._chunk_prefilter_break_even(8, 10**9, 1_250_000_000, 65e9)
    monkeypatch.setattr(D, "_LINK_BYTES_PER_S", 65e9)
    assert slow > fast and D._chunk_prefilter_break_even(8, 10**9, 1_250_000_000, 45e9) == slow and D._chunk_prefilter_break_even(8, 10**9, 1_250_000_000) == fast
    monkeypatch.setattr(D, "_LINK_BYTES_PER_S", 45e9)
    assert D._precheck_threshold("auto", 1, 125_000_000, 1_250_000_000) == (0.0, None, "model: cannot pay")
    D._PRECHECK_MEMO.clear()
