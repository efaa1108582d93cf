"""The building blocks BEHIND the C ABI, one by one: stream joins, the sender side and the owner side of the chunk-form shuffle, the
sender-side prechecks, the build-broadcast form's pack / probe / join, the partition diagnostic (include/flashjoin_lab.h).

Not part of the drop-in boundary: the product library does not export these entry points.  The test-suite and the measurement
tools load the lab build of the library (the same objects linked without the export list: FJ_LIB_VARIANT=lab, set by
tests/conftest.py and by the tools that need it) and drive the pieces through LabEngine."""
from __future__ import annotations

import ctypes
from typing import List

from .distributed import HipEngine


class LabEngine(HipEngine):
    def __init__(self, device=None):
        super().__init__(device)
        if not getattr(self.L, "has_lab", False):
            raise RuntimeError("flash_hash_join_amd.lab needs the lab build of the library: set FJ_LIB_VARIANT=lab before importing the package "
                               "(flash_hash_join_amd/lib/libflashjoin_hip_lab.so, built by `make -C flash_hash_join_amd/csrc`)")
        self._keep = []

    def empty_like(self, t):
        return self.torch.empty_like(t)

    def cat(self, parts):
        return self.torch.cat(list(parts))

    def reserve_cus(self, n: int) -> None:
        self.L.fj_ctx_reserve_cus(self.ctx, n)

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
    def bcast_region_bytes(self, nb_total: int, nkeys: int, with_vals: bool = False) -> int:
        return int(self.L.fj_bcast_region_bytes(nb_total, nkeys, int(with_vals)))

    def bcast_pack(self, keys, nb_total: int, region, pieces: int, vals=None) -> None:
        """Asynchronous: this rank's build keys (and values: a materialising step) -> `region` (a uint8 tensor view of
        bcast_region_bytes(nb_total, keys.numel(), vals is not None) bytes)."""
        s = self.torch.cuda.current_stream(self.index).cuda_stream
        self._lib.check(self.L.fj_bcast_pack(self.ctx, keys.data_ptr(), vals.data_ptr() if vals is not None else None, keys.numel(), nb_total, region.data_ptr(), pieces, int(vals is not None), s))

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

    def chunk_rows(self, dirw) -> int:
        """Rows in a set of chunks, from their directory words (bucket << 9 | count; unused ids are all ones)."""
        if dirw.numel() == 0:
            return 0
        cnt = dirw & 0x1FF
        return int(cnt[dirw != -1].sum().item())

    def stream_abort(self):
        self._keep = []
        super().stream_abort()
