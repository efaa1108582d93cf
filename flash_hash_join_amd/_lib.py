"""ctypes binding of libflashjoin_hip.so (C ABI: include/flashjoin.h, 40 entry points).

There is no CPU fallback: if the HIP library is missing or cannot be loaded, importing the join
API raises, and every call on a box without a HIP device fails with the library's error string.

The building blocks behind the ABI (include/flashjoin_lab.h: stream joins, the pieces of the multi-GPU driver, the partition
diagnostic) are not exported by the product library.  FJ_LIB_VARIANT=lab loads libflashjoin_hip_lab.so - the same objects linked
without the export list - for the test-suite and the measurement tools (flash_hash_join_amd/lab.py).
"""
from __future__ import annotations

import ctypes
import os
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
# FJ_LIB_VARIANT=lab: the library with the internal entry points visible; any other <name>: lib/ab/<name>.so (same-box A/B of two
# builds of the library, tools/mk_lib_variant.sh: linked with everything visible; the in-tree library is never overwritten by a
# measurement script)
VARIANT = os.environ.get("FJ_LIB_VARIANT", "")
LIB_PATH = (os.path.join(_PKG, "lib", "libflashjoin_hip.so") if not VARIANT
            else os.path.join(_PKG, "lib", "libflashjoin_hip_lab.so") if VARIANT == "lab"
            else os.path.join(_PKG, "lib", "ab", VARIANT + ".so"))
PRODUCT_LIB_PATH = os.path.join(_PKG, "lib", "libflashjoin_hip.so")
LAB_LIB_PATH = os.path.join(_PKG, "lib", "libflashjoin_hip_lab.so")
ABI_VERSION = 6                     # include/flashjoin.h: FJ_ABI_VERSION
CSRC = os.path.join(_PKG, "csrc")

# every symbol include/flashjoin.h declares (what libflashjoin_hip.so exports: csrc/exports.map)
SYMBOLS = [
    "fj_initialize", "fj_last_error", "fj_device_count", "fj_version", "fj_abi_version", "fj_set_option", "fj_get_option", "fj_key_mix64", "fj_key_unmix64",
    "fj_ctx_create", "fj_ctx_destroy", "fj_ctx_workspace_bytes", "fj_ctx_trim",
    "fj_join_host", "fj_free_host", "fj_last_timings", "fj_join_device", "fj_emit_pairs",
    "fj_owner_split", "fj_stream_abort", "fj_shuffle_plan", "fj_bcast_plan",
    "fj_dist_comm_set_form", "fj_dist_model", "fj_dist_unique_id", "fj_dist_comm_create", "fj_dist_comm_from_nccl", "fj_dist_comm_from_transport",
    "fj_dist_comm_destroy", "fj_dist_comm_rank", "fj_dist_comm_size", "fj_dist_join_count", "fj_dist_join",
    "fj_generate_build", "fj_generate_probe",
    "fj_device_malloc", "fj_device_free", "fj_memcpy_h2d", "fj_memcpy_d2h", "fj_memcpy_d2d",
]
# ... and include/flashjoin_lab.h (visible in libflashjoin_hip_lab.so only)
LAB_SYMBOLS = [
    "fj_owner_hist", "fj_owner_scatter",
    "fj_shuffle_chunk_bytes", "fj_shuffle_pack_begin", "fj_shuffle_pack_counts", "fj_shuffle_pack_finish", "fj_stream_open_shuffled",
    "fj_stream_append_build_chunks", "fj_stream_append_probe_chunks",
    "fj_shuffle_part_filter_bytes", "fj_shuffle_part_filter_range", "fj_stream_export_part_filters", "fj_shuffle_pack_filter", "fj_shuffle_pack_kept", "fj_part_filter_sample",
    "fj_bcast_region_bytes", "fj_bcast_piece_span", "fj_bcast_pack", "fj_bcast_pack_bounds", "fj_bcast_probe", "fj_bcast_join", "fj_bcast_finish", "fj_bcast_abort", "fj_bcast_emit",
    "fj_bloom_filter_words", "fj_bloom_export", "fj_bloom_prefilter",
    "fj_stream_open", "fj_stream_append_build", "fj_stream_advance_probe", "fj_stream_begin", "fj_stream_append_probe", "fj_stream_finish",
    "fj_ctx_reserve_cus", "fj_debug_partition",
]


class FjTimings(ctypes.Structure):
    _fields_ = [
        ("total_ms", ctypes.c_double), ("build_phase_ms", ctypes.c_double), ("probe_phase_ms", ctypes.c_double),
        ("join_ms", ctypes.c_double), ("emit_ms", ctypes.c_double), ("probe_part_kernel_ms", ctypes.c_double * 4),
        ("h2d_ms", ctypes.c_double), ("d2h_ms", ctypes.c_double),
        ("path", ctypes.c_int), ("passes", ctypes.c_int), ("radix_bits", ctypes.c_int), ("fell_back", ctypes.c_int),
        ("partitions", ctypes.c_uint64), ("overlapped", ctypes.c_int), ("lds_retries", ctypes.c_int),
        ("filter_ms", ctypes.c_double), ("filter_survivors", ctypes.c_uint64), ("bloom_level", ctypes.c_int), ("sampled_hit_bp", ctypes.c_int), ("host_streamed", ctypes.c_int), ("reserved3", ctypes.c_int),
    ]

    def as_dict(self):
        d = {n: getattr(self, n) for n, _ in self._fields_ if n != "probe_part_kernel_ms"}
        d["probe_part_kernel_ms"] = list(self.probe_part_kernel_ms)
        return d


class FjDistTimings(ctypes.Structure):
    _fields_ = [
        ("struct_size", ctypes.c_size_t),
        ("total_ms", ctypes.c_double), ("split_ms", ctypes.c_double), ("exchange_ms", ctypes.c_double), ("join_ms", ctypes.c_double),
        ("local_count", ctypes.c_uint64), ("local_build_chunks", ctypes.c_uint64), ("local_probe_chunks", ctypes.c_uint64),
        ("sent_chunks", ctypes.c_uint64),
        ("pieces", ctypes.c_int), ("nranks", ctypes.c_int), ("fan_log0", ctypes.c_int), ("wire_chunk_bytes", ctypes.c_int),
        ("prefilter", ctypes.c_int), ("prefilter_sampled", ctypes.c_double), ("probe_rows_kept", ctypes.c_uint64), ("filter_bytes", ctypes.c_uint64),
        ("form", ctypes.c_int), ("form_reserved", ctypes.c_int), ("wire_bytes_sent", ctypes.c_uint64),
        ("local", FjTimings),
        ("reserve_cus", ctypes.c_int), ("reserve_how", ctypes.c_int), ("reserve_with_ms", ctypes.c_double), ("reserve_without_ms", ctypes.c_double),
    ]


# fj_dist_transport / fj_dist_engine_ops (include/flashjoin.h): callback tables of fj_dist_comm_from_transport
_vp, _pu64, _pvp = ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_void_p)
AllGatherFn = ctypes.CFUNCTYPE(ctypes.c_int, _vp, _pu64, ctypes.c_int, _pu64)
AllReduceFn = ctypes.CFUNCTYPE(ctypes.c_int, _vp, _pu64, ctypes.c_int)
AllToAllFn = ctypes.CFUNCTYPE(ctypes.c_int, _vp, ctypes.c_int, _pvp, _pu64, _pvp, _pu64)


class FjDistTransport(ctypes.Structure):
    _fields_ = [("user", _vp), ("nranks", ctypes.c_int), ("rank", ctypes.c_int),
                ("all_gather_u64", AllGatherFn), ("all_reduce_sum_u64", AllReduceFn), ("all_to_all_bytes", AllToAllFn)]


EngErrorFn = ctypes.CFUNCTYPE(ctypes.c_char_p, _vp)
EngPlanFn = ctypes.CFUNCTYPE(ctypes.c_int, _vp, ctypes.c_uint64, ctypes.c_int)
EngAllocFn = ctypes.CFUNCTYPE(_vp, _vp, ctypes.c_size_t)
EngReleaseFn = ctypes.CFUNCTYPE(None, _vp, _vp)
EngPackBeginFn = ctypes.CFUNCTYPE(ctypes.c_int, _vp, _vp, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int)
EngPackCountsFn = ctypes.CFUNCTYPE(ctypes.c_int, _vp, _pu64)
EngPackFinishFn = ctypes.CFUNCTYPE(ctypes.c_int, _vp, _pvp, _pvp)
EngOpenFn = ctypes.CFUNCTYPE(ctypes.c_int, _vp, ctypes.c_uint64, ctypes.c_int, ctypes.c_int, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int)
EngAppendFn = ctypes.CFUNCTYPE(ctypes.c_int, _vp, ctypes.c_int, _vp, _vp, ctypes.c_uint64)
EngFinishFn = ctypes.CFUNCTYPE(ctypes.c_int, _vp, _pu64)
EngAbortFn = ctypes.CFUNCTYPE(None, _vp)
EngFilterRangeFn = ctypes.CFUNCTYPE(ctypes.c_int, _vp, ctypes.c_uint64, ctypes.c_int, ctypes.c_int, _pu64, _pu64, _pu64, _pu64)
EngExportFn = ctypes.CFUNCTYPE(ctypes.c_int, _vp, _vp)
EngPackFilterFn = ctypes.CFUNCTYPE(ctypes.c_int, _vp, _vp, _pu64)
EngSampleFn = ctypes.CFUNCTYPE(ctypes.c_int, _vp, _vp, ctypes.c_uint64, ctypes.c_uint64, _vp, ctypes.c_uint64, ctypes.c_int, _pu64)
_u64c, _pu32 = ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint32)
EngBcRegionFn = ctypes.CFUNCTYPE(ctypes.c_uint64, _vp, _u64c, _u64c)
EngBcSpanFn = ctypes.CFUNCTYPE(ctypes.c_int, _vp, _u64c, _u64c, _u64c, _u64c, ctypes.c_int, _pu64, _pu64)
EngBcNpartsFn = ctypes.CFUNCTYPE(ctypes.c_int, _vp, _u64c, _pu32)
EngBcPackFn = ctypes.CFUNCTYPE(ctypes.c_int, _vp, _vp, _u64c, _u64c, _vp, ctypes.c_int, _pu64)
EngBcProbeFn = ctypes.CFUNCTYPE(ctypes.c_int, _vp, _vp, _u64c, _u64c)
EngBcJoinFn = ctypes.CFUNCTYPE(ctypes.c_int, _vp, _vp, ctypes.c_int, _pu64, _pu64, ctypes.c_uint32, ctypes.c_uint32)
EngBcFinishFn = ctypes.CFUNCTYPE(ctypes.c_int, _vp, _pu64)


class FjDistEngineOps(ctypes.Structure):
    _fields_ = [("struct_size", ctypes.c_size_t), ("user", _vp), ("chunk_bytes", ctypes.c_size_t), ("error", EngErrorFn), ("plan", EngPlanFn), ("alloc", EngAllocFn),
                ("release", EngReleaseFn), ("pack_begin", EngPackBeginFn), ("pack_counts", EngPackCountsFn), ("pack_finish", EngPackFinishFn),
                ("open", EngOpenFn), ("append", EngAppendFn), ("finish", EngFinishFn), ("abort", EngAbortFn),
                ("filter_range", EngFilterRangeFn), ("export_filters", EngExportFn), ("pack_filter", EngPackFilterFn), ("sample", EngSampleFn),
                ("bc_region_bytes", EngBcRegionFn), ("bc_span", EngBcSpanFn), ("bc_nparts", EngBcNpartsFn), ("bc_pack", EngBcPackFn),
                ("bc_probe", EngBcProbeFn), ("bc_join", EngBcJoinFn), ("bc_finish", EngBcFinishFn)]


def build_native(force: bool = False) -> str:
    """Compile the HIP library in-tree with hipcc for gfx950 (csrc/Makefile)."""
    cmd = ["make", "-C", CSRC, "-j4", "-s"] + (["-B"] if force else [])
    subprocess.check_call(cmd)
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"hipcc build did not produce {LIB_PATH}")
    return LIB_PATH


_lib = None


def load() -> ctypes.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"flash_hash_join_amd: HIP library {LIB_PATH} is missing. Build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` or `make -C flash_hash_join_amd/csrc`. "
            "There is no CPU fallback.")
    # PyTorch-ROCm wheels bundle their own libamdhip64 under the same soname.  Whichever copy is loaded first
    # serves the whole process, and torch does not find a GPU through /opt/rocm's copy; so when torch is
    # installed, let it load its runtime before this library binds to "libamdhip64".
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = ctypes.CDLL(LIB_PATH)
    u64, sz, vp, i32 = ctypes.c_uint64, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_int
    pu64 = ctypes.POINTER(u64)
    L.fj_initialize.restype = i32
    L.fj_last_error.restype = ctypes.c_char_p
    L.fj_device_count.restype = i32
    L.fj_version.restype = ctypes.c_char_p
    L.fj_key_mix64.restype = u64; L.fj_key_mix64.argtypes = [u64]
    L.fj_key_unmix64.restype = u64; L.fj_key_unmix64.argtypes = [u64]
    L.fj_ctx_create.restype = vp; L.fj_ctx_create.argtypes = [i32]
    L.fj_ctx_destroy.restype = None; L.fj_ctx_destroy.argtypes = [vp]
    L.fj_ctx_workspace_bytes.restype = sz; L.fj_ctx_workspace_bytes.argtypes = [vp]
    L.fj_ctx_trim.restype = i32; L.fj_ctx_trim.argtypes = [vp]
    L.fj_stream_abort.restype = i32; L.fj_stream_abort.argtypes = [vp]
    L.fj_join_host.restype = i32
    L.fj_join_host.argtypes = [i32, i32, i32, vp, vp, sz, vp, sz, pu64, ctypes.POINTER(ctypes.c_double),
                               ctypes.POINTER(vp), ctypes.POINTER(vp)]
    L.fj_free_host.restype = None; L.fj_free_host.argtypes = [vp]
    L.fj_last_timings.restype = i32; L.fj_last_timings.argtypes = [ctypes.POINTER(FjTimings)]
    L.fj_join_device.restype = i32
    L.fj_join_device.argtypes = [vp, i32, i32, i32, vp, vp, sz, vp, sz, vp, i32, pu64, vp, vp, sz,
                                 ctypes.POINTER(FjTimings)]
    L.fj_emit_pairs.restype = i32
    L.fj_emit_pairs.argtypes = [vp, vp, vp, sz, vp, ctypes.POINTER(FjTimings)]
    L.fj_owner_split.restype = i32
    L.fj_owner_split.argtypes = [vp, vp, vp, sz, i32, vp, vp, pu64, vp]
    L.fj_set_option.restype = i32; L.fj_set_option.argtypes = [ctypes.c_char_p, ctypes.c_longlong]
    L.fj_get_option.restype = ctypes.c_longlong; L.fj_get_option.argtypes = [ctypes.c_char_p]
    L.fj_shuffle_plan.restype = i32; L.fj_shuffle_plan.argtypes = [sz, i32, ctypes.POINTER(i32), ctypes.POINTER(i32)]
    L.fj_dist_comm_from_transport.restype = vp
    L.fj_dist_comm_from_transport.argtypes = [vp, ctypes.POINTER(FjDistTransport), ctypes.POINTER(FjDistEngineOps)]
    L.fj_dist_unique_id.restype = i32; L.fj_dist_unique_id.argtypes = [ctypes.c_char_p]
    L.fj_dist_comm_create.restype = vp; L.fj_dist_comm_create.argtypes = [vp, ctypes.c_char_p, i32, i32]
    L.fj_dist_comm_from_nccl.restype = vp; L.fj_dist_comm_from_nccl.argtypes = [vp, vp]
    L.fj_dist_comm_destroy.restype = None; L.fj_dist_comm_destroy.argtypes = [vp]
    L.fj_dist_comm_rank.restype = i32; L.fj_dist_comm_rank.argtypes = [vp]
    L.fj_dist_comm_size.restype = i32; L.fj_dist_comm_size.argtypes = [vp]
    psz, pi32, pu32 = ctypes.POINTER(sz), ctypes.POINTER(i32), ctypes.POINTER(ctypes.c_uint32)
    L.fj_bcast_plan.restype = i32; L.fj_bcast_plan.argtypes = [sz, pi32, pu32, pi32]
    pdbl = ctypes.POINTER(ctypes.c_double)
    L.fj_dist_model.restype = i32; L.fj_dist_model.argtypes = [i32, u64, u64, u64, u64, u64, ctypes.c_double, pdbl, pdbl]
    L.fj_dist_comm_set_form.restype = i32; L.fj_dist_comm_set_form.argtypes = [vp, i32, ctypes.c_double]
    L.fj_dist_join_count.restype = i32; L.fj_dist_join_count.argtypes = [vp, vp, sz, vp, sz, i32, vp, pu64, ctypes.POINTER(FjDistTimings)]
    L.fj_dist_join.restype = i32; L.fj_dist_join.argtypes = [vp, vp, vp, sz, vp, sz, i32, i32, ctypes.c_double, vp, pu64, pu64, ctypes.POINTER(FjDistTimings)]
    L.fj_generate_build.restype = i32; L.fj_generate_build.argtypes = [vp, vp, vp, u64, sz, vp]
    L.fj_generate_probe.restype = i32
    L.fj_generate_probe.argtypes = [vp, vp, u64, sz, u64, u64, ctypes.c_uint32, pu64, vp]
    L.fj_device_malloc.restype = i32; L.fj_device_malloc.argtypes = [ctypes.POINTER(vp), sz]
    L.fj_device_free.restype = i32; L.fj_device_free.argtypes = [vp]
    L.fj_memcpy_h2d.restype = i32; L.fj_memcpy_h2d.argtypes = [vp, vp, sz]
    L.fj_memcpy_d2h.restype = i32; L.fj_memcpy_d2h.argtypes = [vp, vp, sz]
    L.fj_memcpy_d2d.restype = i32; L.fj_memcpy_d2d.argtypes = [vp, vp, sz]

    L.fj_abi_version.restype = i32; L.fj_abi_version.argtypes = []
    if L.fj_abi_version() != ABI_VERSION:
        raise ImportError(f"flash_hash_join_amd: {LIB_PATH} speaks ABI {L.fj_abi_version()}, this binding ABI {ABI_VERSION} (rebuild: make -C flash_hash_join_amd/csrc)")
    L.has_lab = hasattr(L, "fj_debug_partition")          # the internal entry points are visible (libflashjoin_hip_lab.so, A/B variants)
    if L.has_lab:
        L.fj_ctx_reserve_cus.restype = None; L.fj_ctx_reserve_cus.argtypes = [vp, ctypes.c_uint]
        L.fj_owner_hist.restype = i32; L.fj_owner_hist.argtypes = [vp, vp, sz, i32, pu64, vp]
        L.fj_owner_scatter.restype = i32; L.fj_owner_scatter.argtypes = [vp, vp, vp, sz, i32, pu64, vp, vp, vp]
        L.fj_stream_open.restype = i32; L.fj_stream_open.argtypes = [vp, sz, i32, sz, i32, vp, i32]
        L.fj_stream_append_build.restype = i32; L.fj_stream_append_build.argtypes = [vp, vp, sz, vp]
        L.fj_stream_advance_probe.restype = i32; L.fj_stream_advance_probe.argtypes = [vp, vp]
        L.fj_stream_begin.restype = i32; L.fj_stream_begin.argtypes = [vp, vp, vp, sz, sz, i32, vp, i32]
        L.fj_stream_append_probe.restype = i32; L.fj_stream_append_probe.argtypes = [vp, vp, sz, vp]
        L.fj_stream_finish.restype = i32; L.fj_stream_finish.argtypes = [vp, vp, pu64, ctypes.POINTER(FjTimings)]
        L.fj_bloom_filter_words.restype = sz; L.fj_bloom_filter_words.argtypes = []
        L.fj_bloom_export.restype = i32; L.fj_bloom_export.argtypes = [vp, vp, sz, i32, vp, vp]
        L.fj_bloom_prefilter.restype = i32; L.fj_bloom_prefilter.argtypes = [vp, vp, sz, i32, vp, vp, sz, pu64, vp]
        L.fj_shuffle_chunk_bytes.restype = sz; L.fj_shuffle_chunk_bytes.argtypes = [sz, i32]
        L.fj_shuffle_pack_begin.restype = i32; L.fj_shuffle_pack_begin.argtypes = [vp, vp, vp, sz, sz, i32, i32, vp]
        L.fj_shuffle_part_filter_bytes.restype = sz; L.fj_shuffle_part_filter_bytes.argtypes = []
        L.fj_shuffle_part_filter_range.restype = i32; L.fj_shuffle_part_filter_range.argtypes = [sz, i32, i32, ctypes.POINTER(sz), ctypes.POINTER(sz), ctypes.POINTER(sz)]
        L.fj_stream_export_part_filters.restype = i32; L.fj_stream_export_part_filters.argtypes = [vp, vp, vp]
        L.fj_shuffle_pack_filter.restype = i32; L.fj_shuffle_pack_filter.argtypes = [vp, vp, vp]
        L.fj_shuffle_pack_kept.restype = u64; L.fj_shuffle_pack_kept.argtypes = [vp]
        L.fj_part_filter_sample.restype = i32; L.fj_part_filter_sample.argtypes = [vp, vp, sz, sz, vp, sz, i32, vp, pu64]
        L.fj_shuffle_pack_counts.restype = i32; L.fj_shuffle_pack_counts.argtypes = [vp, pu64]
        L.fj_shuffle_pack_finish.restype = i32; L.fj_shuffle_pack_finish.argtypes = [vp, ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(vp), vp]
        L.fj_stream_open_shuffled.restype = i32; L.fj_stream_open_shuffled.argtypes = [vp, sz, i32, i32, sz, i32, sz, i32, i32, vp]
        L.fj_stream_append_build_chunks.restype = i32; L.fj_stream_append_build_chunks.argtypes = [vp, vp, vp, vp, sz, vp]
        L.fj_stream_append_probe_chunks.restype = i32; L.fj_stream_append_probe_chunks.argtypes = [vp, vp, vp, sz, vp]
        L.fj_bcast_region_bytes.restype = sz; L.fj_bcast_region_bytes.argtypes = [sz, sz, i32]
        L.fj_bcast_piece_span.restype = i32; L.fj_bcast_piece_span.argtypes = [sz, sz, sz, sz, i32, psz, psz]
        L.fj_bcast_pack.restype = i32; L.fj_bcast_pack.argtypes = [vp, vp, vp, sz, sz, vp, i32, i32, vp]
        L.fj_bcast_pack_bounds.restype = i32; L.fj_bcast_pack_bounds.argtypes = [vp, pu64]
        L.fj_bcast_probe.restype = i32; L.fj_bcast_probe.argtypes = [vp, vp, sz, sz, vp]
        L.fj_bcast_join.restype = i32; L.fj_bcast_join.argtypes = [vp, vp, i32, pu64, pu64, ctypes.c_uint32, ctypes.c_uint32, vp]
        L.fj_bcast_finish.restype = i32; L.fj_bcast_finish.argtypes = [vp, vp, pu64, ctypes.POINTER(FjTimings)]
        L.fj_bcast_abort.restype = None; L.fj_bcast_abort.argtypes = [vp]
        L.fj_bcast_emit.restype = i32; L.fj_bcast_emit.argtypes = [vp, vp, vp, sz, vp]
        L.fj_debug_partition.restype = i32
        L.fj_debug_partition.argtypes = [vp, vp, vp, sz, i32, i32, vp, vp, vp, vp, pu64]
    _lib = L
    return L


def last_error() -> str:
    return load().fj_last_error().decode("utf-8", "replace")


def check(rc: int) -> None:
    if rc != 0:
        msg = last_error()
        if "out of" in msg and "memory" in msg:
            raise MemoryError(msg)
        raise RuntimeError(msg)
