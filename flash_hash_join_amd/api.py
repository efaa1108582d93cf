"""Host-side mirror of the reference's `flash_join` module (PYBIND11_MODULE, hash_join.cpp:598-640).

Same thirteen names, same keyword arguments (`build_keys, build_values, probe_keys`), same
return value `(total_results: int, core_duration_sec: float)` for every join, `initialize()`
returning None.  The join itself runs in libflashjoin_hip.so (HIP kernels for gfx950) through the
C ABI of include/flashjoin.h; there is no CPU implementation behind these functions.

Differences from the reference, all deliberate:
  * non-contiguous inputs are made contiguous instead of being silently misread;
  * `build_values` shorter than `build_keys` raises ValueError instead of reading out of bounds;
  * `core_duration_sec` is the device-resident time (HIP events); PCIe copies of NumPy inputs
    are reported separately in `last_timings()`;
  * opt-in `return_arrays=True` returns the materialised pairs the reference computes and drops
    (hash_join.cpp:365-380): `(count, seconds, keys, values)`;
  * torch tensors that already live on a ROCm device are joined in place (no PCIe); so is any device array that
    speaks DLPack (`__dlpack__` / `__dlpack_device__`); pandas / Arrow / list inputs go through `np.asarray`;
  * the "scalar" functions (`hash_join*`: ONE table for the whole build side) run the partitioned plan by default:
    a table that does not fit LDS costs a cache-missing 64-B access per probe in HBM, more traffic than two
    streaming partition passes, so on MI355X it is the slower way to the same result at every size.
    `set_option("scalar_hbm_table", 1)` restores the literal algorithm (HBM table with linear probing over 8-slot
    groups, bloom word per group for the `_bloom` variants); it is also the fallback when a partition overflows LDS.
"""
from __future__ import annotations

import ctypes
import threading
from typing import Any, Dict, Optional, Tuple

import numpy as np

from . import _lib
from ._lib import FjTimings, check

ALGO_ADAPTIVE, ALGO_SCALAR, ALGO_RADIX = 0, 1, 2
ALGO_MANY_TO_MANY = 0x10          # FJ_ALGO_MANY_TO_MANY: OR'ed into algo (extension, include/flashjoin.h)

_ctxs: Dict[int, int] = {}
_ctx_locks: Dict[int, Any] = {}
_last: Optional[FjTimings] = None


def _is_torch_tensor(x: Any) -> bool:
    return type(x).__module__.startswith("torch") and hasattr(x, "data_ptr")


def _as_u64_host(a: Any, name: str) -> np.ndarray:
    """array_t<uint64_t> forcecast semantics: uint64 zero-copy, int64 reinterpreted bit-for-bit,
    anything else value-cast; N-D inputs are flattened (hash_join.cpp:317)."""
    arr = np.asarray(a)
    if arr.dtype == np.int64:
        arr = arr.view(np.uint64) if arr.flags.c_contiguous else np.ascontiguousarray(arr).view(np.uint64)
    elif arr.dtype != np.uint64:
        if arr.dtype.kind not in "iufb":
            raise TypeError(f"{name}: cannot convert dtype {arr.dtype} to uint64")
        arr = arr.astype(np.uint64, casting="unsafe")
    arr = np.ascontiguousarray(arr).reshape(-1)
    if arr.ctypes.data % 16:
        arr = np.require(arr.copy(), requirements=["ALIGNED", "C"])
    return arr


def context(device: int) -> int:
    """One native context (workspace, events) per device, created on first use."""
    if device not in _ctxs:
        L = _lib.load()
        h = L.fj_ctx_create(int(device))
        if not h:
            raise RuntimeError(_lib.last_error())
        _ctxs[device] = h
    return _ctxs[device]


def workspace_bytes(device: Optional[int] = None) -> int:
    """Device memory the native contexts keep cached between joins (grow-only until trim_workspace)."""
    L = _lib.load()
    return sum(int(L.fj_ctx_workspace_bytes(h)) for d, h in _ctxs.items() if device is None or d == device)


def trim_workspace(device: Optional[int] = None) -> None:
    """Give the cached workspace back to the device (tens of GB after a 1B-row join); contexts stay usable."""
    L = _lib.load()
    for d, h in list(_ctxs.items()):
        if device is None or d == device:
            with _ctx_locks.setdefault(d, threading.RLock()):     # not between another thread's count and emit calls
                check(L.fj_ctx_trim(h))
    check(L.fj_ctx_trim(None))                  # the context behind the NumPy entry


def set_option(name: str, value: int) -> None:
    """Process-wide dispatch option of the native library: "radix_threshold", "scalar_hbm_table" (include/flashjoin.h)."""
    check(_lib.load().fj_set_option(name.encode(), int(value)))


def get_option(name: str) -> int:
    v = int(_lib.load().fj_get_option(name.encode()))
    if v < 0:
        raise KeyError(_lib.last_error())
    return v


def last_timings() -> Optional[dict]:
    """Phase timings (ms) of the most recent join on this thread."""
    return _last.as_dict() if _last is not None else None


def _join_host(algo: int, bloom: int, materialize: int, bk, bv, pk, return_arrays: bool):
    global _last
    L = _lib.load()
    bk, bv, pk = _as_u64_host(bk, "build_keys"), _as_u64_host(bv, "build_values"), _as_u64_host(pk, "probe_keys")
    if bv.size < bk.size:
        raise ValueError(f"build_values has {bv.size} elements, build_keys has {bk.size}")
    cnt = ctypes.c_uint64(0)
    sec = ctypes.c_double(0.0)
    ok, ov = ctypes.c_void_p(), ctypes.c_void_p()
    want = bool(materialize and return_arrays)
    check(L.fj_join_host(algo, bloom, materialize, bk.ctypes.data, bv.ctypes.data, bk.size, pk.ctypes.data, pk.size,
                         ctypes.byref(cnt), ctypes.byref(sec),
                         ctypes.byref(ok) if want else None, ctypes.byref(ov) if want else None))
    t = FjTimings()
    L.fj_last_timings(ctypes.byref(t))
    _last = t
    n = int(cnt.value)
    if not want:
        return n, float(sec.value)
    try:
        if n:
            keys = np.ctypeslib.as_array(ctypes.cast(ok, ctypes.POINTER(ctypes.c_uint64)), shape=(n,)).copy()
            vals = np.ctypeslib.as_array(ctypes.cast(ov, ctypes.POINTER(ctypes.c_uint64)), shape=(n,)).copy()
        else:
            keys, vals = np.empty(0, np.uint64), np.empty(0, np.uint64)
    finally:
        L.fj_free_host(ok)
        L.fj_free_host(ov)
    return n, float(sec.value), keys, vals


def _dev_tensor(t, name: str):
    import torch
    if t.dtype not in (torch.int64, torch.uint64):
        raise TypeError(f"{name}: device tensors must be int64/uint64, got {t.dtype}")
    t = t.reshape(-1)
    if not t.is_contiguous():
        t = t.contiguous()
    if t.data_ptr() % 16:
        t = t.clone()
    return t


def _room_for(np_rows: int, dev: int) -> bool:
    """Output buffers for ANY result of a materialising join (16 bytes per probe row) are worth allocating when they take at
    most a third of the device memory that is free right now."""
    import torch
    try:
        free, _ = torch.cuda.mem_get_info(dev)
    except Exception:                                          # noqa: BLE001
        return False
    return 16 * np_rows <= free // 3


def join_device(algo: int, bloom: int, materialize: int, bk, bv, pk, return_arrays: bool = False,
                hash_top_bits: int = 64):
    """Device-resident join on torch ROCm tensors (int64 storage, bit-identical to uint64)."""
    global _last
    import torch
    L = _lib.load()
    bk, bv, pk = _dev_tensor(bk, "build_keys"), _dev_tensor(bv, "build_values"), _dev_tensor(pk, "probe_keys")
    if bv.numel() < bk.numel():
        raise ValueError(f"build_values has {bv.numel()} elements, build_keys has {bk.numel()}")
    dev = bk.device.index if bk.device.index is not None else torch.cuda.current_device()
    ctx = context(dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    cnt = ctypes.c_uint64(0)
    t = FjTimings()
    with _ctx_locks.setdefault(dev, threading.RLock()):      # count + emit are two calls on one context: keep other threads out
        out = None
        if (materialize and not (algo & ALGO_MANY_TO_MANY) and pk.numel() > 0 and bk.numel() > 0 and get_option("mat_single_pass")
                and _room_for(pk.numel(), dev)):      # (a many-to-many join can return more pairs than probe rows)
            # room for ANY result (the reference allocates the same, hash_join.cpp:330-334): the join may run in one pass over
            # the probe side; the pairs are the first n rows
            ok = torch.empty(pk.numel(), dtype=torch.int64, device=bk.device)
            ov = torch.empty(pk.numel(), dtype=torch.int64, device=bk.device)
            check(L.fj_join_device(ctx, algo, bloom, materialize, bk.data_ptr(), bv.data_ptr(), bk.numel(), pk.data_ptr(),
                                   pk.numel(), stream, hash_top_bits, ctypes.byref(cnt), ok.data_ptr(), ov.data_ptr(), pk.numel(), ctypes.byref(t)))
            n = int(cnt.value)
            _last = t
            if return_arrays:
                # the buffers hold room for ANY result; a view of their first n rows would keep 16 bytes per PROBE row alive for as
                # long as the caller keeps the pairs (16 GB at config 3 for 8 GB of pairs).  Unless the result fills most of them,
                # hand back exact-size copies (one device-to-device copy of n rows) and let the big buffers go.
                if n * 4 < pk.numel() * 3:
                    ok, ov = ok[:n].clone(), ov[:n].clone()
                else:
                    ok, ov = ok[:n], ov[:n]
                return n, t.total_ms * 1e-3, ok, ov
            return n, t.total_ms * 1e-3
        check(L.fj_join_device(ctx, algo, bloom, materialize, bk.data_ptr(), bv.data_ptr(), bk.numel(), pk.data_ptr(),
                               pk.numel(), stream, hash_top_bits, ctypes.byref(cnt), None, None, 0, ctypes.byref(t)))
        n = int(cnt.value)
        if materialize and n > 0:
            ok = torch.empty(n, dtype=torch.int64, device=bk.device)
            ov = torch.empty(n, dtype=torch.int64, device=bk.device)
            check(L.fj_emit_pairs(ctx, ok.data_ptr(), ov.data_ptr(), n, stream, ctypes.byref(t)))
            out = (ok, ov)
        elif materialize:
            out = (torch.empty(0, dtype=torch.int64, device=bk.device), torch.empty(0, dtype=torch.int64, device=bk.device))
    _last = t
    if materialize and return_arrays:
        return n, t.total_ms * 1e-3, out[0], out[1]
    return n, t.total_ms * 1e-3


_DL_DEVICE_GPU = (2, 10)         # DLPack device types kDLCUDA (what torch-ROCm reports) and kDLROCM


def _from_dlpack_if_device(x: Any):
    """Device arrays of other libraries (CuPy-ROCm, JAX, Arrow-on-GPU, ...) enter through DLPack, zero-copy."""
    if _is_torch_tensor(x) or isinstance(x, np.ndarray) or not hasattr(x, "__dlpack_device__"):
        return x
    try:
        dev_type = int(x.__dlpack_device__()[0])
    except Exception:
        return x
    if dev_type not in _DL_DEVICE_GPU:
        return x
    import torch
    return torch.from_dlpack(x)


def _join(algo: int, bloom: int, materialize: int, build_keys, build_values, probe_keys, return_arrays: bool):
    build_keys, build_values, probe_keys = (_from_dlpack_if_device(x) for x in (build_keys, build_values, probe_keys))
    if _is_torch_tensor(build_keys) and build_keys.is_cuda:
        return join_device(algo, bloom, materialize, build_keys, build_values, probe_keys, return_arrays)
    if _is_torch_tensor(build_keys):
        build_keys, build_values, probe_keys = (x.numpy() for x in (build_keys, build_values, probe_keys))
    return _join_host(algo, bloom, materialize, build_keys, build_values, probe_keys, return_arrays)


# ---- the reference's exported names (hash_join.cpp:603-639) -------------------------------------
def adaptive_join(build_keys, build_values, probe_keys, return_arrays: bool = False):
    """Adaptively chooses between scalar and radix join for materialization. (hash_join.cpp:603)"""
    return _join(ALGO_ADAPTIVE, 0, 1, build_keys, build_values, probe_keys, return_arrays)


def adaptive_join_bloom(build_keys, build_values, probe_keys, return_arrays: bool = False):
    """Adaptive join with bloom filter for materialization. (hash_join.cpp:607)"""
    return _join(ALGO_ADAPTIVE, 1, 1, build_keys, build_values, probe_keys, return_arrays)


def adaptive_join_count(build_keys, build_values, probe_keys):
    """Adaptively chooses between scalar and radix join for counting. (hash_join.cpp:611)"""
    return _join(ALGO_ADAPTIVE, 0, 0, build_keys, build_values, probe_keys, False)


def adaptive_join_count_bloom(build_keys, build_values, probe_keys):
    """Adaptive join with bloom filter for counting. (hash_join.cpp:615)"""
    return _join(ALGO_ADAPTIVE, 1, 0, build_keys, build_values, probe_keys, False)


def hash_join_radix(build_keys, build_values, probe_keys, return_arrays: bool = False):
    """Forces the use of radix join for materialization. (hash_join.cpp:621)"""
    return _join(ALGO_RADIX, 0, 1, build_keys, build_values, probe_keys, return_arrays)


def hash_join(build_keys, build_values, probe_keys, return_arrays: bool = False):
    """Forces the use of scalar (non-partitioned) join for materialization. (hash_join.cpp:624)"""
    return _join(ALGO_SCALAR, 0, 1, build_keys, build_values, probe_keys, return_arrays)


def hash_join_radix_bloom(build_keys, build_values, probe_keys, return_arrays: bool = False):
    """Radix join with bloom tables, materialization. (hash_join.cpp:627)"""
    return _join(ALGO_RADIX, 1, 1, build_keys, build_values, probe_keys, return_arrays)


def hash_join_bloom(build_keys, build_values, probe_keys, return_arrays: bool = False):
    """Scalar join with bloom precheck, materialization. (hash_join.cpp:628)"""
    return _join(ALGO_SCALAR, 1, 1, build_keys, build_values, probe_keys, return_arrays)


def hash_join_count_radix(build_keys, build_values, probe_keys):
    """Forces the use of radix join for counting. (hash_join.cpp:630)"""
    return _join(ALGO_RADIX, 0, 0, build_keys, build_values, probe_keys, False)


def hash_join_count(build_keys, build_values, probe_keys):
    """Forces the use of scalar (non-partitioned) join for counting. (hash_join.cpp:633)"""
    return _join(ALGO_SCALAR, 0, 0, build_keys, build_values, probe_keys, False)


def hash_join_count_radix_bloom(build_keys, build_values, probe_keys):
    """Radix join with bloom tables, counting. (hash_join.cpp:636)"""
    return _join(ALGO_RADIX, 1, 0, build_keys, build_values, probe_keys, False)


def hash_join_count_bloom(build_keys, build_values, probe_keys):
    """Scalar join with bloom precheck, counting. (hash_join.cpp:637)"""
    return _join(ALGO_SCALAR, 1, 0, build_keys, build_values, probe_keys, False)


# ---- extension: many-to-many inner join (the reference deduplicates build keys, hash_join.cpp:125) -------------------
def inner_join_count(build_keys, build_values, probe_keys):
    """Number of (probe row, build row) pairs with equal keys - every duplicate build row counts (SQL inner join).
    Not part of the reference's API; same argument and return conventions as the other joins."""
    return _join(ALGO_RADIX | ALGO_MANY_TO_MANY, 0, 0, build_keys, build_values, probe_keys, False)


def inner_join(build_keys, build_values, probe_keys, return_arrays: bool = False):
    """Materialises every (probe_key, build_value) pair of the many-to-many inner join; `return_arrays=True` returns them."""
    return _join(ALGO_RADIX | ALGO_MANY_TO_MANY, 0, 1, build_keys, build_values, probe_keys, return_arrays)


def sort_pairs(keys, values):
    """The pairs a `return_arrays=True` join handed back, in (key, value) order as unsigned 64-bit integers - the join's own
    output order is unspecified, so comparisons go through this.  NumPy arrays or device tensors."""
    if _is_torch_tensor(keys):
        import torch
        if keys.numel() == 0:
            return keys, values
        flip = torch.tensor(-(1 << 63), dtype=torch.int64, device=keys.device)     # int64 storage: order as uint64
        k, v = keys.reshape(-1) ^ flip, values.reshape(-1) ^ flip
        i = torch.argsort(v, stable=True)
        j = torch.argsort(k[i], stable=True)
        o = i[j]
        return keys.reshape(-1)[o], values.reshape(-1)[o]
    k, v = np.asarray(keys).reshape(-1).view(np.uint64), np.asarray(values).reshape(-1).view(np.uint64)
    o = np.lexsort((v, k))
    return k[o], v[o]


def initialize() -> None:
    """Replaces initialize_memory_system (hash_join.cpp:596, :639): checks that a HIP device is
    usable and warms up the native context. Returns None like the reference."""
    L = _lib.load()
    check(L.fj_initialize())
    context(0)
    return None


# benchmark.py's labels (benchmark.py:240-247) and BASELINE.json's wording, as aliases
flash_join = hash_join
flash_join_radix = hash_join_radix
flash_join_bloom = hash_join_bloom
flash_join_radix_bloom = hash_join_radix_bloom
adaptive_bloom = adaptive_join_bloom

REFERENCE_EXPORTS = [
    "adaptive_join", "adaptive_join_bloom", "adaptive_join_count", "adaptive_join_count_bloom",
    "hash_join_radix", "hash_join", "hash_join_radix_bloom", "hash_join_bloom",
    "hash_join_count_radix", "hash_join_count", "hash_join_count_radix_bloom", "hash_join_count_bloom",
    "initialize",
]
ALIASES = ["flash_join", "flash_join_radix", "flash_join_bloom", "flash_join_radix_bloom", "adaptive_bloom"]
EXTENSIONS = ["inner_join", "inner_join_count"]
__all__ = REFERENCE_EXPORTS + ALIASES + EXTENSIONS + ["last_timings", "join_device", "context", "set_option", "get_option", "sort_pairs", "workspace_bytes", "trim_workspace"]
