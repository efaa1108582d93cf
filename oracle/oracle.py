"""CPU ORACLE wrapper -- test infrastructure, NOT the product.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
Two independent checkers live here:

* `c_join(...)`   -- ctypes binding of oracle/flashjoin_oracle.c, the plain-C restatement
                     of the reference algorithm (hash_join.cpp:38-594).
* `np_join(...)`  -- a NumPy set-membership oracle (sort + searchsorted) that shares no
                     code with either the C restatement or the HIP path.

Parity status: "unpinned" by the reference itself (it has no tests and cannot be built
here without a stand-in for mimalloc.h); pinned against CRC-32C's published check value,
SURVEY.md App. A.5 known answers and `np_join` (tests/test_oracle.py).
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libflashjoin_oracle.so")

ALGO = {"adaptive": 0, "scalar": 1, "radix": 2}


def build(force: bool = False) -> str:
    """Compile the C oracle with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "flashjoin_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


_lib = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        u64, sz, u32, u16 = ctypes.c_uint64, ctypes.c_size_t, ctypes.c_uint32, ctypes.c_uint16
        p64 = ctypes.POINTER(u64)
        L.fjo_hash64.restype = u64; L.fjo_hash64.argtypes = [u64]
        L.fjo_crc32c_sw.restype = u32; L.fjo_crc32c_sw.argtypes = [u32, u64]
        L.fjo_crc32c_buf.restype = u32; L.fjo_crc32c_buf.argtypes = [ctypes.c_char_p, sz]
        L.fjo_uses_hw_crc.restype = ctypes.c_int
        L.fjo_bloom_mask.restype = u16; L.fjo_bloom_mask.argtypes = [u64]
        L.fjo_tags_table.restype = u16; L.fjo_tags_table.argtypes = [u32]
        L.fjo_capacity.restype = sz; L.fjo_capacity.argtypes = [sz]
        L.fjo_default_threads.restype = ctypes.c_int
        L.fjo_partition.restype = ctypes.c_int
        L.fjo_partition.argtypes = [p64, p64, sz, ctypes.c_int, p64, p64, ctypes.POINTER(sz)]
        L.fjo_join.restype = ctypes.c_int
        L.fjo_join.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, p64, p64, sz, p64, sz, ctypes.c_int,
                               p64, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(p64), ctypes.POINTER(p64)]
        L.fjo_free.restype = None; L.fjo_free.argtypes = [ctypes.c_void_p]
        _lib = L
    return _lib


def _as_u64(a) -> np.ndarray:
    a = np.ascontiguousarray(a)
    if a.dtype == np.int64:
        a = a.view(np.uint64)
    elif a.dtype != np.uint64:
        a = a.astype(np.uint64)
    return a.reshape(-1)


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64))


def hash64(key: int) -> int:
    return int(lib().fjo_hash64(ctypes.c_uint64(key & 0xFFFFFFFFFFFFFFFF)))


def capacity(build_size: int) -> int:
    return int(lib().fjo_capacity(build_size))


def partition(keys, vals=None, threads: int = 0):
    """256-way stable radix partition (hash_join.cpp:209-292). Returns (keys, vals|None, offsets[257])."""
    k = _as_u64(keys)
    v = None if vals is None else _as_u64(vals)
    ok = np.empty_like(k)
    ov = None if v is None else np.empty_like(v)
    off = np.zeros(257, dtype=np.uintp)
    rc = lib().fjo_partition(_ptr(k), None if v is None else _ptr(v), k.size, threads, _ptr(ok),
                             None if ov is None else _ptr(ov), off.ctypes.data_as(ctypes.POINTER(ctypes.c_size_t)))
    if rc:
        raise MemoryError("oracle partition failed")
    return ok, ov, off


def c_join(build_keys, build_values, probe_keys, algo: str = "adaptive", bloom: bool = False,
           materialize: bool = False, threads: int = 0, return_arrays: bool = False):
    """Run the C restatement. Returns (count, seconds) or (count, seconds, keys, values)."""
    bk, bv, pk = _as_u64(build_keys), _as_u64(build_values), _as_u64(probe_keys)
    if bv.size < bk.size:
        raise ValueError("build_values shorter than build_keys")
    cnt = ctypes.c_uint64(0)
    sec = ctypes.c_double(0.0)
    pk_out = ctypes.POINTER(ctypes.c_uint64)()
    pv_out = ctypes.POINTER(ctypes.c_uint64)()
    want = bool(materialize and return_arrays)
    rc = lib().fjo_join(ALGO[algo], int(bool(bloom)), int(bool(materialize)), _ptr(bk), _ptr(bv), bk.size,
                        _ptr(pk), pk.size, threads, ctypes.byref(cnt), ctypes.byref(sec),
                        ctypes.byref(pk_out) if want else None, ctypes.byref(pv_out) if want else None)
    if rc:
        raise MemoryError("oracle join failed")
    if not want:
        return int(cnt.value), float(sec.value)
    n = int(cnt.value)
    try:
        keys = np.ctypeslib.as_array(pk_out, shape=(max(n, 1),))[:n].copy()
        vals = np.ctypeslib.as_array(pv_out, shape=(max(n, 1),))[:n].copy()
    finally:
        lib().fjo_free(pk_out)
        lib().fjo_free(pv_out)
    return n, float(sec.value), keys, vals


def np_join(build_keys, build_values, probe_keys, return_arrays: bool = False):
    """Independent NumPy oracle: first occurrence of a duplicate build key wins
    (the reference's radix path, hash_join.cpp:125 + stable partition)."""
    bk, bv, pk = _as_u64(build_keys), _as_u64(build_values), _as_u64(probe_keys)
    if bk.size == 0 or pk.size == 0:
        e = np.empty(0, dtype=np.uint64)
        return (0, e, e.copy()) if return_arrays else 0
    order = np.argsort(bk, kind="stable")
    sk, sv = bk[order], bv[order]
    first = np.ones(sk.size, dtype=bool)
    first[1:] = sk[1:] != sk[:-1]
    uk, uv = sk[first], sv[first]
    pos = np.searchsorted(uk, pk)
    pos_c = np.minimum(pos, uk.size - 1)
    hit = uk[pos_c] == pk
    if not return_arrays:
        return int(hit.sum())
    return int(hit.sum()), pk[hit], uv[pos_c[hit]]


def np_inner_join(build_keys, build_values, probe_keys, return_arrays: bool = False):
    """NumPy oracle of the many-to-many extension (no reference counterpart: the reference dedups build keys): every build
    row with the probe row's key yields a pair.  Sort + searchsorted ranges + repeat."""
    bk, bv, pk = _as_u64(build_keys), _as_u64(build_values), _as_u64(probe_keys)
    if bk.size == 0 or pk.size == 0:
        e = np.empty(0, dtype=np.uint64)
        return (0, e, e.copy()) if return_arrays else 0
    order = np.argsort(bk, kind="stable")
    sk, sv = bk[order], bv[order]
    lo = np.searchsorted(sk, pk, side="left")
    hi = np.searchsorted(sk, pk, side="right")
    mult = (hi - lo).astype(np.int64)
    total = int(mult.sum())
    if not return_arrays:
        return total
    out_k = np.repeat(pk, mult)
    starts = np.repeat(lo, mult)
    within = np.arange(total, dtype=np.int64) - np.repeat(np.cumsum(mult) - mult, mult)
    return total, out_k, sv[starts + within]


def canon_pairs(keys, vals) -> Tuple[np.ndarray, np.ndarray]:
    """Sort (key, value) pairs so two outputs can be compared modulo order."""
    k, v = _as_u64(keys), _as_u64(vals)
    o = np.lexsort((v, k))
    return k[o], v[o]
