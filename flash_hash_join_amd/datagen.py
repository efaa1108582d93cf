"""Deterministic synthetic relations, identical in NumPy and on the device
(fj_generate_build / fj_generate_probe in csrc/fj_join.hip).

    build_keys[i] = (i+1)*M,  build_values[i] = i            M = 0x9E3779B97F4A7C15 (odd => bijection)
    probe j: r = 1 + mix(seed, j) % B;  hit = mix(seed^1, j) % 10000 < hit_bp
             key = (r + (0 if hit else B)) * M                (misses map to ids B+1..2B)
    expected match count = number of hits (closed form, scale-free).
"""
from __future__ import annotations

import ctypes
from typing import Tuple

import numpy as np

M = np.uint64(0x9E3779B97F4A7C15)
_C0 = np.uint64(0xD6E8FEB86659FD93)
_C1 = np.uint64(0xBF58476D1CE4E5B9)
_C2 = np.uint64(0x94D049BB133111EB)


def mix(seed: int, j: np.ndarray) -> np.ndarray:
    """splitmix64-style counter hash; must match fj_mix in csrc/fj_common.h."""
    with np.errstate(over="ignore"):
        z = np.uint64(seed) * _C0 + j.astype(np.uint64) + M
        z = (z ^ (z >> np.uint64(30))) * _C1
        z = (z ^ (z >> np.uint64(27))) * _C2
        return z ^ (z >> np.uint64(31))


def build_numpy(n: int, first: int = 0) -> Tuple[np.ndarray, np.ndarray]:
    i = np.arange(first, first + n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        return (i + np.uint64(1)) * M, i.copy()


def probe_numpy(n: int, build_total: int, seed: int = 1, hit_bp: int = 5000, first: int = 0) -> Tuple[np.ndarray, int]:
    j = np.arange(first, first + n, dtype=np.uint64)
    r = np.uint64(1) + mix(seed, j) % np.uint64(build_total)
    hit = (mix(seed ^ 1, j) % np.uint64(10000)) < np.uint64(hit_bp)
    with np.errstate(over="ignore"):
        keys = (r + np.where(hit, np.uint64(0), np.uint64(build_total))) * M
    return keys, int(hit.sum())


def build_device(n: int, device, first: int = 0):
    """Generate the build relation directly in HBM. Returns (keys, values) int64 torch tensors."""
    import torch
    from . import _lib, api
    L = _lib.load()
    dev = torch.device(device)
    k = torch.empty(n, dtype=torch.int64, device=dev)
    v = torch.empty(n, dtype=torch.int64, device=dev)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    _lib.check(L.fj_generate_build(api.context(idx), k.data_ptr(), v.data_ptr(), first, n,
                                   torch.cuda.current_stream(idx).cuda_stream))
    return k, v


def probe_device(n: int, build_total: int, device, seed: int = 1, hit_bp: int = 5000, first: int = 0):
    """Generate probe keys in HBM. Returns (keys int64 tensor, expected match count)."""
    import torch
    from . import _lib, api
    L = _lib.load()
    dev = torch.device(device)
    k = torch.empty(n, dtype=torch.int64, device=dev)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    exp = ctypes.c_uint64(0)
    _lib.check(L.fj_generate_probe(api.context(idx), k.data_ptr(), first, n, build_total, seed, hit_bp,
                                   ctypes.byref(exp), torch.cuda.current_stream(idx).cuda_stream))
    return k, int(exp.value)
