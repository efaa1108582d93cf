"""NumPy restatement of the device's key mixer (csrc/fj_common.h fj_key_mix / fj_key_unmix; C ABI fj_key_mix64 / fj_key_unmix64)
and of the owner shuffle's 7-byte wire format (csrc/fj_pack.hip) - test infrastructure: what the tests check placements and
packed chunks against.  tests/test_abi.py pins these functions to the library's own."""
import numpy as np

_M32 = np.uint64(0xFFFFFFFF)


def _fmix32(x):
    x = x.astype(np.uint64)
    x ^= x >> np.uint64(16); x = (x * np.uint64(0x85EBCA6B)) & _M32
    x ^= x >> np.uint64(13); x = (x * np.uint64(0xC2B2AE35)) & _M32
    x ^= x >> np.uint64(16)
    return x


def _fmix32_inv(x):
    x = x.astype(np.uint64)
    x ^= x >> np.uint64(16); x = (x * np.uint64(0x7ED1B41D)) & _M32
    x ^= (x >> np.uint64(13)) ^ (x >> np.uint64(26)); x = (x * np.uint64(0xA5CB9243)) & _M32
    x ^= x >> np.uint64(16)
    return x


def mix(k):
    """fj_key_mix64 of a uint64 array: high word = radix digits / owner (hash word 1), low word = table slots (hash word 2)."""
    k = np.asarray(k).astype(np.uint64)
    a = ((k & _M32) * np.uint64(0x9E3779B1)) & _M32
    b = ((k >> np.uint64(32)) * np.uint64(0x85EBCA77)) & _M32
    w1 = _fmix32(a ^ b)
    w2 = _fmix32((a + w1) & _M32)
    return (w1 << np.uint64(32)) | w2


def unmix(h):
    h = np.asarray(h).astype(np.uint64)
    w1, w2 = h >> np.uint64(32), h & _M32
    a = (_fmix32_inv(w2) - w1) & _M32
    b = _fmix32_inv(w1) ^ a
    lo = (a * np.uint64(0x0E8B2F51)) & _M32
    hi = (b * np.uint64(0xB6C92F47)) & _M32
    return (hi << np.uint64(32)) | lo


def hash_w1(k):
    """hash word 1 of raw keys (csrc/fj_common.h fj_hash_w1) as uint32"""
    return (mix(k) >> np.uint64(32)).astype(np.uint32)


def unpack_wire(chunks_u8, dirw, fan_log0):
    """Wire chunks (uint8 array) + directory words -> (raw keys, bucket of every key, keys per chunk).  chunk_bytes = 1792 when
    fan_log0 >= 8 (three planes + the bucket's top bits), else 2048 (whole mixed keys)."""
    d = np.asarray(dirw).astype(np.int64) & 0xFFFFFFFF
    n = d.size
    bucket, cnt = d >> 9, d & 0x1FF
    if fan_log0 >= 8:
        c = np.asarray(chunks_u8, dtype=np.uint8).reshape(n, 1792)
        lo = c[:, :1024].copy().view(np.uint32).astype(np.uint64)
        mid = c[:, 1024:1536].copy().view(np.uint16).astype(np.uint64)
        hi = c[:, 1536:].astype(np.uint64)
        top = (bucket >> (fan_log0 - 8)).astype(np.uint64)[:, None]
        h = (top << np.uint64(56)) | (hi << np.uint64(48)) | (mid << np.uint64(32)) | lo
    else:
        h = np.asarray(chunks_u8, dtype=np.uint8).reshape(n, 2048).copy().view(np.uint64)
    mask = np.arange(256)[None, :] < cnt[:, None]
    return unmix(h[mask]), np.repeat(bucket, cnt), cnt
