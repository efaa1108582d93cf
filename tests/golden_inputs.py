"""Seeded inputs shared by the golden-vector generator and the parity tests."""
import numpy as np

U64MAX = np.uint64(0xFFFFFFFFFFFFFFFF)


def _rand(rng, n):
    return rng.integers(0, 2**64, size=n, dtype=np.uint64)


def make_case(name):
    """Returns (build_keys, build_values, probe_keys) as uint64 arrays."""
    seed = {n: i + 1 for i, n in enumerate(CASES + BIG_CASES)}[name]
    rng = np.random.default_rng(1000 + seed)
    if name == "tiny":
        bk = np.array([5, 1, 9, 0, 2**64 - 1], dtype=np.uint64)
        bv = np.array([50, 10, 90, 7, 8], dtype=np.uint64)
        pk = np.array([1, 1, 2, 0, 2**64 - 1, 9, 42, 5, 5, 5], dtype=np.uint64)
    elif name == "small_unique_50":           # B <= 4096: LDS table without any partition pass
        bk = rng.permutation(np.arange(1, 3001, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15))
        bv = _rand(rng, bk.size)
        pk = np.concatenate([rng.choice(bk, 20000), _rand(rng, 20000)])
    elif name == "mid_unique_50":             # one radix pass
        bk = np.unique(_rand(rng, 200000))
        bv = _rand(rng, bk.size)
        pk = rng.permutation(np.concatenate([rng.choice(bk, 300000), _rand(rng, 300000)]))
    elif name == "two_pass_unique_5":         # > 4096*256 build rows: two radix passes, 5% hits
        bk = np.unique(_rand(rng, 1300000))
        bv = _rand(rng, bk.size)
        pk = rng.permutation(np.concatenate([rng.choice(bk, 100000), _rand(rng, 1900000)]))
    elif name == "dup_build_same_value":      # duplicate build keys (same value): dedup at insert (hash_join.cpp:125)
        base = np.unique(_rand(rng, 50000))
        bk = np.concatenate([base, base[:20000], base[:5000]])
        bv = bk ^ np.uint64(0x5555)
        p = rng.permutation(bk.size)
        bk, bv = bk[p], bv[p]
        pk = np.concatenate([rng.choice(base, 80000), _rand(rng, 40000)])
    elif name == "dup_probe_all_hit":         # every probe row hits, heavy probe duplicates
        bk = np.unique(_rand(rng, 10000))
        bv = _rand(rng, bk.size)
        pk = rng.choice(bk[:100], 150000)
    elif name == "all_miss":
        bk = np.arange(0, 100000, dtype=np.uint64) * np.uint64(2)
        bv = _rand(rng, bk.size)
        pk = np.arange(0, 150000, dtype=np.uint64) * np.uint64(2) + np.uint64(1)
    elif name == "extreme_keys":              # 0, 2^64-1 and int64-negative bit patterns on both sides
        special = np.array([0, 1, 2**63 - 1, 2**63, 2**64 - 2, 2**64 - 1], dtype=np.uint64)
        bk = np.concatenate([special, np.unique(_rand(rng, 70000))])
        bv = _rand(rng, bk.size)
        pk = np.concatenate([special, special, rng.choice(bk, 50000), _rand(rng, 50000)])
    elif name == "sequential_keys":           # dense small integers (db-benchmark J1 style ids)
        bk = rng.permutation(np.arange(1, 400001, dtype=np.uint64))
        bv = bk * np.uint64(3)
        pk = rng.integers(1, 800001, size=900000, dtype=np.uint64)
    elif name == "ragged_sizes":              # sizes that are not multiples of 2 / 16 / 256 / tile
        bk = np.unique(_rand(rng, 70001))[:65537]
        bv = _rand(rng, bk.size)
        pk = np.concatenate([rng.choice(bk, 33333), _rand(rng, 44444)])[:77777 - 2]
    elif name == "big_two_pass_wide":         # 21M x 50M rows: a two-pass plan and <= 3 probe rows per build row (the bucketed wide join kernel by default)
        rng = np.random.default_rng(77)
        bk = np.unique(_rand(rng, 21_000_000))
        bv = bk * np.uint64(0x9E3779B97F4A7C15) + np.uint64(12345)
        pk = np.concatenate([rng.choice(bk, 25_000_000), _rand(rng, 25_000_000)])
        rng.shuffle(pk)
    else:
        raise KeyError(name)
    return bk, bv, pk


# cases of tens of millions of rows: count + pair digest only, run once (not under every dispatch mode)
BIG_CASES = ["big_two_pass_wide"]

CASES = ["tiny", "small_unique_50", "mid_unique_50", "two_pass_unique_5", "dup_build_same_value",
         "dup_probe_all_hit", "all_miss", "extreme_keys", "sequential_keys", "ragged_sizes"]
