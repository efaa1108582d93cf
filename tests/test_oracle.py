"""CPU tests: pin the oracle (oracle/) against every known answer available for this path.

The reference has no tests or fixtures and is unbuildable here (mimalloc),
so the pins are: the published CRC-32C check value, SURVEY App. A.3/A.5 known answers, the
committed golden vectors, and an independent NumPy set oracle."""
import hashlib
import json
import os

import numpy as np
import pytest

from golden_inputs import CASES, make_case

HERE = os.path.dirname(os.path.abspath(__file__))


def _kat():
    with open(os.path.join(HERE, "golden", "kat.json")) as f:
        return json.load(f)


def _golden():
    with open(os.path.join(HERE, "golden", "golden_joins.json")) as f:
        return json.load(f)


def test_crc32c_published_check_value(oracle):
    L = oracle.lib()
    k = _kat()["crc32c_check"]
    assert L.fjo_crc32c_buf(k["input"].encode(), len(k["input"])) == int(k["value"], 16)


def test_hash64_known_answers(oracle):
    L = oracle.lib()
    for row in _kat()["hash64"]:
        key, h = int(row["key"], 16), int(row["hash"], 16)
        assert oracle.hash64(key) == h                                   # hash_join.cpp:40-44
        assert L.fjo_crc32c_sw(0xAAAAAAAA, key) == (h & 0xFFFFFFFF)      # software CRC == SSE4.2 instruction
        assert (h >> 56) == row["partition"]                             # hash_join.cpp:209
        assert ((h & 0xFFFFFFFF) >> 21) == row["bloom_idx"]              # hash_join.cpp:183
        assert (h >> 32) == ((h & 0xFFFFFFFF) * 0x8648DBDB) & 0xFFFFFFFF


def test_bloom_tag_table_known_answers(oracle):
    L = oracle.lib()
    for i, v in _kat()["tags_table"].items():
        assert L.fjo_tags_table(int(i)) == int(v, 16)
    for i in range(2048):                                                # 1..4 bits of 16 (hash_join.cpp:64-71)
        assert 1 <= bin(L.fjo_tags_table(i)).count("1") <= 4


def test_capacity_known_answers(oracle):
    for b, cap in _kat()["capacity"].items():
        assert oracle.capacity(int(b)) == cap                            # hash_join.cpp:96-99


def test_partition_is_stable_and_complete(oracle):
    rng = np.random.default_rng(7)
    keys = rng.integers(0, 2**64, size=50000, dtype=np.uint64)
    vals = np.arange(keys.size, dtype=np.uint64)
    for threads in (1, 3, 8):
        ok, ov, off = oracle.partition(keys, vals, threads=threads)
        assert off[0] == 0 and off[256] == keys.size
        assert np.array_equal(keys[ov.astype(np.int64)], ok)             # values travel with their keys
        for p in (0, 93, 255):
            seg = ov[off[p]:off[p + 1]]
            assert np.all(np.diff(seg.astype(np.int64)) > 0)             # original order kept (hash_join.cpp:227-233)
            assert all((oracle.hash64(int(k)) >> 56) == p for k in ok[off[p]:off[p] + 20])


@pytest.mark.parametrize("name", CASES)
def test_golden_vectors(oracle, name):
    """C restatement and NumPy oracle both reproduce the committed golden vectors, all variants."""
    g = _golden()[name]
    bk, bv, pk = make_case(name)
    assert (bk.size, pk.size) == (g["nb"], g["np"])
    n, ek, ev = oracle.np_join(bk, bv, pk, return_arrays=True)
    assert n == g["count"]

    def digest(k, v):
        k, v = oracle.canon_pairs(k, v)
        return hashlib.sha256(k.tobytes() + v.tobytes()).hexdigest()

    assert digest(ek, ev) == g["pairs_sha256"]
    for algo in ("adaptive", "scalar", "radix"):
        for bloom in (False, True):
            assert oracle.c_join(bk, bv, pk, algo=algo, bloom=bloom)[0] == g["count"]
    n2, _, ck, cv = oracle.c_join(bk, bv, pk, algo="radix", materialize=True, return_arrays=True)
    assert n2 == g["count"] and digest(ck, cv) == g["pairs_sha256"]


def test_duplicate_build_keys_first_occurrence_wins_in_radix_path(oracle):
    """radix path keeps the first occurrence's value (stable partition + insert_local)."""
    rng = np.random.default_rng(3)
    base = rng.integers(0, 2**64, size=5000, dtype=np.uint64)
    bk = np.concatenate([base, base])
    bv = np.concatenate([np.arange(5000, dtype=np.uint64), np.arange(5000, dtype=np.uint64) + np.uint64(10**6)])
    n, _, k, v = oracle.c_join(bk, bv, base, algo="radix", materialize=True, threads=4, return_arrays=True)
    assert n == 5000
    assert np.all(v < 10**6)
    assert oracle.np_join(bk, bv, base) == 5000


def test_empty_inputs(oracle):
    e = np.empty(0, dtype=np.uint64)
    one = np.array([1], dtype=np.uint64)
    for algo in ("scalar", "radix", "adaptive"):
        assert oracle.c_join(e, e, one, algo=algo)[0] == 0
        assert oracle.c_join(one, one, e, algo=algo)[0] == 0
        assert oracle.c_join(e, e, e, algo=algo)[0] == 0


def test_int64_negative_keys_are_reinterpreted(oracle):
    bk = np.array([-1, -2, 5], dtype=np.int64)
    bv = np.array([1, 2, 3], dtype=np.int64)
    pk = np.array([-1, 5, -3, 2**63 - 1], dtype=np.int64)
    assert oracle.c_join(bk, bv, pk, algo="scalar")[0] == 2
    assert oracle.np_join(bk, bv, pk) == 2


def test_adaptive_threshold(oracle):
    """B < 1,000,000 -> scalar, else radix (hash_join.cpp:576-594); both must agree on the count."""
    rng = np.random.default_rng(11)
    for nb in (999_999, 1_000_000):
        bk = np.arange(1, nb + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)
        pk = np.concatenate([bk[:5000], rng.integers(0, 2**64, size=5000, dtype=np.uint64)])
        assert oracle.c_join(bk, bk, pk, algo="adaptive")[0] == oracle.np_join(bk, bk, pk)


def test_baseline_config1_on_cpu(oracle):
    """BASELINE.json configs[0]: scalar path on CPU, 1M build x 10M probe int64 keys, 50 % hit rate (plumbing, no GPU):
    the oracle's restatement, every variant, against the generator's closed-form count."""
    from flash_hash_join_amd import datagen
    bk, bv = datagen.build_numpy(1_000_000)
    pk, exp = datagen.probe_numpy(10_000_000, 1_000_000, seed=1, hit_bp=5000)
    assert abs(exp - 5_000_000) < 20_000
    for algo in ("scalar", "radix", "adaptive"):
        for bloom in (False, True):
            assert oracle.c_join(bk.view(np.int64), bv, pk, algo=algo, bloom=bloom)[0] == exp


def test_numpy_oracle_of_the_many_to_many_extension():
    """np_inner_join against a brute-force double loop on small inputs with duplicates on both sides."""
    from oracle import oracle as O
    rng = np.random.default_rng(5)
    for nb, npk, dom in [(1, 1, 2), (40, 70, 9), (300, 500, 40), (0, 5, 3), (5, 0, 3)]:
        bk = rng.integers(0, dom, size=nb, dtype=np.uint64)
        bv = rng.integers(0, 2**64, size=nb, dtype=np.uint64)
        pk = rng.integers(0, dom + 2, size=npk, dtype=np.uint64)
        exp = sorted((int(k), int(v)) for k in pk for kb, v in zip(bk, bv) if kb == k)
        n, ok, ov = O.np_inner_join(bk, bv, pk, return_arrays=True)
        assert n == len(exp) == O.np_inner_join(bk, bv, pk)
        assert sorted(zip(ok.tolist(), ov.tolist())) == exp
