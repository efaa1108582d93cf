"""CPU tests of the synthetic generators: closed-form expected counts."""
import numpy as np

from flash_hash_join_amd import datagen


def test_build_keys_are_unique_and_bijective():
    k, v = datagen.build_numpy(100000)
    assert np.unique(k).size == k.size
    assert v.tolist()[:3] == [0, 1, 2]
    k2, v2 = datagen.build_numpy(10, first=5)
    assert np.array_equal(k2, k[5:15]) and np.array_equal(v2, v[5:15])              # sharded generation is consistent


def test_probe_expected_count_is_closed_form(oracle):
    B = 5000
    bk, bv = datagen.build_numpy(B)
    for hit_bp in (0, 500, 5000, 10000):
        pk, exp = datagen.probe_numpy(60000, B, seed=1, hit_bp=hit_bp)
        assert oracle.np_join(bk, bv, pk) == exp
        assert abs(exp - 60000 * hit_bp / 10000) <= 5 * (60000 ** 0.5)
    a, ea = datagen.probe_numpy(1000, B, first=0)
    b, eb = datagen.probe_numpy(500, B, first=500)
    assert np.array_equal(a[500:], b)
