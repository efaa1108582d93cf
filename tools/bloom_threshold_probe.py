"""Where does the bloom precheck stop paying?  100M x 1B, hash_join_count_radix vs hash_join_count_radix_bloom by hit rate
(the measurement behind bloom_auto_max_hit_bp)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import flash_join
flash_join.initialize()
from flash_hash_join_amd import datagen
nb, npk = 100_000_000, 1_000_000_000
bk, bv = datagen.build_device(nb, "cuda:0")
print("hit_bp,plain_ms,bloom_ms,survivors")
for hit_bp in (500, 1000, 2000, 2500, 3000, 3500, 4000, 5000):
    pk, exp = datagen.probe_device(npk, nb, "cuda:0", seed=1, hit_bp=hit_bp)
    out = []
    for fn in ("hash_join_count_radix", "hash_join_count_radix_bloom"):
        best = None
        for _ in range(4):
            n, sec = getattr(flash_join, fn)(bk, bv, pk)
            assert n == exp
            best = sec if best is None else min(best, sec)
        out.append(best * 1e3)
    print(f"{hit_bp},{out[0]:.3f},{out[1]:.3f},{flash_join.last_timings()['filter_survivors']}", flush=True)
    del pk
