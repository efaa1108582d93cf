"""Timing probe of the many-to-many extension: 20M build rows (5M distinct keys) x 200M probe rows, next to the N:1 joins."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import flash_join
flash_join.initialize()
from flash_hash_join_amd import datagen
nb, dom, npk = 20_000_000, 5_000_000, 200_000_000
ids = torch.randint(0, dom, (nb,), device="cuda", dtype=torch.int64)
bk = ids * -7046029254386353131
bv = torch.arange(nb, device="cuda", dtype=torch.int64)
pk = torch.randint(0, 2 * dom, (npk,), device="cuda", dtype=torch.int64) * -7046029254386353131
for fn in ("inner_join_count", "inner_join", "hash_join_count_radix", "hash_join_radix"):
    for _ in range(3):
        n, sec = getattr(flash_join, fn)(bk, bv, pk)
    print(fn, n, round(sec * 1e3, 3), "ms", flash_join.last_timings()["partitions"])
