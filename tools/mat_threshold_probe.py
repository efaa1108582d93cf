import sys, os
sys.path.insert(0, '/root/repo')
import torch, flash_join
from flash_hash_join_amd import datagen, api
flash_join.initialize()
for nb, npk in ((1_000_000, 100_000_000), (4_000_000, 100_000_000), (20_000_000, 200_000_000)):
    bk, bv = datagen.build_device(nb, "cuda:0")
    pk, exp = datagen.probe_device(npk, nb, "cuda:0", seed=1, hit_bp=5000)
    for pm in (8192, 512, 1):
        api.set_option("persistent_min_items", pm)
        best = 1e9
        for _ in range(6):
            n, sec = flash_join.hash_join_radix(bk, bv, pk)
            assert n == exp
            best = min(best, sec)
        lt = flash_join.last_timings()
        print(nb, npk, "persistent_min_items", pm, f"{best*1e3:.3f} ms", "emit", round(lt["emit_ms"], 3), "partitions", lt["partitions"], flush=True)
    api.set_option("persistent_min_items", 8192)
