import sys; sys.path.insert(0, "/root/repo")
import torch
from flash_hash_join_amd import api, datagen
api.initialize()
for nb, npk in [(3_000_000, 20_000_000), (20_000_000, 100_000_000), (50_000_000, 200_000_000), (100_000_000, 300_000_000)]:
    bk, bv = datagen.build_device(nb, "cuda:0")
    pk, exp = datagen.probe_device(npk, nb, "cuda:0", seed=1, hit_bp=5000)
    for top in (64, 48):
        n, s = api.join_device(api.ALGO_RADIX, 0, 0, bk, bv, pk, hash_top_bits=top)
        t = api.last_timings()
        print(nb, npk, top, n, exp, n == exp, t["radix_bits"], t["passes"], t["fell_back"])
    del bk, bv, pk; torch.cuda.empty_cache()
