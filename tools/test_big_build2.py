import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flash_hash_join_amd import api, datagen
api.initialize()
nb, npk = 300_000_000, 400_000_000
bk, bv = datagen.build_device(nb, "cuda:0")
pk, exp = datagen.probe_device(npk, nb, "cuda:0", seed=2, hit_bp=5000)
for _ in range(2):
    n, s = api.join_device(api.ALGO_RADIX, 0, 0, bk, bv, pk)
print(n == exp, api.last_timings())
