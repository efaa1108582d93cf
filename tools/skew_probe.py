import sys; sys.path.insert(0, "/root/repo")
import torch, flash_join as fj
from flash_hash_join_amd import datagen
nb, npk = 20_000_000, 400_000_000
dbk, dbv = datagen.build_device(nb, "cuda:0")
dpk, exp = datagen.probe_device(npk, nb, "cuda:0", seed=5, hit_bp=5000)
for i in range(2): n,_ = fj.hash_join_count_radix(dbk, dbv, dpk)
print("uniform", fj.last_timings())
for frac in (0.01, 0.1, 0.5):
    d2 = dpk.clone()
    step = int(1/frac)
    d2[::step] = dbk[12345]
    for i in range(2): n,_ = fj.hash_join_count_radix(dbk, dbv, d2)
    t = fj.last_timings()
    print("hot fraction", frac, round(t["total_ms"],2), [round(x,2) for x in t["probe_part_kernel_ms"]], round(t["join_ms"],2))
