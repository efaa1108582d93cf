"""Plans beyond 16 radix bits (build sides > 268M rows: 512-bucket passes; > 1.07G rows: three passes): count and
materialised pairs against the generator's closed form."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flash_hash_join_amd import api, datagen
api.initialize()
for nb, npk in [(300_000_000, 400_000_000), (600_000_000, 200_000_000), (1_300_000_000, 100_000_000)]:
    bk, bv = datagen.build_device(nb, "cuda:0")
    pk, exp = datagen.probe_device(npk, nb, "cuda:0", seed=2, hit_bp=5000)
    n, s = api.join_device(api.ALGO_RADIX, 0, 0, bk, bv, pk)
    t = api.last_timings()
    print(nb, npk, n, exp, n == exp, "bits", t["radix_bits"], "passes", t["passes"], "ms", round(t["total_ms"], 2), "fell_back", t["fell_back"])
    n, s, k, v = api.join_device(api.ALGO_RADIX, 0, 1, bk, bv, pk, return_arrays=True)
    M = torch.tensor(-7046029254386353131, dtype=torch.int64, device="cuda:0")
    print("  materialize", n == exp, bool(torch.all((v + 1) * M == k)))
    del bk, bv, pk, k, v; torch.cuda.empty_cache()
