"""Sweep build sizes: global-table (scalar) path vs radix path, to place the adaptive threshold on MI355X."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flash_hash_join_amd import api, datagen

api.initialize()
api.set_option("scalar_hbm_table", 1)          # ALGO_SCALAR = the literal one-table algorithm (default: the partitioned plan)
P = int(os.environ.get("P", 100_000_000))
print("B,P,algo,bloom,count_ok,total_ms,build_ms,probe_ms,passes,partitions")
for B in [1000, 4096, 8192, 32768, 131072, 262144, 524288, 1_000_000, 2_000_000, 4_000_000, 10_000_000, 30_000_000]:
    bk, bv = datagen.build_device(B, "cuda:0")
    for hit_bp in (5000, 500):
        pk, exp = datagen.probe_device(P, B, "cuda:0", seed=1, hit_bp=hit_bp)
        for algo, bloom in ((api.ALGO_SCALAR, 0), (api.ALGO_SCALAR, 1), (api.ALGO_RADIX, 0)):
            best = None
            for _ in range(3):
                n, sec = api.join_device(algo, bloom, 0, bk, bv, pk)
                t = api.last_timings()
                if best is None or t["total_ms"] < best["total_ms"]:
                    best = t
            print(f"{B},{P},{['adaptive','scalar','radix'][algo]},{bloom},{hit_bp},{n == exp},{best['total_ms']:.3f},{best['build_phase_ms']:.3f},{best['probe_phase_ms']:.3f},{best['passes']},{best['partitions']}")
        del pk
    del bk, bv
    torch.cuda.empty_cache()
