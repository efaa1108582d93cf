"""Round 6: why is c2's join (1M x 100M: 256 partitions from ONE pass) half as fast per probe key as c3's?  Join kernel time per
probe key over shapes with the same rows per partition but different plans.  usage: python tools/r6_c2_join_probe.py"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flash_hash_join_amd import api, datagen
api.initialize()
for nb, npk in ((1_000_000, 100_000_000), (1_000_000, 200_000_000), (1_000_000, 400_000_000), (4_000_000, 400_000_000),
                (16_000_000, 1_600_000_000), (100_000_000, 1_000_000_000), (2_000_000, 100_000_000)):
    bk, bv = datagen.build_device(nb, "cuda:0")
    pk, exp = datagen.probe_device(npk, nb, "cuda:0", seed=1, hit_bp=5000)
    best = None
    for it in range(6):
        n, sec = api.join_device(1, 0, 0, bk, bv, pk, return_arrays=False)[:2]
        assert n == exp
        t = api.last_timings()
        if it >= 2 and (best is None or t["join_ms"] < best["join_ms"]): best = dict(t)
    print(f"{nb:>11} x {npk:>11}: passes {best['passes']} bits {best['radix_bits']}  join {best['join_ms']:.3f} ms = {best['join_ms'] * 1e9 / npk:.2f} ps per probe key; "
          f"probe-side pass kernels {[round(x, 3) for x in best['probe_part_kernel_ms'][:best['passes']]]}  total {best['total_ms']:.3f}", flush=True)
    del bk, bv, pk
