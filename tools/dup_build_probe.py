#!/usr/bin/env python3
"""Materialising join of a build side with duplicate keys (first-occurrence path) next to a unique-key build of the same size."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, flash_join
flash_join.initialize()
dev = "cuda:0"
nb, npk = 50_000_000, 200_000_000
g = torch.Generator(device=dev); g.manual_seed(1)
for dom in (nb * 4, 35_000_000, 5_000_000):
    domain = torch.randint(-(1 << 62), 1 << 62, (dom,), dtype=torch.int64, device=dev, generator=g)
    if dom >= nb * 4:
        bk = domain[:nb].clone()                       # (practically) unique
    else:
        bk = domain[torch.randint(0, dom, (nb,), device=dev, generator=g)]
    bv = torch.arange(nb, dtype=torch.int64, device=dev)
    pk = torch.where(torch.rand(npk, device=dev, generator=g) < 0.5, domain[torch.randint(0, dom, (npk,), device=dev, generator=g)],
                     torch.randint(-(1 << 62), 1 << 62, (npk,), dtype=torch.int64, device=dev, generator=g))
    del domain
    best_c = best_m = 1e9
    for _ in range(4):
        n, sec = flash_join.hash_join_count_radix(bk, bv, pk); best_c = min(best_c, sec)
        m, sec = flash_join.hash_join_radix(bk, bv, pk); best_m = min(best_m, sec)
        assert n == m
    lt = flash_join.last_timings()
    print(f"distinct ~{min(dom, nb)}: count {best_c*1e3:.2f} ms, materialise {best_m*1e3:.2f} ms (emit {lt['emit_ms']:.2f}), pairs {m}", flush=True)
    del bk, bv, pk
