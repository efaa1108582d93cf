"""Round 6: random counting joins in the shapes that take the bucketed wide join (fj_count_join_wide: fewer than 3 probe rows per
build row, or join_wide=1), duplicate build keys in many multiplicities, keys of the first and the last partition, hit rates from
0 to 100 % - every count against torch.isin.  usage: python tools/r6_wide_fuzz.py [cases=40] [seed=1]"""
import os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flash_hash_join_amd import api

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
api.initialize()
dev = "cuda:0"
t0 = time.time()
for c in range(cases):
    nb = int(10 ** rng.uniform(5.5, 7.9))
    npk = int(nb * rng.choice([0.3, 1.0, 2.0, 2.9, 6.0]))
    kind = rng.choice(["random", "sequential", "dups", "fewdistinct", "highword"])
    g = torch.Generator(device=dev); g.manual_seed(rng.randrange(1 << 30))
    if kind == "sequential":
        bk = torch.arange(1, nb + 1, device=dev, dtype=torch.int64) * rng.choice([1, 3, 1 << 20])
    elif kind == "highword":
        bk = torch.arange(1, nb + 1, device=dev, dtype=torch.int64) << 32
    else:
        bk = torch.randint(-2**62, 2**62, (nb,), device=dev, dtype=torch.int64, generator=g)
    if kind == "dups":                                         # every key 1 .. 300 times
        m = rng.choice([2, 7, 60, 300])
        bk = bk[: max(1, nb // m)].repeat(m)[:nb].contiguous()
    if kind == "fewdistinct":                                  # a handful of keys, thousands of copies each (the walk's limit: retry ladder)
        d = rng.choice([1, 5, 1000])
        bk = bk[:d].repeat(nb // d + 1)[:nb].contiguous()
    nb = int(bk.numel())                                       # (repeat() may have come out a little short)
    bv = torch.arange(nb, device=dev, dtype=torch.int64)
    hit = rng.choice([0.0, 0.05, 0.5, 1.0])
    idx = torch.randint(0, nb, (npk,), device=dev, generator=g)
    miss = torch.randint(-2**62, 2**62, (npk,), device=dev, dtype=torch.int64, generator=g)
    pk = torch.where(torch.rand(npk, device=dev, generator=g) < hit, bk[idx], miss).contiguous()
    exp = int(torch.isin(pk, bk).sum())
    for mode in (2, 1):                                         # the plan's own choice, then the wide kernel forced
        api.set_option("join_wide", mode)
        n = api.join_device(1, 0, 0, bk, bv, pk, return_arrays=False)[0]
        t = api.last_timings()
        assert n == exp, (c, kind, nb, npk, hit, mode, n, exp, t)
    print(f"case {c}: {kind} nb {nb} np {npk} hit {hit}: {exp} ok (path {t['path']}, retries {t['lds_retries']}, fell_back {t['fell_back']})", flush=True)
    del bk, bv, pk, idx, miss
api.set_option("join_wide", 2)
print(f"OK: {cases} cases x 2 modes in {time.time() - t0:.0f} s")
