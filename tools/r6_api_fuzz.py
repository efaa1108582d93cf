"""Round 6: random joins through the twelve functions of the module surface (+ inner_join_count) on device tensors - random sizes
from a thousand to tens of millions of rows, sequential / random / high-word-only keys, duplicate build keys in many
multiplicities, a few distinct keys with thousands of copies, repeated probe keys, hit rates 0 .. 100 %, every dispatch option at
random.  Counts against torch.isin; pairs: as many as the count, the keys are exactly the matching probe rows (as a multiset) and
every value is the FIRST occurrence's (hash_join.cpp:125).  usage: python tools/r6_api_fuzz.py [cases=100] [seed=1] [log10 of the smallest build side=3] [of the largest=7.5]"""
import os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import flash_join as fj

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
fj.initialize()
dev = "cuda:0"
from flash_hash_join_amd import _lib
_EMPTY_RAW = int(_lib.load().fj_key_unmix64(2**64 - 1))                # the raw key whose mixed form is the tables' empty marker
_EMPTY_RAW = _EMPTY_RAW - 2**64 if _EMPTY_RAW >= 2**63 else _EMPTY_RAW
SPECIAL = [0, -1, 1, -2**63, 2**63 - 1, _EMPTY_RAW]
EDGES = [1, 2, 255, 256, 257, 3949, 3950, 3951, 4095, 4096, 4097, 8191, 8192, 8193, 999_999, 1_000_000, 1_000_001]
COUNT = ["adaptive_join_count", "adaptive_join_count_bloom", "hash_join_count_radix", "hash_join_count", "hash_join_count_radix_bloom", "hash_join_count_bloom"]
MAT = ["adaptive_join", "adaptive_join_bloom", "hash_join_radix", "hash_join", "hash_join_radix_bloom", "hash_join_bloom"]
OPTS = {"join_wide": [0, 1, 2], "mat_single_pass": [0, 1], "scalar_hbm_table": [0, 0, 0, 1], "persistent_min_items": [0, 8192, 1 << 30], "plan_target_keys": [4096, 4096, 1024, 256]}
t0 = time.time()
for c in range(cases):
    nb = int(10 ** rng.uniform(float(sys.argv[3]) if len(sys.argv) > 3 else 3.0, float(sys.argv[4]) if len(sys.argv) > 4 else 7.5))
    if rng.random() < 0.2: nb = rng.choice(EDGES)                # sizes at the plans' and the dispatch's thresholds
    npk = max(1, int(nb * 10 ** rng.uniform(-1, 1.3)))
    if nb * 8 + npk * 24 > 2e10: npk = int((2e10 - nb * 8) / 24)
    kind = rng.choice(["random", "sequential", "dups", "fewdistinct", "highword"])
    g = torch.Generator(device=dev); g.manual_seed(rng.randrange(1 << 30))
    if kind == "sequential": bk = torch.arange(1, nb + 1, device=dev, dtype=torch.int64) * rng.choice([1, 3, 1 << 20])
    elif kind == "highword": bk = torch.arange(1, nb + 1, device=dev, dtype=torch.int64) << 32
    else: bk = torch.randint(-2**62, 2**62, (nb,), device=dev, dtype=torch.int64, generator=g)
    if kind == "dups":
        m = rng.choice([2, 7, 60, 300]); bk = bk[: max(1, nb // m)].repeat(m)
        bk = bk[torch.randperm(bk.numel(), device=dev, generator=g)].contiguous()
    if kind == "fewdistinct":
        d = rng.choice([1, 5, 1000]); bk = bk[:d].repeat(nb // d + 1)[:nb].contiguous()
    if rng.random() < 0.3:                                     # special key values on the build side (the probe side draws from it)
        sp = torch.tensor(rng.sample(SPECIAL, rng.randrange(1, len(SPECIAL) + 1)), device=dev, dtype=torch.int64)
        bk = bk.clone(); bk[torch.randint(0, bk.numel(), (sp.numel(),), device=dev, generator=g)] = sp
    nb = int(bk.numel())
    bv = torch.randint(-2**62, 2**62, (nb,), device=dev, dtype=torch.int64, generator=g)
    hit = rng.choice([0.0, 0.05, 0.5, 1.0])
    idx = torch.randint(0, nb, (npk,), device=dev, generator=g)
    miss = torch.randint(-2**62, 2**62, (npk,), device=dev, dtype=torch.int64, generator=g)
    pk = torch.where(torch.rand(npk, device=dev, generator=g) < hit, bk[idx], miss).contiguous()
    if rng.random() < 0.2: pk = pk[: max(1, npk // 50)].repeat(50)[:npk].contiguous()        # repeated probe keys
    if rng.random() < 0.3:                                     # ... and special values among the probe keys whether or not the build side has them
        sp = torch.tensor(SPECIAL, device=dev, dtype=torch.int64)
        pk = pk.clone(); pk[torch.randint(0, pk.numel(), (sp.numel(),), device=dev, generator=g)] = sp
    npk = int(pk.numel())
    hitmask = torch.isin(pk, bk)
    exp = int(hitmask.sum())
    opts = {k: rng.choice(v) for k, v in OPTS.items()}
    for k, v in opts.items(): fj.set_option(k, v)
    name = rng.choice(COUNT + MAT + ["inner_join_count"])
    tag = f"case {c}: {name} {kind} nb {nb} np {npk} hit {hit} {opts}"
    if name == "inner_join_count":
        if kind in ("dups", "fewdistinct"): name = "hash_join_count_radix"     # (many-to-many is limited to 4096 build rows per final partition: refused loudly beyond)
        else:
            uniq, cnt = torch.unique(bk, return_counts=True)
            pos = torch.searchsorted(uniq, pk).clamp(max=uniq.numel() - 1)
            exp_mm = int((cnt[pos] * (uniq[pos] == pk)).sum())
            n = getattr(fj, name)(bk, bv, pk)[0]
            assert n == exp_mm, (tag, n, exp_mm)
    if name in COUNT:
        n = getattr(fj, name)(bk, bv, pk)[0]
        assert n == exp, (tag, n, exp, fj.last_timings())
    elif name in MAT:
        n, _, k, v = getattr(fj, name)(bk, bv, pk, return_arrays=True)
        assert n == exp == k.numel() == v.numel(), (tag, n, exp, k.numel())
        if n:
            assert bool(torch.equal(torch.sort(k)[0], torch.sort(pk[hitmask])[0])), tag                 # the matching probe rows, each once
            uniq, inv = torch.unique(bk, return_inverse=True)
            first = torch.full((uniq.numel(),), nb, device=dev, dtype=torch.int64).scatter_reduce(0, inv, torch.arange(nb, device=dev), "amin")
            pos = torch.searchsorted(uniq, k)
            if opts["scalar_hbm_table"] and name in ("hash_join", "hash_join_bloom"):
                # the literal one-table algorithm (insert_concurrent, hash_join.cpp:96-110): whichever copy's CAS lands first wins, in the
                # reference as here - the value must be SOME copy's
                M = -7046029254386353131
                assert bool(torch.isin(k * M + v, bk * M + bv).all()), (tag, "a value that belongs to no copy of the key")
            else:
                assert bool(torch.all(v == bv[first[pos]])), (tag, "a value that is not the first occurrence's")
        del k, v
    print(tag, "->", exp, "ok", flush=True)
    del bk, bv, pk, idx, miss, hitmask
for k, v in {"join_wide": 2, "mat_single_pass": 1, "scalar_hbm_table": 0, "persistent_min_items": 8192, "plan_target_keys": 4096}.items(): fj.set_option(k, v)
print(f"OK: {cases} cases in {time.time() - t0:.0f} s")
