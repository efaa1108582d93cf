"""Round 6: random joins through the NumPy entry of the module surface (host arrays -> pinned ring -> HBM under the join's first
pass, fj_join_host): int64 and uint64 arrays, strided and N-D views, empty and tiny inputs, duplicate build keys; all twelve
functions.  Counts against numpy.isin, pairs against the first-occurrence rule.  usage: python tools/r6_host_fuzz.py [cases=60] [seed=1]"""
import os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import flash_join as fj

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
fj.initialize()
COUNT = ["adaptive_join_count", "adaptive_join_count_bloom", "hash_join_count_radix", "hash_join_count", "hash_join_count_radix_bloom", "hash_join_count_bloom"]
MAT = ["adaptive_join", "adaptive_join_bloom", "hash_join_radix", "hash_join", "hash_join_radix_bloom", "hash_join_bloom"]
t0 = time.time()
for c in range(cases):
    nb = rng.choice([0, 1, 2, 255, 4097]) if rng.random() < 0.15 else int(10 ** rng.uniform(2, 6.8))
    npk = rng.choice([0, 1, 3]) if rng.random() < 0.1 else int(10 ** rng.uniform(2, 7.2))
    r = np.random.default_rng(rng.randrange(1 << 30))
    dt = rng.choice([np.int64, np.uint64])
    bk = r.integers(0, 2**63, size=nb, dtype=np.uint64).astype(dt)
    if nb and rng.random() < 0.3: bk = np.repeat(bk[: max(1, nb // 5)], 5)[:nb]; r.shuffle(bk)
    nb = bk.size
    bv = r.integers(0, 2**63, size=nb, dtype=np.uint64).astype(dt)
    hit = rng.choice([0.0, 0.05, 0.5, 1.0])
    pk = r.integers(0, 2**63, size=npk, dtype=np.uint64).astype(dt)
    if nb and npk:
        m = r.random(npk) < hit
        pk[m] = bk[r.integers(0, nb, size=int(m.sum()))]
    layout = rng.choice(["plain", "strided", "2d"])
    def lay(a):
        if layout == "strided" and a.size:
            wide = np.zeros(a.size * 3, dtype=a.dtype); wide[::3] = a; return wide[::3]
        if layout == "2d" and a.size % 2 == 0 and a.size: return a.reshape(2, -1)
        return a
    hb, hv, hp = lay(bk), lay(bv), lay(pk)
    if layout == "2d" and (hb.ndim != hv.ndim): hv = bv; hb = bk
    hitmask = np.isin(pk, bk); exp = int(hitmask.sum())
    name = rng.choice(COUNT + MAT)
    tag = f"case {c}: {name} {np.dtype(dt).name} {layout} nb {nb} np {npk} hit {hit}"
    if name in COUNT:
        n = getattr(fj, name)(hb, hv, hp)[0]
        assert n == exp, (tag, n, exp)
    else:
        n, _, k, v = getattr(fj, name)(hb, hv, hp, return_arrays=True)
        k, v = np.asarray(k).view(np.uint64), np.asarray(v).view(np.uint64)
        assert n == exp == k.size == v.size, (tag, n, exp, k.size)
        if n:
            assert np.array_equal(np.sort(k), np.sort(pk[hitmask].view(np.uint64))), tag
            uniq, first = np.unique(bk.view(np.uint64), return_index=True)
            assert np.array_equal(v, bv.view(np.uint64)[first[np.searchsorted(uniq, k)]]), (tag, "a value that is not the first occurrence's")
    print(tag, "->", exp, "ok", flush=True)
print(f"OK: {cases} cases in {time.time() - t0:.0f} s")
