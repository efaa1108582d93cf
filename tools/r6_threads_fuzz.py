"""Round 6: several Python threads calling the module surface at once on one GPU (the ctypes calls release the GIL; calls on one
native context take turns behind its lock; fj_last_error / fj_last_timings are per thread) - random joins, counting and
materialising, device tensors and NumPy arrays; every result checked.  usage: python tools/r6_threads_fuzz.py [threads=4] [joins per thread=40] [seed=1]"""
import os, random, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import flash_join as fj

nthreads = int(sys.argv[1]) if len(sys.argv) > 1 else 4
per = int(sys.argv[2]) if len(sys.argv) > 2 else 40
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1
fj.initialize()
dev = "cuda:0"
errors = []


def work(tidx):
    rng = random.Random(seed * 1000 + tidx)
    try:
        for c in range(per):
            nb = int(10 ** rng.uniform(2, 6.7)); npk = max(1, int(nb * 10 ** rng.uniform(-1, 1.2)))
            g = torch.Generator(device=dev); g.manual_seed(rng.randrange(1 << 30))
            bk = torch.randint(-2**62, 2**62, (nb,), device=dev, dtype=torch.int64, generator=g)
            bv = bk * 7 + 1
            idx = torch.randint(0, nb, (npk,), device=dev, generator=g)
            miss = torch.randint(-2**62, 2**62, (npk,), device=dev, dtype=torch.int64, generator=g)
            pk = torch.where(torch.rand(npk, device=dev, generator=g) < rng.choice([0.0, 0.5, 1.0]), bk[idx], miss).contiguous()
            exp = int(torch.isin(pk, bk).sum())
            mode = rng.choice(["count", "mat", "host"])
            if mode == "count":
                n = getattr(fj, rng.choice(["hash_join_count_radix", "adaptive_join_count", "hash_join_count_radix_bloom", "hash_join_count"]))(bk, bv, pk)[0]
                assert n == exp, (tidx, c, n, exp)
                t = fj.last_timings(); assert t is not None and t["total_ms"] >= 0
            elif mode == "mat":
                n, _, k, v = getattr(fj, rng.choice(["hash_join_radix", "adaptive_join", "hash_join"]))(bk, bv, pk, return_arrays=True)
                assert n == exp == k.numel() and bool(torch.all(v == k * 7 + 1)), (tidx, c, n, exp)
            else:
                hb, hv, hp = bk.cpu().numpy(), bv.cpu().numpy(), pk.cpu().numpy()
                n = fj.hash_join_count_radix(hb, hv, hp)[0]
                assert n == exp, (tidx, c, n, exp)
    except Exception as ex:                                       # noqa: BLE001
        errors.append((tidx, repr(ex)))


t0 = time.time()
ts = [threading.Thread(target=work, args=(i,)) for i in range(nthreads)]
for t in ts: t.start()
for t in ts: t.join(timeout=1200)
alive = [t.is_alive() for t in ts]
if any(alive) or errors:
    raise SystemExit(f"FAILED: alive {alive} errors {errors[:3]}")
print(f"OK: {nthreads} threads x {per} joins in {time.time() - t0:.0f} s")
