import sys, time
sys.path.insert(0, '/root/repo')
import torch, flash_join
flash_join.initialize()
dev = 'cuda:0'
for nb, dom in ((50_000_000, 1), (50_000_000, 3), (20_000_000, 5000)):
    g = torch.Generator(device=dev); g.manual_seed(1)
    domain = torch.randint(-(1 << 62), 1 << 62, (dom,), dtype=torch.int64, device=dev, generator=g)
    bk = domain[torch.randint(0, dom, (nb,), device=dev, generator=g)]
    bv = bk + 1
    pk = torch.cat([domain[torch.randint(0, dom, (30_000_000,), device=dev, generator=g)],
                    torch.randint(-(1 << 62), 1 << 62, (70_000_000,), dtype=torch.int64, device=dev, generator=g)])
    exp = int(torch.isin(pk, domain).sum())
    for fn in ('hash_join_count_radix', 'hash_join_count', 'adaptive_join_count_bloom'):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n, sec = getattr(flash_join, fn)(bk, bv, pk)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        lt = flash_join.last_timings()
        print(nb, dom, fn, n == exp, f"{dt*1e3:.1f} ms", 'path', lt['path'], 'fell_back', lt['fell_back'], 'retries', lt['lds_retries'], flush=True)
    n, sec, k, v = flash_join.hash_join_radix(bk, bv, pk, return_arrays=True)
    print(nb, dom, 'hash_join_radix', n == exp, bool(torch.all(k + 1 == v)), flush=True)
    n = flash_join.inner_join_count(bk[:200_000], bv[:200_000], pk[:1000])[0] if dom >= 5000 else None
