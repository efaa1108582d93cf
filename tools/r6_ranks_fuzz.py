"""Round 6: random multi-rank joins with 2-3 ranks sharing ONE GPU over gloo (the transport of tools/two_ranks_one_gpu.py: device
tensors staged through the host; everything else is the production path - HipEngine, the C++ driver over the callback transport,
the ladder).  Random relation sizes, ragged and EMPTY blocks, duplicate build keys within and across ranks, every starting rung,
counting and materialising.  Every rank checks the global count against torch.isin over the whole relations and - materialising -
that the ranks' pairs together are the matching probe rows, each once, with a value of their key's.
usage: python tools/r6_ranks_fuzz.py [world=2] [cases=12] [seed=1] [log10 of the smallest build side=5] [of the largest=7]"""
import os, random, socket, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from two_ranks_one_gpu import HostStagedDist


def worker(rank, world, port, cases, seed, q, lo=5.0, hi=7.0):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import flash_hash_join_amd.distributed as D
        from flash_hash_join_amd import api
        api.initialize()
        shim = HostStagedDist()
        rng = random.Random(seed)                                   # the same sequence on every rank
        dev = "cuda:0"
        M = -7046029254386353131
        forms = {}
        for c in range(cases):
            nb = int(10 ** rng.uniform(lo, hi))
            npk = int(nb * 10 ** rng.uniform(-0.5, 0.9))
            kind = rng.choice(["random", "random", "sequential", "dups", "fewdistinct"])
            g = torch.Generator(device=dev); g.manual_seed(rng.randrange(1 << 30))      # (same data on every rank; each keeps its block)
            if kind == "sequential": bk = torch.arange(1, nb + 1, device=dev, dtype=torch.int64) * 3
            else: bk = torch.randint(-2**62, 2**62, (nb,), device=dev, dtype=torch.int64, generator=g)
            if kind == "dups":
                m = rng.choice([2, 7, 60]); bk = bk[: max(1, nb // m)].repeat(m)
                bk = bk[torch.randperm(bk.numel(), device=dev, generator=g)].contiguous()
            if kind == "fewdistinct":
                d = rng.choice([5, 1000]); bk = bk[:d].repeat(nb // d + 1)[:nb].contiguous()
            nb = int(bk.numel())
            bv = torch.randint(-2**62, 2**62, (nb,), device=dev, dtype=torch.int64, generator=g)
            hit = rng.choice([0.0, 0.05, 0.5, 1.0])
            idx = torch.randint(0, nb, (npk,), device=dev, generator=g)
            miss = torch.randint(-2**62, 2**62, (npk,), device=dev, dtype=torch.int64, generator=g)
            pk = torch.where(torch.rand(npk, device=dev, generator=g) < hit, bk[idx], miss).contiguous()
            hitmask = torch.isin(pk, bk); exp = int(hitmask.sum())
            # ragged blocks: random cut points; sometimes a rank without build rows / without probe rows
            cb = sorted(rng.randrange(nb + 1) for _ in range(world - 1)); cp = sorted(rng.randrange(npk + 1) for _ in range(world - 1))
            if rng.random() < 0.25: cb[0] = 0
            if rng.random() < 0.25: cp[-1] = npk
            cb = [0] + cb + [nb]; cp = [0] + cp + [npk]
            mybk, mybv, mypk = bk[cb[rank]: cb[rank + 1]].clone(), bv[cb[rank]: cb[rank + 1]].clone(), pk[cp[rank]: cp[rank + 1]].clone()
            strategy = rng.choice(["auto", "broadcast", "shuffle", "scatter"])
            os.environ["FJ_DIST_PREFILTER"] = rng.choice(["0", "auto", "1"]) if strategy == "shuffle" else "0"
            os.environ["FJ_DIST_PIECES"] = str(rng.choice([0, 1, 4]))
            mat = rng.random() < 0.5
            D._FORM_MEMO.clear(); D._PRECHECK_MEMO.clear()
            t = {}
            tag = f"case {c}: world {world} {kind} nb {nb} np {npk} hit {hit} cuts {cb[1:-1]} {cp[1:-1]} start {strategy} mat {mat}"
            if mat:
                n, _, k, v = D.distributed_join(mybk, mybv, mypk, materialize=True, return_arrays=True, timings=t, transport=shim, strategy=strategy)
                assert n == exp, (tag, n, exp, t)
                assert bool(torch.isin(k * M + v, bk * M + bv).all()), (tag, "a value that belongs to no copy of the key")
                # the ranks' pairs together = the matching probe rows, each once: gather the keys (through the host)
                ks = [None] * world
                dist.all_gather_object(ks, k.cpu())
                allk = torch.cat(ks).to(dev)
                assert allk.numel() == exp and bool(torch.equal(torch.sort(allk)[0], torch.sort(pk[hitmask])[0])), tag
                del k, v, allk
            else:
                n, _ = D.distributed_join(mybk, mybv, mypk, timings=t, transport=shim, strategy=strategy)
                assert n == exp, (tag, n, exp, t)
            forms[t.get("strategy")] = forms.get(t.get("strategy"), 0) + 1
            if rank == 0: print(tag, "->", exp, "ok in form", t.get("strategy"), flush=True)
            del bk, bv, pk, idx, miss, hitmask, mybk, mybv, mypk
        q.put((rank, forms))
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    cases = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn"); q = ctx.Queue()
    lo = float(sys.argv[4]) if len(sys.argv) > 4 else 5.0; hi = float(sys.argv[5]) if len(sys.argv) > 5 else 7.0
    ps = [ctx.Process(target=worker, args=(r, world, port, cases, seed, q, lo, hi)) for r in range(world)]
    for p in ps: p.start()
    for p in ps: p.join(timeout=1500)
    codes = [p.exitcode for p in ps]
    for p in ps:
        if p.is_alive(): p.terminate()
    if any(c != 0 for c in codes): raise SystemExit(f"worker exit codes {codes}")
    print("OK:", world, "ranks on one GPU,", cases, "random joins, forms", q.get()[1])
