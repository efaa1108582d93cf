"""Round 6: random joins through the multi-GPU driver on ONE rank (1-rank RCCL group, force_exchange) starting at the
build-broadcast rung: counting and materialising, duplicate build keys (a materialising broadcast refuses them: the ladder must
are served by every form: one pair per matching probe row, some copy's value), a few keys with thousands of copies (the broadcast's partitions overflow: agreed failure, next
rung), random pieces, RCCL / callback transport.  Counts against torch.isin; pairs: the matching probe rows, each once, with a value of their key's.
usage: python tools/r6_bcast_fuzz.py [cases=60] [seed=1]"""
import os, random, socket, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
import flash_hash_join_amd.distributed as D
from flash_hash_join_amd import api

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
dev = "cuda:0"
forms = {}
t0 = time.time()
for c in range(cases):
    nb = int(10 ** rng.uniform(6.4, 7.7))
    npk = rng.choice([0, 5, 1000]) if rng.random() < 0.1 else int(nb * 10 ** rng.uniform(-0.7, 1.0))
    kind = rng.choice(["random", "random", "sequential", "dups", "fewdistinct", "highword"])
    g = torch.Generator(device=dev); g.manual_seed(rng.randrange(1 << 30))
    if kind == "sequential": bk = torch.arange(1, nb + 1, device=dev, dtype=torch.int64) * rng.choice([1, 3, 1 << 20])
    elif kind == "highword": bk = torch.arange(1, nb + 1, device=dev, dtype=torch.int64) << 32
    else: bk = torch.randint(-2**62, 2**62, (nb,), device=dev, dtype=torch.int64, generator=g)
    if kind == "dups":
        m = rng.choice([2, 7, 60]); bk = bk[: max(1, nb // m)].repeat(m)
        bk = bk[torch.randperm(bk.numel(), device=dev, generator=g)].contiguous()
    if kind == "fewdistinct":
        d = rng.choice([5, 1000]); bk = bk[:d].repeat(nb // d + 1)[:nb].contiguous()
    nb = int(bk.numel())
    bv = torch.randint(-2**62, 2**62, (nb,), device=dev, dtype=torch.int64, generator=g)
    hit = rng.choice([0.0, 0.05, 0.5, 1.0])
    idx = torch.randint(0, nb, (max(npk, 1),), device=dev, generator=g)[:npk]
    miss = torch.randint(-2**62, 2**62, (max(npk, 1),), device=dev, dtype=torch.int64, generator=g)[:npk]
    pk = torch.where(torch.rand(npk, device=dev, generator=g) < hit, bk[idx], miss).contiguous() if npk else miss
    hitmask = torch.isin(pk, bk) if npk else torch.zeros(0, dtype=torch.bool, device=dev)
    exp = int(hitmask.sum())
    os.environ["FJ_DIST_PIECES"] = str(rng.choice([0, 1, 4, 7]))
    os.environ["FJ_DIST_NATIVE"] = rng.choice(["1", "1", "0"])
    api.set_option("lab_hooks", rng.choice([0, 0, 1]))           # 1: the rank's own share travels through ncclSend / ncclRecv too (the path every peer's share takes at N > 1)
    mat = rng.random() < 0.4 and exp < 80_000_000
    D._FORM_MEMO.clear()
    t = {}
    tag = f"case {c}: {kind} nb {nb} np {npk} hit {hit} pieces {os.environ['FJ_DIST_PIECES']} native {os.environ['FJ_DIST_NATIVE']} mat {mat}"
    if mat:
        n, _, k, v = D.distributed_join(bk, bv, pk, materialize=True, return_arrays=True, timings=t, strategy="broadcast", force_exchange=True)
        assert n == exp == k.numel(), (tag, n, exp, t)
        if n:
            assert bool(torch.equal(torch.sort(k)[0], torch.sort(pk[hitmask])[0])), tag
            M = -7046029254386353131                          # across GPUs a duplicated key's pair carries the value of ONE of its copies (no global first occurrence)
            assert bool(torch.isin(k * M + v, bk * M + bv).all()), (tag, "a value that belongs to no copy of the key", t.get("strategy"))
        del k, v
    else:
        n, _ = D.distributed_join(bk, bv, pk, timings=t, strategy="broadcast", force_exchange=True)
        assert n == exp, (tag, n, exp, t)
    forms[t.get("strategy")] = forms.get(t.get("strategy"), 0) + 1
    print(tag, "->", exp, "ok in form", t.get("strategy"), "| fell from broadcast:" if "broadcast_form_error" in t else "", str(t.get("broadcast_form_error", ""))[:90], flush=True)
    del bk, bv, pk, idx, miss, hitmask
dist.destroy_process_group()
print(f"OK: {cases} joins through the driver starting at the broadcast rung, 0 mismatches, forms {forms}; {time.time() - t0:.0f} s")
