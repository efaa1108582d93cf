"""Random joins through the multi-GPU driver on ONE rank with the sender-side precheck of the chunk form forced on: random relation
sizes (build sides from the smallest the chunk form takes up to 40M rows, probe sides from empty to 80M), random hit rates, probe
keys repeated, random pieces, counting and materialising, RCCL (plain / loop-back) and the callback transport.  Every count is
checked against torch.isin, every pair against the key -> value rule.  usage: python tools/precheck_fuzz.py [cases=60] [seed=1]"""
import os, random, socket, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from flash_hash_join_amd import api, datagen
from flash_hash_join_amd.distributed import distributed_join

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
os.environ.update(FJ_DIST_STRATEGY="shuffle", FJ_DIST_NO_FALLBACK="1", FJ_DIST_PREFILTER="1")
_dj = distributed_join
distributed_join = lambda *a, **kw: _dj(*a, force_exchange=True, **kw)      # the full protocol on this one rank
M = torch.tensor(-7046029254386353131, dtype=torch.int64, device="cuda:0")
t0 = time.time()
kept_tot = rows_tot = 0
for c in range(cases):
    nb = int(2_200_000 * (40_000_000 / 2_200_000) ** rng.random())
    npk = rng.choice([0, 1, 7, 1000]) if rng.random() < 0.15 else int(10 ** rng.uniform(4, 7.9))
    hit_bp = rng.choice([0, 1, 100, 500, 2500, 5000, 9000, 10000])
    first = rng.randrange(1 << 30)
    bk, bv = datagen.build_device(nb, "cuda:0", first=first)
    if rng.random() < 0.5:                                 # generated probe side (uniform over the build keys + misses)
        pk, _ = datagen.probe_device(max(npk, 1), nb, "cuda:0", seed=rng.randrange(1 << 20), hit_bp=hit_bp)
        pk = pk[:npk]
        # the generator draws from build rows 1..nb of first=0: re-draw against THIS build side
        idx = torch.randint(0, nb, (npk,), device="cuda:0")
        hit = torch.rand(npk, device="cuda:0") < hit_bp / 10000
        pk = torch.where(hit, bk[idx], pk ^ 0x5555555555)
    else:                                                  # few distinct probe keys, repeated many times
        d = max(1, int(10 ** rng.uniform(0, 5)))
        pool = torch.cat([bk[torch.randint(0, nb, (d,), device="cuda:0")], torch.randint(-2**62, 2**62, (d,), device="cuda:0")])
        pk = pool[torch.randint(0, 2 * d, (npk,), device="cuda:0")]
    pk = pk.contiguous()
    exp = int(torch.isin(pk, bk).sum()) if npk else 0
    os.environ["FJ_DIST_PIECES"] = str(rng.choice([1, 2, 4, 7]))
    loop = rng.choice(["0", "1"]); api.set_option("lab_hooks", int(loop))      # (lab_hooks & 1: the loop-back hook)
    os.environ["FJ_DIST_NATIVE"] = rng.choice(["1", "1", "0"])
    mat = rng.random() < 0.3 and exp < 60_000_000
    t = {}
    print(f"case {c}: nb {nb} np {npk} hit_bp {hit_bp} pieces {os.environ['FJ_DIST_PIECES']} loop {loop} native {os.environ['FJ_DIST_NATIVE']} mat {mat} exp {exp}", file=sys.stderr, flush=True)
    if mat:
        n, _, k, v = distributed_join(bk, bv, pk, materialize=True, return_arrays=True, timings=t)
        assert n == exp == k.numel() and bool(torch.all((v + 1) * M == k)) and (exp == 0 or (int(v.min()) >= first and int(v.max()) < first + nb)), (c, n, exp)
        del k, v
    else:
        n, _ = distributed_join(bk, bv, pk, timings=t)
        assert n == exp, (c, nb, npk, hit_bp, n, exp, t)
    assert t["shuffle_form"].startswith("chunks") and t["prefilter"] is True, t
    assert exp == 0 or t["probe_rows_sent"] >= min(exp, 1), t
    kept_tot += t["probe_rows_sent"]; rows_tot += npk
    if c % 10 == 9:
        print(f"case {c + 1}/{cases}: ok ({time.time() - t0:.0f} s; rows kept so far {kept_tot} of {rows_tot})", flush=True)
    del bk, bv, pk
dist.destroy_process_group()
print(f"OK: {cases} random joins through the driver with the precheck on, 0 mismatches; {kept_tot} of {rows_tot} probe rows travelled")
