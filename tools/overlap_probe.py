"""Does an HBM-bound kernel (the probe side's radix passes) share the chip profitably with an instruction-bound one (the wide join)?  Two
contexts run the build-broadcast step of ONE rank of 8 (tools/bcast_one_gpu.py) on two streams, each kernel limited to half the CUs
(fj_ctx_reserve_cus(128): a pass / join workgroup fills a CU, so the two pipelines run side by side), the second pipeline started
half a step behind the first so that one's join meets the other's passes.  Prints: one step on the whole chip, one step on half the
chip, two steps side by side.  usage: python tools/overlap_probe.py [np_rank=1250000000] [lag_ms=8]"""
import os
os.environ.setdefault("FJ_LIB_VARIANT", "lab")           # the building blocks behind the C ABI are visible in the lab build only
import ctypes, os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flash_hash_join_amd import api, datagen, _lib
from flash_hash_join_amd.lab import LabEngine as HipEngine

np_rank = int(sys.argv[1]) if len(sys.argv) > 1 else 1_250_000_000
world, nb_rank, pieces = 8, 125_000_000, 4
api.initialize()
L = _lib.load()
engs = [HipEngine("cuda:0"), HipEngine("cuda:0")]
engs[1].ctx = L.fj_ctx_create(0)                       # a second native context: its own workspace, events, scalars
nb_total = nb_rank * world
bits, nparts, mid = engs[0].bcast_plan(nb_total)
rb = engs[0].bcast_region_bytes(nb_total, nb_rank)
base = torch.empty(rb * world, dtype=torch.uint8, device="cuda:0")
offs = [r * rb for r in range(world)]
empty = torch.empty(16, dtype=torch.int64, device="cuda:0")[:0]
for r in range(world):
    bk, _ = datagen.build_device(nb_rank, "cuda:0", first=r * nb_rank)
    engs[0].bcast_pack(bk, nb_total, base[offs[r]: offs[r] + rb], pieces); engs[0].bcast_pack_bounds(pieces)
    engs[0].bcast_probe(empty, nb_total); engs[0].bcast_finish()
    del bk
bk, _ = datagen.build_device(nb_rank, "cuda:0", first=0)
scratch = [torch.empty(rb, dtype=torch.uint8, device="cuda:0") for _ in range(2)]     # (each pipeline packs its own copy: the step's pack is part of what is timed)
pks, exps = [], []
for i in range(2):
    pk, e = datagen.probe_device(np_rank, nb_total, "cuda:0", seed=1, hit_bp=5000, first=i * np_rank)
    pks.append(pk); exps.append(e)
streams = [torch.cuda.Stream(priority=0), torch.cuda.Stream(priority=-1)]      # (two default-priority torch streams shared ONE hardware queue under rocprofv3: nothing overlapped)
torch.cuda.synchronize()


def enqueue(i):
    with torch.cuda.stream(streams[i]):
        e = engs[i]
        e.bcast_pack(bk, nb_total, scratch[i], pieces)
        e.bcast_probe(pks[i], nb_total)
        for q in range(pieces):
            e.bcast_join(base, offs, [nb_rank] * world, nparts * q // pieces, nparts * (q + 1) // pieces)


def finish(i):
    with torch.cuda.stream(streams[i]):
        n = engs[i].bcast_finish()
    assert n == exps[i], (i, n, exps[i])


def timed(fn, reps=3):
    best = 1e9
    for _ in range(reps + 1):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) * 1e3)
    return best


def one(i):
    enqueue(i); finish(i)


def both(lag_ms):
    enqueue(0)
    if lag_ms > 0:
        torch.cuda._sleep(int(lag_ms * 2.1e6)) if False else time.sleep(lag_ms * 1e-3)
    enqueue(1)
    finish(0); finish(1)


for e in engs:
    L.fj_ctx_reserve_cus(e.ctx, 0)
t_full = timed(lambda: one(0)); timed(lambda: one(1))
for e in engs:
    L.fj_ctx_reserve_cus(e.ctx, 128)
t_half = timed(lambda: one(0))
print(f"one step, whole chip: {t_full:.2f} ms; one step on 128 CUs: {t_half:.2f} ms")
for lag in [float(x) for x in (sys.argv[2:] or ["0", "4", "8", "12"])]:
    t2 = timed(lambda: both(lag))
    print(f"two steps side by side on 128 CUs each, the second {lag:.0f} ms behind: {t2:.2f} ms for both = {t2 / 2:.2f} ms per step ({2 * t_full / t2:.2f}x two steps one after the other on the whole chip)")
