"""Sender side of the owner shuffle in isolation (fj_shuffle_pack_begin / _counts / _finish): time of the first pass + bookkeeping
and of the copy into the wire format, for one piece of n rows.  usage: python tools/pack_probe.py [n] [nb_total] [world]"""
import os
os.environ.setdefault("FJ_LIB_VARIANT", "lab")           # the building blocks behind the C ABI are visible in the lab build only
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flash_hash_join_amd import datagen
from flash_hash_join_amd.lab import LabEngine as HipEngine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 312_500_000
nb_total = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000_000
world = int(sys.argv[3]) if len(sys.argv) > 3 else 8
eng = HipEngine("cuda:0")
L, lib = eng.L, eng._lib
pk, _ = datagen.probe_device(n, nb_total, "cuda:0", seed=1, hit_bp=5000)
cb = eng.shuffle_chunk_bytes(nb_total, world)
stream = torch.cuda.current_stream().cuda_stream
vp = ctypes.c_void_p
for it in range(4):
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    e0.record()
    lib.check(L.fj_shuffle_pack_begin(eng.ctx, pk.data_ptr(), None, n, nb_total, world, 0, stream))
    used = (ctypes.c_uint64 * 64)()
    lib.check(L.fj_shuffle_pack_counts(eng.ctx, used))
    e1.record()
    used = [int(used[r]) for r in range(world)]
    if it == 0:
        chunks = [torch.empty(max(16, u * cb + 4096), dtype=torch.uint8, device="cuda:0") for u in used]
        dirs = [torch.empty(max(4, u + 64), dtype=torch.int32, device="cuda:0") for u in used]
    dk = (vp * 64)(*[c.data_ptr() for c in chunks]); dd = (vp * 64)(*[d.data_ptr() for d in dirs])
    torch.cuda.synchronize()
    e1.record()
    lib.check(L.fj_shuffle_pack_finish(eng.ctx, dk, None, dd, stream))
    e2.record()
    torch.cuda.synchronize()
    tot = sum(used)
    print(f"n={n} plan for {nb_total} rows, world {world}: chunk_bytes {cb}, {tot} wire chunks ({tot * (cb + 4) / n:.4f} B/key); "
          f"first pass + bookkeeping {e0.elapsed_time(e1):.3f} ms (incl. host sync), copy {e1.elapsed_time(e2):.3f} ms = "
          f"{n * (8 + cb / 256) / e1.elapsed_time(e2) / 1e6:.0f} GB/s", flush=True)
