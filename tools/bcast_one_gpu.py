"""The multi-GPU build-broadcast join (csrc/fj_bcast.hip) as ONE rank of an N-rank job sees it, on one GPU: every rank's build block
is packed into its region once (what the peers would have sent), then rank `rank`'s step is timed: pack of its own build rows ->
both passes over its own probe rows -> the join of every partition range against all N regions.  No transport: what this measures
is the kernel time per rank and step that tools/scale_model.py puts beside the wire time.
usage: python tools/bcast_one_gpu.py [world=8] [nb_rank=125000000] [np_rank=1250000000] [pieces=4] [steps=5] [hit_bp=5000] [reserve_cus=0] [mat=0]
(mat=1: the MATERIALISING step - regions with values, fj_dense_mat_join, then fj_emit_pairs writes the rank's pairs;
reserve_cus: CUs the passes and the join leave free, as they do while RCCL's kernels share the GPU at N > 1: csrc/fj_dist.hip reserves 32)"""
import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("FJ_LIB_VARIANT", "lab")           # the building blocks behind the C ABI are visible in the lab build only
import torch
from flash_hash_join_amd import api, datagen
from flash_hash_join_amd.lab import LabEngine as HipEngine

args = [int(x) for x in sys.argv[1:]]
world, nb_rank, np_rank, pieces, steps, hit_bp, reserve, mat = (args + [8, 125_000_000, 1_250_000_000, 4, 5, 5000, 0, 0][len(args):])[:8]
rank = 0
api.initialize(); eng = HipEngine("cuda:0")
if reserve:
    eng.L.fj_ctx_reserve_cus(eng.ctx, reserve)
nb_total = nb_rank * world
bits, nparts, mid = eng.bcast_plan(nb_total)
rb = eng.bcast_region_bytes(nb_total, nb_rank, bool(mat))
base = torch.empty(rb * world, dtype=torch.uint8, device="cuda:0")
offs = [r * rb for r in range(world)]
for r in range(world):
    bk, bv = datagen.build_device(nb_rank, "cuda:0", first=r * nb_rank)
    eng.bcast_pack(bk, nb_total, base[offs[r]: offs[r] + rb], pieces, vals=bv if mat else None)
    bounds = eng.bcast_pack_bounds(pieces)
    pk0 = torch.empty(16, dtype=torch.int64, device="cuda:0")
    eng.bcast_probe(pk0[:0], nb_total); eng.bcast_finish()       # (closes the step the pack opened)
    del bk, bv
bk, bv = datagen.build_device(nb_rank, "cuda:0", first=rank * nb_rank)
pk, expected = datagen.probe_device(np_rank, nb_total, "cuda:0", seed=1, hit_bp=hit_bp, first=rank * np_rank)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
print(f"world {world}: {nb_rank} x {np_rank} rows per rank, plan {bits} bits = {nparts} partitions, {rb / nb_rank:.3f} wire bytes per build key, "
      f"{rb * (world - 1) / 1e9:.3f} GB received per rank = {rb / 1e9:.3f} GB per link and step")
for it in range(steps + 1):
    t0 = time.perf_counter()
    ev[0].record()
    eng.bcast_pack(bk, nb_total, base[offs[rank]: offs[rank] + rb], pieces, vals=bv if mat else None)
    ev[1].record()
    eng.bcast_probe(pk, nb_total)
    ev[2].record()
    for q in range(pieces):
        eng.bcast_join(base, offs, [nb_rank] * world, nparts * q // pieces, nparts * (q + 1) // pieces)
    ev[3].record()
    n = eng.bcast_finish()
    emit = ""
    if mat:
        k, v = eng.emit_pairs(n)                               # (allocates the pair buffers: their first-touch cost is inside this time on step 0 only)
        ev[4].record(); torch.cuda.synchronize()
        emit = f"  emit {ev[3].elapsed_time(ev[4]):.2f} ({n} pairs)"
        assert k.numel() == n
        del k, v
    dt = (time.perf_counter() - t0) * 1e3
    assert n == expected, (n, expected)
    if it:
        print(f"step {it}: wall {dt:.2f} ms  pack {ev[0].elapsed_time(ev[1]):.2f}  probe passes {ev[1].elapsed_time(ev[2]):.2f}  join {ev[2].elapsed_time(ev[3]):.2f}{emit}  "
              f"= {ev[0].elapsed_time(ev[3]):.2f} ms of kernels; {np_rank / dt / 1e6:.1f} G probes/s on this rank; count exact")
