"""The sender-side precheck of the chunk form at the plan of an N-rank job, on ONE GPU: every owner's per-partition Bloom filters
are built in turn (the whole build side is generated here, packed, and each owner's share appended to a shuffled stream join of
that rank: fj_stream_export_part_filters), then one probe piece is packed without and with the precheck (fj_shuffle_pack_filter):
kernel-side time of the first pass + precheck + bookkeeping, of the copy, rows kept, false-positive rate.
usage: python tools/precheck_probe.py [piece_rows] [nb_total] [world] [hit_bp]"""
import os, sys, time
os.environ.setdefault("FJ_LIB_VARIANT", "lab")           # the building blocks behind the C ABI are visible in the lab build only
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flash_hash_join_amd import datagen
from flash_hash_join_amd.lab import LabEngine as HipEngine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 312_500_000
nb_total = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000_000
world = int(sys.argv[3]) if len(sys.argv) > 3 else 8
hit_bp = int(sys.argv[4]) if len(sys.argv) > 4 else 5000
eng = HipEngine("cuda:0")
first, count, total, fb = eng.part_filter_range(nb_total, world, 0)
filters = torch.zeros(total * fb, dtype=torch.uint8, device="cuda:0")
bk, _ = datagen.build_device(nb_total, "cuda:0")
t0 = time.perf_counter()
chunks, dirs, used = eng.shuffle_pack(bk, None, nb_total, world)
for r in range(world):
    first, count, _, _ = eng.part_filter_range(nb_total, world, r)
    eng.stream_open_shuffled(nb_total, world, r, used[r] * 256, 1, 1 << 20, 1)
    eng.stream_append_chunks(0, chunks[r], dirs[r])
    eng.stream_export_part_filters(filters[first * fb: (first + count) * fb])
    torch.cuda.synchronize()
    eng.L.fj_stream_abort(eng.ctx)
torch.cuda.synchronize()
print(f"filters of {total} partitions ({total * fb / 2**20:.0f} MiB = {total * fb / nb_total:.2f} B per build key) built in {time.perf_counter() - t0:.2f} s; "
      f"bits set: {float((torch.bitwise_count(filters.view(torch.int64)) if hasattr(torch, 'bitwise_count') else torch.zeros(1)).sum()) / (total * fb * 8):.3f}", flush=True)
del bk, chunks, dirs
pk, hits = datagen.probe_device(n, nb_total, "cuda:0", seed=1, hit_bp=hit_bp)
for label, f in (("without", None), ("with", filters)):
    for it in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        ch, dr, used = eng.shuffle_pack(pk, None, nb_total, world, filters=f)
        e1.record()
        torch.cuda.synchronize()
        kept = eng.last_pack_kept
    print(f"{label:7s} the precheck: piece of {n} rows at {hit_bp / 100:.0f} % hits, plan for {nb_total} build rows on {world} ranks: "
          f"pack (first pass [+ precheck] + bookkeeping + copy, incl. allocations and one host sync) {e0.elapsed_time(e1):.2f} ms; "
          f"{sum(used)} wire chunks; rows kept {kept} = {kept / n:.4f} (hits {hits}; {(kept - hits) / max(1, n - hits):.4f} of the misses pass)", flush=True)
