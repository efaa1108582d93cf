"""Soak test of the multi-GPU driver on ONE rank (1-rank RCCL group): many consecutive steps alternating counting / materialising
joins, the loop-back hook, the callback transport, an injected local failure, and the sender-side precheck off / on / auto; counts must stay exact and device memory must
not creep.  usage: python tools/dist_soak.py [rounds=40]"""
import os, socket, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from flash_hash_join_amd import api, datagen
from flash_hash_join_amd.distributed import distributed_join

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
os.environ["FJ_DIST_STRATEGY"] = "shuffle"
import flash_hash_join_amd.distributed as D
_dj = distributed_join
distributed_join = lambda *a, **kw: _dj(*a, force_exchange=True, **kw)      # the full protocol on this one rank
nb, npk = 20_000_000, 150_000_000
bk, bv = datagen.build_device(nb, "cuda:0")
pk, exp = datagen.probe_device(npk, nb, "cuda:0", seed=3, hit_bp=5000)
M = torch.tensor(-7046029254386353131, dtype=torch.int64, device="cuda:0")
free0 = None
t0 = time.time()
for r in range(rounds):
    mode = r % 5
    hooks = 1 if mode == 1 else 0           # lab_hooks: FJ_HOOK_LOOPBACK
    os.environ["FJ_DIST_NATIVE"] = "0" if mode == 2 else "1"
    api.set_option("lab_hooks", hooks)
    os.environ["FJ_DIST_PREFILTER"] = ("0", "1", "auto")[(r // 5) % 3]      # the sender-side precheck: off / forced / by a sample (threshold below)
    D.PREFILTER_BELOW_OVERRIDE = 0.7
    if mode == 3:
        api.set_option("lab_hooks", hooks | 2)              # the chunk form fails (agreed), the step reruns in the owner-scatter form
    t = {}
    if mode == 4:
        n, sec, k, v = distributed_join(bk, bv, pk, materialize=True, return_arrays=True, timings=t)
        assert n == exp and k.numel() == exp and bool(torch.all((v + 1) * M == k)), (r, n, exp)
        del k, v
    else:
        n, sec = distributed_join(bk, bv, pk, timings=t)
        assert n == exp, (r, mode, n, exp)
    assert (mode == 3) == ("chunk_form_error" in t), (r, mode, t)
    torch.cuda.synchronize()
    free, total = torch.cuda.mem_get_info()
    if r == 9:
        free0 = free                                     # (workspaces and caches have reached their sizes)
    if r >= 10 and r % 10 == 9:
        print(f"round {r + 1}: counts exact; free memory {free / 2**30:.2f} GiB (after warm-up {free0 / 2**30:.2f})", flush=True)
        assert free > free0 - (256 << 20), "device memory is creeping"
print(f"OK: {rounds} steps in {time.time() - t0:.1f} s")
dist.destroy_process_group()
