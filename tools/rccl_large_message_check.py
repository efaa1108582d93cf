"""Reproduces the RCCL finding behind distributed._exchange: an int64 all_to_all_single is exact at 0.8 GB per peer and
wrong at 4.8 GB per peer (MI355X, RCCL 2.26.6, torch 2.10 ROCm 7.0); also checks fj_owner_split at 1e9 rows."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("FJ_LIB_VARIANT", "lab")           # the building blocks behind the C ABI are visible in the lab build only
import torch, torch.distributed as dist
from flash_hash_join_amd import api, datagen
from flash_hash_join_amd.lab import LabEngine as HipEngine
api.initialize()
eng = HipEngine("cuda:0")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29588")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
for n in (100_000_000, 600_000_000, 1_000_000_000):
    pk, exp = datagen.probe_device(n, 1000, "cuda:0", seed=1, hit_bp=5000)
    s0, x0 = int(pk.sum()), int(torch.bitwise_xor(pk[: n // 2], pk[n - n // 2:]).sum())
    for world in (1, 8):
        out, _, counts = eng.owner_split(pk, None, world)
        ok_sum = int(out.sum()) == s0
        # multiset check via sort of a strided sample is too weak; compare full sorted arrays in two halves
        a = torch.sort(pk)[0]; b = torch.sort(out)[0]
        same = bool(torch.equal(a, b))
        del a, b
        print("split", n, world, sum(counts) == n, ok_sum, same)
    recv = torch.empty_like(pk)
    dist.all_to_all_single(recv, pk, output_split_sizes=[n], input_split_sizes=[n])
    torch.cuda.synchronize()
    print("a2a", n, bool(torch.equal(recv, pk)))
    del pk, recv, out; torch.cuda.empty_cache()
dist.destroy_process_group()
