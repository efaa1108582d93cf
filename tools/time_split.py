import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("FJ_LIB_VARIANT", "lab")           # the building blocks behind the C ABI are visible in the lab build only
import torch
from flash_hash_join_amd import api, datagen
from flash_hash_join_amd.lab import LabEngine as HipEngine
api.initialize(); eng = HipEngine("cuda:0")
pk, _ = datagen.probe_device(1_000_000_000, 1000, "cuda:0")
for world in (1, 2, 4, 8):
    for _ in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out, _, c = eng.owner_split(pk, None, world)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        del out
    print("world", world, "split 1B keys: %.2f ms" % (dt * 1e3), "max/min share %.4f" % (max(c) / min(c)))
