"""One oversized final partition at config-3 sizes (100M x 1B + 27K build keys whose hash word 1 agrees in its top 15 bits):
time of hash_join_count_radix next to the uniform join.  `python tools/skew_build_partition_probe.py [reps]`"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, flash_join
from flash_hash_join_amd import datagen
flash_join.initialize()
dev = "cuda:0"
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 4

def hash_w1(k):
    m32 = 0xFFFFFFFF
    lo, hi = k & m32, (k >> 32) & m32
    x = ((lo * 0x9E3779B1) & m32) ^ ((hi * 0x85EBCA77) & m32)
    x = x ^ (x >> 16); x = (x * 0x85ebca6b) & m32
    x = x ^ (x >> 13); x = (x * 0xc2b2ae35) & m32
    return x ^ (x >> 16)

found, base = [], 1 << 40
for c0 in range(0, 1 << 30, 1 << 26):
    cand = torch.arange(base + c0, base + c0 + (1 << 26), dtype=torch.int64, device=dev)
    found.append(cand[(hash_w1(cand) >> 17) == 12345]); del cand
extra = torch.cat(found)[:27_000]
bk, bv = datagen.build_device(100_000_000, dev)
pk, exp = datagen.probe_device(1_000_000_000, 100_000_000, dev, seed=1, hit_bp=5000)
def timed(b, v, p):
    best = None
    for _ in range(reps):
        n, sec = flash_join.hash_join_count_radix(b, v, p)
        best = sec if best is None else min(best, sec)
    return n, best, flash_join.last_timings()
n, tu, lt = timed(bk, bv, pk)
print("uniform", n == exp, "%.3f ms" % (tu * 1e3))
bk2, bv2 = torch.cat([bk, extra]), torch.cat([bv, extra + 1]); del bk, bv
pk[:500_000] = extra.repeat(19)[:500_000]
n, ts, lt = timed(bk2, bv2, pk)
print("one 30K-key partition", "%.3f ms" % (ts * 1e3), "= %.3fx" % (ts / tu), "fell_back", lt["fell_back"], "lds_retries", lt["lds_retries"])
