"""Randomised large-size cross-check of the partitioned plans against the literal one-table algorithm (an independent kernel
path: fj_gt_build / fj_gt_probe, `scalar_hbm_table=1`) on device-resident inputs.  No CPU oracle: at 10M-400M rows the NumPy
oracle takes minutes per case; what is compared is two independent GPU implementations of the same join, plus order-free
checksums of the materialised pairs (values are a function of the key, so duplicate build keys agree on their value).

    python tools/stress_large.py [cases=16] [seed0=0]
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import flash_join as fj  # noqa: E402

M = 0x9E3779B97F4A7C15
MASK = (1 << 64) - 1


def _mix(x: torch.Tensor) -> torch.Tensor:          # a bijection of int64 (odd multiplier; wraps)
    return x * torch.tensor(M - (1 << 64), dtype=torch.int64, device=x.device)


def make_case(seed: int, dev):
    g = torch.Generator(device=dev); g.manual_seed(1000 + seed)
    nb = int(torch.randint(2_000_000, 120_000_000, (1,), generator=g, device=dev).item())
    np_ = int(torch.randint(10_000_000, 400_000_000, (1,), generator=g, device=dev).item())
    kind = seed % 5
    if kind == 0:       # unique build keys, ~50 % hits
        bk = _mix(torch.arange(1, nb + 1, dtype=torch.int64, device=dev))
        pk = _mix(torch.randint(1, 2 * nb + 1, (np_,), generator=g, device=dev, dtype=torch.int64))
    elif kind == 1:     # duplicate-heavy build side (about nb/3 distinct keys)
        dom = max(1000, nb // 3)
        bk = _mix(torch.randint(1, dom + 1, (nb,), generator=g, device=dev, dtype=torch.int64))
        pk = _mix(torch.randint(1, 3 * dom + 1, (np_,), generator=g, device=dev, dtype=torch.int64))
    elif kind == 2:     # hot probe keys: a third of the probe side is 16 keys
        bk = _mix(torch.arange(1, nb + 1, dtype=torch.int64, device=dev))
        pk = _mix(torch.randint(1, 4 * nb + 1, (np_,), generator=g, device=dev, dtype=torch.int64))
        hot = _mix(torch.randint(1, nb + 1, (16,), generator=g, device=dev, dtype=torch.int64))
        idx = torch.randint(0, np_, (np_ // 3,), generator=g, device=dev)
        pk[idx] = hot[torch.randint(0, 16, (np_ // 3,), generator=g, device=dev)]
    elif kind == 3:     # sequential (unmixed) keys: consecutive integers, few hits
        bk = torch.arange(7, nb + 7, dtype=torch.int64, device=dev)
        pk = torch.randint(0, 20 * nb, (np_,), generator=g, device=dev, dtype=torch.int64)
    else:               # build keys concentrated in few values of the high word (stresses the radix digits' source)
        bk = (torch.arange(1, nb + 1, dtype=torch.int64, device=dev) << 20) | 5
        pk = (torch.randint(1, 2 * nb + 1, (np_,), generator=g, device=dev, dtype=torch.int64) << 20) | 5
    bv = bk ^ 0x5555                                  # value = f(key)
    return nb, np_, kind, bk, bv, pk


def checksum(k: torch.Tensor, v: torch.Tensor):
    return int(k.sum().item()) & MASK, int(v.sum().item()) & MASK, int((k ^ (v * 3)).sum().item()) & MASK


def main():
    ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    dev = torch.device("cuda", 0)
    fj.initialize()
    bad = 0
    for seed in range(seed0, seed0 + ncases):
        nb, np_, kind, bk, bv, pk = make_case(seed, dev)
        t0 = time.time()
        fj.set_option("scalar_hbm_table", 1)
        ref = fj.hash_join_count(bk, bv, pk)[0]                      # literal one-table algorithm in HBM
        fj.set_option("scalar_hbm_table", 0)
        got = {fn: getattr(fj, fn)(bk, bv, pk)[0] for fn in ("hash_join_count_radix", "hash_join_count_radix_bloom", "adaptive_join_count", "hash_join_count")}
        ok = all(v == ref for v in got.values())
        msg = ""
        if np_ <= 150_000_000:                                       # pairs: two materialising plans, order-free checksums
            n1, _, k1, v1 = fj.hash_join_radix(bk, bv, pk, return_arrays=True)
            c1 = checksum(k1, v1); del k1, v1
            n2, _, k2, v2 = fj.hash_join_radix_bloom(bk, bv, pk, return_arrays=True)
            c2 = checksum(k2, v2)
            okp = n1 == ref and n2 == ref and c1 == c2 and bool(((k2 ^ 0x5555) == v2).all().item())
            del k2, v2
            ok = ok and okp
            msg = " pairs " + ("ok" if okp else "MISMATCH")
        lt = fj.last_timings()
        print(f"seed {seed} kind {kind} nb {nb} np {np_} count {ref} {'ok' if ok else 'MISMATCH ' + str(got)}{msg}  ({time.time() - t0:.1f} s, last path {lt['path']} fell_back {lt['fell_back']})", flush=True)
        bad += 0 if ok else 1
        del bk, bv, pk
        torch.cuda.empty_cache()
    print("FAILED" if bad else "OK", f"{ncases} cases, {bad} mismatches")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
