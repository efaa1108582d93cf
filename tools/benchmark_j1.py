#!/usr/bin/env python3
"""J1-shaped benchmark: the reference's benchmark.py cases on synthetic data.

benchmark.py joins db-benchmark "J1" tables (`benchmark.py:166-171, 202-207`): probe table x with N rows against
build tables small / medium / big with N/1e6, N/1e3 and N rows, N in {1e7, 2e7, 4e7}, six implementations x
{join_count, join_materialize}, and prints `RESULT,Library=..,Task=..,Threads=..,Time=..,Result=..` lines
(`benchmark.py:83`).  The R data generator is not available, so keys are synthetic with the same shapes:
build keys are unique ids, ~90 % of the probe rows hit (db-benchmark joins are mostly-matching).
Inputs live in HBM (torch tensors); Time is wall time of the call, Core is the device time it returns.
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import flash_join                              # imported first, like benchmark.py:13-18
flash_join.initialize()
import torch                                   # noqa: E402
from flash_hash_join_amd import datagen        # noqa: E402

IMPLS = [("adaptive_join", "adaptive_join_count", "adaptive_join"), ("adaptive_bloom", "adaptive_join_count_bloom", "adaptive_join_bloom"),
         ("flash_join", "hash_join_count", "hash_join"), ("flash_join_bloom", "hash_join_count_bloom", "hash_join_bloom"),
         ("flash_join_radix", "hash_join_count_radix", "hash_join_radix"),
         ("flash_join_radix_bloom", "hash_join_count_radix_bloom", "hash_join_radix_bloom")]     # benchmark.py:240-247


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="1e7,2e7,4e7")
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    for n_s in args.sizes.split(","):
        n = int(float(n_s))
        for q, nb in (("Q1", max(1, n // 1_000_000)), ("Q2", max(1, n // 1000)), ("Q5", n)):
            bk, bv = datagen.build_device(nb, "cuda:0")
            pk, exp = datagen.probe_device(n, nb, "cuda:0", seed=7, hit_bp=9000)
            for label, fcount, fmat in IMPLS:
                for task, fn in (("join_count", fcount), ("join_materialize", fmat)):
                    best_wall, best_core, res = None, None, None
                    for _ in range(args.reps):
                        torch.cuda.synchronize()
                        t0 = time.perf_counter()
                        res, core = getattr(flash_join, fn)(bk, bv, pk)
                        torch.cuda.synchronize()
                        wall = time.perf_counter() - t0
                        if best_wall is None or wall < best_wall:
                            best_wall, best_core = wall, core
                    assert res == exp, (label, task, res, exp)
                    print(f"RESULT,Library={label},Task={task},Case={n_s}-{q},Threads=gpu,Time={best_wall:.4f},Core={best_core:.5f},Result={res}")
            del bk, bv, pk
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
