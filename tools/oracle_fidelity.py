#!/usr/bin/env python3
"""Fidelity of the oracle's C restatement as a stand-in for the reference's CPU path: its core_duration_sec on the
configurations SURVEY.md 6.3 measured with the COMPILED reference in this container class (8 vCPU Xeon, -O3 -march=native),
side by side.  CPU only (test infrastructure; never the product path).  The compiled reference itself cannot be rebuilt
inside the repo's rules (hash_join.cpp:31 needs the un-vendored mimalloc.h), so SURVEY's numbers are the reference column.
usage: python tools/oracle_fidelity.py > profiles/r02_oracle_fidelity.txt"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from flash_hash_join_amd import datagen
from oracle import oracle as O

# (build rows, probe rows, hit bp, function, (algo, bloom, materialize), SURVEY 6.3 core seconds of the compiled reference)
CASES = [
    (1_000_000, 10_000_000, 5000, "hash_join_count", ("scalar", False, False), 0.294),
    (1_000_000, 10_000_000, 5000, "hash_join_count_bloom", ("scalar", True, False), 0.138),
    (1_000_000, 10_000_000, 5000, "hash_join_count_radix", ("radix", False, False), 0.151),
    (1_000_000, 10_000_000, 5000, "adaptive_join_count", ("adaptive", False, False), 0.111),
    (1_000_000, 10_000_000, 5000, "hash_join", ("scalar", False, True), 0.244),
    (1_000_000, 100_000_000, 5000, "hash_join_count", ("scalar", False, False), 0.868),
    (1_000_000, 100_000_000, 5000, "hash_join_count_bloom", ("scalar", True, False), 0.671),
    (1_000_000, 100_000_000, 5000, "hash_join_count_radix", ("radix", False, False), 1.506),
    (10_000_000, 100_000_000, 5000, "hash_join_count", ("scalar", False, False), 2.57),
    (10_000_000, 100_000_000, 5000, "hash_join_count_radix", ("radix", False, False), 1.78),
    (10_000_000, 100_000_000, 5000, "hash_join_count_radix_bloom", ("radix", True, False), 1.61),
    (10_000_000, 100_000_000, 500, "hash_join_count_bloom", ("scalar", True, False), 1.81),
    (10_000_000, 100_000_000, 500, "hash_join_count_radix_bloom", ("radix", True, False), 1.35),
]


def main():
    O.build()
    print(f"# host: {os.cpu_count()} logical CPUs, oracle threads = {O.lib().fjo_default_threads()}, hw crc32c = {bool(O.lib().fjo_uses_hw_crc())}")
    print("build_rows,probe_rows,hit_bp,function,count_ok,oracle_port_s(best of 3),reference_s(SURVEY 6.3),ratio_port_over_reference")
    cache = {}
    for nb, npk, hit_bp, fn, (algo, bloom, mat), ref_s in CASES:
        key = (nb, npk, hit_bp)
        if key not in cache:
            cache.clear()
            bk, bv = datagen.build_numpy(nb)
            pk, exp = datagen.probe_numpy(npk, nb, seed=1, hit_bp=hit_bp)
            cache[key] = (bk, bv, pk, exp)
        bk, bv, pk, exp = cache[key]
        best, ok = None, True
        for _ in range(3):
            n, sec = O.c_join(bk, bv, pk, algo=algo, bloom=bloom, materialize=mat, threads=0)[:2]
            ok = ok and n == exp
            best = sec if best is None else min(best, sec)
        print(f"{nb},{npk},{hit_bp},{fn},{ok},{best:.3f},{ref_s},{best / ref_s:.2f}", flush=True)


if __name__ == "__main__":
    main()
