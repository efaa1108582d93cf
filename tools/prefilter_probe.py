#!/usr/bin/env python3
"""Cost of the owner shuffle's sender-side precheck on one MI355X at BASELINE configs[4] per-GPU sizes (N = 8):
an owner's 125M build keys -> fj_bloom_export; one of a sender's 8 owner segments (156M probe rows) -> fj_bloom_prefilter,
at several hit rates.  Prints the times, the survivors, and the per-link bytes with / without the precheck.
usage: python tools/prefilter_probe.py [--scale 1.0]"""
import argparse
import os
os.environ.setdefault("FJ_LIB_VARIANT", "lab")           # the building blocks behind the C ABI are visible in the lab build only
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                            # noqa: E402
from flash_hash_join_amd import api, datagen           # noqa: E402
from flash_hash_join_amd.lab import LabEngine as HipEngine  # noqa: E402


def timed(fn, reps=5):
    best, out = None, None
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); out = fn(); e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        best = ms if best is None else min(best, ms)
    return best, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=float, default=1.0)
    a = ap.parse_args()
    api.initialize()
    eng = HipEngine("cuda:0")
    world = 8
    nb, seg = int(125_000_000 * a.scale), int(156_250_000 * a.scale)
    bk, bv = datagen.build_device(nb, "cuda:0")
    ms, filt = timed(lambda: eng.bloom_export(bk, 48))
    fbytes = filt.numel() * 4
    print(f"export: {nb} build keys -> {fbytes / 1e6:.1f} MB of filters in {ms:.3f} ms")
    for hit_bp in (100, 500, 2500, 5000, 10000):
        pk, exp = datagen.probe_device(seg, nb, "cuda:0", seed=3, hit_bp=hit_bp)
        ms, kept = timed(lambda: eng.bloom_prefilter(pk, filt, 48))
        k = kept.numel()
        plain = seg * 8
        pre = k * 8 + fbytes
        print(f"prefilter hit {hit_bp / 100:5.1f} %: {seg} rows -> {k} survivors ({k / seg:.3f}; hits {exp / seg:.3f}) in {ms:.3f} ms "
              f"= {seg / ms / 1e6:.1f} G rows/s; per link {plain / 1e9:.3f} GB -> {pre / 1e9:.3f} GB; "
              f"x{world} segments per sender = {ms * world:.1f} ms")
        del pk, kept


if __name__ == "__main__":
    main()
