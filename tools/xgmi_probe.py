#!/usr/bin/env python3
"""All-to-all bandwidth per xGMI link direction, as the join's exchange sees it.

    python -m torch.distributed.run --nproc-per-node N tools/xgmi_probe.py [--mb 64,256]

Every rank sends `mb` MiB to every other rank with one all_to_all_single (RCCL); a fully connected xGMI mesh carries one
peer per link, so  bytes sent to ONE peer / time  is the per-link, per-direction rate the exchange strategies are priced
with (flash_hash_join_amd/distributed.py: _LINK_BYTES_PER_S).  bench.py calls measure() once at N > 1 and feeds the
result to the strategy model instead of the built-in guess."""
from __future__ import annotations

import argparse
import os
import time


def measure(dist, device, mb: int = 128, reps: int = 5, group=None) -> dict:
    """Returns {"link_GBps": per-peer per-direction GB/s (min over ranks), "mb_per_peer": mb, "world": N}."""
    import torch
    world = dist.get_world_size(group)
    n = mb * (1 << 20) // 8
    send = torch.empty(n * world, dtype=torch.int64, device=device).fill_(dist.get_rank(group))
    recv = torch.empty_like(send)
    for _ in range(2):
        dist.all_to_all_single(recv, send, group=group)
    torch.cuda.synchronize(device)
    dist.barrier(group)
    best = None
    for _ in range(reps):
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        dist.all_to_all_single(recv, send, group=group)
        torch.cuda.synchronize(device)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    rate = torch.tensor([n * 8 / best / 1e9], dtype=torch.float64, device=device)
    dist.all_reduce(rate, op=dist.ReduceOp.MIN, group=group)
    return {"link_GBps": round(float(rate.item()), 2), "mb_per_peer": mb, "world": world}


def main() -> None:
    import torch
    import torch.distributed as dist
    ap = argparse.ArgumentParser()
    ap.add_argument("--mb", default="16,64,256,1024")
    args = ap.parse_args()
    rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    for mb in (int(x) for x in args.mb.split(",")):
        r = measure(dist, dev, mb)
        if rank == 0:
            print(f"world={world} MiB_per_peer={mb} link_GBps_per_direction={r['link_GBps']}", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
