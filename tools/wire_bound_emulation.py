"""A step of the multi-GPU driver where the transport is the bottleneck, with real multi-rank control flow: `world` ranks share ONE
GPU and exchange through gloo with host staging (a few GB/s - far slower than the kernels, as one xGMI link is for 2 ranks).
Every rank joins a 15M x 150M-row shard of a world x (15M x 150M) join at 50 % and at 5 % hits, sender-side precheck of the chunk
form off / on / auto; prints ms per step (max over ranks), rows and bytes that travelled.  What it shows is the policy end to end
(filters exported, all-gathered through the transport, probe pieces compacted, fewer bytes -> shorter step); the rates are not xGMI's.
usage: python tools/wire_bound_emulation.py [world=2]"""
import os, sys, socket, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from two_ranks_one_gpu import HostStagedDist


def worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import flash_hash_join_amd.distributed as D
        from flash_hash_join_amd import datagen, api
        api.initialize()
        shim = HostStagedDist()
        nb_r, np_r = 15_000_000, 150_000_000
        bk, bv = datagen.build_device(nb_r, "cuda:0", first=rank * nb_r)
        os.environ["FJ_DIST_STRATEGY"] = "shuffle"
        out = []
        for hit_bp in (5000, 500):
            pk, exp_local = datagen.probe_device(np_r, nb_r * world, "cuda:0", seed=1, hit_bp=hit_bp, first=rank * np_r)
            e = torch.tensor([exp_local]); dist.all_reduce(e); exp = int(e.item())
            # the link rate this "fabric" delivers, as bench.py measures it at N > 1 (here: bytes through gloo + host copies)
            t0 = time.perf_counter(); probe = torch.empty(8 << 20, dtype=torch.int64, device="cuda:0"); recv = torch.empty_like(probe)
            shim.all_to_all_single(recv, probe); rate = probe.numel() * 8 * (world - 1) / world / (time.perf_counter() - t0)
            r = torch.tensor([rate]); dist.all_reduce(r, op=dist.ReduceOp.MIN); D.set_link_rate(float(r.item()) / max(1, world - 1))
            for mode in ("0", "1", "auto"):
                os.environ["FJ_DIST_PREFILTER"] = mode
                D._PRECHECK_MEMO.clear()
                best, t = None, {}
                for it in range(3):
                    dist.barrier(); torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    n, _ = D.distributed_join(bk, bv, pk, timings=t, transport=shim)
                    torch.cuda.synchronize(); dt = time.perf_counter() - t0
                    assert n == exp, (mode, n, exp)
                    m = torch.tensor([dt]); dist.all_reduce(m, op=dist.ReduceOp.MAX)
                    best = float(m.item()) if best is None else min(best, float(m.item()))
                out.append((hit_bp, mode, best * 1e3, t.get("prefilter"), t.get("prefilter_decision"), t.get("prefilter_sampled_survivors"), t.get("probe_rows_sent"),
                            t.get("wire_bytes_sent"), t.get("filter_bytes_received"), t.get("shuffle_form")))
            del pk
        q.put((rank, D._LINK_BYTES_PER_S, out))
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn"); q = ctx.Queue()
    ps = [ctx.Process(target=worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps: p.start()
    import queue as _queue
    rows, t_end = [], time.time() + 900
    while len(rows) < world:
        try:
            rows.append(q.get(timeout=2))
        except _queue.Empty:
            if any(p.exitcode not in (None, 0) for p in ps) or time.time() > t_end:
                for p in ps:
                    if p.is_alive(): p.terminate()
                raise SystemExit(f"worker exit codes {[p.exitcode for p in ps]}")
    for p in ps: p.join(timeout=60)
    rank, rate, out = sorted(rows)[0]
    print(f"{world} ranks on one GPU over gloo + host staging: {rate / 1e9:.2f} GB/s per 'link' as the model sees it")
    for hit_bp, mode, ms, pf, how, sampled, rows_sent, wire, fbytes, form in out:
        print(f"  {hit_bp / 100:4.0f} % hits  precheck={mode:4s} {ms:8.1f} ms/step  ran={pf} ({how}; sampled {sampled})  rows sent by rank 0: {rows_sent}  wire bytes: {wire}  filter bytes received: {fbytes}  [{form}]")
