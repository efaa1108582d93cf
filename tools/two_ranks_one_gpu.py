"""Two ranks sharing ONE GPU over the gloo backend: runs distributed_join with the real HipEngine under genuine
multi-rank control flow (RCCL refuses two ranks on one device, and only 1-GPU boxes were available to the builder).
Collectives that gloo cannot run on device tensors are staged through the host by a thin wrapper - only the transport
differs from the production path.  usage: python tools/two_ranks_one_gpu.py [world]"""
import os, sys, socket
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


class HostStagedDist:
    """torch.distributed look-alike that moves device tensors through pinned host copies for gloo."""
    ReduceOp = dist.ReduceOp

    class _Done:
        def wait(self):
            return True

    def __getattr__(self, name):
        return getattr(dist, name)

    def _h(self, t):
        return t.cpu()

    def all_reduce(self, t, op=dist.ReduceOp.SUM, group=None, async_op=False):
        h = self._h(t); dist.all_reduce(h, op=op, group=group); t.copy_(h); return self._Done()

    def all_gather_into_tensor(self, out, t, group=None, async_op=False):
        ho, hi = self._h(out), self._h(t); dist.all_gather_into_tensor(ho, hi, group=group); out.copy_(ho); return self._Done()

    def all_gather(self, outs, t, group=None, async_op=False):
        hos = [self._h(o) for o in outs]
        dist.all_gather(hos, self._h(t), group=group)
        for o, h in zip(outs, hos):
            o.copy_(h)
        return self._Done()

    def all_to_all_single(self, out, inp, output_split_sizes=None, input_split_sizes=None, group=None, async_op=False):
        ho, hi = self._h(out), self._h(inp)
        dist.all_to_all_single(ho, hi, output_split_sizes=output_split_sizes, input_split_sizes=input_split_sizes, group=group)
        out.copy_(ho); return self._Done()

    def batch_isend_irecv(self, ops):
        """Grouped point-to-point transfers of device-tensor views (distributed._views_all_to_all), through host copies."""
        reqs, backs = [], []
        for op in ops:
            if op.op is dist.isend:
                reqs.append(dist.isend(op.tensor.cpu(), op.peer, group=op.group))
            else:
                h = torch.empty(op.tensor.shape, dtype=op.tensor.dtype)
                reqs.append(dist.irecv(h, op.peer, group=op.group)); backs.append((op.tensor, h))

        class _Work:
            def wait(self_inner):
                for r in reqs:
                    r.wait()
                for t, h in backs:
                    t.copy_(h)
                return True
        return [_Work()]


def worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import flash_hash_join_amd.distributed as D
        from flash_hash_join_amd import datagen, api
        api.initialize()
        shim = HostStagedDist()                            # handed to distributed_join as its transport
        nb, npk = 6_000_000, 40_000_000                    # global rows; block-distributed
        b0, b1 = rank * nb // world, (rank + 1) * nb // world
        p0, p1 = rank * npk // world, (rank + 1) * npk // world
        bk, bv = datagen.build_device(b1 - b0, "cuda:0", first=b0)
        pk, exp_local = datagen.probe_device(p1 - p0, nb, "cuda:0", seed=1, hit_bp=5000, first=p0)
        e = torch.tensor([exp_local]); dist.all_reduce(e); exp = int(e.item())
        res = {}
        # the pre-flight check of a multi-rank job (distributed.self_check): passes, and reports a transport that moves wrong data
        sk, sv = datagen.build_device(1_500_000, "cuda:0", first=rank * 1_500_000)      # (3M+ build rows in all: the chunk form applies)
        sp, se = datagen.probe_device(1_500_000, 1_500_000 * world, "cuda:0", seed=3, hit_bp=5000, first=rank * 1_500_000)
        et = torch.tensor([se]); dist.all_reduce(et)
        chk = D.self_check(shim, None, D.HipEngine("cuda:0"), (sk, sv, sp), int(et.item()), 300_000, transport=shim)
        assert chk["ok"] and chk["failed_ranks"] == 0, chk
        assert chk["precheck"] and chk["precheck"]["ok"] and chk["precheck"]["ran"] and chk["precheck"]["form"].startswith("chunks"), chk   # the forced-precheck leg
        chk = D.self_check(shim, None, D.HipEngine("cuda:0"), (sk, sv, sp), int(et.item()), 300_000, transport=shim, corrupt=True)
        assert not chk["ok"] and chk["failed_ranks"] == 1 and (rank != 0 or "elements received from rank" in chk["error"]), chk
        for strategy, variant in (("shuffle", ""), ("scatter", ""), ("shuffle", "prefilter"), ("shuffle", "auto"), ("broadcast", ""), ("auto", "")):
            os.environ["FJ_DIST_STRATEGY"] = strategy
            os.environ["FJ_DIST_PREFILTER"] = "1" if variant == "prefilter" else "auto" if variant == "auto" else "0"
            D._PRECHECK_MEMO.clear(); D._FORM_MEMO.clear()
            t = {}
            n, sec = D.distributed_join(bk, bv, pk, timings=t, transport=shim)
            assert n == exp, (strategy, n, exp)
            if strategy in ("broadcast", "auto"):           # the build-broadcast form through the same driver (forced / picked by its model: 6.7 probe rows per build row)
                assert t["strategy"] == "broadcast" and t["probe_rows_sent"] == 0 and t["local_build_rows"] == nb and t["local_probe_rows"] == p1 - p0, t
                assert 0 < t["wire_bytes_sent"] <= (world - 1) * (8.1 * (b1 - b0) + 300_000), t        # (6M build rows in all: 11 bits, a 4-byte high-word plane)
                assert "broadcast_form_error" not in t and "chunk_form_error" not in t, t
            if variant == "prefilter":                      # half the probe rows miss; the owners' per-partition filters (all-gathered inside the driver: uneven ranges at 3 ranks) stop nearly all of them
                assert t["prefilter"] and t["shuffle_form"].startswith("chunks") and t["filter_bytes_received"] > 0 and t["probe_rows_sent"] < 0.53 * (p1 - p0), t
            if variant == "auto":                           # every rank reaches the same verdict from the all-reduced sample
                assert t["prefilter_mode"] == "auto" and t["shuffle_form"].startswith("chunks"), t
                vs = [None] * world
                dist.all_gather_object(vs, (t["prefilter"], t["prefilter_sampled_survivors"]))
                assert len(set(vs)) == 1, vs
            tm = {}
            n2, sec, k, v = D.distributed_join(bk, bv, pk, materialize=True, return_arrays=True, transport=shim, timings=tm)
            M = torch.tensor(-7046029254386353131, dtype=torch.int64, device="cuda:0")
            assert n2 == exp and bool(torch.all((v + 1) * M == k)), strategy
            if strategy in ("broadcast", "auto"):            # materialising joins take the same form as counting ones: the regions carry the values too
                assert tm["strategy"] == "broadcast" and tm["shuffle_form"].startswith("build broadcast") and tm["probe_rows_sent"] == 0, tm
                assert (world - 1) * 14 * (b1 - b0) <= tm["wire_bytes_sent"] <= (world - 1) * (16.1 * (b1 - b0) + 300_000), tm     # (keys 6 - 8 bytes as above + the 8-byte value)
                assert "broadcast_form_error" not in tm and "chunk_form_error" not in tm, tm
            elif strategy != "scatter":                      # the chunk-form shuffle (values travel with the build rows)
                assert tm["shuffle_form"].startswith("chunks (fj_dist_join over a callback transport"), tm
                assert tm["prefilter"] == (variant == "prefilter") or variant == "auto", tm
                assert "chunk_form_error" not in tm, tm
            else:
                assert tm["shuffle_form"] == "owner-scatter" and t["shuffle_form"] == "owner-scatter", (t, tm)
            tot = torch.tensor([k.numel()]); dist.all_reduce(tot)
            assert int(tot.item()) == exp                   # the ranks' pair sets add up to the global result
            if strategy == "shuffle":
                assert t["shuffle_form"].startswith("chunks (fj_dist_join over a callback transport"), t
            res[strategy + variant] = (t["strategy"], t["pieces"] if "pieces" in t else None, t["local_build_rows"], t["local_probe_rows"])
        q.put((rank, exp, res))
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn"); q = ctx.Queue()
    ps = [ctx.Process(target=worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps: p.start()
    import queue as _queue
    import time as _time
    rows, t_end = [], _time.time() + 600
    while len(rows) < world:                       # (a worker that died says so at once instead of after the queue's timeout)
        try:
            rows.append(q.get(timeout=2))
        except _queue.Empty:
            dead = [p.exitcode for p in ps if p.exitcode not in (None, 0)]
            if dead or _time.time() > t_end:
                for p in ps:
                    if p.is_alive(): p.terminate()
                raise SystemExit(f"worker exit codes {[p.exitcode for p in ps]}" if dead else "timed out")
    for p in ps:
        p.join(timeout=60); assert p.exitcode == 0
    for r in sorted(rows): print(r)
    print("OK: %d ranks on one GPU, all strategies, counts and pairs exact" % world)
