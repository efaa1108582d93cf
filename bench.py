#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X hash-join hot path.

  python bench.py --gpus N --steps K --warmup W            (N=1 directly; N>1 under torch.distributed.run)

A "step" is one whole join (build phase + probe phase, count) of the workload over synthetic
uniform-random int64 keys that are already resident in HBM when the timed region starts:

  N = 1 : BASELINE.json configs[2], the configuration the metric's target is quoted on:
          hash_join_count_radix, 100M build x 1B probe rows, 50 % hit rate, one MI355X.
  N > 1 : workload "c5" = BASELINE.json configs[4] cut into its per-GPU shards: 125M build x 1.25B probe rows
          PER GPU (1B x 10B at N = 8; the same shard size at N = 2 and 4: weak scaling), block-distributed,
          joined in the form the C++ driver's cost model picks for those sizes and the link rate measured at
          start-up (flash_hash_join_amd/distributed.py, csrc/fj_dist.hip) - for configs[4] the BUILD
          BROADCAST (the probe rows stay, every rank's partitioned build rows travel to every peer over
          xGMI, 6 bytes per key); north_star's all-to-all of radix partitions, the OWNER SHUFFLE, is
          measured beside it in the same line (`alt_strategy`, with its own roofline block and wire bytes).
          FJ_DIST_STRATEGY=broadcast|shuffle|scatter pins the timed form.  `python bench.py --gpus N` without
          a torchrun environment starts `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a
          child process (before anything touches the GPU) and relays its output and exit code.

value = probes processed by all ranks / wall time of the K timed steps (max over ranks), in
G probes/s, end to end (build side included).  The probe-phase-only rate, the build time, the
roofline of the dominant kernel (probe-side partition pass) and the CPU baseline (the oracle's
restatement of the reference algorithm on this box's host cores, bounded sample; at N > 1: rank 0
runs the config-3-size join, labelled as such) ride along in the same JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
HBM_COPY_GBS = 6290.0          # measured float4-copy ceiling, same guide
HBM_POOL_COPY_GBS = 5745.0     # this pool's MI355X boxes: best plain read-one-write-one copy of a 158-point sweep (tools/ubench_copy2.hip, profiles/r06_ubench_copy.csv:
                               # 8-32 GiB, grid-stride and persistent-slice kernels, 1-8 loads in flight, nontemporal or not; hipMemcpyDtoD: 4.8-5.3 TB/s; round 1's 4 x 4 sweep: 5621).
                               # A radix pass streams 8 B in and 8 B out per key: a copy is its ceiling.  (Rounds 1-3 also quoted a
                               # "scatter ceiling" of 5038 GB/s from one grid size of tools/ubench_scatter.hip; the same benchmark reaches
                               # 5.2-5.5 TB/s with other grids - profiles/r02_ubench_scatter_by_grid.csv - and the pass ran at 1.008 of
                               # it: not a ceiling, dropped.)

WORKLOADS = {
    # name: (build rows per GPU, probe rows per GPU, hit rate in basis points, function)
    "c2": (1_000_000, 100_000_000, 5000, "hash_join_count"),
    "c2_hbm_table": (1_000_000, 100_000_000, 5000, "hash_join_count"),           # literal one-table algorithm (option)
    "c2_radix": (1_000_000, 100_000_000, 5000, "hash_join_count_radix"),
    "c3": (100_000_000, 1_000_000_000, 5000, "hash_join_count_radix"),
    "c4": (100_000_000, 1_000_000_000, 500, "hash_join_count_radix_bloom"),
    "c4_scalar_bloom": (100_000_000, 1_000_000_000, 500, "hash_join_count_bloom"),
    "c4_adaptive": (100_000_000, 1_000_000_000, 500, "adaptive_join_count"),             # the sampled hit rate turns the precheck on
    "c3_adaptive": (100_000_000, 1_000_000_000, 5000, "adaptive_join_count"),            # ... and leaves it off at 50 % hits
    "c4_hbm_table_bloom": (100_000_000, 1_000_000_000, 500, "hash_join_count_bloom"),   # literal one-table algorithm + bloom precheck
    "c3_mat": (100_000_000, 1_000_000_000, 5000, "hash_join_radix"),
    "small": (1_000_000, 10_000_000, 5000, "hash_join_count_radix"),
    # BASELINE configs[4] (1B x 10B over 8 GPUs) as its per-GPU shard; the default for --gpus > 1
    "c5": (125_000_000, 1_250_000_000, 5000, "hash_join_count_radix"),
    # the same shards at config 4's 5 % hit rate through the *_bloom function: at N > 1 the owner shuffle's sender-side
    # precheck decides from a sample whether to filter the probe exchange (distributed._prefilter_mode)
    "c5_bloom": (125_000_000, 1_250_000_000, 500, "hash_join_count_radix_bloom"),
    "c5_mat": (125_000_000, 1_250_000_000, 5000, "hash_join_radix"),                      # ... materialising (625M pairs per rank)
    # what ONE rank joins locally under the replicate-build multi-GPU strategy at N = 2, 4, 8 (c3 rows per GPU)
    "rep2": (200_000_000, 1_000_000_000, 5000, "hash_join_count_radix"),
    "rep4": (400_000_000, 1_000_000_000, 5000, "hash_join_count_radix"),
    "rep8": (800_000_000, 1_000_000_000, 5000, "hash_join_count_radix"),
    # ... and at config 5's shard sizes (125M x 1.25B per GPU): N = 2, 4, 8 ranks' build keys against one rank's probe rows
    "c5_rep2": (250_000_000, 1_250_000_000, 5000, "hash_join_count_radix"),
    "c5_rep4": (500_000_000, 1_250_000_000, 5000, "hash_join_count_radix"),
    "c5_rep8": (1_000_000_000, 1_250_000_000, 5000, "hash_join_count_radix"),
}


def _flush_c_stdio() -> None:
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()


def _self_launch(ngpus: int) -> int:
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ngpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    proc = subprocess.Popen(cmd, env=env, cwd=ROOT)
    return proc.wait()


def _host_cpu_facts() -> dict:
    """CPU model and the affinity mask of this process (BASELINE.md section 4 asks for both beside the core count)."""
    model = None
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    try:
        aff = sorted(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        aff = []
    ranges, i = [], 0
    while i < len(aff):                                  # compact "0-63,128-191" form
        j = i
        while j + 1 < len(aff) and aff[j + 1] == aff[j] + 1:
            j += 1
        ranges.append(str(aff[i]) if i == j else f"{aff[i]}-{aff[j]}")
        i = j + 1
    # physical cores and sockets (north_star asks for the core count; os_cpu_count counts hardware threads)
    cores, sockets = set(), set()
    try:
        phys = core = None
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("physical id"):
                    phys = line.split(":", 1)[1].strip()
                elif line.startswith("core id"):
                    core = line.split(":", 1)[1].strip()
                elif not line.strip():
                    if phys is not None and core is not None:
                        cores.add((phys, core)); sockets.add(phys)
                    phys = core = None
    except OSError:
        pass
    return {"cpu_model": model, "os_cpu_count": os.cpu_count(), "physical_cores": len(cores) or None, "sockets": len(sockets) or None,
            "affinity_cpus": len(aff), "affinity_mask": ",".join(ranges)}


def cpu_baseline(device, sample_b: int, sample_p: int, hit_bp: int, algo: str = "adaptive", budget_s: float = 25.0) -> dict:
    """Oracle restatement of the reference CPU path (hash_join.cpp:498-534 radix count; algo="scalar": :536-567), all host
    cores, on a bounded sample of the same generator.  A reported baseline, not a target."""
    import numpy as np
    from flash_hash_join_amd import datagen
    from oracle import oracle as O
    O.build()
    bk, bv = datagen.build_device(sample_b, device)
    pk, exp = datagen.probe_device(sample_p, sample_b, device, seed=1, hit_bp=hit_bp)
    hbk, hbv, hpk = (x.cpu().numpy().view(np.uint64) for x in (bk, bv, pk))
    del bk, bv, pk
    cores = O.lib().fjo_default_threads()
    best = None
    t_all = time.perf_counter()
    for _ in range(3):                                   # bounded: stop after ~25 s of CPU runs
        n, sec = O.c_join(hbk, hbv, hpk, algo=algo, bloom=False, materialize=False, threads=0)
        assert n == exp, (n, exp)
        best = sec if best is None else min(best, sec)
        if time.perf_counter() - t_all > budget_s:
            break
    fn = {"adaptive": "adaptive_join_count (radix path)", "scalar": "hash_join_count (one table)", "radix": "hash_join_count_radix"}[algo]
    return {"value": round(sample_p / best / 1e9, 5), "unit": "Gprobes/s", "cores": int(cores), "kind": "port",
            "core_seconds": round(best, 4),
            "sample": f"{fn} restatement, {sample_b} build x {sample_p} probe rows, "
                      f"{hit_bp / 100:.0f}% hits, best of <=3, core_duration_sec={best:.3f}s",
            "hw_crc32c": bool(O.lib().fjo_uses_hw_crc()), **_host_cpu_facts()}


def host_entry(device) -> dict:
    """The drop-in NumPy entry (fj_join_host) at BASELINE configs[1] sizes (1M x 100M rows, 816 MB in): wall time of
    flash_join.hash_join_count on pageable NumPy arrays - copy through the pinned ring with the join's first pass running
    under it - next to the time a plain pinned-memory H2D copy of the same bytes takes on this box.  Never `value`."""
    import numpy as np
    import torch
    import flash_join
    from flash_hash_join_amd import api, datagen
    nb, npk = 1_000_000, 100_000_000
    bk, bv = datagen.build_device(nb, device)
    pk, exp = datagen.probe_device(npk, nb, device, seed=1, hit_bp=5000)
    hbk, hbv, hpk = (x.cpu().numpy().view(np.uint64) for x in (bk, bv, pk))
    del bk, bv, pk
    walls = []
    for _ in range(4):
        t0 = time.perf_counter()
        n, sec = flash_join.hash_join_count(hbk, hbv, hpk)
        walls.append((time.perf_counter() - t0) * 1e3)
        assert n == exp, (n, exp)
    lt = api.last_timings()
    nbytes = (2 * nb + npk) * 8
    pin = torch.empty(nbytes // 8, dtype=torch.int64).pin_memory()
    dst = torch.empty(nbytes // 8, dtype=torch.int64, device=device)
    h2d = []
    for _ in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        dst.copy_(pin, non_blocking=True)
        torch.cuda.synchronize()
        h2d.append((time.perf_counter() - t0) * 1e3)
    w, h = min(walls[1:]), min(h2d[1:])
    return {"workload": "hash_join_count on NumPy arrays, 1000000 build x 100000000 probe rows (BASELINE configs[1] sizes)",
            "bytes_in": nbytes, "wall_incl_pcie_ms": round(w, 3), "pinned_h2d_only_ms": round(h, 3), "wall_over_h2d": round(w / h, 3),
            "h2d_GBps": round(nbytes / h / 1e6, 1), "gprobes_per_s_incl_pcie": round(npk / w / 1e6, 3),
            "device_resident_ms": round(lt["total_ms"], 3), "streamed_under_copy": bool(lt["host_streamed"])}


def dist_roofline(strategy: str, part_ms, launches: int, units: float, wire_chunk_bytes) -> dict:
    """Roofline block of a multi-GPU step's dominant kernel, the probe-side radix pass: over this rank's own probe rows (build
    broadcast, owner scatter: 8 B read + 8 B written per key) or over a received piece in the wire format (owner shuffle: 7 B read
    when the chunks are 1792 bytes, + 8 B written); launch times by HIP events on the join stream (fj_timings.probe_part_kernel_ms)."""
    avg_ms = sum(part_ms) / len(part_ms)
    bpu = 8.0 + (float(wire_chunk_bytes) / 256.0 if (strategy == "owner-shuffle" and wire_chunk_bytes) else 8.0)
    alg_bytes = bpu * units
    achieved = alg_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms else 0.0
    return {"bound": "hbm", "kernel": "fj_partition_kernel<keys-only> (the owner's radix pass over a received piece)" if strategy == "owner-shuffle" else
            "fj_partition_kernel<keys> (a probe-side radix pass over this rank's own probe rows)" if strategy == "build-broadcast" else
            "fj_partition_kernel<keys> (a probe-side radix pass of the owner's join over the rows it received)",
            "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
            "frac_of_copy_ceiling": round(achieved / HBM_COPY_GBS, 4),
            "frac_of_this_pools_copy_rate": round(achieved / HBM_POOL_COPY_GBS, 4),
            "algorithmic_bytes_per_launch": alg_bytes, "bytes_per_unit": bpu, "units_per_launch": units,
            "note": "launch time by HIP events on the join stream: at N > 1 the pass shares the GPU with the exchange's kernels and the next piece's packing kernels (other streams), so this is an upper bound of the kernel's own time",
            "avg_launch_ms": round(avg_ms, 4), "launches_timed": launches, "traffic": None}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS),
                    help="default: c3 (BASELINE configs[2]) at --gpus 1, c5 (configs[4], 125M x 1.25B rows per GPU) at --gpus > 1")
    ap.add_argument("--scale", type=float, default=1.0, help="scale the row counts (debugging)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-entry", action="store_true", help="skip the PCIe-inclusive measurement of the NumPy entry")
    ap.add_argument("--selfcheck-timeout", type=float, default=240.0, help="N > 1: seconds the pre-flight may take before the run is declared stuck")
    ap.add_argument("--selfcheck-corrupt", action="store_true", help=argparse.SUPPRESS)      # test hook: rank 0 flips one received bit in the pre-flight exchange
    args = ap.parse_args()
    if args.workload is None:
        args.workload = "c3" if args.gpus == 1 else "c5"

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # launched the way the driver launches the 1-GPU bench (`python bench.py --gpus N ...`): become the launcher.
        # Nothing in this process has touched the GPU (torch is not even imported yet); the ranks run in a child
        # process tree, their output and exit code are relayed.
        sys.exit(_self_launch(args.gpus))

    import torch
    import torch.distributed as dist
    from flash_hash_join_amd import api, datagen
    from flash_hash_join_amd.distributed import HipEngine, distributed_join

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if args.gpus > 1:
            raise SystemExit(f"--gpus {args.gpus} needs `python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py ...` "
                             f"(WORLD_SIZE={world})")
    # FJ_BENCH_SHARE_GPU=1 (self-test on a 1-GPU box): every rank uses cuda:0 and the collectives travel over gloo through
    # the host (RCCL refuses two ranks on one device) - exercises this file's N>1 logic, says nothing about speed
    share_gpu = bool(os.environ.get("FJ_BENCH_SHARE_GPU")) and world > 1
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    # FJ_BENCH_FORCE_DIST=1 drives the multi-GPU code path (owner split -> RCCL all-to-all -> join -> all-reduce)
    # on a single rank: a self-test of the N>1 branch on boxes with one GPU; results are identical.
    force_dist = bool(os.environ.get("FJ_BENCH_FORCE_DIST")) and world == 1
    transport = None                             # what distributed_join uses instead of torch.distributed (share_gpu only)
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        if share_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            from two_ranks_one_gpu import HostStagedDist
            transport = dist = HostStagedDist()
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        _flush_c_stdio()
        # `value` at N > 1: what an unconfigured job runs - the form the C++ driver's cost model picks for the step's sizes with the
        # link rate measured below (fj_dist_model: build broadcast or owner shuffle); the other form is measured beside it as
        # `alt_strategy`, never instead of it.  FJ_DIST_STRATEGY=broadcast|shuffle|scatter pins the timed form.
        pinned_strategy = os.environ.get("FJ_DIST_STRATEGY")
        if pinned_strategy in ("", "auto"):
            pinned_strategy = None

    nb_gpu, np_gpu, hit_bp, fn_name = WORKLOADS[args.workload]
    nb_gpu, np_gpu = max(1, int(nb_gpu * args.scale)), max(1, int(np_gpu * args.scale))
    nb_total, np_total = nb_gpu * world, np_gpu * world
    algo = {"hash_join_count": api.ALGO_SCALAR, "hash_join_count_bloom": api.ALGO_SCALAR, "adaptive_join_count": api.ALGO_ADAPTIVE,
            "adaptive_join_count_bloom": api.ALGO_ADAPTIVE}.get(fn_name, api.ALGO_RADIX)
    bloom = int("bloom" in fn_name)
    materialize = int(fn_name in ("hash_join_radix", "hash_join"))

    api.initialize() if local_rank == 0 else api.context(local_rank)

    api.set_option("scalar_hbm_table", int("hbm_table" in args.workload))
    bk, bv = datagen.build_device(nb_gpu, device, first=rank * nb_gpu)
    pk, exp_local = datagen.probe_device(np_gpu, nb_total, device, seed=1, hit_bp=hit_bp, first=rank * np_gpu)
    exp_total = exp_local
    if world > 1:
        e = torch.tensor([exp_local], dtype=torch.int64, device=device)
        dist.all_reduce(e)
        exp_total = int(e.item())
    engine = HipEngine(device) if (world > 1 or force_dist) else None
    link = None
    if world > 1 and not share_gpu:
        # one all-to-all of 128 MiB per peer: the per-link rate this node delivers (recorded; what FJ_DIST_STRATEGY=auto and the
        # sender-side precheck's break-even price with).  Every rank must end up with the same rate, or with none.
        from flash_hash_join_amd import distributed as _D
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import xgmi_probe
            link = xgmi_probe.measure(dist, device, 128)
        except Exception as ex:
            link = {"error": repr(ex)}
        okf = torch.tensor([0 if "error" in link else 1], dtype=torch.int64, device=device)
        dist.all_reduce(okf, op=dist.ReduceOp.MIN)
        if int(okf.item()) == 1:
            _D.set_link_rate(link["link_GBps"] * 1e9)
        elif "error" not in link:
            link = {"error": "the link probe failed on another rank", **link}

    form_pick = None
    if world > 1 or force_dist:
        from flash_hash_join_amd import distributed as _D
        if pinned_strategy is None:
            # the driver's model for THIS step's sizes and the measured link rate (every rank computes the same: the rate was agreed on above);
            # pinned for the self-check and the timed steps so that what was checked is what is timed
            form_pick = _D.form_model(world, nb_gpu, np_gpu, materialize=bool(materialize))
            if world == 1:
                form_pick = dict(form_pick, pick="shuffle", note="one rank: the shuffle (FJ_DIST_STRATEGY=broadcast times the other form on one rank)")
            os.environ["FJ_DIST_STRATEGY"] = form_pick["pick"]
        timed_strategy = os.environ["FJ_DIST_STRATEGY"]

    # ---- self-check before anything is timed (N > 1): the exchange transport at the step's largest message size, and one
    #      small join against its closed-form count; a failure is ONE JSON line with `error` and a non-zero exit code ----------
    selfcheck = None
    if world > 1 or force_dist:
        import threading
        from flash_hash_join_amd.distributed import self_check
        nb_s, np_s = max(400_000, 3_200_000 // world), 4_000_000
        sbk, sbv = datagen.build_device(nb_s, device, first=rank * nb_s)
        spk, sexp = datagen.probe_device(np_s, nb_s * world, device, seed=3, hit_bp=5000, first=rank * np_s)
        e = torch.tensor([sexp], dtype=torch.int64, device=device)
        dist.all_reduce(e)
        pieces = int(os.environ.get("FJ_DIST_PIECES", "0")) or 4       # (0 / unset: the driver decides - 4, or 8 for a wire-bound broadcast step: smaller messages)
        msg = int(1.3 * np_gpu / max(1, pieces) / world) + 4096          # int64 per peer and piece in the chunk form

        def error_line(err, sc):
            return json.dumps({"error": err, "self_check": sc, "rank": rank, "n_gpus": world, "torch": torch.__version__,
                               "hip": getattr(torch.version, "hip", None),
                               "rccl": ".".join(str(x) for x in torch.cuda.nccl.version()) if hasattr(torch.cuda, "nccl") else None,
                               "metric": "probe throughput (billion probes/sec), int64 keys, whole join (build + probe phases) per step",
                               "value": None})

        # a collective that never returns (mismatched sends on some rank) must not become a silent driver timeout
        done = threading.Event()

        def watchdog():
            if not done.wait(args.selfcheck_timeout):
                if rank == 0:
                    print(error_line("the multi-rank self-check did not finish (a collective is stuck)", None), flush=True)
                os._exit(3)
        threading.Thread(target=watchdog, daemon=True).start()
        # the default protocol first (chunk-form shuffle through the native entry); a form that fails its check - the ranks agree
        # on that - is replaced by the next simpler one, and the line says which one was timed
        scatter_form = [("owner-scatter", {"FJ_DIST_STRATEGY": "scatter"})]
        shuffle_forms = [("chunks (fj_dist_join over RCCL)", {"FJ_DIST_STRATEGY": "shuffle"}),
                         ("chunks (fj_dist_join over torch.distributed callbacks)", {"FJ_DIST_STRATEGY": "shuffle", "FJ_DIST_NATIVE": "0"})] + scatter_form
        forms = {"broadcast": [("build broadcast (fj_dist_join over RCCL)", {"FJ_DIST_STRATEGY": "broadcast"}),
                               ("build broadcast (fj_dist_join over torch.distributed callbacks)", {"FJ_DIST_STRATEGY": "broadcast", "FJ_DIST_NATIVE": "0"})] + shuffle_forms,
                 "scatter": scatter_form}.get(timed_strategy, shuffle_forms)
        user_pins = {k: os.environ[k] for k in ("FJ_DIST_NATIVE",) if k in os.environ}
        if pinned_strategy is not None:
            user_pins["FJ_DIST_STRATEGY"] = pinned_strategy
        tried = []
        for name, env in forms:
            if any(k in user_pins and user_pins[k] != v for k, v in env.items()):
                continue                                     # the user pinned a form: honour it
            os.environ.update(env)
            selfcheck = self_check(dist, None, engine, (sbk, sbv, spk), int(e.item()), msg, transport=transport, corrupt=args.selfcheck_corrupt)
            selfcheck["shuffle_form_checked"] = name
            tried.append({"form": name, "ok": selfcheck["ok"], "error": selfcheck["error"]})
            if selfcheck["ok"]:
                break
        timed_strategy = os.environ["FJ_DIST_STRATEGY"]
        selfcheck["forms_tried"] = tried
        if selfcheck.get("precheck") and selfcheck["precheck"].get("action"):      # the precheck failed its check (the ranks agree): off for the timed steps
            os.environ["FJ_DIST_PREFILTER"] = "0"
        done.set()
        del sbk, sbv, spk
        if not selfcheck["ok"]:
            if rank == 0:
                print(error_line(selfcheck["error"] or "self-check failed on another rank", selfcheck), flush=True)
            try:
                dist.destroy_process_group()
            except Exception:
                pass
            sys.exit(3)

    units_per_launch = [float(np_gpu)]
    strategy_seen = ["single GPU"]
    part_ms, part_launches, phase = [], 0, {"build_ms": [], "probe_ms": [], "join_ms": [], "total_ms": [], "emit_ms": [], "filter_ms": []}
    kern = {}                       # single GPU: kernel name -> [(launch ms, algorithmic bytes)] over the timed steps
    dtimes = {"split_s": [], "exchange_s": [], "join_s": []}
    dlast = {}
    dsampled = {}

    def record_kernels(lt) -> None:
        """Algorithmic HBM bytes of every timed kernel launch of one single-GPU join: a plain probe-side
        pass reads and writes every key it is given (16 B); the bloom filter kernel reads every probe key, writes the
        survivors and reads the level's build keys once (8 P + 8 S + 8 B); passes behind it see survivors only; the join reads its probe keys and the build keys once (+ 16 B per emitted pair in the emit pass)."""
        P, B, S = float(np_gpu), float(nb_gpu), float(lt["filter_survivors"])
        L = lt["bloom_level"]
        for i in range(min(4, lt["passes"])):
            ms = lt["probe_part_kernel_ms"][i]
            if L and i >= L:
                kern.setdefault("fj_partition_kernel<keys> over the filter's survivors", []).append((ms, 16 * S))
            else:
                kern.setdefault("fj_partition_kernel<keys> (probe-side radix pass)", []).append((ms, 16 * P))
        if L:
            kern.setdefault("fj_bloom_filter_kernel (bloom precheck between the probe-side passes)", []).append((lt["filter_ms"], 8 * P + 8 * S + 8 * B))
        if lt["path"] == 0 and materialize and lt["emit_ms"] == 0 and exp_local > 0:
            # the single-pass materialising join: one kernel builds, probes and emits (16 B per pair)
            kern.setdefault("fj_emit_join_persistent, single-pass form (table build + probe + emit)", []).append((lt["join_ms"], 8 * (S if L else P) + 16 * B + 16.0 * exp_local))
        elif lt["path"] == 0:
            kern.setdefault("join kernel (per-partition LDS table build + probe)", []).append((lt["join_ms"], 8 * (S if L else P) + (16 if materialize else 8) * B))
            if materialize and lt["emit_ms"] > 0:
                kern.setdefault("emitting join kernel (second pass of a materialising join)", []).append((lt["emit_ms"], 8 * (S if L else P) + 16 * B + 16.0 * exp_local))

    def step(record: bool) -> int:
        nonlocal part_launches
        if world == 1 and not force_dist:
            res = api.join_device(algo, bloom, materialize, bk, bv, pk, return_arrays=False)
        else:
            t = {}
            # a materialising step writes its pairs inside the timed region (return_arrays: the pairs this rank keeps; dropped at once)
            res = distributed_join(bk, bv, pk, materialize=bool(materialize), bloom=bool(bloom), engine=engine, timings=t, transport=transport, force_exchange=force_dist,
                                   return_arrays=bool(materialize))
            if materialize:
                res = (res[0], res[1])
            if t.get("prefilter_sampled_survivors") is not None:       # (warm-up steps included: "auto" samples once per shape, then remembers)
                dsampled.update(survivors=t["prefilter_sampled_survivors"], below=t.get("prefilter_below"))
            if record:
                for k in dtimes:
                    dtimes[k].append(t.get(k, 0.0))
                dlast.update(t)
        if record:
            lt = api.last_timings()
            if world == 1 and not force_dist:
                npart = 0
                record_kernels(lt)
            elif t.get("strategy") == "broadcast":
                npart = min(4, int(lt["passes"]))          # probe rows never move: the plan's passes, each over all of this rank's probe rows
                units_per_launch[0] = float(np_gpu)
                strategy_seen[0] = "build-broadcast"
            elif t.get("strategy") == "scatter":
                npart = min(4, int(lt["passes"]))          # the owner's plain single-GPU join of what it received
                units_per_launch[0] = float(t.get("local_probe_rows", np_gpu))
                strategy_seen[0] = "owner-scatter"
            else:
                npart = min(4, int(t.get("pieces", 0)))    # pipelined exchange: first-pass launches, one per received piece
                units_per_launch[0] = t.get("local_probe_rows", np_gpu) / max(1, int(t.get("pieces", 1)))
                strategy_seen[0] = "owner-shuffle"
            for i in range(npart):
                part_ms.append(lt["probe_part_kernel_ms"][i]); part_launches += 1
            phase["build_ms"].append(lt["build_phase_ms"]); phase["probe_ms"].append(lt["probe_phase_ms"])
            phase["join_ms"].append(lt["join_ms"]); phase["total_ms"].append(lt["total_ms"]); phase["emit_ms"].append(lt["emit_ms"])
            phase["filter_ms"].append(lt["filter_ms"])
        return int(res[0])

    def sync():
        if world > 1 or force_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # --warmup 0: one untimed priming step anyway, so that the workspace allocation (hipMalloc of ~20 GB of pools, ~0.5 s) and
    # at N>1 RCCL's communicator / channel set-up (seconds) are not what the first timed step measures
    priming = 1 if args.warmup == 0 else 0
    for _ in range(args.warmup + priming):
        got = step(False)
        assert got == exp_total or os.environ.get("FJ_JOIN_ABLATE"), f"warmup count {got} != expected {exp_total}"
    sync()
    t0 = time.perf_counter()
    bad = 0
    for _ in range(args.steps):
        got = step(True)
        bad += int(got != exp_total)               # every timed step is checked against the closed-form count
    sync()
    elapsed = time.perf_counter() - t0
    assert bad == 0 or os.environ.get("FJ_JOIN_ABLATE"), f"{bad} of {args.steps} timed steps returned a wrong count (last {got}, expected {exp_total})"
    if world > 1:
        e = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(e, op=dist.ReduceOp.MAX)
        elapsed = float(e.item())

    lt = api.last_timings()
    ms_per_step = elapsed / args.steps * 1e3
    value = np_total * args.steps / elapsed / 1e9
    mean = lambda xs: (sum(xs) / len(xs)) if xs else 0.0

    # ---- roofline of the dominant kernel --------------------------------------------------------------
    # radix workloads: the probe-side partition pass, 8 B read + 8 B written per key, one
    # launch per pass over all of this rank's probe rows; its duration comes from HIP events recorded around
    # each launch on the join's stream (fj_timings.probe_part_kernel_ms).
    roof = None
    roof_all = None
    if kern:
        # single GPU: every timed kernel with its own bytes; the DOMINANT kernel is the one with the largest share of the step
        rows = []
        for name, xs in kern.items():
            avg_ms = mean([x[0] for x in xs]); alg = mean([x[1] for x in xs])
            per_step = len(xs) / args.steps
            ach = alg / (avg_ms * 1e-3) / 1e9 if avg_ms else 0.0
            rows.append({"kernel": name, "avg_launch_ms": round(avg_ms, 4), "launches_per_step": round(per_step, 2),
                         "algorithmic_bytes_per_launch": alg, "achieved": round(ach, 1), "frac": round(ach / HBM_PEAK_GBS, 4)})
        for r in rows:
            if r["frac"] > 1.0 and not (os.environ.get("FJ_JOIN_ABLATE") or os.environ.get("FJ_BLOOM_ABLATE")):   # a kernel cannot beat the HBM peak on its own bytes: the accounting above must be wrong
                raise SystemExit(f"bench.py: roofline accounting error, {r['kernel']} shows {r['frac']} of the HBM peak: {r}")
        rows.sort(key=lambda r: -r["avg_launch_ms"] * r["launches_per_step"])
        roof_all = rows
        d = rows[0]
        traffic = None
        tp = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tp) and args.workload == "c3" and args.scale == 1.0:
            try:      # HBM bytes per launch from rocprofv3 PMC passes (tools/pmc.sh + tools/traffic_json.py), committed together
                      # with the hash of the kernel sources they measured: stale numbers (a kernel changed since) are not reported
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                from source_hash import kernel_source_hash
                tj = json.load(open(tp))
                traffic = tj.get("fj_partition_kernel_keys_bytes_per_launch") if tj.get("source_sha256") == kernel_source_hash() else None
            except Exception:
                traffic = None
        roof = {"bound": "hbm", "kernel": d["kernel"], "achieved": d["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": d["frac"],
                "frac_of_copy_ceiling": round(d["achieved"] / HBM_COPY_GBS, 4),
                "frac_of_this_pools_copy_rate": round(d["achieved"] / HBM_POOL_COPY_GBS, 4),
                "algorithmic_bytes_per_launch": d["algorithmic_bytes_per_launch"], "avg_launch_ms": d["avg_launch_ms"],
                "launches_timed": int(round(d["launches_per_step"] * args.steps)), "traffic": traffic}
    elif part_ms:
        roof = dist_roofline(strategy_seen[0], part_ms, part_launches, units_per_launch[0], dlast.get("wire_chunk_bytes"))
    else:
        # non-partitioned workloads (one table in HBM): the probe kernel dominates; 8 B per probe key
        avg_ms = mean(phase["join_ms"])
        alg_bytes = 8.0 * np_gpu
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms else 0.0
        roof = {"bound": "hbm", "kernel": "fj_gt_probe_kernel (one table in HBM)", "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                "algorithmic_bytes_per_launch": alg_bytes, "bytes_per_unit": 8, "units_per_launch": np_gpu,
                "avg_launch_ms": round(avg_ms, 4), "launches_timed": args.steps, "traffic": None}

    # phase-level accounting with SURVEY 8(d)'s formula for the declared k
    k = lt["passes"]
    probe_ms, build_ms = mean(phase["probe_ms"]), mean(phase["build_ms"])
    phase_schedule = "as timed (build relation first, then the probe relation, one stream: disjoint phases)"
    serial_total_ms = None
    phases = {
        "k_radix_passes": k, "radix_bits": lt["radix_bits"], "partitions": lt["partitions"], "path": lt["path"],
        "build_phase_ms": round(build_ms, 3), "probe_phase_ms": round(probe_ms, 3), "join_kernel_ms": round(mean(phase["join_ms"]), 3),
        "device_total_ms": round(mean(phase["total_ms"]), 3),
        "bloom_level": lt["bloom_level"], "bloom_filter_kernel_ms": round(mean(phase["filter_ms"]), 3) if lt["bloom_level"] else None,
        "bloom_survivors": lt["filter_survivors"] if lt["bloom_level"] else None, "sampled_hit_bp": lt["sampled_hit_bp"],
        "phase_schedule": phase_schedule, "device_total_serial_ms": round(serial_total_ms, 3) if serial_total_ms else None,
        "probe_phase_gprobes_per_s": round(np_gpu / (probe_ms * 1e-3) / 1e9, 2) if probe_ms else None,
        "probe_phase_algorithmic_bytes": (24 * k + 8) * np_gpu,
        "probe_phase_frac_of_hbm_peak": round((24 * k + 8) * np_gpu / (probe_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if probe_ms else None,
        # the implementation reads no histogram pass: it moves 16 B per key and pass + 8 B per probe
        # (a filtered plan moves fewer: not quoted there)
        "probe_phase_bytes_moved": (16 * k + 8) * np_gpu if not lt["bloom_level"] else None,
        "probe_phase_frac_of_hbm_peak_by_bytes_moved": round((16 * k + 8) * np_gpu / (probe_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        if probe_ms and not lt["bloom_level"] else None,
        "build_phase_algorithmic_bytes": (40 * k + 16) * nb_gpu,
    }
    if world > 1 or force_dist:
        phases.update({kk: round(mean(v) * 1e3, 3) for kk, v in (("split_ms", dtimes["split_s"]), ("exchange_ms", dtimes["exchange_s"]),
                                                                 ("local_join_ms", dtimes["join_s"]))})
        phases.update({"shuffle_prefilter": dlast.get("prefilter"), "shuffle_prefilter_mode": dlast.get("prefilter_mode"),
                       "shuffle_prefilter_decision": dlast.get("prefilter_decision"),
                       "shuffle_prefilter_sampled_survivors": dsampled.get("survivors"), "shuffle_prefilter_break_even": dsampled.get("below"),
                       "filter_bytes_received_rank0": dlast.get("filter_bytes_received"),
                       "probe_rows_sent_rank0": dlast.get("probe_rows_sent"), "shuffle_form": dlast.get("shuffle_form"),
                       # what rank 0 put on its links per step (chunks + directory words; its own share never travels) and per key sent
                       "wire_chunk_bytes": dlast.get("wire_chunk_bytes"), "wire_bytes_sent_rank0": dlast.get("wire_bytes_sent"),
                       "chunk_form_error": dlast.get("chunk_form_error"), "broadcast_form_error": dlast.get("broadcast_form_error"),
                       "strategy_timed": dlast.get("strategy"), "form_model": form_pick,
                       # what the passes left to the transport's kernels in the last timed step, and the measurements behind that choice
                       "cu_reserve": dlast.get("cu_reserve"),
                       "pieces": dlast.get("pieces")})            # (FJ_DIST_PIECES unset: the driver's own choice - 4, or 8 for a wire-bound broadcast step)

    out = {
        "metric": "probe throughput (billion probes/sec), int64 keys, whole join (build + probe phases) per step",
        "value": round(value, 3), "unit": "Gprobes/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "int64", "data": "synthetic",
        "config": {"workload": f"{fn_name}: {nb_gpu} build x {np_gpu} probe int64 rows per GPU, {hit_bp / 100:.0f}% hit rate"
                               + (" (BASELINE configs[2])" if args.workload == "c3" and args.scale == 1.0 else "")
                               + (f" (BASELINE configs[4] cut into per-GPU shards: {nb_total} x {np_total} rows over {world} GPUs"
                                  + (" = the full 1B x 10B)" if world == 8 else ")") if args.workload == "c5" and args.scale == 1.0 else ""),
                   "function": fn_name, "build_rows_total": nb_total, "probe_rows_total": np_total,
                   "matches": exp_total, "bench_workload": args.workload,
                   "options": {k: api.get_option(k) for k in ("scalar_hbm_table", "persistent_min_items", "radix_threshold", "bloom_auto", "bloom_variant")},
                   "parallelism": f"{strategy_seen[0]} x{world}" if (world > 1 or force_dist) else "single GPU"},
        "build_time_ms": round(build_ms, 3),
        "phases": phases,
        "roofline": roof,
    }
    if roof_all:
        out["roofline_kernels"] = roof_all
    if rank == 0 and world == 1 and not args.no_host_entry and not force_dist:
        try:
            del bk, bv, pk
        except NameError:
            pass
        torch.cuda.empty_cache()
        try:
            out["host_entry"] = host_entry(device)
        except Exception as ex:
            out["host_entry"] = {"error": repr(ex)}
    if link is not None:
        out["xgmi_all_to_all"] = link
    if selfcheck is not None:
        out["self_check"] = selfcheck
    if (world > 1 or force_dist) and not materialize and timed_strategy in ("broadcast", "shuffle"):
        # the OTHER form of the step - the owner shuffle (north_star's all-to-all of radix partitions) when the broadcast was timed, and
        # the other way round - as a second, labelled measurement with its own roofline block and wire bytes; never `value`
        main = dlast.get("strategy", timed_strategy)
        alt = "shuffle" if main == "broadcast" else "broadcast"
        try:
            asteps = max(1, min(3, args.steps))
            ta_ = {}
            a_part, a_launch, a_units, a_label = [], 0, float(np_gpu), "owner-shuffle" if alt == "shuffle" else "build-broadcast"
            got = distributed_join(bk, bv, pk, bloom=bool(bloom), engine=engine, transport=transport, timings=ta_, strategy=alt, force_exchange=force_dist)[0]      # untimed (workspace growth)
            sync()
            ta = time.perf_counter()
            for _ in range(asteps):
                got = distributed_join(bk, bv, pk, bloom=bool(bloom), engine=engine, transport=transport, timings=ta_, strategy=alt, force_exchange=force_dist)[0]
                lta = api.last_timings()
                if ta_.get("strategy") == "broadcast":
                    npart_a, a_units, a_label = min(4, int(lta["passes"])), float(np_gpu), "build-broadcast"
                elif ta_.get("strategy") == "scatter":
                    npart_a, a_units, a_label = min(4, int(lta["passes"])), float(ta_.get("local_probe_rows", np_gpu)), "owner-scatter"
                else:
                    npart_a, a_units, a_label = min(4, int(ta_.get("pieces", 0))), ta_.get("local_probe_rows", np_gpu) / max(1, int(ta_.get("pieces", 1))), "owner-shuffle"
                for i in range(npart_a):
                    a_part.append(lta["probe_part_kernel_ms"][i]); a_launch += 1
            sync()
            ea = torch.tensor([time.perf_counter() - ta], dtype=torch.float64, device=device)
            if world > 1:
                dist.all_reduce(ea, op=dist.ReduceOp.MAX)
            out["alt_strategy"] = {"strategy": a_label, "asked_for": alt,
                                   "form": ta_.get("shuffle_form"), "steps": asteps, "ms_per_step": round(float(ea.item()) / asteps * 1e3, 3),
                                   "value": round(np_total * asteps / float(ea.item()) / 1e9, 3), "unit": "Gprobes/s", "count_ok": int(got) == exp_total,
                                   "wire_bytes_sent_rank0": ta_.get("wire_bytes_sent"), "wire_chunk_bytes": ta_.get("wire_chunk_bytes"),
                                   "roofline": dist_roofline(a_label, a_part, a_launch, a_units, ta_.get("wire_chunk_bytes")) if a_part else None,
                                   "fell_to": None if ta_.get("strategy") == alt else ta_.get("strategy"),
                                   "errors": {k: ta_[k] for k in ("broadcast_form_error", "chunk_form_error") if ta_.get(k)} or None,
                                   "note": "not `value`: the other form of the multi-GPU step, measured beside the timed one"}
        except Exception as ex:
            out["alt_strategy"] = {"strategy": alt, "error": repr(ex)}
    if rank == 0 and not args.no_cpu_baseline and not force_dist:
        try:
            # N = 1: the full workload on the host when it fits (the oracle needs ~2.5x the input bytes), else a 1/10 sample.
            # N > 1: rank 0 runs the config-3-size join (100M x 1B) and the block says so ("c5's CPU column is the
            # config-3-size run, labelled as such"); the other ranks wait at the final barrier meanwhile.
            import psutil
            cb_, cp_ = (nb_gpu, np_gpu) if world == 1 else (int(100_000_000 * args.scale) or 1, int(1_000_000_000 * args.scale) or 1)
            need = (cb_ * 16 + cp_ * 8) * 2.5
            full = psutil.virtual_memory().available > need + (8 << 30)
            sb, sp = (cb_, cp_) if full else (max(1, cb_ // 10), max(1, cp_ // 10))
            if world == 1:
                try:
                    del bk, bv, pk
                except NameError:
                    pass
                torch.cuda.empty_cache()
            out["cpu_baseline"] = cpu_baseline(device, sb, sp, hit_bp)
            if world > 1:
                out["cpu_baseline"]["sample"] = "config-3-size run on rank 0's host cores (not this step's 1/N share of config 5): " + out["cpu_baseline"]["sample"]
        except Exception as ex:      # the baseline never blocks the GPU measurement
            out["cpu_baseline"] = {"error": repr(ex)}
        # "Configs 1-2 always" - the reference's own CPU-runnable cases (hash_join_count, one table),
        # a few seconds each
        for tag, cb, cp in (("cpu_baseline_c1", 1_000_000, 10_000_000), ("cpu_baseline_c2", 1_000_000, 100_000_000)) if world == 1 else ():
            try:
                out[tag] = cpu_baseline(device, cb, cp, 5000, algo="scalar", budget_s=6.0)
            except Exception as ex:
                out[tag] = {"error": repr(ex)}
    if priming:
        out["priming_steps"] = priming
    if share_gpu:
        out["note"] = "FJ_BENCH_SHARE_GPU: ranks share cuda:0, gloo + host-staged collectives - a functional self-test, not a measurement"
    if world > 1 or force_dist:
        dist.barrier()
        dist.destroy_process_group()
    _flush_c_stdio()                      # RCCL's banner sits in C stdio buffers: get it out before the result line
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
