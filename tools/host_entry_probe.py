"""The NumPy entry (fj_join_host) at BASELINE configs[1] sizes by number of copy threads, next to this box's NUMA facts:
wall time incl. PCIe vs. a plain pinned H2D copy (bench.host_entry).  usage: python tools/host_entry_probe.py [threads ...]"""
import glob, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def numa_facts():
    nodes = {}
    for d in sorted(glob.glob("/sys/devices/system/node/node[0-9]*")):
        try:
            nodes[os.path.basename(d)] = open(os.path.join(d, "cpulist")).read().strip()
        except OSError:
            pass
    gpus = {}
    for d in glob.glob("/sys/class/drm/card*/device"):
        try:
            if open(os.path.join(d, "vendor")).read().strip() == "0x1002":
                gpus[os.path.basename(os.path.realpath(d))] = open(os.path.join(d, "numa_node")).read().strip()
        except OSError:
            pass
    return {"nodes": nodes, "gpu_numa_node": gpus, "affinity": len(os.sched_getaffinity(0))}


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--one":
        sys.path.insert(0, ROOT)
        import torch
        import bench
        print(json.dumps(bench.host_entry(torch.device("cuda", 0))))
        sys.exit(0)
    print(json.dumps(numa_facts()))
    for th in (sys.argv[1:] or ["default", "4", "8", "12", "16", "24", "32"]):
        env = dict(os.environ)
        if th != "default":
            env["FJ_HOST_COPY_THREADS"] = th
        for extra in ({}, {"FJ_HOST_COPY_BIND": "0"}):
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--one"], capture_output=True, text=True, env=dict(env, **extra))
            line = [l for l in out.stdout.splitlines() if l.startswith("{")]
            d = json.loads(line[-1]) if line else {"error": out.stderr[-300:]}
            print("threads", th, extra, {k: d.get(k) for k in ("wall_incl_pcie_ms", "pinned_h2d_only_ms", "wall_over_h2d", "copy_threads", "error")}, flush=True)
