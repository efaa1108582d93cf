#!/bin/bash
# same-box A/B of the counting join kernels: FJ_OPTIONS=join_wide=0 (8192-slot cuckoo, two workgroups per CU) / 1 (16384-slot kernel
# wherever eligible) / 2 (auto) over several workloads.  usage (on the GPU box): [WLS="rep8 c5_rep8"] [MODES="0 2"] bash tools/r5_wide_ab.sh
cd "${GRAFT_REPO_ROOT:-$PWD}" || exit 1
mkdir -p gpurun_out
for wl in ${WLS:-rep8 c5_rep8 c2 c4 c3}; do
  for r in 1 2; do for m in ${MODES:-0 1}; do
    FJ_OPTIONS=join_wide=$m python bench.py --workload $wl --steps ${ST:-10} --warmup 2 --no-cpu-baseline --no-host-entry 2>/tmp/err.txt | tail -1 > /tmp/b.json
    python - <<PY
import json
try:
    d=json.load(open("/tmp/b.json")); ph=d["phases"]
    print("$wl wide=$m", d["value"], "G/s", d["ms_per_step"], "ms  build", ph.get("build_phase_ms"), "probe", ph.get("probe_phase_ms"), "join", ph.get("join_kernel_ms"), "part", d["roofline"]["avg_launch_ms"], flush=True)
except Exception as ex:
    print("$wl wide=$m FAILED", ex, open("/tmp/err.txt").read()[-600:], flush=True)
PY
  done; done
done 2>&1 | tee gpurun_out/r5_wide_ab_${TAG:-x}.txt
