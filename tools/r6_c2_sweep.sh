#!/bin/bash
# round 6: c2 (1M x 100M, one pass of 256 partitions) - the join's slice count (join_items_target) x one workgroup per item /
# resident workgroups (persistent_min_items), and the same for the 1M x 10M case.  usage (GPU box): bash tools/r6_c2_sweep.sh
cd "${GRAFT_REPO_ROOT:-$PWD}" || exit 1
mkdir -p gpurun_out
{
for wl in ${WLS:-c2 small}; do
  for tgt in ${TGTS:-1024 2048 4096 8192 16384}; do
    for pm in ${PMS:-0 1000000000}; do
      FJ_OPTIONS=join_items_target=$tgt,persistent_min_items=$pm python bench.py --workload $wl --steps ${ST:-30} --warmup 3 --no-cpu-baseline --no-host-entry 2>/tmp/err.txt | tail -1 > /tmp/b.json
      python - <<PY
import json
try:
    d=json.load(open("/tmp/b.json")); ph=d["phases"]
    print("$wl items_target $tgt persistent_min $pm:", d["ms_per_step"], "ms  build", ph.get("build_phase_ms"), "probe", ph.get("probe_phase_ms"), "join", ph.get("join_kernel_ms"), "pass", d["roofline"]["avg_launch_ms"], flush=True)
except Exception as ex:
    print("$wl $tgt $pm FAILED", ex, open("/tmp/err.txt").read()[-400:], flush=True)
PY
    done
  done
done
} 2>&1 | tee gpurun_out/r06_c2_sweep.txt
