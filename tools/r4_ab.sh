#!/bin/bash
# same-box A/B of two library builds (flash_hash_join_amd/lib/ab/{old,new}.so) over several workloads: WLS="c3 c4 c2 c3_mat"
# (the build under test is chosen with FJ_LIB_VARIANT - flash_hash_join_amd/_lib.py - the in-tree library is never touched)
cd "${GRAFT_REPO_ROOT:-$PWD}" || exit 1
mkdir -p gpurun_out
for wl in ${WLS:-c3 c4 c2 c3_mat}; do
  for r in 1 2; do for v in old new; do
    FJ_LIB_VARIANT=$v python bench.py --workload $wl --steps ${ST:-10} --warmup 2 --no-cpu-baseline --no-host-entry 2>/tmp/err.txt | tail -1 > /tmp/b.json
    python - <<PY
import json
try:
    d=json.load(open("/tmp/b.json")); ph=d["phases"]
    print("$wl $v", d["value"], "G/s", d["ms_per_step"], "ms  build", ph.get("build_phase_ms"), "probe", ph.get("probe_phase_ms"), "join", ph.get("join_kernel_ms"), "filter", ph.get("bloom_filter_ms"), "surv", ph.get("bloom_survivors"), "part", d["roofline"]["avg_launch_ms"], flush=True)
except Exception as ex:
    print("$wl $v FAILED", ex, open("/tmp/err.txt").read()[-600:], flush=True)
PY
  done; done
done 2>&1 | tee gpurun_out/r4_ab_${TAG:-x}.txt
