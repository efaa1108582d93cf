#!/bin/bash
# same-box A/B over library variants flash_hash_join_amd/lib/ab/<name>.so (loaded through FJ_LIB_VARIANT; nothing is copied over
# the in-tree library).  usage (on the GPU box): VARS="v4 v5" [WLS="c5_rep8 rep8"] [WIDE=1] bash tools/r5_variants.sh
cd "${GRAFT_REPO_ROOT:-$PWD}" || exit 1
mkdir -p gpurun_out
for wl in ${WLS:-c5_rep8 rep8}; do
  for r in 1 2; do for v in $VARS; do
    FJ_LIB_VARIANT=$v FJ_OPTIONS=join_wide=${WIDE:-1} python bench.py --workload $wl --steps ${ST:-10} --warmup 2 --no-cpu-baseline --no-host-entry 2>/tmp/err.txt | tail -1 > /tmp/b.json
    python - <<PY
import json
try:
    d=json.load(open("/tmp/b.json")); ph=d["phases"]
    print("$wl $v", d["value"], "G/s", d["ms_per_step"], "ms  build", ph.get("build_phase_ms"), "probe", ph.get("probe_phase_ms"), "join", ph.get("join_kernel_ms"), "part", d["roofline"]["avg_launch_ms"], flush=True)
except Exception as ex:
    print("$wl $v FAILED", ex, open("/tmp/err.txt").read()[-600:], flush=True)
PY
  done; done
done 2>&1 | tee gpurun_out/r5_variants_${TAG:-x}.txt
