#!/bin/bash
# round 6: the bucketed wide join against round 5's (lib/ab/r5.so) on one box - one rank of the 8-rank broadcast step, and the
# single-GPU shapes whose partitions are not thin.  usage (on the GPU box): [VARS="r5 -"] [WLS="rep8 c5_rep8 c4"] [BC=1] bash tools/r6_wide_ab.sh   ("-" = the in-tree library)
cd "${GRAFT_REPO_ROOT:-$PWD}" || exit 1
mkdir -p gpurun_out
{
for v in ${VARS:-r5 -}; do
  [ "$v" = "-" ] && unset FJ_LIB_VARIANT || export FJ_LIB_VARIANT=$v
  if [ "${BC:-1}" = "1" ]; then
    for world in ${WORLDS:-8}; do
      echo "== bcast_one_gpu world $world variant $v"; python tools/bcast_one_gpu.py $world 125000000 1250000000 4 ${BST:-4} 2>&1 | grep -v amdgpu.ids | tail -$((${BST:-4} + 1))
    done
  fi
  for wl in ${WLS:-rep8 c5_rep8}; do
    for m in ${MODES:-1}; do
      FJ_OPTIONS=join_wide=$m python bench.py --workload $wl --steps ${ST:-8} --warmup 2 --no-cpu-baseline --no-host-entry 2>/tmp/err.txt | tail -1 > /tmp/b.json
      python - <<PY
import json
try:
    d=json.load(open("/tmp/b.json")); ph=d["phases"]
    print("$wl variant $v wide=$m", d["value"], "G/s", d["ms_per_step"], "ms  build", ph.get("build_phase_ms"), "probe", ph.get("probe_phase_ms"), "join", ph.get("join_kernel_ms"), "part", d["roofline"]["avg_launch_ms"], flush=True)
except Exception as ex:
    print("$wl variant $v wide=$m FAILED", ex, open("/tmp/err.txt").read()[-600:], flush=True)
PY
    done
  done
done
} 2>&1 | tee gpurun_out/r6_wide_ab_${TAG:-x}.txt
