#!/bin/bash
# same-box A/B of csrc/fj_join_wide.hip variants (tools/mk_wide_variant.sh): the build-broadcast step of one rank of 8 and the
# fat-build-side bench workloads.  usage (on the GPU box): VARS="c0r0 c2r1" bash tools/r5_claim_ab.sh   ("-" = the in-tree library)
cd "${GRAFT_REPO_ROOT:-$PWD}" || exit 1
mkdir -p gpurun_out
for r in 1 2; do for v in ${VARS:--}; do
  lv=$v; [ "$v" = "-" ] && lv=""
  echo "== variant $v"
  FJ_LIB_VARIANT=$lv python tools/bcast_one_gpu.py 8 125000000 1250000000 4 3 5000 0 2>&1 | grep "^step" | tail -2
  for wl in ${WLS:-c5_rep8 rep8}; do
    m=1; [ $wl = c5_rep8 ] && m=2
    FJ_LIB_VARIANT=$lv FJ_OPTIONS=join_wide=$m python bench.py --workload $wl --steps 8 --warmup 2 --no-cpu-baseline --no-host-entry 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); ph=d['phases']; print('$wl wide=$m', d['ms_per_step'],'ms join',ph.get('join_kernel_ms'))"
  done
done; done
