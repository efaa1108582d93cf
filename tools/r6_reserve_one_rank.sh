#!/bin/bash
# The measured CU reserve (csrc/fj_dist.hip: reserve_for_step) on ONE rank: the loop-back hook (lab_hooks & 1) makes the rank's own
# share travel through ncclSend / ncclRecv, so an exchange IS in flight while the passes run; lab_hooks & 16 lets the reserve apply
# on one rank.  The bench line's phases.cu_reserve shows the two measuring steps' pass times and the choice.  usage (GPU box): bash tools/r6_reserve_one_rank.sh
cd "${GRAFT_REPO_ROOT:-$PWD}" || exit 1
mkdir -p gpurun_out
{
for strat in shuffle broadcast; do
  for pin in "" 0 32; do
    echo "== strategy $strat, FJ_DIST_RESERVE_CUS=${pin:-<measured>}"
    ( [ -n "$pin" ] && export FJ_DIST_RESERVE_CUS=$pin; FJ_OPTIONS=lab_hooks=17 FJ_BENCH_FORCE_DIST=1 FJ_DIST_STRATEGY=$strat FJ_DIST_PREFILTER=0 python bench.py --workload c5 --steps 5 --warmup 4 --no-host-entry --no-cpu-baseline 2>/dev/null | tail -1 ) | python -c "
import json,sys
d=json.loads(sys.stdin.read()); ph=d['phases']
print(' ', d['ms_per_step'], 'ms per step;', d['config']['parallelism'], '; pass', d['roofline']['avg_launch_ms'], 'ms; cu_reserve', ph.get('cu_reserve'))"
  done
done
} 2>&1 | tee gpurun_out/r06_cu_reserve_one_rank.txt
