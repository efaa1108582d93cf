#!/bin/bash
# the multi-GPU step of config 5 driven on ONE rank (FJ_BENCH_FORCE_DIST=1): kernel statistics of the chunk-form shuffle next to
# the owner-scatter form;  usage: tools/r3_dist_one_gpu.sh <tag>
cd ${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-r03}
O=gpurun_out/$TAG; mkdir -p $O
for form in 1 0; do
  FJ_BENCH_FORCE_DIST=1 FJ_DIST_CHUNK_SHUFFLE=$form tools/prof_stats.sh ${TAG}_c5_dist_form$form --workload c5 --steps 5 --warmup 2 --no-host-entry > $O/c5_dist_form${form}_kernel_stats.txt 2>&1
  cp $(find gpurun_out/stats_${TAG}_c5_dist_form$form -name "*kernel_stats.csv" | head -1) $O/c5_dist_form${form}_kernel_stats.csv
  grep "^{\"metric\|^{\"error" gpurun_out/stats_${TAG}_c5_dist_form$form.log | tail -1 > $O/c5_dist_form${form}_bench_under_rocprof.json
  tail -3 gpurun_out/stats_${TAG}_c5_dist_form$form.log | cut -c1-600
  cat $O/c5_dist_form${form}_kernel_stats.txt | cut -c1-200
done
for form in 1 0; do
  FJ_BENCH_FORCE_DIST=1 FJ_DIST_CHUNK_SHUFFLE=$form timeout 600 python bench.py --workload c5 --steps 5 --warmup 2 --no-host-entry --no-cpu-baseline 2>&1 | tail -1 > $O/c5_dist_form${form}_bench.json
  python tools/show_bench.py < $O/c5_dist_form${form}_bench.json
done
