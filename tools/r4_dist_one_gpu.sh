#!/bin/bash
# the multi-GPU step of config 5 driven on ONE rank (FJ_BENCH_FORCE_DIST=1) through the C++ driver (csrc/fj_dist.hip): bench line
# (wall per step, wire bytes), then kernel statistics under rocprofv3;  usage: tools/r4_dist_one_gpu.sh <tag>
cd ${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-r04}
O=gpurun_out/$TAG; mkdir -p $O
export FJ_OPTIONS=lab_hooks=16 FJ_DIST_RESERVE_CUS=32 FJ_DIST_STRATEGY=shuffle      # the owner shuffle in chunk form, with the 32-CU reserve pinned as in rounds 4-5 (round 6 measures the reserve per communicator: tools/r6_reserve_one_rank.sh)
FJ_BENCH_FORCE_DIST=1 timeout 600 python bench.py --workload c5 --steps 8 --warmup 2 --no-host-entry --no-cpu-baseline 2>$O/c5_one_rank_bench.err | tail -1 > $O/c5_one_rank_bench.json
python tools/show_bench.py c5_one_rank < $O/c5_one_rank_bench.json
FJ_BENCH_FORCE_DIST=1 FJ_OPTIONS=lab_hooks=17 timeout 600 python bench.py --workload c5 --steps 5 --warmup 2 --no-host-entry --no-cpu-baseline 2>/dev/null | tail -1 > $O/c5_one_rank_loopback_bench.json
python tools/show_bench.py c5_one_rank_loopback < $O/c5_one_rank_loopback_bench.json
timeout 600 python bench.py --workload c5 --steps 8 --warmup 2 --no-host-entry --no-cpu-baseline 2>/dev/null | tail -1 > $O/c5_plain_join_bench.json
python tools/show_bench.py c5_plain_join < $O/c5_plain_join_bench.json
FJ_BENCH_FORCE_DIST=1 tools/prof_stats.sh ${TAG}_c5_one_rank --workload c5 --steps 5 --warmup 2 --no-host-entry > $O/c5_one_rank_kernel_stats.txt 2>&1
cp $(find gpurun_out/stats_${TAG}_c5_one_rank -name "*kernel_stats.csv" | head -1) $O/c5_one_rank_kernel_stats.csv
grep -v "^[EW]2026" $O/c5_one_rank_kernel_stats.txt | cut -c1-200
