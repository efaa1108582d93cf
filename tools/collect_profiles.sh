#!/bin/bash
# One GPU call that produces what is committed under profiles/ for a kernel version:  tools/collect_profiles.sh <tag>
#   PMC passes (c3, c4, c3_mat, c2), rocprofv3 --kernel-trace --stats (c3, c4, c3_mat, c2 and the literal HBM-table forms),
#   the default bench line, the other workloads' bench lines, the J1 harness log, traffic_latest.json, the one-rank multi-GPU step
#   (tools/r4_dist_one_gpu.sh <tag>d), tools/pack_probe.py, the precheck (tools/r4_prefilter_one_gpu.sh <tag>p, tools/precheck_probe.py).  Then: tools/publish_profiles.sh <tag>.
TAG=${1:-r03}
cd ${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/$TAG; mkdir -p $O
for w in c3 c4 c3_mat c2 c2_hbm_table; do
  ./tools/pmc.sh $O/pmc_$w --workload $w --no-host-entry > $O/${w}_pmc_summary.txt 2>&1
done
for w in c3 c4 c3_mat c2 c2_hbm_table c4_hbm_table_bloom; do
  ./tools/prof_stats.sh ${TAG}_$w --workload $w --steps 20 --warmup 3 --no-host-entry > $O/${w}_kernel_stats.txt 2>&1
  cp $(find gpurun_out/stats_${TAG}_$w -name "*kernel_stats.csv" | head -1) $O/${w}_kernel_stats.csv
  grep "^{\"metric" gpurun_out/stats_${TAG}_$w.log | tail -1 > $O/${w}_bench_under_rocprof.json
done
python tools/trace_timeline.py gpurun_out/stats_${TAG}_c3 > $O/c3_timeline.txt 2>&1
python tools/trace_timeline.py gpurun_out/stats_${TAG}_c4 > $O/c4_timeline.txt 2>&1
python tools/trace_timeline.py gpurun_out/stats_${TAG}_c3_mat > $O/c3_mat_timeline.txt 2>&1
python tools/traffic_json.py $O/c3_pmc_summary.txt $O/traffic_latest.json > /dev/null 2>&1
timeout 900 python bench.py 2>&1 | tail -1 > $O/c3_bench.json
: > $O/other_workloads.jsonl
for w in c2 c2_hbm_table c4 c4_scalar_bloom c4_adaptive c4_hbm_table_bloom c3_adaptive c3_mat small rep8 c5_rep8 c5; do
  timeout 300 python bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline --no-host-entry 2>&1 | tail -1 >> $O/other_workloads.jsonl
done
FJ_OPTIONS=mat_single_pass=0 timeout 300 python bench.py --workload c3_mat --steps 10 --warmup 2 --no-cpu-baseline --no-host-entry 2>&1 | tail -1 > $O/c3_mat_two_pass_bench.json
timeout 600 python tools/skew_build_partition_probe.py 4 > $O/skew_build_partition_probe.txt 2>&1
timeout 900 python tools/benchmark_j1.py --sizes 1e7,4e7 --cpu --duckdb > $O/j1_shaped_benchmark.log 2>&1
# the multi-GPU step of config 5 on one rank (C++ driver) and the sender side in isolation
bash tools/r4_dist_one_gpu.sh ${TAG}d > $O/dist_one_gpu.txt 2>&1
timeout 300 python tools/pack_probe.py > $O/pack_probe.txt 2>&1
# the sender-side precheck in chunk form: one-rank steps off / on / auto, and the kernel at the 8-rank plan
bash tools/r4_prefilter_one_gpu.sh ${TAG}p > $O/prefilter_one_rank.txt 2>&1
tools/prof_py.sh ${TAG}_precheck_probe tools/precheck_probe.py > $O/precheck_probe_kernel_stats.txt 2>&1
grep "the precheck:\|^filters of" gpurun_out/stats_${TAG}_precheck_probe.log > $O/precheck_probe.txt
# round 5: the build-broadcast form as one rank of 8 sees it (8-rank plan: 262144 partitions; all peers' regions resident), with and without
# the CU reserve of an N > 1 step; its kernel statistics; the step through the driver on one rank; the 16384-slot join kernel's counters;
# the model table
( python tools/bcast_one_gpu.py 8 125000000 1250000000 4 5 5000 0 | tail -7; echo "-- with the 32-CU reserve of an N > 1 step over RCCL:"; python tools/bcast_one_gpu.py 8 125000000 1250000000 4 5 5000 32 | tail -5 ) > $O/bcast_one_rank.txt 2>&1
tools/prof_py.sh ${TAG}_bcast tools/bcast_one_gpu.py 8 125000000 1250000000 4 5 > $O/bcast_one_rank_kernel_stats.txt 2>&1
cp $(find gpurun_out/stats_${TAG}_bcast -name "*kernel_stats.csv" | head -1) $O/bcast_one_rank_kernel_stats.csv
FJ_BENCH_FORCE_DIST=1 FJ_DIST_STRATEGY=broadcast timeout 600 python bench.py --workload c5 --steps 5 --warmup 2 --no-host-entry --no-cpu-baseline 2>&1 | tail -1 > $O/c5_one_rank_broadcast_bench.json
FJ_OPTIONS=join_wide=1 ./tools/pmc.sh $O/pmc_wide --workload c5_rep8 --no-host-entry > $O/c5_rep8_wide_pmc_summary.txt 2>&1
FJ_OPTIONS=join_wide=0 timeout 300 python bench.py --workload c5_rep8 --steps 10 --warmup 2 --no-cpu-baseline --no-host-entry 2>&1 | tail -1 > $O/c5_rep8_narrow_table_bench.json
python tools/scale_model.py > $O/scale_model.txt 2>&1
# ... and as one rank of 2 and of 4 (16- and 17-bit plans: three and two items per partition, dealt in runs)
( for w in 2 4; do python tools/bcast_one_gpu.py $w 125000000 1250000000 4 5 5000 0 2>&1 | grep "^world\|^step"; done ) > $O/bcast_one_rank_of_2_and_4.txt 2>&1
# round 6: the bucketed wide join by shape (join_wide=0: the narrow-table kernel); the materialising build-broadcast step on one rank
# beside the same join in the chunk-form shuffle; the measured CU reserve
TAG=by_shape VARS="-" BC=0 WLS="c3 c4 rep8 c5_rep8" MODES="0 1" bash tools/r6_wide_ab.sh > /dev/null 2>&1; cp gpurun_out/r6_wide_ab_by_shape.txt $O/wide_by_shape.txt
for st in broadcast shuffle; do
  FJ_BENCH_FORCE_DIST=1 FJ_DIST_STRATEGY=$st timeout 600 python bench.py --workload c5_mat --steps 5 --warmup 2 --no-host-entry --no-cpu-baseline 2>/dev/null | tail -1 > $O/c5_mat_one_rank_${st}_bench.json
done
bash tools/r6_reserve_one_rank.sh > /dev/null 2>&1; cp gpurun_out/r06_cu_reserve_one_rank.txt $O/cu_reserve_one_rank.txt
ls -la $O | head -80
