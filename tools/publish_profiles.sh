#!/bin/bash
# Copy what tools/collect_profiles.sh <tag> and tools/r3_dist_one_gpu.sh <tag>d left under gpurun_out/ (scratch) into profiles/
# (tracked):  tools/publish_profiles.sh <tag>
TAG=${1:-r03}
cd "$(dirname "$0")/.."
S=gpurun_out/$TAG
for f in c2_bench_under_rocprof.json c2_hbm_table_kernel_stats.csv c2_kernel_stats.csv c2_pmc_summary.txt c3_bench.json \
         c3_bench_under_rocprof.json c3_kernel_stats.csv c3_mat_bench_under_rocprof.json c3_mat_kernel_stats.csv c3_mat_pmc_summary.txt \
         c3_mat_timeline.txt c3_mat_two_pass_bench.json c3_pmc_summary.txt c3_timeline.txt c4_bench_under_rocprof.json \
         c4_hbm_table_bloom_kernel_stats.csv c4_kernel_stats.csv c4_pmc_summary.txt c4_timeline.txt j1_shaped_benchmark.log \
         other_workloads.jsonl skew_build_partition_probe.txt; do
  [ -s $S/$f ] && cp $S/$f profiles/${TAG}_$f || echo "missing: $S/$f"
done
[ -s $S/traffic_latest.json ] && cp $S/traffic_latest.json profiles/traffic_latest.json
D=gpurun_out/${TAG}d
if [ -d $D ]; then
  if [ -s $D/c5_one_rank_bench.json ]; then          # tools/r4_dist_one_gpu.sh: the step through the C++ driver
    for f in c5_one_rank_bench.json c5_one_rank_loopback_bench.json c5_plain_join_bench.json c5_one_rank_kernel_stats.csv; do cp $D/$f profiles/${TAG}_$f; done
  else                                                # tools/r3_dist_one_gpu.sh
    cp $D/c5_dist_form1_bench.json profiles/${TAG}_c5_one_rank_shuffle_chunks_bench.json
    cp $D/c5_dist_form1_kernel_stats.csv profiles/${TAG}_c5_one_rank_shuffle_chunks_kernel_stats.csv
    cp $D/c5_dist_form0_bench.json profiles/${TAG}_c5_one_rank_shuffle_owner_scatter_bench.json
    cp $D/c5_dist_form0_kernel_stats.csv profiles/${TAG}_c5_one_rank_shuffle_owner_scatter_kernel_stats.csv
  fi
fi
for f in pack_probe.txt prefilter_one_rank.txt precheck_probe.txt precheck_probe_kernel_stats.txt dist_one_gpu.txt bcast_one_rank.txt bcast_one_rank_kernel_stats.csv \
         c5_one_rank_broadcast_bench.json c5_rep8_wide_pmc_summary.txt c5_rep8_narrow_table_bench.json scale_model.txt bcast_one_rank_of_2_and_4.txt wide_by_shape.txt c5_mat_one_rank_broadcast_bench.json c5_mat_one_rank_shuffle_bench.json cu_reserve_one_rank.txt; do [ -s $S/$f ] && cp $S/$f profiles/${TAG}_$f; done
P=gpurun_out/${TAG}p
[ -s $P/pf_c5_bloom_kernel_stats.csv ] && cp $P/pf_c5_bloom_kernel_stats.csv profiles/${TAG}_c5_bloom_one_rank_precheck_kernel_stats.csv
[ -s $P/pf_c5_kernel_stats.csv ] && cp $P/pf_c5_kernel_stats.csv profiles/${TAG}_c5_one_rank_precheck_kernel_stats.csv
python3 tools/source_hash.py | tail -1; grep source_sha256 profiles/traffic_latest.json
