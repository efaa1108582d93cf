#!/bin/bash
# the sender-side precheck in chunk form (fj_dist_join(prefilter_below)) on ONE rank through the C++ driver: config-5 shards at 5 %
# hits (c5_bloom) and 50 % hits (c5), precheck off / on, wall per step; then kernel statistics.  usage: tools/r4_prefilter_one_gpu.sh <tag>
cd ${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-r04}
O=gpurun_out/$TAG; mkdir -p $O
export FJ_OPTIONS=lab_hooks=16 FJ_DIST_RESERVE_CUS=32 FJ_DIST_STRATEGY=shuffle FJ_BENCH_FORCE_DIST=1
for w in c5_bloom c5; do for pf in 0 1 auto; do
  FJ_DIST_PREFILTER=$pf timeout 600 python bench.py --workload $w --steps 8 --warmup 2 --no-host-entry --no-cpu-baseline 2>$O/pf_${w}_$pf.err | tail -1 > $O/pf_${w}_$pf.json
  python - $O/pf_${w}_$pf.json $w $pf <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); p = d.get("phases", {})
    print(f"{sys.argv[2]:9s} precheck={sys.argv[3]:4s} {d['ms_per_step']:.2f} ms/step  prefilter={p.get('shuffle_prefilter')} sampled={p.get('shuffle_prefilter_sampled_survivors')} rows_sent={p.get('probe_rows_sent_rank0')} wire_bytes={p.get('wire_bytes_sent_rank0')} form={p.get('shuffle_form')}")
except Exception as e:
    print(sys.argv[1], "unreadable:", e)
PY
done; done
FJ_DIST_PREFILTER=1 tools/prof_stats.sh ${TAG}_pf_c5_bloom --workload c5_bloom --steps 5 --warmup 2 --no-host-entry > $O/pf_c5_bloom_kernel_stats.txt 2>&1
cp $(find gpurun_out/stats_${TAG}_pf_c5_bloom -name "*kernel_stats.csv" | head -1) $O/pf_c5_bloom_kernel_stats.csv
FJ_DIST_PREFILTER=1 tools/prof_stats.sh ${TAG}_pf_c5 --workload c5 --steps 5 --warmup 2 --no-host-entry > $O/pf_c5_kernel_stats.txt 2>&1
cp $(find gpurun_out/stats_${TAG}_pf_c5 -name "*kernel_stats.csv" | head -1) $O/pf_c5_kernel_stats.csv
grep -v "^[EW]2026" $O/pf_c5_bloom_kernel_stats.txt | cut -c1-200 | head -16
grep -v "^[EW]2026" $O/pf_c5_kernel_stats.txt | cut -c1-200 | head -16
