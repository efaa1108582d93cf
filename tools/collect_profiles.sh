set -x
cd $GRAFT_REPO_ROOT
./tools/pmc.sh gpurun_out/pmc10 > gpurun_out/pmc10_summary.txt 2>&1
R=$PWD; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/stats10 -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $R/gpurun_out/bench10_under_rocprof.log 2>&1
cd $R
python bench.py > gpurun_out/bench10.log 2>&1
grep "^{\"metric" gpurun_out/bench10.log | tail -1
find gpurun_out/stats10 -name "*kernel_stats.csv" | head
