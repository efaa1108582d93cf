set -x
cd $GRAFT_REPO_ROOT
./tools/pmc.sh gpurun_out/pmc11 > gpurun_out/pmc11_summary.txt 2>&1
R=$PWD; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/stats11 -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $R/gpurun_out/bench11_under_rocprof.log 2>&1
cd $R
python bench.py > gpurun_out/bench11.log 2>&1
grep "^{\"metric" gpurun_out/bench11.log | tail -1
find gpurun_out/stats11 -name "*kernel_stats.csv" | head
