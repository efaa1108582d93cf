#!/bin/bash
# usage: tools/prof_py.sh <tag> <script.py> [args...]  -- rocprofv3 kernel-trace statistics of any python tool -> gpurun_out/stats_<tag>/
TAG=$1; shift; S=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}; export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/stats_$TAG -- python3 $R/$S "$@" > $R/gpurun_out/stats_$TAG.log 2>&1
cd $R
f=$(find gpurun_out/stats_$TAG -name "*kernel_stats.csv" | head -1)
echo "== $TAG: $f"; python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print("%-90s calls=%-6s total_ms=%9.3f avg_us=%10.2f  %5s%%" % (r["Name"][:90], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
