#!/bin/bash
# round 3: config 4 by filter variant (one box);  usage: tools/r3_c4_ab.sh <outdir under gpurun_out>
cd ${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/${1:-r3a}; mkdir -p $O
B="python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-host-entry"
: > $O/c4_ab.jsonl
for v in ${VARIANTS:-2 0 1}; do
  echo "variant=$v" >> $O/c4_ab.jsonl
  FJ_OPTIONS=bloom_variant=$v timeout 300 $B --workload c4 2>&1 | tail -1 >> $O/c4_ab.jsonl
done
timeout 300 $B --workload c4_adaptive 2>&1 | tail -1 > $O/c4_adaptive.json
timeout 300 $B --workload c3 2>&1 | tail -1 > $O/c3.json
python - $O <<'PY'
import json, sys
o = sys.argv[1]
for line in open(o + "/c4_ab.jsonl"):
    line = line.strip()
    if not line.startswith("{"):
        print(line); continue
    d = json.loads(line)
    print("  ms/step %.3f  build %.3f probe %.3f join %.3f filter %s surv %s" % (d["ms_per_step"], d["phases"]["build_phase_ms"], d["phases"]["probe_phase_ms"], d["phases"]["join_kernel_ms"], d["phases"]["bloom_filter_kernel_ms"], d["phases"]["bloom_survivors"]))
    for r in d.get("roofline_kernels", []):
        print("     %-100s %8.3f ms x%.1f  frac %.3f" % (r["kernel"][:100], r["avg_launch_ms"], r["launches_per_step"], r["frac"]))
for f in ("c4_adaptive", "c3"):
    try:
        d = json.loads(open(o + "/" + f + ".json").read())
        print(f, "ms/step %.3f" % d["ms_per_step"], "value", d["value"], "roofline", d["roofline"]["kernel"][:50], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"])
    except Exception as e:
        print(f, "failed", e)
PY
