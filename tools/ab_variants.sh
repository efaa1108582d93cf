# usage (on the GPU box): VARS="old new" [WL=c3] [ST=10] bash tools/ab_variants.sh -- alternate bench runs over the library variants flash_hash_join_amd/lib/ab/<name>.so
# (the variant is chosen with FJ_LIB_VARIANT, flash_hash_join_amd/_lib.py: the in-tree library is never overwritten)
cd "${GRAFT_REPO_ROOT:-$PWD}" || exit 1
for r in 1 2 3; do for v in $VARS; do FJ_LIB_VARIANT=$v python bench.py --workload ${WL:-c3} --steps ${ST:-10} --warmup 2 --no-cpu-baseline 2>&1 | tail -1 > /tmp/b.json; python - <<PY
import json
import re
t=open("/tmp/b.json").read(); d=json.loads(t); m=re.search(r'"bloom_filter_kernel_ms": ([0-9.]+)', t)
print("$v", d["value"], d["ms_per_step"], d["phases"]["build_phase_ms"], d["phases"]["probe_phase_ms"], d["phases"]["join_kernel_ms"], d["roofline"]["avg_launch_ms"], "filter", m.group(1) if m else None)
PY
done; done
