# (the variant is chosen with FJ_LIB_VARIANT, flash_hash_join_amd/_lib.py: the in-tree library is never overwritten)
cd "${GRAFT_REPO_ROOT:-$PWD}" || exit 1
for r in 1 2 3; do for v in old new; do FJ_LIB_VARIANT=$v python bench.py --workload ${WL:-c3} --steps ${ST:-10} --warmup 2 --no-cpu-baseline 2>&1 | tail -1 > /tmp/b.json; python - <<PY
import json
d=json.load(open("/tmp/b.json")); print("$v", d["value"], d["ms_per_step"], d["phases"]["build_phase_ms"], d["phases"]["probe_phase_ms"], d["phases"]["join_kernel_ms"], d["roofline"]["avg_launch_ms"])
PY
done; done
