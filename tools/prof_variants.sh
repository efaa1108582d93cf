# usage (on the GPU box): VARS="old new" [WL=c3] bash tools/prof_variants.sh -- mean kernel durations of a bench run per library
# variant in flash_hash_join_amd/lib/ab/<name>.so (rocprofv3 kernel trace, same box, one after the other, twice)
# (the variant is chosen with FJ_LIB_VARIANT, flash_hash_join_amd/_lib.py: the in-tree library is never overwritten)
cd "${GRAFT_REPO_ROOT:-$PWD}" || exit 1
for rep in 1 2; do for v in $VARS; do
  FJ_LIB_VARIANT=$v bash tools/prof_stats.sh pv_${v}_$rep --workload ${WL:-c3} --steps 12 --warmup 2 > /dev/null 2>&1
  python3 - $v gpurun_out/stats_pv_${v}_$rep <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[2] + "/**/*kernel_trace.csv", recursive=True)[0]
acc = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    if "gen_" in n or "rocclr" in n: continue
    g = r.get("Grid_Size_X", "")
    acc.setdefault((n[:58], g), []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("==", sys.argv[1])
for (n, g), v in acc.items():
    if len(v) >= 10: v = sorted(v)[1:-1]
    if sum(v) / len(v) > 15: print("  %-60s grid %-8s n=%-4d mean %9.1f us" % (n, g, len(v), sum(v) / len(v)))
PY
done; done
rm -rf gpurun_out/stats_pv_*
