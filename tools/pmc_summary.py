"""Summarise the rocprofv3 --pmc passes collected by tools/pmc.sh: mean counter value per launch and kernel.
A kernel whose launches fall into two duration classes (the keys-only partition kernel runs over the 100M-row build
side AND the 1B-row probe side of a counting join) is reported per class: [long] = launches longer than the geometric
mean of the shortest and longest launch, [short] = the rest."""
import csv, glob, collections, math, sys
out = sys.argv[1]
for p in ("p1", "p2", "p3", "p4"):
    f = glob.glob(f"{out}/{p}/*/*counter_collection.csv")
    if not f:
        print(p, "missing"); continue
    rows = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        name = r["Kernel_Name"]
        if "fj_" in name and "gen_" not in name:
            key = name.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0]
            rows[key].append((r["Counter_Name"], float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for key, rs in rows.items():
        durs = [d for _, _, d in rs]
        lo, hi = min(durs), max(durs)
        split = math.sqrt(lo * hi) if lo > 0 and hi / lo > 3 else None
        for c, v, d in rs:
            k = key if split is None else key + (" [long]" if d > split else " [short]")
            agg[k][c].append(v)
    for k in sorted(agg):
        for c, v in sorted(agg[k].items()):
            print("%-4s %-58s %-24s launches=%d mean_per_launch=%.6g" % (p, k, c, len(v), sum(v) / len(v)))
