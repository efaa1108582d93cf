import csv, glob, collections, sys
out = sys.argv[1]
for p in ("p1", "p2", "p3", "p4"):
    f = glob.glob(f"{out}/{p}/*/*counter_collection.csv")
    if not f:
        print(p, "missing"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        name = r["Kernel_Name"]
        if "fj_" in name and "gen_" not in name:
            key = name.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0]
            agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in sorted(agg):
        for c, v in sorted(agg[k].items()):
            print("%-4s %-50s %-24s launches=%d mean_per_launch=%.6g" % (p, k, c, len(v), sum(v) / len(v)))
