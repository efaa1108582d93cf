"""Timeline of the LAST join in a rocprofv3 kernel trace: python tools/trace_timeline.py <dir with *_kernel_trace.csv> [n_kernels]"""
import csv, glob, os, sys
f = max(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)      # (the newest, where several runs left traces)
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# a join starts with the memset of its plan's scalars: the last fill that directly follows a copyBuffer (the read-back that
# ended the previous join; a materialising join reads back more than once, its later stages do not start with a fill)
idx = [i for i, r in enumerate(rows) if "copyBuffer" in r["Kernel_Name"]]
starts = [i for i, r in enumerate(rows) if "fillBuffer" in r["Kernel_Name"] and (i == 0 or "copyBuffer" in rows[i - 1]["Kernel_Name"])]
start = starts[-1] if starts else 0
last = rows[start: idx[-1] + 1] if idx and idx[-1] > start else rows[-int(sys.argv[2]) if len(sys.argv) > 2 else -40:]
t0 = int(last[0]["Start_Timestamp"]); prev_end = t0; busy = 0
for r in last:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:70]
    print("%9.1f  dur %8.1f  gap %6.1f  %s  [grid %s]" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, name, r.get("Grid_Size_X", r.get("Grid_Size", ""))))
    prev_end = max(prev_end, e); busy += e - s
print("span %.1f us, sum of kernel durations %.1f us" % ((prev_end - t0) / 1e3, busy / 1e3))
