"""Print the interesting fields of a bench.py JSON line: `python bench.py ... | python tools/show_bench.py [label]`, or
`python tools/show_bench.py label file.json`."""
import json, sys
src = open(sys.argv[2]).read() if len(sys.argv) > 2 else sys.stdin.read()
lines = [l for l in src.strip().splitlines() if l.startswith("{")]
if not lines:
    print(sys.argv[1] if len(sys.argv) > 1 else "", "no JSON line:", src[-400:])
    sys.exit(0)
d = json.loads(lines[-1])
ph = d["phases"]
print(sys.argv[1] if len(sys.argv) > 1 else "", d["value"], "Gprobes/s", d["ms_per_step"], "ms/step",
      {k: v for k, v in ph.items() if k.endswith("_ms") or k.startswith("bloom")}, "part_launch_ms", d["roofline"]["avg_launch_ms"])
