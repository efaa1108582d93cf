"""Turn a tools_pmc.sh summary into profiles/traffic_latest.json (HBM bytes per launch of the dominant kernel).
gfx950: FETCH_SIZE counts 1/2 of wide coalesced read bytes (MI355X_MICROARCH.md, HBM section) -> x2; WRITE_SIZE exact;
both are reported in KiB and were collected in separate passes."""
import json, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from source_hash import kernel_source_hash
summary, out = sys.argv[1], sys.argv[2]
fetch, write = [], []
for line in open(summary):
    m = re.match(r"p\d\s+(\S.*?)\s+(FETCH_SIZE|WRITE_SIZE)\s+launches=(\d+)\s+mean_per_launch=(\S+)", line)
    if not m:
        continue
    k, c, n, v = m.group(1), m.group(2), int(m.group(3)), float(m.group(4))
    # keys-only passes over the PROBE side (last template argument), flat and chunk-list input
    if re.search(r"fj_partition_kernel<(512|1024), 8, 4, false, (true|false), true(, false)?(, [02])?(, false)?>", k):      # (..., PROBE_SIDE = true, [OWN = false: rounds 1-3,] RLOG, PK7 = false)
        (fetch if c == "FETCH_SIZE" else write).append(v)
f = sum(fetch) / len(fetch) * 1024 * 2
w = sum(write) / len(write) * 1024
json.dump({"fj_partition_kernel_keys_bytes_per_launch": round(f + w), "read_bytes": round(f), "write_bytes": round(w),
           "source": summary, "source_sha256": kernel_source_hash(), "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; FETCH_SIZE x2 (gfx950 half-count of wide reads); mean over the two probe-side passes"},
          open(out, "w"), indent=1)
print(open(out).read())
