"""Per-kernel instruction mix of a hipcc -save-temps .s file: python tools/isa_stats.py file.s [name-substring]
(vector / scalar / LDS / vector-memory instruction counts, waits, scratch use; static counts, not executed ones)."""
import re, sys
src = open(sys.argv[1]).read().split("\n")
pat = sys.argv[2] if len(sys.argv) > 2 else ""
name, body, out = None, [], {}
for ln in src:
    m = re.match(r"^(_Z\w+):", ln)
    if m:
        name, body = m.group(1), []
        continue
    if name is not None:
        body.append(ln)
        if "s_endpgm" in ln:
            out[name] = body
            name = None
for k, b in out.items():
    if pat not in k:
        continue
    ins = [x.strip().split()[0] for x in b if x.startswith("\t") and not x.strip().startswith((".", ";"))]
    def c(p): return sum(1 for i in ins if re.match(p, i))
    print(k)
    print(f"  total {len(ins)}  v_ {c(r'v_')}  s_ {c(r's_')}  ds_ {c(r'ds_')} (b128 {c(r'ds_read_b128')}, b64 {c(r'ds_read_b64')}, add_rtn {c(r'ds_add_rtn')}, write_b64 {c(r'ds_write_b64')})"
          f"  global_ {c(r'global_')}  flat_ {c(r'flat_')}  scratch_ {c(r'scratch_')}  waitcnt {c(r's_waitcnt')}  barrier {c(r's_barrier')}  cbranch {c(r's_cbranch')}")
