import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(sys.argv[1] if len(sys.argv) > 1 else "", d["value"], d["ms_per_step"], {k: v for k, v in d["phases"].items() if k.endswith("_ms")})
