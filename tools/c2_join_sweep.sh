#!/bin/bash
# sweep of the join's item target (FJ_OPTIONS=join_items_target=) on one workload: WL=c2|small|...
for it in ${ITEMS:-256 512 768 1024 1536 2048 3072}; do
echo "WL=${WL:-c2} items_target=$it: $(FJ_OPTIONS=join_items_target=$it timeout 200 python bench.py --workload ${WL:-c2} --steps 30 --warmup 3 --no-cpu-baseline --no-host-entry 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['phases']['join_kernel_ms'], d['phases']['probe_phase_ms'])")"
done
