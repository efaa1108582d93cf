for it in 1024 2048 3072; do for pm in 8192; do
echo "items_target=$it persistent_min=$pm: $(FJ_JOIN_ITEMS_TARGET=$it FJ_PERSISTENT_MIN_ITEMS=$pm timeout 200 python bench.py --workload c2 --steps 20 --warmup 3 --no-cpu-baseline --no-host-entry 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['phases']['join_kernel_ms'], d['phases']['probe_phase_ms'])")"
done; done
