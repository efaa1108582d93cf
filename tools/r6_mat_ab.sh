cd "${GRAFT_REPO_ROOT:-$PWD}"
for rep in 1 2; do for v in prev -; do
  [ "$v" = "-" ] && unset FJ_LIB_VARIANT || export FJ_LIB_VARIANT=$v
  for wl in c3_mat c3; do
    python bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline --no-host-entry 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); ph=d['phases']; print('$wl variant $v:', d['ms_per_step'], 'ms  join', ph.get('join_kernel_ms'), 'emit', ph.get('emit_kernel_ms'), 'pass', d['roofline']['avg_launch_ms'])"
  done
done; done
