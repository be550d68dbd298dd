export TMPDIR=/tmp
T="timeout -s KILL"
$T 600 python -m pytest tests/test_gpu_round5.py -q -x -s -k extrapolated 2>&1 | grep -v "Warn\|warn\|amdgpu" | tail -25
( for r in 1 2; do for we in 0.5 0.6 0.75 0.9; do
  KP_WARM_EXTRAP=$we $T 300 python bench.py --workload objects --steps 60 --warmup 20 --no-secondary --no-cpu-baseline --no-parity-live 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('objects warm_extrap=$we value %.0f launch_ms %.4f sum/2048 %.3f newton/substep %.3f' % (d['value'], d['roofline']['launch_ms'], d['launch_balance']['sum_env_cycles_over_2048_slots_ms'], d['newton_iters_per_substep']))"
done; done
for we in 0.25 0.5 0.75; do for wl in tracked random_init; do KP_WARM_EXTRAP=$we $T 300 python bench.py --workload $wl --steps 60 --warmup 20 --no-secondary --no-cpu-baseline --no-parity-live 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$wl warm_extrap=$we value %.0f launch_ms %.4f newton/substep %.3f' % (d['value'], d['roofline']['launch_ms'], d['newton_iters_per_substep']))"
done; done ) 2>&1 | tee gpurun_out/r05/warm_extrap_option_ab2.log
