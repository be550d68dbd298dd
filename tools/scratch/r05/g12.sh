export TMPDIR=/tmp
mkdir -p gpurun_out/r05
T="timeout -s KILL"
$T 1200 python -m pytest tests -m gpu -q 2>&1 | grep -v "Warn\|warn\|sched_\|^$" | tail -40
$T 300 python tools/floor_fuzz.py 320 2>&1 | tail -2
( for wl in tracked objects random_init wild_eval; do $T 300 python bench.py --workload $wl --steps 60 --warmup 20 --no-secondary --no-cpu-baseline --no-parity-live 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$wl default value %.0f launch_ms %.4f newton/substep %.3f' % (d['value'], d['roofline']['launch_ms'], d['newton_iters_per_substep']))"
done ) 2>&1 | tee gpurun_out/r05/warm_extrap_defaults.log
