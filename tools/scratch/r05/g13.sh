export TMPDIR=/tmp
mkdir -p gpurun_out/r05
T="timeout -s KILL"
( for thr in 0 20 100 500; do for wl in tracked wild_eval random_init objects; do KP_WARM_EXTRAP=0.75 KP_WARM_EXTRAP_MIN=$thr $T 300 python bench.py --workload $wl --steps 60 --warmup 20 --no-secondary --no-cpu-baseline --no-parity-live 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$wl beta=0.75 min=$thr value %.0f launch_ms %.4f newton/substep %.3f fact/substep %.3f' % (d['value'], d['roofline']['launch_ms'], d['newton_iters_per_substep'], d['hessian_factorisations_per_substep']))"
done; done
for wl in tracked wild_eval; do KP_WARM_EXTRAP=0 $T 300 python bench.py --workload $wl --steps 60 --warmup 20 --no-secondary --no-cpu-baseline --no-parity-live 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$wl beta=0 value %.0f launch_ms %.4f newton/substep %.3f fact/substep %.3f' % (d['value'], d['roofline']['launch_ms'], d['newton_iters_per_substep'], d['hessian_factorisations_per_substep']))"
done ) 2>&1 | tee gpurun_out/r05/warm_extrap_threshold.log
