export TMPDIR=/tmp
mkdir -p gpurun_out/r05
T="timeout -s KILL"
$T 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -5
( for r in 1 2 3; do for we in 0 1 1.5 0.75; do
  KP_WARM_EXTRAP=$we $T 300 python bench.py --workload objects --steps 60 --warmup 20 --no-secondary --no-cpu-baseline --no-parity-live 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('objects warm_extrap=$we value %.0f launch_ms %.4f sum/2048 %.3f newton/substep %.3f fact/substep %.3f cap %d bad %d' % (d['value'], d['roofline']['launch_ms'], d['launch_balance']['sum_env_cycles_over_2048_slots_ms'], d['newton_iters_per_substep'], d['hessian_factorisations_per_substep'], d['newton_cap_hits'], d['bad_envs']))"
done; done
for we in 0 1; do KP_WARM_EXTRAP=$we $T 300 python bench.py --workload tracked --steps 60 --warmup 20 --no-secondary --no-cpu-baseline --no-parity-live 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('tracked warm_extrap=$we value %.0f launch_ms %.4f newton/substep %.3f' % (d['value'], d['roofline']['launch_ms'], d['newton_iters_per_substep']))"
done ) 2>&1 | tee gpurun_out/r05/warm_extrap_option_ab.log
( for s in 0 1 2; do $T 200 python tools/obj_fuzz.py 64 3 $s; done ) 2>&1 | grep "scenes x" 
( for s in 0 1; do $T 200 python tools/contact_compare.py 64 $s; done ) 2>&1 | grep -v amdgpu | tail -2
$T 400 python tools/substep_parity.py bench:objects 512 4 15 2>&1 | grep -v amdgpu | head -4
