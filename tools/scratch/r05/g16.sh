export TMPDIR=/tmp
mkdir -p gpurun_out/r05
T="timeout -s KILL"
( for r in 1 2 3; do for lib in tools/micro/bin/libkp_base.so kinpoly_amd/libkinpoly_sim.so; do
  KP_SIM_LIBRARY=$PWD/$lib $T 300 python bench.py --workload objects --steps 60 --warmup 20 --no-secondary --no-cpu-baseline --no-parity-live 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('objects $lib value %.0f launch_ms %.4f sum/2048 %.3f newton/substep %.3f fact/substep %.3f' % (d['value'], d['roofline']['launch_ms'], d['launch_balance']['sum_env_cycles_over_2048_slots_ms'], d['newton_iters_per_substep'], d['hessian_factorisations_per_substep']))"
done; done ) 2>&1 | tee gpurun_out/r05/warm_extrap_objects_too_ab.log
$T 900 python -m pytest tests/test_gpu_round5.py tests/test_gpu_parity.py tests/test_gpu_round3.py tests/test_gpu_round4.py -q -x 2>&1 | grep -v "Warn\|sched_" | tail -5
( for s in 0 1 2; do $T 200 python tools/obj_fuzz.py 64 3 $s; done ) 2>&1 | grep -o "seed [0-9]).*max [0-9.e+-]*; objects\|scenes above 1e-4: [0-9]*"
