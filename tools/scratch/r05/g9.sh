export TMPDIR=/tmp
mkdir -p gpurun_out/r05
( for r in 1 2; do for lib in kinpoly_amd/libkinpoly_sim.so tools/micro/bin/libkp_extrap_1.0_min0.so tools/micro/bin/libkp_extrap_1.0_min3.so tools/micro/bin/libkp_extrap_0.5_min3.so; do for wl in tracked objects random_init; do
  KP_SIM_LIBRARY=$PWD/$lib timeout -s KILL 300 python bench.py --workload $wl --steps 60 --warmup 20 --no-secondary --no-cpu-baseline --no-parity-live 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$wl $lib value %.0f launch_ms %.4f longest %.3f sum/2048 %.3f newton/substep %.3f fact/substep %.3f cap %d bad %d' % (d['value'], d['roofline']['launch_ms'], d['launch_balance']['longest_env_ms'], d['launch_balance']['sum_env_cycles_over_2048_slots_ms'], d['newton_iters_per_substep'], d['hessian_factorisations_per_substep'], d['newton_cap_hits'], d['bad_envs']))"
done; done; done ) 2>&1 | tee gpurun_out/r05/warm_extrap_ab2.log
