export TMPDIR=/tmp
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_gpu_round5.py tests/test_gpu_sampler.py -q -x 2>&1 | tail -4
( for r in 1 2 3; do for lib in kinpoly_amd/libkinpoly_sim.so tools/micro/bin/libkp_pk.so; do for wl in tracked objects; do
  KP_SIM_LIBRARY=$PWD/$lib timeout -s KILL 300 python bench.py --workload $wl --steps 60 --warmup 20 --no-secondary --no-cpu-baseline --no-parity-live 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$wl $lib value %.0f ms_per_step %.3f launch_ms %.4f sum/2048 %.3f' % (d['value'], d['ms_per_step'], d['roofline']['launch_ms'], d['launch_balance']['sum_env_cycles_over_2048_slots_ms']))"
done; done; done ) 2>&1 | tee gpurun_out/r05/pk_elim_ab.log
timeout -s KILL 600 python tools/micro/occupancy_premise.py 2>&1 | grep -v amdgpu | tee gpurun_out/r05/occupancy_premise.log
for a in "" "--pool_depth 8" "--pool_depth 12"; do timeout -s KILL 300 python tools/sampler_regime.py $a 2>/dev/null | head -c 420; echo; done | tee gpurun_out/r05/sampler_regime_variants2.log
