#!/bin/bash
# A/B of the library at HEAD (tools/micro/bin/libkinpoly_sim_base.so) against the working tree's: bit-identity of end states, then launch times
export TMPDIR=/tmp; mkdir -p gpurun_out/r05; O=gpurun_out/r05/schur_prefetch_ab.log; : > $O
B=tools/micro/bin/libkinpoly_sim_base.so
for wl in objects; do
  KP_SIM_LIBRARY=$B timeout -s KILL 300 python tools/micro/lib_ab_state.py $wl /tmp/a_$wl.npz 12 2>&1 | tail -1 >> $O
  timeout -s KILL 300 python tools/micro/lib_ab_state.py $wl /tmp/b_$wl.npz 12 2>&1 | tail -1 >> $O
  python tools/micro/lib_ab_state.py cmp /tmp/a_$wl.npz /tmp/b_$wl.npz >> $O
done
for r in 1 2 3; do for lib in $B ""; do
  KP_SIM_LIBRARY=$lib timeout -s KILL 300 python bench.py --workload objects --steps 60 --warmup 20 --no-secondary --no-cpu-baseline --no-parity-live 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('objects lib=${lib:-tree} value %.0f ms_per_step %.3f launch_ms %.4f newton %.3f' % (d['value'], d['ms_per_step'], d['roofline']['launch_ms'], d['newton_iters_per_substep']))" >> $O
done; done
cat $O
