#!/bin/bash
# round 6, calls 2+: the lean queue layout (stage given by the tree) -- its new tests first, then every GPU test, then the A/B on the metric's workload
set -u
export TMPDIR=/tmp
O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_round6.py -x -q -k "lean or overflow" > $O/g2_round6.log 2>&1; echo "rc $?" >> $O/g2_round6.log; tail -15 $O/g2_round6.log
timeout 900 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_round6.py::test_whole_episodes_match_the_cpu_episode_loop > $O/g2_pytest.log 2>&1; echo "pytest rc $?" >> $O/g2_pytest.log; tail -5 $O/g2_pytest.log
for r in 1 2 3; do for v in 0 1; do
  KP_LEAN_QUEUE=$v timeout -s KILL 300 python bench.py --workload tracked --no-secondary --no-cpu-baseline --no-parity-live --repeats 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('tracked KP_LEAN_QUEUE=$v value %.0f ms_per_step %.3f [%.3f %.3f] launch_ms %.3f contacts %.2f newton %.2f bad %d' % (d['value'], d['ms_per_step'], d['ms_per_step_min'], d['ms_per_step_max'], d['roofline']['launch_ms'], d['contacts_mean'], d['newton_iters_per_substep'], d['bad_envs']))"
done; done 2>&1 | tee $O/lean_queue_ab_stageD.log
