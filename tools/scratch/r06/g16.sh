#!/bin/bash
# round 6: overflow kernel grid sized from the last overflow count seen (empty launch 34 us -> ?), then the tests that exercise the hand-over
set -u
export TMPDIR=/tmp
O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_round6.py -x -q > $O/g16_round6.log 2>&1; echo "rc $?" >> $O/g16_round6.log; tail -4 $O/g16_round6.log
for r in 1 2 3; do timeout -s KILL 300 python bench.py --workload tracked --no-secondary --no-cpu-baseline --no-parity-live --repeats 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('tracked value %.0f ms_per_step %.3f [%.3f %.3f] launch_ms %.3f' % (d['value'], d['ms_per_step'], d['ms_per_step_min'], d['ms_per_step_max'], d['roofline']['launch_ms']))"; done 2>&1 | tee $O/overflow_grid_ab.log
timeout 600 python - <<'PY' 2>&1 | grep -v amdgpu | tee -a $O/overflow_grid_ab.log
import json, os, sys
sys.path.insert(0, os.getcwd())
import bench
r = bench.train_iteration(0, 4, bench.TRAIN_HORIZON, 2, 1)
print(json.dumps({"train_iteration_random_init": {k: r[k] for k in ("T_sample", "T_update", "fail_rate")}}), flush=True)
PY
