#!/bin/bash
# round 6: adaptive lean / full fallback -- its tests, the random-init training iteration that exposed the overflow cost, objects episode parity
set -u
export TMPDIR=/tmp
O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_round6.py -x -q > $O/g13_round6.log 2>&1; echo "rc $?" >> $O/g13_round6.log; tail -6 $O/g13_round6.log
timeout 600 python - <<'PY' 2>&1 | grep -v amdgpu | tee $O/train_iter_lean_adaptive.log
import json, os, sys
sys.path.insert(0, os.getcwd())
import bench
for lean, adaptive in ((1, 1), (1, 0), (0, 1)):
    os.environ["KP_LEAN_QUEUE"] = str(lean)
    import kinpoly_amd.sim as kpsim
    r = bench.train_iteration(0, 4, bench.TRAIN_HORIZON, 2, 1, model_opts={"lean_queue": lean, "lean_adaptive": adaptive})
    print(json.dumps({"lean_queue": lean, "lean_adaptive": adaptive, **{k: r[k] for k in ("T_sample", "T_update", "samples_per_s_per_gpu", "fail_rate")}}), flush=True)
PY
timeout 900 python tools/episode_parity.py --envs 32 --steps 99 --objects --json $O/episode_parity_objects.json > $O/episode_parity_objects.log 2>&1; tail -30 $O/episode_parity_objects.log | head -40
