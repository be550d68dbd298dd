#!/bin/bash
# round 6: the lean queue with its schedule defaults -- every GPU test, the env-count sweep against the full layout, the default bench line
set -u
export TMPDIR=/tmp
O=gpurun_out/r06; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/g9_pytest.log 2>&1; echo "pytest rc $?" >> $O/g9_pytest.log; tail -5 $O/g9_pytest.log
for v in 0 1; do KP_LEAN_QUEUE=$v timeout -s KILL 600 python tools/envs_sweep.py tracked,random_init,wild_eval 3072,4096,6144,8192,12288 2>/dev/null | sed "s/^/lean=$v /"; done 2>&1 | tee $O/lean_envs_sweep_defaults.log
( time timeout -s KILL 900 python bench.py > $O/bench_default_stage_lean.json 2> $O/bench_default.err ) 2> $O/bench_default.time; cut -c1-1500 $O/bench_default_stage_lean.json; cat $O/bench_default.time
