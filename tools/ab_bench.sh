#!/bin/bash
# A/B of a model option on bench workloads: tools/ab_bench.sh VAR "v1 v2 ..." "workloads" [repeats]
set -u
export TMPDIR=/tmp
VAR=$1; VALS=$2; WLS=${3:-"tracked objects"}; REP=${4:-2}
for wl in $WLS; do for r in $(seq $REP); do for v in $VALS; do
  env $VAR=$v timeout -s KILL 300 python bench.py --workload $wl --no-secondary --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$wl $VAR=$v value %.0f ms_per_step %.3f launch_ms %.3f longest_env_ms %.3f sum/slots %.3f' % (d['value'], d['ms_per_step'], d['roofline']['launch_ms'], d['launch_balance']['longest_env_ms'], d['launch_balance']['sum_env_cycles_over_slots_ms']))"
done; done; done
