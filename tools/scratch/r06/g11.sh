#!/bin/bash
# round 6: the lean launch at 4096 envs ends on late-started / heavy envs (sum / slots 1.83 ms, launch 2.44 ms): job schedules and issue priorities
set -u
export TMPDIR=/tmp
O=gpurun_out/r06; mkdir -p $O
run() {
  L=$1; shift
  env "$@" timeout -s KILL 300 python bench.py --workload tracked --no-secondary --no-cpu-baseline --no-parity-live --repeats 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); b=d['launch_balance']; q=d['queue']
print('%-28s value %.0f ms_per_step %.3f launch_ms %.3f [%.3f %.3f] longest_env_ms %.3f median_env_ms %.3f sum/slots %.3f slots %d kept %d' % ('$L', d['value'], d['ms_per_step'], d['roofline']['launch_ms'], d['roofline']['launch_ms_min'], d['roofline']['launch_ms_max'], b['longest_env_ms'], b['median_env_ms'], b['sum_env_cycles_over_slots_ms'], b['slots'], q['kept_by_their_wave']))"
}
{
run default
run late2 KP_QUEUE_LATE=2
run late2_456 KP_QUEUE_LATE=2 KP_JOB_SCHEDULE=4,5,6
run late2_366 KP_QUEUE_LATE=2 KP_JOB_SCHEDULE=3,6,6
run late2_3444 KP_QUEUE_LATE=2 KP_JOB_SCHEDULE=3,4,4,4
run late2_654 KP_QUEUE_LATE=2 KP_JOB_SCHEDULE=6,5,4
run late2_447 KP_QUEUE_LATE=2 KP_JOB_SCHEDULE=4,4,7
run late2_heavy0 KP_QUEUE_LATE=2 KP_QUEUE_HEAVY=0
run late2_heavy130 KP_QUEUE_LATE=2 KP_QUEUE_HEAVY=130
run default_again
run late2_again KP_QUEUE_LATE=2
} 2>&1 | tee $O/lean_schedule_knobs5.log
