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
run lean_366_prio2 KP_LEAN_QUEUE=1 KP_JOB_SCHEDULE=3,6,6 KP_QUEUE_PRIO=2
run lean_366_prio2_late KP_LEAN_QUEUE=1 KP_JOB_SCHEDULE=3,6,6 KP_QUEUE_PRIO=2 KP_QUEUE_LATE=1
run lean_366_late KP_LEAN_QUEUE=1 KP_JOB_SCHEDULE=3,6,6 KP_QUEUE_LATE=1
run lean_555_prio2_late KP_LEAN_QUEUE=1 KP_JOB_SCHEDULE=5,5,5 KP_QUEUE_PRIO=2 KP_QUEUE_LATE=1
run lean_456_prio2_late KP_LEAN_QUEUE=1 KP_JOB_SCHEDULE=4,5,6 KP_QUEUE_PRIO=2 KP_QUEUE_LATE=1
run lean_654_prio2_late KP_LEAN_QUEUE=1 KP_QUEUE_PRIO=2 KP_QUEUE_LATE=1
run lean_2445_prio2_late KP_LEAN_QUEUE=1 KP_JOB_SCHEDULE=2,4,4,5 KP_QUEUE_PRIO=2 KP_QUEUE_LATE=1
run lean_366_prio2_late_lpt KP_LEAN_QUEUE=1 KP_JOB_SCHEDULE=3,6,6 KP_QUEUE_PRIO=2 KP_QUEUE_LATE=1 KP_LPT_ORDER=1
run lean_366_prio2_lpt KP_LEAN_QUEUE=1 KP_JOB_SCHEDULE=3,6,6 KP_QUEUE_PRIO=2 KP_LPT_ORDER=1
run lean_366_prio2_heavy0 KP_LEAN_QUEUE=1 KP_JOB_SCHEDULE=3,6,6 KP_QUEUE_PRIO=2 KP_QUEUE_HEAVY=0
run lean_348_prio2 KP_LEAN_QUEUE=1 KP_JOB_SCHEDULE=3,4,8 KP_QUEUE_PRIO=2
run lean_357_prio2 KP_LEAN_QUEUE=1 KP_JOB_SCHEDULE=3,5,7 KP_QUEUE_PRIO=2
run lean_2_6_7_prio2 KP_LEAN_QUEUE=1 KP_JOB_SCHEDULE=2,6,7 KP_QUEUE_PRIO=2
run full_366_prio2 KP_LEAN_QUEUE=0 KP_JOB_SCHEDULE=3,6,6 KP_QUEUE_PRIO=2
run lean_366_prio2_again KP_LEAN_QUEUE=1 KP_JOB_SCHEDULE=3,6,6 KP_QUEUE_PRIO=2
} 2>&1 | tee $O/lean_schedule_knobs3.log
