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
run lean_555_prio2_late KP_LEAN_QUEUE=1 KP_JOB_SCHEDULE=5,5,5 KP_QUEUE_PRIO=2 KP_QUEUE_LATE=1
run lean_555_prio3_late KP_LEAN_QUEUE=1 KP_JOB_SCHEDULE=5,5,5 KP_QUEUE_PRIO=3 KP_QUEUE_LATE=1
run lean_555_prio3 KP_LEAN_QUEUE=1 KP_JOB_SCHEDULE=5,5,5 KP_QUEUE_PRIO=3
run lean_654_prio3_late KP_LEAN_QUEUE=1 KP_QUEUE_PRIO=3 KP_QUEUE_LATE=1
run lean_654_prio3 KP_LEAN_QUEUE=1 KP_QUEUE_PRIO=3
run lean_366_prio3 KP_LEAN_QUEUE=1 KP_JOB_SCHEDULE=3,6,6 KP_QUEUE_PRIO=3
run lean_366_prio3_late KP_LEAN_QUEUE=1 KP_JOB_SCHEDULE=3,6,6 KP_QUEUE_PRIO=3 KP_QUEUE_LATE=1
run lean_555_prio3_late_heavy0 KP_LEAN_QUEUE=1 KP_JOB_SCHEDULE=5,5,5 KP_QUEUE_PRIO=3 KP_QUEUE_LATE=1 KP_QUEUE_HEAVY=0
run lean_78_prio3_late KP_LEAN_QUEUE=1 KP_JOB_SCHEDULE=7,8 KP_QUEUE_PRIO=3 KP_QUEUE_LATE=1
run lean_5x3_prio3_late KP_LEAN_QUEUE=1 KP_JOB_SCHEDULE=3,3,3,3,3 KP_QUEUE_PRIO=3 KP_QUEUE_LATE=1
run full_prio3 KP_LEAN_QUEUE=0 KP_QUEUE_PRIO=3
run lean_555_prio3_late_again KP_LEAN_QUEUE=1 KP_JOB_SCHEDULE=5,5,5 KP_QUEUE_PRIO=3 KP_QUEUE_LATE=1
} 2>&1 | tee $O/lean_schedule_knobs4.log

runo() {
  L=$1; shift
  env "$@" timeout -s KILL 300 python bench.py --workload objects --no-secondary --no-cpu-baseline --no-parity-live --repeats 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); b=d['launch_balance']; q=d['queue']
print('%-28s value %.0f ms_per_step %.3f launch_ms %.3f [%.3f %.3f] longest_env_ms %.3f median_env_ms %.3f sum/slots %.3f slots %d kept %d' % ('$L', d['value'], d['ms_per_step'], d['roofline']['launch_ms'], d['roofline']['launch_ms_min'], d['roofline']['launch_ms_max'], b['longest_env_ms'], b['median_env_ms'], b['sum_env_cycles_over_slots_ms'], b['slots'], q['kept_by_their_wave']))"
}
{
runo objects
runo objects_prio2 KP_QUEUE_PRIO=2
runo objects_prio3 KP_QUEUE_PRIO=3
runo objects_prio3_late KP_QUEUE_PRIO=3 KP_QUEUE_LATE=1
runo objects_prio3_555 KP_QUEUE_PRIO=3 KP_JOB_SCHEDULE=5,5,5
runo objects_prio3_3x5 KP_QUEUE_PRIO=3 KP_JOB_SCHEDULE=3,3,3,3,3
runo objects_again
} 2>&1 | tee $O/objects_schedule_knobs.log
