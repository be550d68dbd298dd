#!/bin/bash
# round 6: why 12 envs per CU buys the contact kernel 6.5 % where it bought the free-fall kernel 19 % -- schedule knobs and env counts on the lean layout
set -u
export TMPDIR=/tmp
O=gpurun_out/r06; mkdir -p $O
run() {  # label, env assignments...
  L=$1; shift
  env "$@" timeout -s KILL 300 python bench.py --workload tracked --no-secondary --no-cpu-baseline --no-parity-live --repeats 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); b=d['launch_balance']; q=d['queue']
print('%-28s value %.0f ms_per_step %.3f launch_ms %.3f [%.3f %.3f] longest_env_ms %.3f median_env_ms %.3f sum/slots %.3f slots %d kept %d overflow %d contacts_max %d' % ('$L', d['value'], d['ms_per_step'], d['roofline']['launch_ms'], d['roofline']['launch_ms_min'], d['roofline']['launch_ms_max'], b['longest_env_ms'], b['median_env_ms'], b['sum_env_cycles_over_slots_ms'], b['slots'], q['kept_by_their_wave'], q['lean_overflow_jobs'], d['contacts_max_in_a_substep']))"
}
{
run full KP_LEAN_QUEUE=0
run lean KP_LEAN_QUEUE=1
run lean_prio KP_LEAN_QUEUE=1 KP_QUEUE_PRIO=1
run lean_spj2 KP_LEAN_QUEUE=1 KP_SUBSTEPS_PER_JOB=2
run lean_spj3 KP_LEAN_QUEUE=1 KP_SUBSTEPS_PER_JOB=3
run lean_spj5 KP_LEAN_QUEUE=1 KP_SUBSTEPS_PER_JOB=5
run lean_heavy0 KP_LEAN_QUEUE=1 KP_QUEUE_HEAVY=0
run lean_heavy120 KP_LEAN_QUEUE=1 KP_QUEUE_HEAVY=120
run lean_heavy250 KP_LEAN_QUEUE=1 KP_QUEUE_HEAVY=250
run lean_lpt KP_LEAN_QUEUE=1 KP_LPT_ORDER=1
run lean_slots2816 KP_LEAN_QUEUE=1 KP_QUEUE_SLOTS=2816
run lean_slots2560 KP_LEAN_QUEUE=1 KP_QUEUE_SLOTS=2560
run lean_slots2048 KP_LEAN_QUEUE=1 KP_QUEUE_SLOTS=2048
run full_again KP_LEAN_QUEUE=0
run lean_again KP_LEAN_QUEUE=1
} 2>&1 | tee $O/lean_schedule_knobs.log
for v in 0 1; do KP_LEAN_QUEUE=$v timeout -s KILL 600 python tools/envs_sweep.py tracked,random_init 4096,6144,8192,12288 2>/dev/null | sed "s/^/lean=$v /"; done 2>&1 | tee $O/lean_envs_sweep.log
