# job sizes of kp_step_queue_kernel (KP_JOB_SCHEDULE overrides the tapered default) on the bench workloads
for sch in "7,5,3" "8,7" "9,6" "6,5,4" "8,4,3" "6,4,3,2" "5,4,3,2,1" "15"; do
  for wl in tracked random_init; do
    KP_JOB_SCHEDULE=$sch timeout 200 python bench.py --workload $wl --no-cpu-baseline --no-secondary --steps 40 --warmup 15 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('schedule $sch $wl', round(d['value']), 'env-steps/s, launch', round(d['roofline']['launch_ms'],3), 'ms')"
  done
done
