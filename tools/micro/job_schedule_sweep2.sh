# second sweep (round 3, after the torque hand-over made a hand-over cheaper): finer schedules on the three bench workloads
for sch in "6,5,4" "5,4,3,2,1" "6,4,3,2" "5,4,3,3" "4,4,4,3" "5,4,4,2" "6,5,3,1" "4,3,3,3,2" "7,5,3"; do
  for wl in tracked random_init objects; do
    KP_JOB_SCHEDULE=$sch timeout 200 python bench.py --workload $wl --no-cpu-baseline --no-secondary --steps 40 --warmup 15 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('schedule $sch $wl', round(d['value']), 'env-steps/s, launch', round(d['roofline']['launch_ms'],3), 'ms')"
  done
done
