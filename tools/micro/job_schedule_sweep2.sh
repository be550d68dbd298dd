for sch in "6,5,4" "5,5,5" "6,5,3,1" "7,4,4" "6,6,3" "5,5,3,2" "4,4,4,3" "5,4,3,3" "6,5,4"; do
  for wl in tracked random_init objects; do
    KP_JOB_SCHEDULE=$sch timeout 200 python bench.py --workload $wl --no-cpu-baseline --no-secondary --steps 40 --warmup 15 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('schedule $sch $wl', round(d['value']), 'env-steps/s, launch', round(d['roofline']['launch_ms'],3), 'ms')"
  done
done
