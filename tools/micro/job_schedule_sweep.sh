for sch in "3,3,3,3,3" "5,5,5" "6,4,3,2" "5,4,3,3" "7,5,3" "8,4,3" "6,5,4" "4,4,4,3" "9,6"; do
  echo "schedule $sch"
  KP_JOB_SCHEDULE=$sch timeout 100 python tools/pmc_step.py 2>&1 | grep ms/launch
  KP_JOB_SCHEDULE=$sch timeout 200 python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', round(d['value']), round(d['roofline']['launch_ms'],3))"
done
