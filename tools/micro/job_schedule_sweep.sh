for sch in "7,5,3" "5,4,3,3" "4,4,3,2,2" "5,4,3,2,1" "3,3,3,3,3" "6,4,3,2" "4,3,3,3,2" "3,3,3,2,2,2"; do
  echo "schedule $sch"
  KP_JOB_SCHEDULE=$sch timeout 100 python tools/pmc_step.py 2>&1 | grep ms/launch
  KP_JOB_SCHEDULE=$sch timeout 200 python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', round(d['value']), round(d['roofline']['launch_ms'],3))"
done
