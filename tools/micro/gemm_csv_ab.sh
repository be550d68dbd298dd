export TMPDIR=/tmp
for r in 1 2; do for f in "" tools/micro/tunableop_candidate.csv; do
  KP_TUNED_GEMMS=$f timeout -s KILL 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
t=d['train_iteration']['4096x24']
print('csv=[$f] value %.0f ms %.3f mfma_ms %.4f T_update %.4f T_sample %.3f' % (d['value'], d['ms_per_step'], d['mfma']['ms_per_step_isolated'], t['T_update'], t['T_sample']))"
done; done
