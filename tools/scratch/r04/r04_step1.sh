#!/bin/bash
# round 4, first measurement: the dataset-driven training iteration from random init (every episode through init_context), launches per env-step with objects
set -u
export TMPDIR=/tmp
E=gpurun_out/r04_a
mkdir -p $E
timeout -s KILL 300 python scripts/train_ar_policy.py --num_envs 4096 --iters 3 --horizon 24 > $E/train_h24.log 2>&1
timeout -s KILL 300 python scripts/train_ar_policy.py --num_envs 4096 --iters 2 --horizon 99 > $E/train_h99.log 2>&1
timeout -s KILL 600 python bench.py --workload train_iter --steps 2 --warmup 1 > $E/bench_train_iter.json 2> $E/bench_train_iter.err
timeout -s KILL 600 rocprofv3 --kernel-trace --stats --output-format csv -d $E/prof_objects -o stats -- python bench.py --workload objects --steps 30 --warmup 10 --no-cpu-baseline --no-secondary > $E/bench_objects_prof.log 2>&1
cp $(find $E/prof_objects -name "*kernel_stats.csv" | head -1) $E/r04_kernel_stats_objects_a.csv
find $E/prof_objects -type f -size +1000k -delete
timeout -s KILL 600 python bench.py --workload objects --no-secondary --no-cpu-baseline > $E/bench_objects.json 2> $E/bench_objects.err
grep -h "iter" $E/train_h24.log $E/train_h99.log | cut -c1-700
cut -c1-1500 $E/bench_train_iter.json
python - <<'PY'
import csv,sys
rows=list(csv.DictReader(open('gpurun_out/r04_a/r04_kernel_stats_objects_a.csv')))
tot=sum(int(r['Calls']) for r in rows)
print('launches total', tot, 'per env-step (40 steps)', tot/40.0)
for r in rows[:12]: print(r['Name'][:70], r['Calls'], r['AverageNs'], r['Percentage'])
PY
cut -c1-400 $E/bench_objects.json
