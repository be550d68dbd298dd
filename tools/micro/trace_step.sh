cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -- python /root/repo/bench.py --workload tracked --steps 6 --warmup 4 --no-cpu-baseline --no-secondary > /dev/null 2>&1
f=$(find /tmp/tr -name "*kernel_trace.csv" | head -1)
python - "$f" <<'P'
import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
names=[r['Kernel_Name'] for r in rows]
# find the last two physics launches, print kernels between them
idx=[i for i,n in enumerate(names) if 'kp_step_queue_kernel' in n]
a,b=idx[-2],idx[-1]
for r in rows[a:b+1]:
    d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
    print(f"{d:8.1f} us  {r['Kernel_Name'][:100]}")
P
