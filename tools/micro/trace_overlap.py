"""Concurrency summary of a rocprofv3 --kernel-trace CSV: for the control-step kernel, how many launches overlap in time."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
ph = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r.get('Queue_Id'), r.get('Stream_Id')) for r in rows if 'kp_step_kernel' in r['Kernel_Name'])
ph = ph[len(ph) // 2:]
t0 = ph[0][0]
for s, e, q, st in ph[:24]:
    print(f"phys q{q} s{st}: start {(s - t0) / 1e3:9.1f} us  end {(e - t0) / 1e3:9.1f}  dur {(e - s) / 1e3:8.1f}")
allk = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows)
busy = 0; cur_s, cur_e = allk[0]
for s, e in allk[1:]:
    if s > cur_e: busy += cur_e - cur_s; cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
busy += cur_e - cur_s
print("span", (allk[-1][1] - allk[0][0]) / 1e6, "ms; some kernel running", busy / 1e6, "ms")
