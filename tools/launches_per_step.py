"""Steady-state kernel launches per env-step of a bench.py workload: two rocprofv3 --kernel-trace --stats runs that differ only in --steps; the
difference of the per-kernel call counts / the difference of the step counts is what ONE timed env-step launches (set-up, warm-up and the
profiler's own start-up cancel).      python tools/launches_per_step.py <workload> <out_dir> [steps_a steps_b]"""
import csv
import glob
import os
import subprocess
import sys

wl, out = sys.argv[1], sys.argv[2]
sa, sb = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (10, 50)
os.makedirs(out, exist_ok=True)
counts = {}
for steps in (sa, sb):
    d = os.path.join(out, f"steps{steps}")
    cmd = ["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", d, "-o", "stats", "--", sys.executable, "bench.py", "--workload", wl,
           "--steps", str(steps), "--warmup", "10", "--no-cpu-baseline", "--no-secondary"]
    subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, env=dict(os.environ, TMPDIR="/tmp"), timeout=900)
    path = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)[0]
    with open(path) as f:
        counts[steps] = {r["Name"]: (int(r["Calls"]), float(r["TotalDurationNs"])) for r in csv.DictReader(f)}
    for big in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        os.remove(big)
rows = []
for name in set(counts[sa]) | set(counts[sb]):
    ca, ta = counts[sa].get(name, (0, 0.0)); cb, tb = counts[sb].get(name, (0, 0.0))
    if cb != ca:
        rows.append((name, (cb - ca) / (sb - sa), (tb - ta) / (sb - sa) * 1e-3))
rows.sort(key=lambda r: -r[2])
tot = sum(r[1] for r in rows)
print(f"# workload {wl}: kernel launches per env-step = {tot:.2f}  (call counts of --steps {sb} minus --steps {sa}, / {sb - sa}); GPU time per env-step {sum(r[2] for r in rows):.1f} us")
print("launches_per_step,us_per_step,kernel")
for name, c, us in rows:
    print(f"{c:.3f},{us:.1f},{name[:150]}")
