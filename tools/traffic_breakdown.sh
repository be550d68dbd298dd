#!/bin/bash
# HBM traffic of the control-step kernel with and without the job queue (KP_SUBSTEPS_PER_JOB=0: one workgroup per env, no hand-overs through memory):
# what part of traffic_over_algorithmic is the queue's deliberate state round trips, what part is the kernel's own (scratch, tables).
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r04_traffic
mkdir -p $OUT
for WL in tracked objects; do for SPJ in 0 4; do
  for C in FETCH_SIZE WRITE_SIZE; do
    KP_SUBSTEPS_PER_JOB=$SPJ timeout -s KILL 300 rocprofv3 --pmc $C --output-format csv -d $OUT/${WL}_spj${SPJ}_$C -o pmc -- python bench.py --workload $WL --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > /dev/null 2>&1
  done
done; done
python - <<'PY'
import csv, glob, os
out = "gpurun_out/r04_traffic"
algo = {"tracked": 2772 * 4096, "objects": 3292 * 4096}
for wl in ("tracked", "objects"):
    for spj in (0, 4):
        tot = {}
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            vals = {}
            for path in glob.glob(os.path.join(out, f"{wl}_spj{spj}_{c}", "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(path)):
                    if "kp_step" in row.get("Kernel_Name", "") and "forward" not in row.get("Kernel_Name", ""):
                        vals.setdefault(row.get("Dispatch_Id", "0"), 0.0)
                        vals[row["Dispatch_Id"]] += float(row["Counter_Value"])
            v = sorted(vals.values())
            tot[c] = v[len(v) // 2] if v else float("nan")
        hbm = (tot["FETCH_SIZE"] * 2 + tot["WRITE_SIZE"]) * 1024
        print(f"{wl} substeps_per_job={spj}: FETCH_SIZE {tot['FETCH_SIZE'] / 1024:.1f} MB (x2 per the guide) WRITE_SIZE {tot['WRITE_SIZE'] / 1024:.1f} MB -> {hbm / 1e6:.1f} MB per launch = {hbm / algo[wl]:.2f} x algorithmic")
PY
find $OUT -type f -size +500k -delete
