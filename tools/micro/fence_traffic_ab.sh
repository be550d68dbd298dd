export TMPDIR=/tmp
OUT=gpurun_out/r04_traffic_fence; mkdir -p $OUT
for F in 0 1; do for C in FETCH_SIZE WRITE_SIZE; do
  KP_QUEUE_FENCE=$F timeout -s KILL 300 rocprofv3 --pmc $C --output-format csv -d $OUT/objects_f${F}_$C -o pmc -- python bench.py --workload objects --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > /dev/null 2>&1
done; done
python - <<'PY'
import csv, glob, os
out = "gpurun_out/r04_traffic_fence"
for f in (0, 1):
    tot = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        vals = {}
        for path in glob.glob(os.path.join(out, f"objects_f{f}_{c}", "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(path)):
                if "kp_step" in row.get("Kernel_Name", "") and "forward" not in row.get("Kernel_Name", ""):
                    vals.setdefault(row.get("Dispatch_Id", "0"), 0.0); vals[row["Dispatch_Id"]] += float(row["Counter_Value"])
        v = sorted(vals.values()); tot[c] = v[len(v) // 2] if v else float("nan")
    hbm = (tot["FETCH_SIZE"] * 2 + tot["WRITE_SIZE"]) * 1024
    print(f"objects queue_fence={f}: FETCH_SIZE {tot['FETCH_SIZE'] / 1024:.1f} MB (x2) WRITE_SIZE {tot['WRITE_SIZE'] / 1024:.1f} MB -> {hbm / 1e6:.1f} MB per launch = {hbm / (3292 * 4096):.2f} x algorithmic")
PY
find $OUT -type f -size +500k -delete
