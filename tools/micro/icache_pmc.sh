export TMPDIR=/tmp
OUT=gpurun_out/r04_icache; mkdir -p $OUT
for WL in tracked objects; do
  timeout -s KILL 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_INSTS_VALU --output-format csv -d $OUT/$WL -o pmc -- python bench.py --workload $WL --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $OUT/$WL.log 2>&1
  timeout -s KILL 300 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_IFETCH SQ_WAIT_ANY --output-format csv -d $OUT/${WL}_b -o pmc -- python bench.py --workload $WL --steps 20 --warmup 5 --no-cpu-baseline --no-secondary >> $OUT/$WL.log 2>&1
done
python - <<'PY'
import csv, glob, os, collections
out = "gpurun_out/r04_icache"
for wl in ("tracked", "objects"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for d in (wl, wl + "_b"):
        for path in glob.glob(os.path.join(out, d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(path)):
                if "kp_step_queue" in row.get("Kernel_Name", ""):
                    acc[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
    med = {k: sorted(v.values())[len(v) // 2] for k, v in acc.items()}
    print(wl, {k: f"{v:.4g}" for k, v in med.items()})
    if "SQC_ICACHE_REQ" in med and med["SQC_ICACHE_REQ"] > 0:
        print("   icache hit rate %.4f, misses per 1000 VALU insts %.2f" % (med.get("SQC_ICACHE_HITS", 0) / med["SQC_ICACHE_REQ"], 1000 * med.get("SQC_ICACHE_MISSES", 0) / max(med.get("SQ_INSTS_VALU", 1), 1)))
PY
tail -3 $OUT/tracked.log | cut -c1-300
find $OUT -type f -size +500k -delete
