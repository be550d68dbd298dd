#!/bin/bash
# round 6: what do three waves per SIMD wait for?  LDS-side counters of the lean queue kernel (tracked workload)
set -u
export TMPDIR=/tmp
O=gpurun_out/r06; mkdir -p $O
CMD="python bench.py --workload tracked --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-parity-live --repeats 1"
rocprofv3 -L 2>/dev/null | grep -i -E "^\s*(SQ_LDS|SQ_WAIT|SQ_ACTIVE_INST|SQ_INST_CYCLES|SQ_INSTS_LDS|SQ_LDS_)" | head -40 > $O/pmc_available.txt
for C in "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD"; do
  TAG=$(echo "$C" | tr ' ' '+')
  for lean in 1 0; do
    KP_LEAN_QUEUE=$lean timeout -s KILL 300 rocprofv3 --pmc $C --output-format csv -d $O/pmcx_$lean -o pmc -- $CMD > /dev/null 2>&1
    F=$(find $O/pmcx_$lean -name "*counter_collection.csv" | head -1)
    [ -n "$F" ] && python - "$F" "$lean" <<'PY'
import csv, sys, collections
acc=collections.defaultdict(lambda: collections.defaultdict(float))
for row in csv.DictReader(open(sys.argv[1])):
    if "kp_step_queue" in row.get("Kernel_Name",""):
        acc[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
print("lean=%s" % sys.argv[2], {k: "%.4g" % sorted(v.values())[len(v)//2] for k,v in acc.items()})
PY
    rm -rf $O/pmcx_$lean
  done
done 2>&1 | tee $O/pmc_lds_side.log
