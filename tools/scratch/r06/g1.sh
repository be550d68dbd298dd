#!/bin/bash
# round 6, call 1: sanity of the tree (GPU tests), then the occupancy experiment with its PMC passes
set -u
export TMPDIR=/tmp
O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/g1_pytest.log 2>&1; echo "pytest rc $?" >> $O/g1_pytest.log; tail -3 $O/g1_pytest.log
timeout 1500 python tools/micro/occupancy_lean.py run > $O/occupancy_lean.log 2>&1; tail -14 $O/occupancy_lean.log
for V in "20480 2048" "0 3072"; do
  set -- $V
  for N in 4096 6144; do
    timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $O/occ_pmc_$1_$2_$N -o pmc -- python tools/micro/occupancy_lean.py one lean $1 $2 $N > $O/occ_pmc_$1_$2_$N.log 2>&1
    F=$(find $O/occ_pmc_$1_$2_$N -name "*counter_collection.csv" | head -1)
    [ -n "$F" ] && python tools/summarize_pmc.py $F $O/occ_pmc_$1_$2_$N.json | grep step_queue | head -2
    rm -rf $O/occ_pmc_$1_$2_$N
  done
done
