#!/bin/bash
# rocprofv3 passes of the bench command itself (run ON the GPU box through gpurun, from the repo root):
#   kernel trace + stats, then one --pmc pass per counter group (FETCH_SIZE and WRITE_SIZE do not fit one pass; no trace domains with --pmc).
# Results land in gpurun_out/${KP_ROUND:-r05}_prof/ ; tools/make_pmc_summary.py turns them into profiles/${KP_ROUND:-r05}/pmc_bench_<workload>.json.
#   usage: tools/profile_bench.sh [workload]
set -u
WL=${1:-tracked}
OUT=gpurun_out/${KP_ROUND:-r05}_prof/$WL
mkdir -p "$OUT"
export TMPDIR=/tmp
CMD="python bench.py --workload $WL --steps 30 --warmup 10 --no-cpu-baseline --no-secondary --no-parity-live --repeats 1"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o stats -- $CMD > "$OUT/bench_stats.log" 2>&1
for C in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU"; do
    TAG=$(echo "$C" | tr ' ' '+')
    rocprofv3 --pmc $C --output-format csv -d "$OUT/pmc_$TAG" -o pmc -- $CMD > "$OUT/pmc_$TAG.log" 2>&1
done
find "$OUT" -name "*.csv" | head -20
python tools/make_pmc_summary.py "$OUT" "$WL" "$CMD" > "$OUT/summary.log" 2>&1
tail -5 "$OUT/summary.log"
