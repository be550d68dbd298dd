#!/bin/bash
# round 6: refresh of the evidence that is stamped with the device-code fingerprint (the last change touched kp_sim.hip), on the final tree
set -u
export TMPDIR=/tmp
export KP_ROUND=r06
E=gpurun_out/r06_evidence
mkdir -p $E profiles/r06
T="timeout -s KILL"
SHA=$(python -c "from kinpoly_amd.build import kernel_source_sha256 as k; print(k())")
echo "kernel_source_sha256 $SHA" > $E/kernel_source_sha256.txt
$T 1500 python -m pytest tests -m gpu -q > $E/pytest_gpu.log 2>&1; tail -2 $E/pytest_gpu.log
$T 120 python -c "import __graft_entry__ as g; g.smoke()" > $E/smoke.log 2>&1; tail -1 $E/smoke.log
( echo "kernel_source_sha256 $SHA"; $T 600 python tools/substep_parity.py bench:tracked 2048; $T 600 python tools/substep_parity.py bench:random_init 2048; $T 600 python tools/substep_parity.py bench:objects 1024 ) 2>&1 | grep -v amdgpu.ids > $E/substep_parity_bench.log
cp $E/substep_parity_bench.log profiles/r06/
$T 900 tools/profile_bench.sh tracked > $E/profile_tracked.log 2>&1
$T 900 tools/profile_bench.sh objects > $E/profile_objects.log 2>&1
cp gpurun_out/r06_prof/summary/* $E/ 2>/dev/null
cp gpurun_out/r06_prof/summary/pmc_bench_*.json profiles/r06/ 2>/dev/null
( time $T 900 python bench.py > $E/bench_default.json 2> $E/bench_default.err ) 2> $E/bench_default.time
$T 300 python bench.py --workload objects --no-secondary --no-cpu-baseline > $E/bench_objects.json 2> $E/bench_objects.err
KP_LEAN_QUEUE=0 $T 300 python bench.py --workload tracked --no-secondary --no-cpu-baseline --no-parity-live > $E/bench_tracked_full_layout.json 2> /dev/null
$T 300 python tools/soak.py 120 > $E/soak.log 2>&1; tail -1 $E/soak.log
find gpurun_out/r06_prof $E -type f -size +2000k -delete
cut -c1-400 $E/bench_default.json; cat $E/bench_default.time
