#!/bin/bash
# more of the same on the final kernels (stamp in the log): twelve more seeds of the object sweep, 960 floor scenes, a 600 s soak
export TMPDIR=/tmp; E=gpurun_out/r05; mkdir -p $E; T="timeout -s KILL"
python -c "from kinpoly_amd.build import kernel_source_sha256 as k; print('kernel_source_sha256', k())" > $E/more_sweeps_stamp.txt
( cat $E/more_sweeps_stamp.txt; for s in 9 10 11 12 13 14 15 16 17 18 19 20; do $T 200 python tools/obj_fuzz.py 64 3 $s; done ) 2>&1 | grep "scenes x\|sha256\|  scene " > $E/obj_fuzz_seeds9to20.log
( cat $E/more_sweeps_stamp.txt; $T 900 python tools/floor_fuzz.py 960 ) 2>&1 | grep -v "amdgpu.ids" > $E/floor_fuzz_960.log
( cat $E/more_sweeps_stamp.txt; $T 900 python tools/soak.py 600 ) 2>&1 | grep -v "amdgpu.ids" > $E/soak_600s.log
tail -3 $E/obj_fuzz_seeds9to20.log | cut -c1-300; tail -2 $E/floor_fuzz_960.log | cut -c1-300; tail -1 $E/soak_600s.log | cut -c1-400
