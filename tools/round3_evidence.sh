#!/bin/bash
# One gpurun call that regenerates the round-3 evidence under gpurun_out/r03_evidence/ (copy the summaries to profiles/r03/ afterwards).
# The instrumented library behind obj_profile_newton.log is built on the CPU side first: python tools/micro/obj_instr.py tools/micro/bin/libkinpoly_sim_objnewton.so
set -u
export TMPDIR=/tmp
E=gpurun_out/r03_evidence
mkdir -p $E
python -m pytest tests -m gpu -q > $E/pytest_gpu.log 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > $E/smoke.log 2>&1
python tools/floor_fuzz.py 320 > $E/floor_fuzz.log 2>&1
for s in 0 1 2; do python tools/obj_fuzz.py 64 3 $s; done > $E/obj_fuzz.log 2>&1
for s in 3 4 5 6 7 8; do python tools/obj_fuzz.py 64 3 $s; done 2>&1 | grep "scenes x" > $E/obj_fuzz_seeds3to8.log
( python tools/obj_fuzz_trace.py 2 59; python tools/obj_fuzz_trace.py 2 40 ) > $E/obj_fuzz_trace.log 2>&1      # the one scene left above 1e-4: growth of rounding, every substep agrees from a common state
( python tools/substep_parity.py floor 640; for s in 0 1 2 3 4 5 6 7 8; do python tools/substep_parity.py objects 64 $s; done ) 2>&1 | grep -v amdgpu.ids > $E/substep_parity.log
( python tools/substep_parity.py bench:tracked 2048; python tools/substep_parity.py bench:random_init 2048; python tools/substep_parity.py bench:objects 1024 ) 2>&1 | grep -v amdgpu.ids > $E/substep_parity_bench.log
python tools/obs_reward_errors.py 2>&1 | grep -v amdgpu.ids > $E/obs_reward_errors.log
for s in 0 2; do KP_PLANEMESH=4,0.001 python tools/obj_fuzz.py 64 3 $s; done > $E/obj_fuzz_round2_rule.log 2>&1
for s in 0 1; do python tools/contact_compare.py 64 $s; done > $E/contact_compare.log 2>&1
python tools/obj_bench.py > $E/obj_bench.log 2>&1
python tools/micro/obj_profile.py > $E/obj_profile_phases.log 2>&1
KP_OBJ_NEWTON=1 python tools/micro/obj_profile.py > $E/obj_profile_newton.log 2>&1
python tools/micro/obj_tail.py > $E/obj_tail.log 2>&1
python tools/micro/obj_heavy.py > $E/obj_heavy_phases.log 2>&1
KP_OBJ_NEWTON=1 python tools/micro/obj_heavy.py > $E/obj_heavy_newton.log 2>&1
bash tools/micro/trace_step.sh > $E/trace_step.log 2>&1
python tools/phase_profile.py > $E/phase_cycles.log 2>&1
python tools/update_bench.py > $E/update_bench.log 2>&1
# the three passes of profile_bench.sh per workload: kernel trace + stats, then the --pmc passes (never combined with trace domains)
tools/profile_bench.sh tracked > $E/profile_tracked.log 2>&1
tools/profile_bench.sh objects > $E/profile_objects.log 2>&1
cp gpurun_out/r03_prof/summary/* $E/ 2>/dev/null
mkdir -p profiles/r03 && cp gpurun_out/r03_prof/summary/pmc_bench_*.json profiles/r03/ 2>/dev/null      # bench.py reads the PMC summaries of ITS OWN command from there
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03_prof/update -o stats -- python tools/update_profile.py > $E/update_profile.log 2>&1
cp gpurun_out/r03_prof/update/stats_kernel_stats.csv $E/r03_kernel_stats_update.csv 2>/dev/null
( time python bench.py > $E/bench_default.json 2> $E/bench_default.err ) 2> $E/bench_default.time
python bench.py --workload objects --no-secondary --no-cpu-baseline > $E/bench_objects.json 2> $E/bench_objects.err
KP_BENCH_FORCE_PG=1 MASTER_PORT=29561 python bench.py --workload train_iter --steps 2 --warmup 1 > $E/bench_train_iter_1rank_nccl.json 2> $E/bench_train_iter.err
KP_BENCH_SHARED_DEVICE=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29557 bench.py --gpus 2 --steps 20 --warmup 5 2> $E/bench_2rank.err | grep '^{' > $E/bench_2rank_shared_device.json     # gloo prints its own lines on stdout
find gpurun_out/r03_prof -type f -size +2000k -delete
for f in pytest_gpu smoke floor_fuzz obj_fuzz contact_compare obj_bench obj_profile_phases obj_profile_newton update_bench; do echo "== $f"; grep -v Warn $E/$f.log | tail -4 | cut -c1-400; done
cut -c1-600 $E/bench_default.json
