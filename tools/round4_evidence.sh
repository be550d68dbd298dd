#!/bin/bash
# One gpurun call that regenerates the round-4 evidence under gpurun_out/r04_evidence/ (copy the summaries to profiles/r04/ afterwards).
# Instrumented libraries are built on the CPU side first (tools/micro/flip_instr.py, obj_instr.py, obj_instr2.py, newton_instr.py).
set -u
export TMPDIR=/tmp
export KP_ROUND=r04
E=gpurun_out/r04_evidence
mkdir -p $E
T="timeout -s KILL"
$T 900 python -m pytest tests -m gpu -q > $E/pytest_gpu.log 2>&1
$T 120 python -c "import __graft_entry__ as g; g.smoke()" > $E/smoke.log 2>&1
# parity sweeps (HIP vs fp64 oracle)
$T 300 python tools/floor_fuzz.py 320 > $E/floor_fuzz.log 2>&1
( for s in 0 1 2; do $T 200 python tools/obj_fuzz.py 64 3 $s; done ) > $E/obj_fuzz.log 2>&1
( for s in 3 4 5 6 7 8; do $T 200 python tools/obj_fuzz.py 64 3 $s; done ) 2>&1 | grep "scenes x" > $E/obj_fuzz_seeds3to8.log
( $T 300 python tools/substep_parity.py floor 640; for s in 0 1 2 3 4 5 6 7 8; do $T 200 python tools/substep_parity.py objects 64 $s; done ) 2>&1 | grep -v amdgpu.ids > $E/substep_parity.log
( $T 400 python tools/substep_parity.py bench:tracked 2048; $T 400 python tools/substep_parity.py bench:random_init 2048; $T 400 python tools/substep_parity.py bench:objects 1024 ) 2>&1 | grep -v amdgpu.ids > $E/substep_parity_bench.log
mkdir -p profiles/r04 && cp $E/substep_parity_bench.log profiles/r04/      # bench.py's `parity` block reads it from there
$T 200 python tools/obs_reward_errors.py 2>&1 | grep -v amdgpu.ids > $E/obs_reward_errors.log
( for s in 0 1; do $T 200 python tools/contact_compare.py 64 $s; done ) > $E/contact_compare.log 2>&1
# where the cycles go
$T 200 python tools/obj_bench.py > $E/obj_bench.log 2>&1
$T 200 python tools/micro/obj_heavy.py > $E/obj_heavy_phases.log 2>&1
( KP_FINE=A $T 200 python tools/micro/obj_heavy2.py; KP_FINE=B $T 200 python tools/micro/obj_heavy2.py ) 2>&1 | grep -v amdgpu.ids > $E/obj_heavy_fine.log
$T 200 python tools/micro/obj_tail.py > $E/obj_tail.log 2>&1
( KP_PROFILE=1 $T 200 python tools/micro/obj_tail.py; KP_PROFILE=1 KP_OBJ_NEWTON=1 $T 200 python tools/micro/obj_tail.py ) 2>&1 | grep "cycles per substep\|launch ms" > $E/obj_tail_phases.log
$T 200 python tools/micro/collide_profile.py 2>&1 | grep -v amdgpu.ids > $E/collide_profile.log
$T 200 python tools/phase_profile.py > $E/phase_cycles.log 2>&1
$T 200 python tools/micro/flip_profile.py 2>&1 | grep -v amdgpu.ids > $E/flip_profile.log
$T 200 python tools/micro/factcost_profile.py 2>&1 | grep -v amdgpu.ids > $E/factorisation_cost.log
$T 200 python tools/micro/context_time.py 2>&1 | grep -v "amdgpu.ids\|Warning\|sched_" > $E/context_time.log
# the three passes of profile_bench.sh per workload: kernel trace + stats, then the --pmc passes (never combined with trace domains)
$T 900 tools/profile_bench.sh tracked > $E/profile_tracked.log 2>&1
$T 900 tools/profile_bench.sh objects > $E/profile_objects.log 2>&1
cp gpurun_out/r04_prof/summary/* $E/ 2>/dev/null
cp gpurun_out/r04_prof/summary/pmc_bench_*.json profiles/r04/ 2>/dev/null      # bench.py reads the PMC summaries of ITS OWN command from there
$T 900 tools/traffic_breakdown.sh > $E/traffic_with_and_without_queue_final.log 2>&1
$T 600 python tools/launches_per_step.py objects $E/lps_objects > $E/launches_per_step_objects.csv 2> /dev/null
$T 600 python tools/launches_per_step.py tracked $E/lps_tracked > $E/launches_per_step_tracked.csv 2> /dev/null
$T 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04_prof/update -o stats -- python tools/update_profile.py > $E/update_profile.log 2>&1
cp gpurun_out/r04_prof/update/stats_kernel_stats.csv $E/r04_kernel_stats_update.csv 2>/dev/null
$T 300 python tools/update_bench.py > $E/update_bench.log 2>&1
# the driver's commands
( time $T 600 python bench.py > $E/bench_default.json 2> $E/bench_default.err ) 2> $E/bench_default.time
$T 300 python bench.py --workload objects --no-secondary --no-cpu-baseline > $E/bench_objects.json 2> $E/bench_objects.err
KP_BENCH_FORCE_PG=1 MASTER_PORT=29561 $T 300 python bench.py --workload train_iter --steps 2 --warmup 1 > $E/bench_train_iter_1rank_nccl.json 2> $E/bench_train_iter.err
KP_BENCH_SHARED_DEVICE=1 $T 600 python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline 2> $E/bench_2rank.err | grep '^{' > $E/bench_2rank_self_launched_shared_device.json
# the three scripts end to end
( $T 400 python scripts/train_ar_policy.py --num_envs 4096 --iters 3 --horizon 24; $T 400 python scripts/train_ar_policy.py --num_envs 4096 --iters 3 --horizon 24 --cache_init_context; \
  $T 400 python scripts/train_ar_policy.py --num_envs 4096 --iters 2 --horizon 99; $T 300 python scripts/train_uhc.py --iters 2; $T 300 python scripts/eval_ar_policy.py ) 2>&1 | grep -v "amdgpu.ids\|Warning\|sched_" > $E/scripts_run.log
# soaks: the control-step launch on mixed scenes, and the training pipeline (every episode on a fresh clip, pool top-ups, update) for 40 iterations
$T 400 python tools/soak.py 180 > $E/soak.log 2>&1
$T 600 python scripts/train_ar_policy.py --num_envs 4096 --iters 40 --horizon 24 2>&1 | grep -v "amdgpu.ids\|Warning\|sched_" | tail -3 >> $E/soak.log
# env count, warm start, and the end-to-end learning check from random init
$T 500 python tools/envs_sweep.py > $E/envs_sweep.log 2>/dev/null
$T 500 python tools/warm_start_time.py 3 2>&1 | grep -v amdgpu.ids > $E/warm_start_time.log
$T 1500 tools/learning_demo.sh 2>&1 | grep -v amdgpu.ids > $E/learning_demo.log
mkdir -p $E/learning_demo && cp gpurun_out/learning_demo/*.log $E/learning_demo/ 2>/dev/null
find gpurun_out/r04_prof $E -type f -size +2000k -delete
for f in pytest_gpu smoke floor_fuzz obj_fuzz contact_compare obj_bench update_bench flip_profile factorisation_cost; do echo "== $f"; grep -v Warn $E/$f.log | tail -4 | cut -c1-400; done
cut -c1-600 $E/bench_default.json
