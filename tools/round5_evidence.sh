#!/bin/bash
# One gpurun call that regenerates the round-5 evidence under gpurun_out/r05_evidence/ (copy the summaries to profiles/r05/ afterwards).
set -u
export TMPDIR=/tmp
export KP_ROUND=r05
E=gpurun_out/r05_evidence
mkdir -p $E profiles/r05
T="timeout -s KILL"
SHA=$(python -c "from kinpoly_amd.build import kernel_source_sha256 as k; print(k())")
echo "kernel_source_sha256 $SHA" > $E/kernel_source_sha256.txt
$T 1200 python -m pytest tests -m gpu -q > $E/pytest_gpu.log 2>&1
$T 120 python -c "import __graft_entry__ as g; g.smoke()" > $E/smoke.log 2>&1
# parity sweeps (HIP vs fp64 oracle) on the final build
$T 300 python tools/floor_fuzz.py 320 > $E/floor_fuzz.log 2>&1
( for s in 0 1 2; do $T 200 python tools/obj_fuzz.py 64 3 $s; done ) > $E/obj_fuzz.log 2>&1
( for s in 3 4 5 6 7 8; do $T 200 python tools/obj_fuzz.py 64 3 $s; done ) 2>&1 | grep "scenes x" > $E/obj_fuzz_seeds3to8.log
( $T 400 python tools/substep_parity.py floor 640; for s in 0 1 2 3 4 5 6 7 8; do $T 200 python tools/substep_parity.py objects 64 $s; done ) 2>&1 | grep -v amdgpu.ids > $E/substep_parity.log
( echo "kernel_source_sha256 $SHA"; $T 600 python tools/substep_parity.py bench:tracked 2048; $T 600 python tools/substep_parity.py bench:random_init 2048; $T 600 python tools/substep_parity.py bench:objects 1024 ) 2>&1 | grep -v amdgpu.ids > $E/substep_parity_bench.log
cp $E/substep_parity_bench.log profiles/r05/      # bench.py's `parity` block reads it from there (stamped with the kernel source fingerprint)
( for s in 0 1; do $T 200 python tools/contact_compare.py 64 $s; done ) > $E/contact_compare.log 2>&1
$T 200 python tools/obs_reward_errors.py 2>&1 | grep -v amdgpu.ids > $E/obs_reward_errors.log
# concurrency, sampler regime, occupancy
( for a in "2 4096 50 2" "3 4096 50 2" "3 4096 50 7"; do $T 240 python tools/micro/concurrent_handles.py $a; done ) 2>&1 | grep -v amdgpu.ids > $E/concurrent_handles.log
( for a in "" "--blocking" "--pool_depth 12"; do $T 300 python tools/sampler_regime.py $a 2>/dev/null | head -c 900; echo; done ) > $E/sampler_regime.log
$T 300 python tools/sampler_regime.py --profile 2>/dev/null > $E/sampler_regime_profile.log
$T 200 python tools/obj_bench.py > $E/obj_bench.log 2>&1
$T 200 python tools/phase_profile.py > $E/phase_cycles.log 2>&1
# the constraint solve's starting point (model option warm_extrap): beta sweep on three workloads, 60 timed steps each
( for we in 0 0.5 0.75 1; do for wl in objects tracked random_init wild_eval; do KP_WARM_EXTRAP=$we $T 300 python bench.py --workload $wl --steps 60 --warmup 20 --no-secondary --no-cpu-baseline --no-parity-live 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$wl warm_extrap=$we value %.0f launch_ms %.4f newton/substep %.3f fact/substep %.3f' % (d['value'], d['roofline']['launch_ms'], d['newton_iters_per_substep'], d['hessian_factorisations_per_substep']))"
done; done ) > $E/warm_extrap_sweep.log 2>&1
# rocprofv3: kernel trace + stats, then the --pmc passes (never combined with trace domains), per workload
$T 900 tools/profile_bench.sh tracked > $E/profile_tracked.log 2>&1
$T 900 tools/profile_bench.sh objects > $E/profile_objects.log 2>&1
cp gpurun_out/r05_prof/summary/* $E/ 2>/dev/null
cp gpurun_out/r05_prof/summary/pmc_bench_*.json profiles/r05/ 2>/dev/null      # bench.py reads the PMC summaries of ITS OWN command from there
$T 600 python tools/launches_per_step.py tracked $E/lps_tracked > $E/launches_per_step_tracked.csv 2> /dev/null
$T 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r05_prof/update -o stats -- python tools/update_profile.py > $E/update_profile.log 2>&1
cp gpurun_out/r05_prof/update/stats_kernel_stats.csv $E/r05_kernel_stats_update.csv 2>/dev/null
# the driver's commands
( time $T 900 python bench.py > $E/bench_default.json 2> $E/bench_default.err ) 2> $E/bench_default.time
$T 300 python bench.py --workload objects --no-secondary --no-cpu-baseline > $E/bench_objects.json 2> $E/bench_objects.err
KP_BENCH_FORCE_PG=1 MASTER_PORT=29561 $T 300 python bench.py --workload train_iter --steps 2 --warmup 1 > $E/bench_train_iter_1rank_nccl.json 2> $E/bench_train_iter.err
KP_BENCH_SHARED_DEVICE=1 $T 600 python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline 2> $E/bench_2rank.err | grep '^{' > $E/bench_2rank_self_launched_shared_device.json
$T 60 python bench.py --gpus 4 > $E/bench_gpus4_on_one_gpu.log 2>&1; echo "exit code $?" >> $E/bench_gpus4_on_one_gpu.log
( $T 400 python scripts/train_ar_policy.py --num_envs 4096 --iters 3 --horizon 24; $T 400 python scripts/train_ar_policy.py --num_envs 4096 --iters 2 --horizon 24 --update_dtype fp64; \
  $T 300 python scripts/train_uhc.py --iters 2; $T 300 python scripts/eval_ar_policy.py ) 2>&1 | grep -v "amdgpu.ids\|Warning\|sched_" > $E/scripts_run.log
$T 300 python tools/soak.py 120 > $E/soak.log 2>&1
$T 200 python tools/mujoco_pin.py > $E/mujoco_pin.log 2>&1
find gpurun_out/r05_prof $E -type f -size +2000k -delete
for f in pytest_gpu smoke floor_fuzz obj_fuzz contact_compare concurrent_handles soak; do echo "== $f"; grep -v Warn $E/$f.log | tail -4 | cut -c1-400; done
cut -c1-500 $E/bench_default.json
