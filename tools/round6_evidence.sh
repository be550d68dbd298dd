#!/bin/bash
# One gpurun call that regenerates the round-6 evidence under gpurun_out/r06_evidence/ (summaries are copied to profiles/r06/ by the caller).
set -u
export TMPDIR=/tmp
export KP_ROUND=r06
E=gpurun_out/r06_evidence
mkdir -p $E profiles/r06
T="timeout -s KILL"
SHA=$(python -c "from kinpoly_amd.build import kernel_source_sha256 as k; print(k())")
echo "kernel_source_sha256 $SHA" > $E/kernel_source_sha256.txt
$T 1500 python -m pytest tests -m gpu -q > $E/pytest_gpu.log 2>&1
$T 120 python -c "import __graft_entry__ as g; g.smoke()" > $E/smoke.log 2>&1
# parity sweeps (HIP vs fp64 oracle) on the final build: the lean queue layout is what 4096-env launches of floor scenes run on
$T 300 python tools/floor_fuzz.py 320 > $E/floor_fuzz.log 2>&1
( for s in 0 1 2; do $T 200 python tools/obj_fuzz.py 64 3 $s; done ) > $E/obj_fuzz.log 2>&1
( echo "kernel_source_sha256 $SHA"; $T 600 python tools/substep_parity.py bench:tracked 2048; $T 600 python tools/substep_parity.py bench:random_init 2048; $T 600 python tools/substep_parity.py bench:objects 1024 ) 2>&1 | grep -v amdgpu.ids > $E/substep_parity_bench.log
cp $E/substep_parity_bench.log profiles/r06/      # bench.py's `parity` block reads it from there (stamped with the kernel source fingerprint)
( for s in 0 1; do $T 200 python tools/contact_compare.py 64 $s; done ) > $E/contact_compare.log 2>&1
# whole-episode parity: configs[2] 128 x 99, configs[3] 32 x 99
$T 900 python tools/episode_parity.py --envs 128 --steps 99 --json $E/episode_parity_floor.json > $E/episode_parity_floor.log 2>&1
$T 900 python tools/episode_parity.py --envs 32 --steps 99 --objects --json $E/episode_parity_objects.json > $E/episode_parity_objects.log 2>&1
# rocprofv3: kernel trace + stats, then the --pmc passes (never combined with trace domains), per workload
$T 900 tools/profile_bench.sh tracked > $E/profile_tracked.log 2>&1
$T 900 tools/profile_bench.sh objects > $E/profile_objects.log 2>&1
cp gpurun_out/r06_prof/summary/* $E/ 2>/dev/null
cp gpurun_out/r06_prof/summary/pmc_bench_*.json profiles/r06/ 2>/dev/null      # bench.py reads the PMC summaries of ITS OWN command from there
$T 200 python tools/phase_profile.py > $E/phase_cycles.log 2>&1
# the driver's commands
( time $T 900 python bench.py > $E/bench_default.json 2> $E/bench_default.err ) 2> $E/bench_default.time
$T 300 python bench.py --workload objects --no-secondary --no-cpu-baseline > $E/bench_objects.json 2> $E/bench_objects.err
KP_LEAN_QUEUE=0 $T 300 python bench.py --workload tracked --no-secondary --no-cpu-baseline --no-parity-live > $E/bench_tracked_full_layout.json 2> /dev/null
KP_BENCH_FORCE_PG=1 MASTER_PORT=29561 $T 300 python bench.py --workload train_iter --steps 2 --warmup 1 > $E/bench_train_iter_1rank_nccl.json 2> $E/bench_train_iter.err
KP_BENCH_SHARED_DEVICE=1 $T 600 python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline 2> $E/bench_2rank.err | grep '^{' > $E/bench_2rank_self_launched_shared_device.json
( $T 400 python scripts/train_ar_policy.py --num_envs 4096 --iters 8 --horizon 24 --min_batch_size 10000; $T 400 python scripts/train_ar_policy.py --num_envs 4096 --iters 2 --horizon 24 --update_dtype fp64; \
  $T 300 python scripts/train_uhc.py --iters 2; $T 300 python scripts/eval_ar_policy.py ) 2>&1 | grep -v "amdgpu.ids\|Warning\|sched_" | cut -c1-1200 > $E/scripts_run.log
$T 300 python tools/soak.py 120 > $E/soak.log 2>&1
$T 200 python tools/mujoco_pin.py --report > $E/mujoco_pin.log 2>&1
find gpurun_out/r06_prof $E -type f -size +2000k -delete
for f in pytest_gpu smoke floor_fuzz obj_fuzz contact_compare soak; do echo "== $f"; grep -v Warn $E/$f.log | tail -4 | cut -c1-400; done
cut -c1-600 $E/bench_default.json
bash tools/trained_policy_bench.sh gpurun_out/r06_evidence/trained > gpurun_out/r06_evidence/trained_policy_bench.log 2>&1; tail -4 gpurun_out/r06_evidence/trained_policy_bench.log | cut -c1-1500
