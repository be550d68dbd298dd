#!/bin/bash
# One gpurun call that regenerates the round-2 evidence under gpurun_out/r02_evidence/ (copy the summaries to profiles/r02/ afterwards).
set -u
export TMPDIR=/tmp
E=gpurun_out/r02_evidence
mkdir -p $E
python -m pytest tests -m gpu -q > $E/pytest_gpu.log 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > $E/smoke.log 2>&1
tools/micro/valu_probe > $E/valu_probe.log 2>&1
python tools/floor_fuzz.py 320 > $E/floor_fuzz.log 2>&1
for s in 0 1 2; do python tools/obj_fuzz.py 64 3 $s; done > $E/obj_fuzz.log 2>&1
for s in 0 1; do python tools/contact_compare.py 64 $s; done > $E/contact_compare.log 2>&1
python tools/obj_bench.py > $E/obj_bench.log 2>&1
python tools/phase_profile.py > $E/phase_cycles.log 2>&1
python tools/queue_fence_bench.py > $E/queue_fence_bench.log 2>&1
python tools/update_bench.py > $E/update_bench.log 2>&1
# micro probes behind DESIGN.md section 6 (the instrumented libraries under tools/micro/bin are built by the scripts' "build" step / newton_instr.py)
tools/micro/bin/lds_probe > $E/lds_granule_probe.log 2>&1
python tools/micro/occupancy_premise.py > $E/occupancy_premise.log 2>&1
python tools/micro/queue_timeline.py > $E/queue_timeline.log 2>&1
python tools/micro/queue_policy_sim.py > $E/queue_policy_sim.log 2>&1
python tools/micro/newton_profile.py > $E/newton_profile.log 2>&1
python bench.py > $E/bench_default.json 2> $E/bench_default.err
KP_BENCH_SHARED_DEVICE=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29557 bench.py --gpus 2 --steps 20 --warmup 5 2> $E/bench_2rank.err | grep '^{' > $E/bench_2rank_shared_device.json     # gloo prints its own lines on stdout
tools/profile_bench.sh tracked > $E/profile_tracked.log 2>&1
cp gpurun_out/r02_prof/summary/* $E/ 2>/dev/null
cp gpurun_out/r02_prof/tracked/stats/stats_kernel_stats.csv $E/r02_kernel_stats_tracked_full.csv 2>/dev/null
find gpurun_out/r02_prof -type f -size +2000k -delete
for f in pytest_gpu smoke valu_probe floor_fuzz obj_fuzz contact_compare queue_fence_bench update_bench lds_granule_probe occupancy_premise queue_timeline queue_policy_sim newton_profile; do echo "== $f"; tail -4 $E/$f.log | cut -c1-400; done
cut -c1-600 $E/bench_default.json
