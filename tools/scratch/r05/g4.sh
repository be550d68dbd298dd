export TMPDIR=/tmp
mkdir -p gpurun_out/r05
timeout 600 python -m pytest tests/test_gpu_round5.py tests/test_gpu_sampler.py -q -x 2>&1 | tail -5
R="timeout -s KILL 150 python tools/micro/concurrent_handles.py"
( $R 2 4096 50 2; KP_QUEUE_HEAVY=0 $R 2 4096 50 2; KP_LPT_ORDER=0 $R 2 4096 50 2; KP_QUEUE_HEAVY=0 KP_LPT_ORDER=0 $R 2 4096 50 2; KP_SUBSTEPS_PER_JOB=0 $R 2 1024 50 2; $R 2 4096 50 3; $R 1 4096 50 1 ) > gpurun_out/r05/concurrent_diag.log 2>&1
grep -v amdgpu.ids gpurun_out/r05/concurrent_diag.log
timeout -s KILL 600 python tools/sampler_regime.py --profile > gpurun_out/r05/sampler_regime_after.log 2>&1; head -c 1200 gpurun_out/r05/sampler_regime_after.log
