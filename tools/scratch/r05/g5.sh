export TMPDIR=/tmp
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_gpu_round5.py tests/test_gpu_sampler.py tests/test_gpu_round2.py tests/test_gpu_driver.py -q -x 2>&1 | tail -4
R="timeout -s KILL 200 python tools/micro/concurrent_handles.py"
( $R 2 4096 50 2; $R 3 4096 50 2; $R 2 4096 50 3; $R 3 4096 50 7 ) > gpurun_out/r05/concurrent_handles.log 2>&1
grep -v amdgpu.ids gpurun_out/r05/concurrent_handles.log
for a in "" "--blocking" "--pool_depth 8" "--pool_depth 12"; do timeout -s KILL 300 python tools/sampler_regime.py $a 2>/dev/null | head -c 700; echo; done | tee gpurun_out/r05/sampler_regime_variants.log
timeout -s KILL 900 python tools/substep_parity.py bench:tracked 1024 4 15 2>&1 | grep -v amdgpu | tee gpurun_out/r05/substep_parity_tracked_flips.log | head -8
timeout -s KILL 900 python tools/substep_parity.py floor 320 2024 45 2>&1 | grep -v amdgpu | tee gpurun_out/r05/substep_parity_floor_flips.log | head -6
