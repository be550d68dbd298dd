set -x
export TMPDIR=/tmp
mkdir -p gpurun_out/r05
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -15 > gpurun_out/r05/t3.log
tail -6 gpurun_out/r05/t3.log
timeout -s KILL 900 python bench.py --steps 60 --warmup 20 > gpurun_out/r05/bench_default_a.json 2> gpurun_out/r05/bench_default_a.err; tail -c 600 gpurun_out/r05/bench_default_a.err
timeout -s KILL 600 python tools/sampler_regime.py --profile > gpurun_out/r05/sampler_regime_before.log 2>&1; head -c 1500 gpurun_out/r05/sampler_regime_before.log
DTYPES=fp32 bash tools/update_ablation.sh 2>&1 | tail -8
DTYPES=fp64 VARIANTS=both bash tools/update_ablation.sh 2>&1 | tail -8
for K in 2 3; do timeout -s KILL 150 python tools/micro/concurrent_handles.py $K 4096 50 2 > gpurun_out/r05/concurrent_K$K.log 2>&1; echo "K=$K rc=$?"; tail -3 gpurun_out/r05/concurrent_K$K.log; done
GPU_MAX_HW_QUEUES=8 timeout -s KILL 150 python tools/micro/concurrent_handles.py 3 4096 50 2 > gpurun_out/r05/concurrent_K3_hwq8.log 2>&1; echo "K=3 hwq8 rc=$?"; tail -3 gpurun_out/r05/concurrent_K3_hwq8.log
timeout -s KILL 150 python tools/micro/concurrent_handles.py 3 4096 50 0 > gpurun_out/r05/concurrent_K3_floor.log 2>&1; echo "K=3 floor rc=$?"; tail -3 gpurun_out/r05/concurrent_K3_floor.log
