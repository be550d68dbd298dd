set -x
export TMPDIR=/tmp
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_gpu_round5.py tests/test_gpu_driver.py tests/test_gpu_sampler.py -x -q -s 2>&1 | tail -40 > gpurun_out/r05/t1.log
tail -30 gpurun_out/r05/t1.log
UHC_ITERS=5 AR_ITERS=2 WARM_INIT=2 WARM_FULL=1 bash tools/update_ablation.sh 2>&1 | tail -20
for f in gpurun_out/update_ablation/*.log; do echo $f; tail -c 600 $f; done
mv gpurun_out/update_ablation gpurun_out/update_ablation_smoke
