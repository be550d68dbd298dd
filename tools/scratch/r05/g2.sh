set -x
export TMPDIR=/tmp
mkdir -p gpurun_out/r05
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r05/t2.log
tail -15 gpurun_out/r05/t2.log
bash tools/update_ablation.sh 2>&1 | tail -12
