export TMPDIR=/tmp
mkdir -p gpurun_out/r05
T="timeout -s KILL"
$T 1200 python -m pytest tests -m gpu -q 2>&1 | tail -4
$T 300 python tools/floor_fuzz.py 320 2>&1 | grep -v amdgpu | tail -4
( for s in 0 1 2; do $T 200 python tools/obj_fuzz.py 64 3 $s; done ) 2>&1 | grep "scenes x\|above" | head -8
( for s in 0 1; do $T 200 python tools/contact_compare.py 64 $s; done ) 2>&1 | grep -v amdgpu | tail -4
KP_DEBUG_RESET_PPO_MOMENTUM=1 DTYPES=fp32 VARIANTS=both TAG=_momentum_reset bash tools/update_ablation.sh 2>&1 | tail -12
