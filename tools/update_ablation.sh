#!/bin/bash
# which term of the update drives `surr_loss > 0` and the reward drop of the warm-started policy?
# One UHC, one warm start (saved), then {fp32, fp64} x {PPO only, step update only, both} from that SAME checkpoint, AR_ITERS iterations each at
# 4096 x 24, kin_poly.yml's rates.  gpurun_out/update_ablation/*.log hold the per-iteration records; tools/update_ablation_table.py prints the table.
set -u
export TMPDIR=/tmp
O=gpurun_out/update_ablation; mkdir -p $O
T="timeout -s KILL"
UHC_ITERS=${UHC_ITERS:-300}; AR_ITERS=${AR_ITERS:-40}; AMP=${AMP:-0.1}
if [ ! -f /tmp/uhc_demo.p ]; then
  $T 900 python scripts/train_uhc.py --num_envs 4096 --iters $UHC_ITERS --save /tmp/uhc_demo.p 2>&1 | grep '^{' > $O/uhc.log
fi
$T 1200 python scripts/train_ar_policy.py --num_envs 4096 --horizon 24 --iters 0 --synthetic_amp $AMP --cc_ckpt /tmp/uhc_demo.p --warm_start --warm_update_init ${WARM_INIT:-150} \
   --warm_update_full ${WARM_FULL:-12} --num_sample 2000 --batch_size 256 --save /tmp/ar_warm.p 2>&1 | grep '^{' > $O/warm_start.log
for dt in ${DTYPES:-fp32 fp64}; do
  for v in ${VARIANTS:-none ppo step both}; do
    case $v in none) F="--rl_update 0 --step_update 0";; ppo) F="--rl_update 1 --step_update 0";; step) F="--rl_update 0 --step_update 1";; both) F="--rl_update 1 --step_update 1";; esac
    $T 1500 python scripts/train_ar_policy.py --num_envs 4096 --horizon 24 --iters $AR_ITERS --synthetic_amp $AMP --cc_ckpt /tmp/uhc_demo.p --load /tmp/ar_warm.p \
       --update_dtype $dt $F --eval_first_last ${EXTRA:-} 2>&1 | grep '^{' > $O/${dt}_${v}${TAG:-}.log
    echo "$dt $v done: $(wc -l < $O/${dt}_${v}${TAG:-}.log) records"
  done
done
python tools/update_ablation_table.py $O | tee $O/table.txt
