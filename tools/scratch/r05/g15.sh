export TMPDIR=/tmp
O=gpurun_out/update_ablation_repeat; mkdir -p $O
T="timeout -s KILL"
$T 900 python scripts/train_uhc.py --num_envs 4096 --iters 300 --save /tmp/uhc_demo.p 2>&1 | grep '^{' > $O/uhc.log
$T 1200 python scripts/train_ar_policy.py --num_envs 4096 --horizon 24 --iters 0 --synthetic_amp 0.1 --cc_ckpt /tmp/uhc_demo.p --warm_start --warm_update_init 150 --warm_update_full 12 --num_sample 2000 --batch_size 256 --save /tmp/ar_warm.p 2>&1 | grep '^{' > $O/warm_start.log
for rep in a b; do for v in step both ppo; do
  case $v in ppo) F="--rl_update 1 --step_update 0";; step) F="--rl_update 0 --step_update 1";; both) F="--rl_update 1 --step_update 1";; esac
  $T 1500 python scripts/train_ar_policy.py --num_envs 4096 --horizon 24 --iters 40 --synthetic_amp 0.1 --cc_ckpt /tmp/uhc_demo.p --load /tmp/ar_warm.p $F --eval_first_last --eval_every 10 2>&1 | grep '^{' > $O/fp32_${v}_$rep.log
  echo "$v $rep:"; grep fixed_eval $O/fp32_${v}_$rep.log | cut -c1-200
done; done
