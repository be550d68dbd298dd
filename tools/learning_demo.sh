#!/bin/bash
# End-to-end learning check of the whole path on one MI355X (no reference data or weights exist here, so everything starts from random init):
#   1. scripts/train_uhc.py: PPO on the UHC (PolicyMCP) against standing + small-sinusoid clips          -> a UHC that holds and tracks
#   2. scripts/train_ar_policy.py --cc_ckpt <that UHC> --warm_start: supervised warm start of the kinematic policy (AgentAR.train_init, shortened),
#      then dynamics-regulated PPO + supervised step updates on the synthetic takes; and the same WITHOUT the trained UHC / the warm start for contrast.
# The synthetic takes' joint sinusoids are bounded by AMP (default 0.1 rad, the range the four-minute UHC was trained on; at SURVEY's 0.3 rad the fixed pelvis makes the
# legs swing the feet through the floor and no episode outlives ~30 frames, with any controller).
# Prints one line per stage; gpurun_out/learning_demo/*.log hold the per-iteration records.
set -u
export TMPDIR=/tmp
O=gpurun_out/learning_demo; mkdir -p $O
T="timeout -s KILL"
UHC_ITERS=${UHC_ITERS:-300}; AR_ITERS=${AR_ITERS:-40}      # VARIANTS="trained_uhc_warm_start" AR_ITERS=250 for a longer curve of the working configuration
$T 900 python scripts/train_uhc.py --num_envs 4096 --iters $UHC_ITERS --save /tmp/uhc_demo.p 2>&1 | grep '^{' > $O/uhc.log
python - <<PY
import json
r=[json.loads(l) for l in open("$O/uhc.log")]
print("UHC PPO: iter 0 avg_reward %.3f fail_rate %.4f -> iter %d avg_reward %.3f fail_rate %.4f" % (r[0]["avg_reward"], r[0]["fail_rate"], r[-1]["iter"], r[-1]["avg_reward"], r[-1]["fail_rate"]))
PY
for variant in ${VARIANTS:-trained_uhc_warm_start trained_uhc_only random_uhc}; do
  case $variant in
    trained_uhc_warm_start) FLAGS="--cc_ckpt /tmp/uhc_demo.p --warm_start --warm_update_init ${WARM_INIT:-150} --warm_update_full ${WARM_FULL:-12} --num_sample 2000 --batch_size 256";;
    trained_uhc_only) FLAGS="--cc_ckpt /tmp/uhc_demo.p";;
    random_uhc) FLAGS="";;
  esac
  $T 1200 python scripts/train_ar_policy.py --num_envs 4096 --horizon 24 --iters $AR_ITERS --synthetic_amp ${AMP:-0.1} $FLAGS 2>&1 | grep '^{' > $O/ar_$variant.log
  python - <<PY
import json
rows=[json.loads(l) for l in open("$O/ar_$variant.log")]
ws=[r for r in rows if "warm_start" in r]; it=[r for r in rows if "iter" in r]
def s(r): return "avg_reward %.3f fail_rate %.3f eps_len %.1f" % (r["avg_reward"], r["fail_rate"], r["log"]["avg_episode_len"])
print("$variant:", ("warm start losses %s; " % ws[0]["warm_start"]) if ws else "", "iter 0:", s(it[0]), "| iter %d:" % it[-1]["iter"], s(it[-1]))
PY
done
