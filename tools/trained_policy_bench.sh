#!/bin/bash
# A headline with nothing overridden: train a UHC and a kinematic policy with this engine (tools/learning_demo.sh's working configuration), then run
# bench.py on THAT checkpoint -- the policy's own sampled output drives env.step -- next to the `tracked` stand-in, and the whole-episode parity on the same networks.
#   usage (on the GPU box, from the repo root): tools/trained_policy_bench.sh [out_dir]
set -u
export TMPDIR=/tmp
O=${1:-gpurun_out/r06/trained}; mkdir -p $O
CK=${KP_CKPT_DIR:-/tmp/kp_trained}; mkdir -p $CK      # the checkpoints (2 x 45 MB) stay off gpurun_out/ (64 MiB are copied back)
T="timeout -s KILL"
UHC_ITERS=${UHC_ITERS:-300}; AR_ITERS=${AR_ITERS:-40}
$T 900 python scripts/train_uhc.py --num_envs 4096 --iters $UHC_ITERS --save $CK/uhc.p 2>&1 | grep '^{' > $O/uhc.log
$T 1500 python scripts/train_ar_policy.py --num_envs 4096 --horizon 24 --iters $AR_ITERS --synthetic_amp ${AMP:-0.1} --cc_ckpt $CK/uhc.p --warm_start --warm_update_init ${WARM_INIT:-150} --warm_update_full ${WARM_FULL:-12} \
   --num_sample 2000 --batch_size 256 --save $CK/ar.p 2>&1 | grep '^{' > $O/ar.log
python - <<PY
import json
r=[json.loads(l) for l in open("$O/uhc.log")]
print("UHC PPO: iter 0 avg_reward %.3f fail_rate %.4f -> iter %d avg_reward %.3f fail_rate %.4f" % (r[0]["avg_reward"], r[0]["fail_rate"], r[-1]["iter"], r[-1]["avg_reward"], r[-1]["fail_rate"]))
rows=[json.loads(l) for l in open("$O/ar.log")]; it=[x for x in rows if "iter" in x]
print("kinematic policy: iter 0 avg_reward %.3f fail_rate %.4f -> iter %d avg_reward %.3f fail_rate %.4f" % (it[0]["avg_reward"], it[0]["fail_rate"], it[-1]["iter"], it[-1]["avg_reward"], it[-1]["fail_rate"]))
PY
for wl in trained tracked; do
  $T 300 python bench.py --workload $wl --policy-ckpt $CK/ar.p --cc-ckpt $CK/uhc.p --no-secondary --no-cpu-baseline > $O/bench_$wl.json 2> $O/bench_$wl.err
done
python - <<PY
import json
out={}
for wl in ("trained","tracked"):
    d=json.loads(open("$O/bench_%s.json" % wl).read().strip().splitlines()[-1])
    out[wl]={k: d[k] for k in ("value","ms_per_step","ms_per_step_min","ms_per_step_max","contacts_mean","contacts_max_in_a_substep","newton_iters_per_substep","episodes_ended_per_step_frac","bad_envs")}
    out[wl]["launch_ms"]=d["roofline"]["launch_ms"]; out[wl]["workload"]=d["config"]["workload"]; out[wl]["parity_live"]=d.get("parity_live")
out["trained_over_tracked"]=out["trained"]["value"]/out["tracked"]["value"]
json.dump(out, open("$O/bench_trained_policy.json","w"), indent=1)
print(json.dumps({k:(v if not isinstance(v,dict) else {kk:vv for kk,vv in v.items() if kk not in ("workload","parity_live")}) for k,v in out.items()}))
PY
$T 900 python tools/episode_parity.py --envs 128 --steps 99 --policy-ckpt $CK/ar.p --cc-ckpt $CK/uhc.p --json $O/episode_parity_trained.json > $O/episode_parity_trained.log 2>&1
python -c "
import json; d=json.load(open('$O/episode_parity_trained.json'))
print({k:d[k] for k in ('first_termination_step_equal_frac','done_flags_equal_frac_of_rows','episodes_ended','failures','mean_reward','dqpos_aligned_rows')})"
