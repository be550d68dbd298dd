#!/bin/bash
# Which takes do the episodes of the learning demo's working configuration fail on?  UHC PPO, warm start, a few AR iterations with a result_dir,
# then freq_dict.pt ([percent, fr_start] per finished episode and take) summarised per take.
set -u
export TMPDIR=/tmp
T="timeout -s KILL"
$T 900 python scripts/train_uhc.py --num_envs 4096 --iters ${UHC_ITERS:-300} --save /tmp/uhc_demo.p > /dev/null 2>&1
$T 900 python scripts/train_ar_policy.py --num_envs 4096 --horizon 24 --iters ${AR_ITERS:-20} --cc_ckpt /tmp/uhc_demo.p --warm_start --warm_update_init 150 --warm_update_full 12 \
   --num_sample 2000 --batch_size 256 --result_dir /tmp/fail_by_take > /dev/null 2>&1
python - <<PY
import joblib, numpy as np
fd = joblib.load("/tmp/fail_by_take/freq_dict.pt")
print("take, episodes kept, share that reached the clip's end, mean percent, mean first frame of the failed ones")
for k, v in sorted(fd.items()):
    a = np.asarray(v, float)
    if len(a) == 0:
        print(k, 0); continue
    ok = a[:, 0] == 1
    print(f"{k:28s} {len(a):6d}  {ok.mean():.3f}  {a[:,0].mean():.3f}  {a[~ok,1].mean() if (~ok).any() else float('nan'):.1f}")
PY
