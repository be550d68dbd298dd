export TMPDIR=/tmp
python - <<PY
import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
from kinpoly_amd import dataset as D, sim as kpsim
from kinpoly_amd.model_compiler import read_kpm
std = np.load("tests/golden/standing_neutral.npz")
fk = kpsim.KpSim(kpsim.KpModel(kpsim.STEP_KPM), 8, 0)
takes = D.synthetic_takes(fk, std["qpos"], n_per_action=1, T_range=(30, 50), body_mass=read_kpm(kpsim.STEP_KPM)["body_mass"], seed=2, amp_max=0.1)
D.write_features("/tmp/feats.p", takes)
print("features written", sorted(takes))
PY
timeout -s KILL 200 python scripts/eval_ar_policy.py --data /tmp/feats.p --num_seq 3 --clip_len 20 --result_dir /tmp/ev1 --metrics 2>&1 | grep -v amdgpu | tail -3
timeout -s KILL 200 python scripts/eval_ar_policy.py --data /tmp/feats.p --num_seq 4 --clip_len 20 --result_dir /tmp/ev2 --fail_safe --wild 2>&1 | grep -v amdgpu | tail -2
ls /tmp/ev1 /tmp/ev2
