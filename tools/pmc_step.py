"""Workload for `rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE`: 4096 envs, standing with contact, 10 control-step launches of
kp_step_kernel (plus the one forward-only launch of set_state, which the median in summarize_pmc.py ignores)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kinpoly_amd.sim import KpModel, KpSim  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
std = np.load(os.path.join(ROOT, "tests/golden/standing_neutral.npz"))
n = 4096
rng = np.random.default_rng(3)
qpos = np.tile(std["qpos"], (n, 1)); qpos[:, 7:] += rng.normal(size=(n, 69)) * 0.2
qvel = rng.normal(size=(n, 75)) * 0.5
sim = KpSim(KpModel(), n)
q = torch.tensor(qpos, dtype=torch.float32, device="cuda"); v = torch.tensor(qvel, dtype=torch.float32, device="cuda")
sim.set_state(q, v); sim.set_target(q.clone())
a = torch.tensor(rng.normal(size=(n, 75)) * 0.2, dtype=torch.float32, device="cuda")
ms = []
for _ in range(10):
    sim.step_ctrl(a, 15)
    ms.append(sim.last_step_seconds() * 1e3)
torch.cuda.synchronize()
print("ms/launch last", ms[-1], "median", float(np.median(ms[2:])))
