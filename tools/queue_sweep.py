"""Launch time of the control-step kernel against the job size of kp_step_queue_kernel (4096 envs, standing + contact workload of
tools/pmc_step.py).    python tools/queue_sweep.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kinpoly_amd.sim import KpModel, KpSim  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
std = np.load(os.path.join(ROOT, "tests/golden/standing_neutral.npz"))
n = 4096
for spj in (0, 8, 5, 4, 3, 2, 1):
    rng = np.random.default_rng(3)
    qpos = np.tile(std["qpos"], (n, 1)); qpos[:, 7:] += rng.normal(size=(n, 69)) * 0.2
    qvel = rng.normal(size=(n, 75)) * 0.5
    sim = KpSim(KpModel(substeps_per_job=spj), n)
    q = torch.tensor(qpos, dtype=torch.float32, device="cuda"); v = torch.tensor(qvel, dtype=torch.float32, device="cuda")
    sim.set_state(q, v); sim.set_target(q.clone())
    a = torch.tensor(rng.normal(size=(n, 75)) * 0.2, dtype=torch.float32, device="cuda")
    ms = []
    for it in range(12):
        sim.step_ctrl(a, 15)
        ms.append(sim.last_step_seconds() * 1e3)
    c = sim.launch_cost().astype(np.float64)
    print(f"substeps_per_job={spj}: launch ms median {np.median(ms[2:]):.3f} (min {min(ms[2:]):.3f} max {max(ms[2:]):.3f}); "
          f"sum of env cycles / 2048 slots = {c.sum() / 2048 / 2.38e6:.3f} ms", flush=True)
