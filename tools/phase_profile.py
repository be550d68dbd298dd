"""Per-phase shader-clock breakdown of kp_step_kernel (KP_PROFILE=1).  python tools/phase_profile.py"""
import os
import sys

os.environ["KP_PROFILE"] = "1"
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kinpoly_amd.sim import KpModel, KpSim  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
std = np.load(os.path.join(ROOT, "tests/golden/standing_neutral.npz"))
for contact, lift, n in ((0, 10.0, 4096), (1, 0.0, 4096), (1, 0.0, 256)):
    rng = np.random.default_rng(3)
    qpos = np.tile(std["qpos"], (n, 1)); qpos[:, 2] += lift; qpos[:, 7:] += rng.normal(size=(n, 69)) * 0.2
    qvel = rng.normal(size=(n, 75)) * 0.5
    sim = KpSim(KpModel(contact=contact), n)
    q = torch.tensor(qpos, dtype=torch.float32, device="cuda"); v = torch.tensor(qvel, dtype=torch.float32, device="cuda")
    sim.set_state(q, v); sim.set_target(q.clone())
    a = torch.zeros((n, 75), device="cuda")
    for _ in range(3):
        sim.step_ctrl(a, 15)
    pc = sim.phase_cycles()
    ms = sim.last_step_seconds() * 1e3
    print(f"contact={contact} n={n}: launch {ms:.3f} ms; cycles/env/control-step:", {k: int(v) for k, v in pc.items()},
          " per substep:", {k: int(v / 15) for k, v in pc.items()}, flush=True)
