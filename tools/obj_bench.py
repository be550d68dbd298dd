"""Launch time of the object variant of the physics kernel (kp_step_kernel<64, true>): 4096 envs, (a) standing on the dynamic
step box, (b) the push scene (box on table next to the humanoid), (c) same poses with the objects parked (floor-only kernel)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kinpoly_amd.sim import STEP_KPM, KpModel, KpSim  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
std = np.load(os.path.join(ROOT, "tests/golden/standing_neutral.npz"))
n = 4096
rng = np.random.default_rng(3)
x0, y0 = std["qpos"][0], std["qpos"][1]


def block(active):
    blk = np.zeros((n, 35))
    for i in range(5):
        blk[:, 7 * i: 7 * i + 3] = [(i + 1) * 100, 100, 0]
    for oi, pose in active.items():
        blk[:, 7 * oi: 7 * oi + 7] = pose
    return blk


for name, active, lift in (("standing on the step box", {4: [x0, y0, 0.3705, 1, 0, 0, 0]}, 0.341),
                           ("push scene (box on table, 0.45 m ahead)", {1: [x0 + 0.75, y0, 0.921, 1, 0, 0, 0], 2: [x0 + 0.75, y0, 0.7905, 1, 0, 0, 0]}, 0.0),
                           ("push scene 2 m away (box on table, far from the humanoid)", {1: [x0 + 2.5, y0, 0.921, 1, 0, 0, 0], 2: [x0 + 2.5, y0, 0.7905, 1, 0, 0, 0]}, 0.0),
                           ("table alone 2 m away (1 object, 16 floor contacts)", {2: [x0 + 2.0, y0, 0.7905, 1, 0, 0, 0]}, 0.0),
                           ("box + step on the floor 2 m away (2 objects, 8 floor contacts)", {1: [x0 + 2.0, y0 + 1.0, 0.2205, 1, 0, 0, 0], 4: [x0 + 2.0, y0, 0.3705, 1, 0, 0, 0]}, 0.0),
                           ("step box 2 m away (dynamic, untouched)", {4: [x0 + 2.0, y0, 0.3705, 1, 0, 0, 0]}, 0.0),
                           ("step box 2 m away, dynamic_objects=0 (static obstacle)", {4: [x0 + 2.0, y0, 0.3705, 1, 0, 0, 0]}, -1.0),
                           ("no object (floor-only kernel)", {}, 0.0)):
    model = KpModel(STEP_KPM)
    if lift < 0:
        model.set_option("dynamic_objects", 0); lift = 0.0
    sim = KpSim(model, n)
    qpos = np.tile(std["qpos"], (n, 1)); qpos[:, 2] += lift; qpos[:, 7:] += rng.normal(size=(n, 69)) * 0.05
    qvel = rng.normal(size=(n, 75)) * 0.2
    q = torch.tensor(qpos, dtype=torch.float32, device="cuda"); v = torch.tensor(qvel, dtype=torch.float32, device="cuda")
    if active:
        sim.set_objects(torch.tensor(block(active), dtype=torch.float32, device="cuda"))
    sim.set_state(q, v); sim.set_target(q.clone())
    a = torch.tensor(rng.normal(size=(n, 75)) * 0.1, dtype=torch.float32, device="cuda")
    ts = []
    for _ in range(8):
        sim.step_ctrl(a, 15)
        ts.append(sim.last_step_seconds())
    dg = sim.diag()
    if os.environ.get("KP_PROFILE") == "1":
        print("   cycles per substep:", {k: int(v / 15) for k, v in sim.phase_cycles().items()})
    print(f"{name}: launch {np.mean(ts[2:]) * 1e3:.3f} ms ({n / np.mean(ts[2:]):.0f} env-steps/s physics only), contacts {dg[:, 0].mean():.1f}, "
          f"newton it/substep {dg[:, 1].mean() / 15:.2f}, factorisations/substep {(dg[:, 3] >> 8).mean() / 15:.2f}, bad {int((dg[:, 2] != 0).sum())}", flush=True)
