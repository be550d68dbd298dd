"""Per-env cost of the control-step launch inside the training sampler, grouped by the take's action class (sit / push / avoid / step):
shader-clock cycles (kp_sim_launch_cost), contacts, Newton iterations and Hessian factorisations per substep."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kinpoly_amd import dataset as D  # noqa: E402
from kinpoly_amd import sim as kpsim  # noqa: E402
from kinpoly_amd.agent import AgentAR  # noqa: E402
from kinpoly_amd.model_compiler import read_kpm  # noqa: E402

n = 4096
std = np.load(os.path.join(ROOT, "tests", "golden", "standing_neutral.npz"))
fk_sim = kpsim.KpSim(kpsim.KpModel(kpsim.STEP_KPM), n, 0)
takes = D.synthetic_takes(fk_sim, std["qpos"], n_per_action=4, T_range=(42, 92), body_mass=read_kpm(kpsim.STEP_KPM)["body_mass"], seed=4)
ds = D.StateARDataset(takes, fr_num=32, seed=4, device=fk_sim.device)
agent = AgentAR(n, lambda k: ds.sample_batch(k), device=0, horizon=24, num_optim_epoch=0, num_step_update=0)
agent.optimize_policy(0)
env = agent.env
cost = env.sim.launch_cost().astype(np.float64); dg = env.sim.diag()
cls = env.ctx["action_one_hot"].argmax(1).cpu().numpy()
has = (env.ctx["action_one_hot"].sum(1) > 0).cpu().numpy()
print(f"last launch: {env.sim.last_step_seconds() * 1e3:.2f} ms; sum of env cycles / 1536 slots = {cost.sum() / 1536 / 2.38e6:.2f} ms")
for a, name in enumerate(("sit", "push", "avoid", "step")):
    m = (cls == a) & has
    if m.any():
        print(f"{name:6s}: {int(m.sum()):5d} envs, cycles mean {cost[m].mean() / 1e6:.2f} M (max {cost[m].max() / 1e6:.2f} M), contacts {dg[m, 0].mean():.1f}, "
              f"newton it/substep {dg[m, 1].mean() / 15:.2f}, factorisations/substep {(dg[m, 3] >> 8).mean() / 15:.2f}, non-finite {int((dg[m, 2] != 0).sum())}")
