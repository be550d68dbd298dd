"""How many Newton iterations per substep does the fp32 kernel spend vs the fp64 oracle on the SAME states / actions?
States are taken from a bench-like rollout (random-init policies, falling humanoids)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from kinpoly_amd.sim import KpModel, KpSim  # noqa: E402
from oracle.kpo import OracleSim  # noqa: E402

bench.ENVS_PER_GPU = 256
env, policy, sampler, std = bench.build_engine(0, 4, 64)
bench.rollout_steps(sampler, 12)
sim = env.sim
q, v = sim.get("qpos").clone(), sim.get("qvel").clone()
qd, vd = sim.get("qpos_d").clone(), sim.get("qvel_d").clone()
tq = sim.get("target_qpos").clone()
n = 16
rng = np.random.default_rng(0)
act = torch.tensor(rng.normal(size=(256, 75)) * 0.3, dtype=torch.float32, device="cuda")
s2 = KpSim(KpModel(), 256)
s2.set_full_state(q, v, qd, vd); s2.set_target(tq)
s2.step_ctrl(act, 15)
dg = s2.diag()
hip_it = dg[:n, 1]
ora_it = []
for e in range(n):
    o = OracleSim()
    o.reset(qd[e].double().cpu().numpy(), vd[e].double().cpu().numpy())          # derived quantities of the stale state
    o.set_state_raw(q[e].double().cpu().numpy(), v[e].double().cpu().numpy())
    tot = 0
    for _ in range(15):
        o.do_simulation(act[e].double().cpu().numpy(), tq[e].double().cpu().numpy(), 1)
        tot += o.niter
    ora_it.append(tot)
    err = np.abs(o.get("qpos") - s2.get("qpos")[e].double().cpu().numpy()).max()
    print(e, "hip iters", int(hip_it[e]), "oracle iters", tot, "contacts", int(dg[e, 0]), "|dqpos| %.2e" % err)
print("mean per substep: hip %.2f oracle %.2f" % (hip_it.mean() / 15, np.mean(ora_it) / 15))
