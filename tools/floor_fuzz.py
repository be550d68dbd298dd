"""Randomised parity sweep of the floor / joint-limit path: HIP vs oracle on wild states -- any root orientation, heights 0.15 .. 1.6 m,
joint angles up to the +-pi wrap, fast tumbling -- for 3 control steps (45 substeps) each.    python tools/floor_fuzz.py [n_scenes]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kinpoly_amd.sim import KpModel, KpSim  # noqa: E402
from oracle.kpo import OracleSim  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STD = np.load(os.path.join(ROOT, "tests/golden/standing_neutral.npz"))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
rng = np.random.default_rng(2024)
qpos = np.tile(STD["qpos"], (n, 1)); qvel = np.zeros((n, 75)); act = np.zeros((n, 75)); tgt = np.tile(STD["qpos"], (n, 1))
for e in range(n):
    kind = e % 5
    q = rng.normal(size=4); q /= np.linalg.norm(q)
    if kind == 0:      # upright-ish, hard landing
        qpos[e, 2] += rng.uniform(0.0, 0.6); qpos[e, 7:] += rng.normal(size=69) * 0.3
        qvel[e] = rng.normal(size=75) * 1.0; qvel[e, 2] -= rng.uniform(0, 4)
    elif kind == 1:    # any orientation near the floor (lying / head first), many contacts
        qpos[e, 3:7] = q; qpos[e, 2] = rng.uniform(0.15, 0.5); qpos[e, 7:] += rng.normal(size=69) * 0.5
        qvel[e] = rng.normal(size=75) * 1.5
    elif kind == 2:    # tumbling in the air, joints at the limits
        qpos[e, 3:7] = q; qpos[e, 2] = rng.uniform(0.8, 1.6); qpos[e, 7:] = rng.uniform(-3.1, 3.1, size=69)
        qvel[e] = rng.normal(size=75) * 3.0
    elif kind == 4:    # half-buried start (what a random-init context network predicts): dozens of deep contacts
        qpos[e, 3:7] = q if rng.uniform() < 0.5 else qpos[e, 3:7]
        qpos[e, 2] = rng.uniform(-0.1, 0.35); qpos[e, 7:] += rng.normal(size=69) * 0.3
        qvel[e] = rng.normal(size=75) * 0.5
    else:              # standing, violent controller actions and a far target
        qpos[e, 7:] += rng.normal(size=69) * 0.1; qvel[e] = rng.normal(size=75) * 0.3
        tgt[e, 7:] += rng.normal(size=69) * 1.0
    act[e] = rng.normal(size=75) * (1.0 if kind == 3 else 0.3)
dev = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device="cuda")  # noqa: E731
sim = KpSim(KpModel(), n)
sim.set_state(dev(qpos), dev(qvel)); sim.set_target(dev(tgt))
a = dev(act)
q32, v32, t32, a32 = (x.double().cpu().numpy() for x in (dev(qpos), dev(qvel), dev(tgt), a))
steps = 3
for _ in range(steps):
    sim.step_ctrl(a, 15)
got = sim.get("qpos").double().cpu().numpy(); gotv = sim.get("qvel").double().cpu().numpy()
dg = sim.diag()
err = np.zeros(n); errv = np.zeros(n); ncon = np.zeros(n)
o = OracleSim()
for e in range(n):
    o.reset(q32[e], v32[e])
    for _ in range(steps):
        o.do_simulation(a32[e], t32[e], 15)
    want = o.get("qpos"); wantv = o.get("qvel")
    err[e] = np.abs(got[e] - want).max(); errv[e] = np.abs(gotv[e] - wantv).max() / max(1.0, np.abs(wantv).max())
for kind, name in enumerate(("hard landing", "lying / any orientation", "tumbling, joints at limits", "standing, violent control", "half-buried start")):
    m = np.arange(n) % 5 == kind
    print(f"{name:28s}: |dqpos| median {np.median(err[m]):.1e} p90 {np.percentile(err[m], 90):.1e} max {err[m].max():.1e}; rel |dqvel| max {errv[m].max():.1e}; "
          f"contacts max {int(dg[m, 3].max() & 255)}, newton it/substep {dg[m, 1].mean() / 15 / 1:.2f}, flagged {int((dg[m, 2] != 0).sum())}")
print(f"all {n} scenes x {steps} control steps: worst |dqpos| {err.max():.2e} (scene {int(err.argmax())}), above 1e-4: {int((err > 1e-4).sum())}, above 1e-3: {int((err > 1e-3).sum())}")
