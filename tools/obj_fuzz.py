"""Randomised parity sweep of the free-object path: N scenes (random action class, objects dropped / tilted / overlapping
the humanoid's reach), HIP kernel vs the fp64 oracle after a few control steps.  Prints the error distribution."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kinpoly_amd.model_compiler import read_kpm  # noqa: E402
from kinpoly_amd.sim import STEP_KPM, KpModel, KpSim  # noqa: E402
from oracle.kpo import OracleSim  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
nstep = int(sys.argv[2]) if len(sys.argv) > 2 else 3
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 0
kpm = read_kpm(STEP_KPM)
std = np.load(os.path.join(ROOT, "tests/golden/standing_neutral.npz"))
rng = np.random.default_rng(seed)
x0, y0 = std["qpos"][0], std["qpos"][1]
nominal = {0: [[0.0, -0.45, 0.3805]], 1: [[0.0, 0.55, 0.921], [0.0, 0.55, 0.7905]], 2: [[0.0, 0.45, 0.69]], 3: [[0.0, 0.0, 0.3705]]}   # per action: objects (local x, y, z)
obj_of_action = {0: [0], 1: [1, 2], 2: [3], 3: [4]}


def rquat(scale):
    ax = rng.normal(size=3); ax /= np.linalg.norm(ax)
    a = rng.normal() * scale
    return np.concatenate([[np.cos(a / 2)], np.sin(a / 2) * ax])


blk = np.zeros((n, 35))
for i in range(5):
    blk[:, 7 * i: 7 * i + 3] = [(i + 1) * 100, 100, 0]
qpos = np.tile(std["qpos"], (n, 1)); qvel = rng.normal(size=(n, 75)) * 0.2
scenes = []
for e in range(n):
    a = int(rng.integers(0, 4))
    objs = {}
    shift = rng.normal(size=2) * 0.15
    lift = rng.uniform(0, 0.25) if rng.uniform() < 0.5 else 0.0
    tilt = rquat(0.25 if rng.uniform() < 0.5 else 0.0)
    for oi, (lx, ly, lz) in zip(obj_of_action[a], nominal[a]):
        objs[oi] = [x0 + lx + shift[0], y0 + ly + shift[1], lz + lift + 0.0003, *tilt]   # 0.3 mm: clear of the knife edge dist == margin (the table rests 0.5 mm above its nominal height)
        blk[e, 7 * oi: 7 * oi + 7] = objs[oi]
    if a == 3:
        qpos[e, 2] += 0.341 + lift + 0.02
    qpos[e, 7:] += rng.normal(size=69) * 0.1
    scenes.append(objs)
action = rng.normal(size=(n, 75)) * 0.2
dev = lambda x: torch.tensor(x, dtype=torch.float32, device="cuda")  # noqa: E731
_pm = [float(x) for x in os.environ['KP_PLANEMESH'].split(',')] if 'KP_PLANEMESH' in os.environ else None      # (maxplanemesh, tolplanemesh) on both sides
sim = KpSim(KpModel(STEP_KPM, **({'planemesh_max': _pm[0], 'planemesh_tol': _pm[1]} if _pm else {})), n)
# both sides get the SAME inputs: what the device holds after the fp32 conversion (the oracle used to get the unrounded doubles)
r32 = lambda x: np.asarray(x, np.float32).astype(np.float64)  # noqa: E731
blk, qpos, qvel, action = r32(blk), r32(qpos), r32(qvel), r32(action)
scenes = [{oi: list(blk[e, 7 * oi: 7 * oi + 7]) for oi in sc} for e, sc in enumerate(scenes)]
sim.set_objects(dev(blk)); sim.set_state(dev(qpos), dev(qvel)); sim.set_target(dev(qpos))
a_t = dev(action)
maxc = np.zeros(n, int); its = np.zeros(n, int)
traj_h, traj_o = [], []
for _ in range(nstep):
    sim.step_ctrl(a_t, 15)
    dg = sim.diag(); maxc = np.maximum(maxc, dg[:, 3] & 255); its += dg[:, 1]
    assert dg[:, 2].max() == 0, "non-finite state"
    traj_h.append(sim.get("qpos").double().cpu().numpy()); traj_o.append(sim.get("obj_qpos").double().cpu().numpy())
eh, eo = np.zeros((nstep, n)), np.zeros((nstep, n))
for e in range(n):
    o = OracleSim(kpm=STEP_KPM, planemesh=_pm)
    for slot, oi in enumerate(sorted(scenes[e])):
        o.set_object(slot, kpm, oi, scenes[e][oi])
    o.reset(qpos[e], qvel[e])
    for t in range(nstep):
        o.do_simulation(action[e], qpos[e], 15)
        eh[t, e] = np.abs(o.get("qpos") - traj_h[t][e]).max()
        eo[t, e] = max(np.abs(o.get_object(slot)[0] - traj_o[t][e, 7 * oi: 7 * oi + 7]).max() for slot, oi in enumerate(sorted(scenes[e])))
step1 = np.maximum(eh[0], eo[0])
ehl, eol = eh[-1], eo[-1]
print(f"{n} scenes x {nstep} control steps (seed {seed}): humanoid |dqpos| median {np.median(ehl):.2e} p90 {np.quantile(ehl, .9):.2e} max {ehl.max():.2e}; "
      f"objects median {np.median(eol):.2e} p90 {np.quantile(eol, .9):.2e} max {eol.max():.2e}; contacts max {maxc.max()} mean {maxc.mean():.1f}; "
      f"newton it/substep {its.mean() / 15 / nstep:.2f}; scenes above 1e-4: {int(((ehl > 1e-4) | (eol > 1e-4)).sum())}; "
      f"after the FIRST control step: median {np.median(step1):.2e} max {step1.max():.2e}, above 1e-4: {int((step1 > 1e-4).sum())}, above 1e-3: {int((step1 > 1e-3).sum())}")
worst = np.argsort(-np.maximum(ehl, eol))[:5]
for e in worst:
    print(f"  scene {e}: objects {sorted(scenes[e])} err per control step humanoid {['%.1e' % x for x in eh[:, e]]} object {['%.1e' % x for x in eo[:, e]]} contacts {maxc[e]}")
