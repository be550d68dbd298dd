"""Contact-level parity of the narrow phases: the contact set (entities, dist, position, normal) the HIP kernel builds for a state vs
the fp64 oracle's, on the randomised object scenes of tools/obj_fuzz.py (same generator).    python tools/contact_compare.py [n] [seed]"""
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
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
kpm = read_kpm(STEP_KPM)
std = np.load(os.path.join(ROOT, "tests/golden/standing_neutral.npz"))
rng = np.random.default_rng(seed)
x0, y0 = std["qpos"][0], std["qpos"][1]
nominal = {0: [[0.0, -0.45, 0.3805]], 1: [[0.0, 0.55, 0.921], [0.0, 0.55, 0.7905]], 2: [[0.0, 0.45, 0.69]], 3: [[0.0, 0.0, 0.3705]]}
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
        objs[oi] = [x0 + lx + shift[0], y0 + ly + shift[1], lz + lift + 0.0003, *tilt]
        blk[e, 7 * oi: 7 * oi + 7] = objs[oi]
    if a == 3:
        qpos[e, 2] += 0.341 + lift + 0.02
    qpos[e, 7:] += rng.normal(size=69) * 0.1
    scenes.append(objs)
dev = lambda x: torch.tensor(x, dtype=torch.float32, device="cuda")  # noqa: E731
sim = KpSim(KpModel(STEP_KPM), n)
sim.record_contacts()
sim.set_objects(dev(blk)); sim.set_state(dev(qpos), dev(qvel)); sim.set_target(dev(qpos))
sim.step_ctrl(dev(np.zeros((n, 75))), 1)
hip = sim.contacts()
q32, v32, b32 = dev(qpos).double().cpu().numpy(), dev(qvel).double().cpu().numpy(), dev(blk).double().cpu().numpy()
bad = 0
worst = dict(dist=0.0, pos=0.0, normal=0.0)
for e in range(n):
    o = OracleSim(kpm=STEP_KPM)
    for slot, oi in enumerate(sorted(scenes[e])):
        o.set_object(slot, kpm, oi, b32[e, 7 * oi:7 * oi + 7])
    o.reset(q32[e], v32[e])
    c = o.contacts_full(); h = hip[e]
    same = len(c["body"]) == len(h["body"]) and np.array_equal(c["body"], h["body"]) and np.array_equal(c["b2"], h["b2"])
    if not same:
        bad += 1
        print(f"scene {e} objects {sorted(scenes[e])}: entity lists differ\n   oracle {list(zip(c['body'], c['b2']))}\n   hip    {list(zip(h['body'], h['b2']))}")
        continue
    if len(c["body"]) == 0:
        continue
    # inside one geom pair the ORDER of the contacts carries no meaning (box - box clipping may start its polygon at another vertex in fp32):
    # sort each run of equal (entity, entity) by position before comparing
    def canon(x):
        key = np.lexsort((np.round(x["pos"][:, 2], 4), np.round(x["pos"][:, 1], 4), np.round(x["pos"][:, 0], 4), x["b2"], x["body"]))
        return {k: v[key] for k, v in x.items()}
    c, h = canon(c), canon(h)
    dd, dp, dn = np.abs(c["dist"] - h["dist"]), np.abs(c["pos"] - h["pos"]).max(1), np.abs(c["normal"] - h["normal"]).max(1)
    worst = dict(dist=max(worst["dist"], dd.max()), pos=max(worst["pos"], dp.max()), normal=max(worst["normal"], dn.max()))
    if dd.max() > 1e-5 or dp.max() > 1e-4 or dn.max() > 1e-4:
        bad += 1
        k = int(np.argmax(np.maximum(dd * 10, np.maximum(dp, dn))))
        print(f"scene {e} objects {sorted(scenes[e])}: contact {k} ({c['body'][k]}, {c['b2'][k]}) dist {c['dist'][k]:.6f} / {h['dist'][k]:.6f}  "
              f"pos {np.round(c['pos'][k], 5)} / {np.round(h['pos'][k], 5)}  normal {np.round(c['normal'][k], 5)} / {np.round(h['normal'][k], 5)}")
print(f"{n} scenes (seed {seed}): {bad} with a contact-set difference; worst |d dist| {worst['dist']:.2e} |d pos| {worst['pos']:.2e} |d normal| {worst['normal']:.2e} over the matching ones")
