"""Scene generators shared by the parity tools (the same draws, in the same order, as tools/floor_fuzz.py and tools/obj_fuzz.py)."""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STD = np.load(os.path.join(ROOT, "tests/golden/standing_neutral.npz"))
r32 = lambda x: np.asarray(x, np.float32).astype(np.float64)  # noqa: E731
FLOOR_KINDS = ("hard landing", "lying / any orientation", "tumbling, joints at limits", "standing, violent control", "half-buried start")


def floor_scenes(n, seed=2024):
    """tools/floor_fuzz.py: wild humanoid states on the floor.  dict(qpos, qvel, action, target, kind) already rounded to fp32 values."""
    rng = np.random.default_rng(seed)
    qpos = np.tile(STD["qpos"], (n, 1)); qvel = np.zeros((n, 75)); act = np.zeros((n, 75)); tgt = np.tile(STD["qpos"], (n, 1))
    for e in range(n):
        kind = e % 5
        q = rng.normal(size=4); q /= np.linalg.norm(q)
        if kind == 0:
            qpos[e, 2] += rng.uniform(0.0, 0.6); qpos[e, 7:] += rng.normal(size=69) * 0.3
            qvel[e] = rng.normal(size=75) * 1.0; qvel[e, 2] -= rng.uniform(0, 4)
        elif kind == 1:
            qpos[e, 3:7] = q; qpos[e, 2] = rng.uniform(0.15, 0.5); qpos[e, 7:] += rng.normal(size=69) * 0.5
            qvel[e] = rng.normal(size=75) * 1.5
        elif kind == 2:
            qpos[e, 3:7] = q; qpos[e, 2] = rng.uniform(0.8, 1.6); qpos[e, 7:] = rng.uniform(-3.1, 3.1, size=69)
            qvel[e] = rng.normal(size=75) * 3.0
        elif kind == 4:
            qpos[e, 3:7] = q if rng.uniform() < 0.5 else qpos[e, 3:7]
            qpos[e, 2] = rng.uniform(-0.1, 0.35); qpos[e, 7:] += rng.normal(size=69) * 0.3
            qvel[e] = rng.normal(size=75) * 0.5
        else:
            qpos[e, 7:] += rng.normal(size=69) * 0.1; qvel[e] = rng.normal(size=75) * 0.3
            tgt[e, 7:] += rng.normal(size=69) * 1.0
        act[e] = rng.normal(size=75) * (1.0 if kind == 3 else 0.3)
    return dict(qpos=r32(qpos), qvel=r32(qvel), action=r32(act), target=r32(tgt), kind=np.arange(n) % 5, objects=[{} for _ in range(n)], blk=None)


NOMINAL = {0: [[0.0, -0.45, 0.3805]], 1: [[0.0, 0.55, 0.921], [0.0, 0.55, 0.7905]], 2: [[0.0, 0.45, 0.69]], 3: [[0.0, 0.0, 0.3705]]}   # per action: objects (local x, y, z)
OBJ_OF_ACTION = {0: [0], 1: [1, 2], 2: [3], 3: [4]}


def object_scenes(n, seed=0):
    """tools/obj_fuzz.py: random action class, objects dropped / tilted / overlapping the humanoid's reach.  objects[e] = {object index: qpos7}."""
    rng = np.random.default_rng(seed)
    x0, y0 = STD["qpos"][0], STD["qpos"][1]

    def rquat(scale):
        ax = rng.normal(size=3); ax /= np.linalg.norm(ax)
        a = rng.normal() * scale
        return np.concatenate([[np.cos(a / 2)], np.sin(a / 2) * ax])

    blk = np.zeros((n, 35))
    for i in range(5):
        blk[:, 7 * i: 7 * i + 3] = [(i + 1) * 100, 100, 0]
    qpos = np.tile(STD["qpos"], (n, 1)); qvel = rng.normal(size=(n, 75)) * 0.2
    scenes, kinds = [], []
    for e in range(n):
        a = int(rng.integers(0, 4))
        objs = {}
        shift = rng.normal(size=2) * 0.15
        lift = rng.uniform(0, 0.25) if rng.uniform() < 0.5 else 0.0
        tilt = rquat(0.25 if rng.uniform() < 0.5 else 0.0)
        for oi, (lx, ly, lz) in zip(OBJ_OF_ACTION[a], NOMINAL[a]):
            objs[oi] = [x0 + lx + shift[0], y0 + ly + shift[1], lz + lift + 0.0003, *tilt]
            blk[e, 7 * oi: 7 * oi + 7] = objs[oi]
        if a == 3:
            qpos[e, 2] += 0.341 + lift + 0.02
        qpos[e, 7:] += rng.normal(size=69) * 0.1
        scenes.append(objs); kinds.append(a)
    action = rng.normal(size=(n, 75)) * 0.2
    blk, qpos, qvel, action = r32(blk), r32(qpos), r32(qvel), r32(action)
    scenes = [{oi: blk[e, 7 * oi: 7 * oi + 7].copy() for oi in sc} for e, sc in enumerate(scenes)]
    return dict(qpos=qpos, qvel=qvel, action=action, target=qpos.copy(), kind=np.asarray(kinds), objects=scenes, blk=blk)
