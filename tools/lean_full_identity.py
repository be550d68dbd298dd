"""Longer bit-identity run of the two floor-scene layouts: 4096 envs, `steps` control steps with fresh random actions every step, a share of the envs thrown onto
the floor at random moments (resets onto lying / half-buried poses: the contact counts the lean layout hands over), lean queue vs full-layout queue vs one
workgroup per env.  Prints the number of differing words per field (0 expected) and the hand-over counters.    python tools/lean_full_identity.py [steps] [seed]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kinpoly_amd import sim as kp  # noqa: E402

STD = np.load(os.path.join(ROOT, "tests", "golden", "standing_neutral.npz"))
steps, seed = int(sys.argv[1]) if len(sys.argv) > 1 else 30, int(sys.argv[2]) if len(sys.argv) > 2 else 0
n = 4096
dev = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device="cuda")      # noqa: E731


def run(**opts):
    rng = np.random.default_rng(seed)
    qpos = np.tile(STD["qpos"], (n, 1)); qpos[:, 7:] += np.clip(rng.normal(size=(n, 69)) * 0.2, -np.pi, np.pi)
    qvel = rng.normal(size=(n, 75)) * 0.5
    sim = kp.KpSim(kp.KpModel(**opts), n)
    sim.set_state(dev(qpos), dev(qvel)); sim.set_target(dev(np.tile(STD["qpos"], (n, 1))))
    over = 0
    for t in range(steps):
        a = dev(rng.normal(size=(n, 75)) * 0.3)
        if t % 5 == 2:                                   # a tenth of the envs restart lying on the floor, pelvis 5 - 25 cm up, any heading
            idx = rng.choice(n, n // 10, replace=False)
            q = np.tile(STD["qpos"], (idx.size, 1)); q[:, 7:] += rng.normal(size=(idx.size, 69)) * 0.3
            ang = rng.uniform(-np.pi, np.pi, idx.size); c, s_ = np.cos(np.pi / 4), np.sin(np.pi / 4)
            q[:, 3] = np.cos(ang / 2) * c; q[:, 4] = np.cos(ang / 2) * s_; q[:, 5] = np.sin(ang / 2) * s_; q[:, 6] = np.sin(ang / 2) * c
            q[:, 2] = rng.uniform(0.05, 0.25, idx.size)
            mask = np.zeros(n, np.uint8); mask[idx] = 1
            full_q = sim.get("qpos").cpu().numpy(); full_v = sim.get("qvel").cpu().numpy()
            full_q[idx] = q; full_v[idx] = rng.normal(size=(idx.size, 75)) * 0.3
            sim.set_state(dev(full_q), dev(full_v), torch.tensor(mask, device="cuda"))
        sim.step_ctrl(a, 15)
        c = sim.queue_counters() if opts.get("substeps_per_job", 4) else {"lean_overflow_jobs": 0, "fallbacks_to_full_layout": 0}
        over += c["lean_overflow_jobs"]
    dg = sim.diag()
    return [sim.get(k).cpu().numpy() for k in ("qpos", "qvel", "xpos", "xquat", "xipos", "qpos_d")] + [dg], over, (c if opts.get("substeps_per_job", 4) else None), int((dg[:, 3] & 255).max())


lean, over, c, mc = run(lean_queue=1)
print(f"lean queue: {steps} control steps, overflow jobs handed over {over}, counters at the end {c}, largest contact count of the last step {mc}", flush=True)
lean_na, over2, _, _ = run(lean_queue=1, lean_adaptive=0)
full, _, _, _ = run(lean_queue=0)
one, _, _, _ = run(substeps_per_job=0)
names = ("qpos", "qvel", "xpos", "xquat", "xipos", "qpos_d", "diag")
for tag, other in (("lean (adaptive off, %d hand-overs) vs full-layout queue" % over2, lean_na), ("lean vs full-layout queue", full), ("lean vs one workgroup per env", one)):
    diffs = {k: int((a != b).sum()) for k, a, b in zip(names, lean, other)}
    if tag.endswith("per env"):
        diffs.pop("diag")                                # factorisation counters are per launch form
    print(tag, "-- differing words:", diffs, flush=True)
print("finite:", bool(np.isfinite(lean[0]).all()), "flagged envs:", int(((lean[6][:, 2] & 255) != 0).sum()))
