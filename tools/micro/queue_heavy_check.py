"""The heavy-env continuation of kp_step_queue_kernel (model option queue_heavy) changes who runs a job, never its result: 4096 envs on mixed floor / object scenes,
5 control steps with queue_heavy = 0, 130 and with the job queue off -- final states must be bit-identical; prints how many queue entries were never published."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from kinpoly_amd.sim import STEP_KPM, KpModel, KpSim
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
std = np.load(os.path.join(ROOT, "tests/golden/standing_neutral.npz"))
n = 4096
rng = np.random.default_rng(11)
x0, y0 = std["qpos"][0], std["qpos"][1]
scenes = [({}, 0.0), ({4: [x0, y0, 0.3705, 1, 0, 0, 0]}, 0.341), ({1: [x0 + 0.75, y0, 0.921, 1, 0, 0, 0], 2: [x0 + 0.75, y0, 0.7905, 1, 0, 0, 0]}, 0.0),
          ({3: [x0 + 0.36, y0 + 0.05, 0.69, 1, 0, 0, 0]}, 0.0)]
blk = np.zeros((n, 35))
for i in range(5):
    blk[:, 7 * i: 7 * i + 3] = [(i + 1) * 100, 100, 0]
qpos = np.tile(std["qpos"], (n, 1)); qpos[:, 7:] += rng.normal(size=(n, 69)) * 0.1
for e in range(n):
    act, lift = scenes[e % len(scenes)]
    qpos[e, 2] += lift
    for oi, pose in act.items():
        blk[e, 7 * oi: 7 * oi + 7] = pose
def status64(sim):
    ptr = sim.L.kp_sim_status_device(sim.h)
    iface = {"shape": (64,), "typestr": "<u4", "data": (int(ptr), False), "version": 3, "strides": None}
    return torch.as_tensor(type("_S", (), {"__cuda_array_interface__": iface})(), device="cuda").cpu().numpy()


dev = lambda a: torch.tensor(a, dtype=torch.float32, device="cuda")   # noqa: E731
qvel = rng.normal(size=(n, 75)) * 0.3
acts = [dev(rng.normal(size=(n, 75)) * 0.3) for _ in range(5)]
out = {}
for name, opts in (("queue off", dict(substeps_per_job=0)), ("queue_heavy 0", dict(queue_heavy=0)), ("queue_heavy 130", dict(queue_heavy=130)), ("queue_heavy 105", dict(queue_heavy=105)), ("queue_heavy 160", dict(queue_heavy=160)), ("queue_heavy 200", dict(queue_heavy=200)), ("queue_heavy 250", dict(queue_heavy=250))):
    for kpm, tag in ((None, "floor"), (STEP_KPM, "objects")):
        sim = KpSim(KpModel(kpm, **opts) if kpm else KpModel(**opts), n)
        if kpm:
            sim.set_objects(dev(blk))
        q = dev(qpos)
        sim.set_state(q, dev(qvel)); sim.set_target(q.clone())
        ts = []
        for a in acts:
            sim.step_ctrl(a, 15); ts.append(sim.last_step_seconds() * 1e3)
        st = sim.status_tensor().cpu().numpy()
        st16 = status64(sim)[16]
        out[(name, tag)] = (sim.get("qpos").cpu().numpy(), sim.get("qvel").cpu().numpy(), sim.get("obj_qpos").cpu().numpy() if kpm else None)
        print(f"{name:16s} {tag:8s} launch {np.mean(ts[1:]):.3f} ms; jobs run by the finishing wave in the last launch: {int(st16)} of {n * 3}; stalled {int(st[2])}", flush=True)
        del sim
for tag in ("floor", "objects"):
    ref = out[("queue off", tag)]
    for name in ("queue_heavy 0", "queue_heavy 130", "queue_heavy 105", "queue_heavy 160", "queue_heavy 200", "queue_heavy 250"):
        got = out[(name, tag)]
        same = all(r is None or np.array_equal(r, g) for r, g in zip(ref, got))
        print(f"{tag}: {name} bit-identical to the plain launch: {same}")
        assert same
