"""What would splitting the `objects` env-step into its costliest envs and the rest buy?  The physics launch alone, on the bench's objects engine:
all 4096 envs; the 128 / 256 costliest of the previous launch only (env_mask); everybody else only.  If T(rest) + the 0.87 ms of policy / bookkeeping work is
below T(costliest), overlapping them hides the policy work inside the stragglers' chain; the launch itself cannot get shorter than T(costliest).
    python tools/micro/objects_split_probe.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

torch.cuda.set_device(0)
env, policy, sampler, std = bench.build_engine(0, 4, 64, "objects")
bench.stagger_episodes(env, sampler, 4, True)
bench.rollout_steps(sampler, 15, None, False, True)
sim = env.sim
act = torch.zeros((env.n, 75), device=env.device)


def snapshot():
    return {k: sim.get(k).clone() for k in ("qpos", "qvel", "qpos_d", "qvel_d")}, sim.get("obj_qpos").clone(), sim.get("obj_qvel").clone()


def restore(s):
    st, oq, ov = s
    sim.set_full_state(st["qpos"], st["qvel"], st["qpos_d"], st["qvel_d"])
    sim.set_obj_state(oq, ov)


base = snapshot()
sim.step_ctrl(act, 15)
cost = sim.launch_cost().astype(np.float64)
print(f"per-env cost of one launch: longest {cost.max() / 2.38e6:.3f} ms, median {np.median(cost) / 2.38e6:.3f} ms, sum / 1792 slots {cost.sum() / 1792 / 2.38e6:.3f} ms")
order = np.argsort(-cost)
for k in (0, 128, 256, 512):
    for tag in (("all",) if k == 0 else ("costliest", "rest")):
        mask = np.ones(env.n, np.uint8)
        if tag == "costliest":
            mask[:] = 0; mask[order[:k]] = 1
        elif tag == "rest":
            mask[order[:k]] = 0
        m = torch.tensor(mask, device=env.device)
        ts = []
        for _ in range(6):
            restore(base)
            sim.step_ctrl(act, 15, None if k == 0 else m)
            ts.append(sim.last_step_seconds() * 1e3)
        print(f"{tag:10s} k={k:4d}: envs stepped {int(mask.sum()):5d}, launch {np.median(ts[1:]):.3f} ms (min {min(ts[1:]):.3f})", flush=True)
