"""Cost of the memory-model-correct hand-over (model option queue_fence = 1: agent-scope release / acquire fences around the job publish)
against the default relaxed write-through hand-over of kp_step_queue_kernel, on the bench's standing + contact states.
    python tools/queue_fence_bench.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kinpoly_amd.sim import KpModel, KpSim  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
std = np.load(os.path.join(ROOT, "tests/golden/standing_neutral.npz"))
n = 4096
rng = np.random.default_rng(3)
qpos = np.tile(std["qpos"], (n, 1)); qpos[:, 7:] += rng.normal(size=(n, 69)) * 0.2
qvel = rng.normal(size=(n, 75)) * 0.5
res = {}
for fence in (0, 1, 0, 1):
    sim = KpSim(KpModel(queue_fence=fence), n)
    q = torch.tensor(qpos, dtype=torch.float32, device="cuda"); v = torch.tensor(qvel, dtype=torch.float32, device="cuda")
    sim.set_state(q, v); sim.set_target(q.clone())
    a = torch.tensor(rng.normal(size=(n, 75)) * 0.0, dtype=torch.float32, device="cuda")
    for _ in range(3):
        sim.step_ctrl(a, 15)
    sim.timing_reset()
    for _ in range(20):
        sim.step_ctrl(a, 15)
    ms, k = sim.timing_mean_seconds()
    res.setdefault(fence, []).append(ms * 1e3)
    out = sim.get("qpos").clone()
    res.setdefault(("q", fence), out)
print(f"relaxed sc1 hand-over : {np.mean(res[0]):.3f} ms / launch   (runs {res[0]})")
print(f"release/acquire fences: {np.mean(res[1]):.3f} ms / launch   (runs {res[1]})   -> x{np.mean(res[1]) / np.mean(res[0]):.3f}")
print("bit-identical states:", bool(torch.equal(res[("q", 0)], res[("q", 1)])))
