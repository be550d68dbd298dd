"""How long do the 4096 envs of one control-step launch take each, and what does the order they start in cost?
Prints the distribution of per-env cycles (kp_sim_launch_cost) of the bench-like workload, the makespan a greedy list
scheduler gets on 2048 wave slots in index order vs longest-first, and the measured launch time with / without "lpt_order".
    python tools/launch_balance.py"""
import heapq
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kinpoly_amd.sim import KpModel, KpSim  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
std = np.load(os.path.join(ROOT, "tests/golden/standing_neutral.npz"))
n = 4096


def makespan(cost, slots):
    h = [0.0] * slots
    heapq.heapify(h)
    for c in cost:
        heapq.heappush(h, heapq.heappop(h) + float(c))
    return max(h)


for lpt in (0, 1):
    rng = np.random.default_rng(3)
    qpos = np.tile(std["qpos"], (n, 1)); qpos[:, 7:] += rng.normal(size=(n, 69)) * 0.2
    qvel = rng.normal(size=(n, 75)) * 0.5
    sim = KpSim(KpModel(lpt_order=lpt), n)
    q = torch.tensor(qpos, dtype=torch.float32, device="cuda"); v = torch.tensor(qvel, dtype=torch.float32, device="cuda")
    sim.set_state(q, v); sim.set_target(q.clone())
    a = torch.tensor(rng.normal(size=(n, 75)) * 0.2, dtype=torch.float32, device="cuda")
    ms = []
    prev = None
    for it in range(12):
        sim.step_ctrl(a, 15)
        ms.append(sim.last_step_seconds() * 1e3)
        c = sim.launch_cost().astype(np.float64)
        dg = sim.diag().astype(np.float64)
        work = dg[:, 1] + 0.0
        if prev is not None and it >= 9:
            print(f"   launch {it}: corr(cost, newton iterations of the same launch) = {np.corrcoef(c, work)[0, 1]:.3f}; corr(iterations, previous iterations) = "
                  f"{np.corrcoef(work, prev_work)[0, 1]:.3f}; corr(cost, previous iterations) = {np.corrcoef(c, prev_work)[0, 1]:.3f}")
            print(f"   launch {it}: corr(cost, previous cost) = {np.corrcoef(c, prev)[0, 1]:.3f}; model makespan with the previous launch's "
                  f"order {makespan(c[np.argsort(-prev, kind='stable')], 2048) / 2.38e6:.3f} ms, index order {makespan(c, 2048) / 2.38e6:.3f} ms")
        prev = c; prev_work = work
    print(f"lpt_order={lpt}: launch ms (last 6) {np.round(ms[-6:], 3)}; env cycles mean {c.mean():.0f} p50 {np.percentile(c, 50):.0f} "
          f"p90 {np.percentile(c, 90):.0f} p99 {np.percentile(c, 99):.0f} max {c.max():.0f}; sum/2048 slots = {c.sum() / 2048 / 2.38e6:.3f} ms, "
          f"max env = {c.max() / 2.38e6:.3f} ms", flush=True)
    if lpt == 0:
        print(f"   list-scheduling model on 2048 slots: index order {makespan(c, 2048) / 2.38e6:.3f} ms, longest first {makespan(np.sort(c)[::-1], 2048) / 2.38e6:.3f} ms")
