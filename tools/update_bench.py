"""Time of one AgentAR.optimize_policy iteration at the reference's batch shape (4096 envs x 24 steps = 98 k samples, 10 PPO epochs + 20
supervised steps): sampling vs update, with the fused GRU re-unroll and with the per-step GRUCell loop it replaced.   python tools/update_bench.py"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kinpoly_amd import sim as kpsim  # noqa: E402
from kinpoly_amd.agent import AgentAR  # noqa: E402
from kinpoly_amd.env import standing_context  # noqa: E402
from kinpoly_amd.nets import KinPolicy  # noqa: E402

std = np.load(os.path.join(ROOT, "tests", "golden", "standing_neutral.npz"))
n, T = int(os.environ.get("KP_N", 4096)), 24
fk_sim = kpsim.KpSim(kpsim.KpModel(), n, 0)


def context_fn(m):
    ctx = standing_context(m, 100, std["qpos"], std["qvel"], fk_sim, torch.zeros(m))
    ctx["obj_pose"] = torch.tensor([0.0, 0, 0, 1, 0, 0, 0], device=ctx["qpos"].device).repeat(m, 100, 1)
    return ctx


agent = AgentAR(n, context_fn, device=0, horizon=T, use_init_context=False, pool_depth=2)
for it in range(3):
    info = agent.optimize_policy(it)
    print(f"fused unroll   iter {it}: T_sample {info['T_sample']:.3f} s  T_update {info['T_update']:.3f} s  ({info['num_steps']} samples)", flush=True)
fused = KinPolicy.unroll
KinPolicy.unroll = KinPolicy.unroll_reference
for it in range(2):
    info = agent.optimize_policy(it)
    print(f"GRUCell loop   iter {it}: T_sample {info['T_sample']:.3f} s  T_update {info['T_update']:.3f} s", flush=True)
KinPolicy.unroll = fused
# loss parity of one PPO + supervised pass between the two unrolls on the same batch
batch = agent.sampler.sample(T)
with torch.no_grad():
    a, b = agent.policy_net.unroll(batch.states, batch.episode_start, batch.hx0), agent.policy_net.unroll_reference(batch.states, batch.episode_start, batch.hx0)
    lp_a = agent.policy_net.log_prob(a.reshape(n * T, -1), batch.actions.reshape(n * T, -1)); lp_b = agent.policy_net.log_prob(b.reshape(n * T, -1), batch.actions.reshape(n * T, -1))
print(f"means max |diff| {float((a - b).abs().max()):.2e}; mean log-prob {float(lp_a.mean()):.6f} vs {float(lp_b.mean()):.6f} (|diff| {abs(float(lp_a.mean()) - float(lp_b.mean())):.2e})")
