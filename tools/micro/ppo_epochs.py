#!/usr/bin/env python
"""The clipped surrogate epoch by epoch on ONE batch (policy lr 1e-5, Adam, clip 40): does it go down from 0?  Twice: with a fresh optimiser and with the
optimiser state a few iterations of training leave behind.   python tools/micro/ppo_epochs.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def epochs(agent, batch, n_ep):
    from kinpoly_amd.rollout import estimate_advantages, ppo_surrogate
    tr, pol = agent.trainer, agent.policy_net
    N, T, _ = batch.states.shape
    flat = batch.states.reshape(N * T, -1)
    with torch.no_grad():
        values = agent.value_net(flat).view(N, T); last_v = agent.value_net(batch.last_states).view(N)
    adv, _ = estimate_advantages(batch.rewards, batch.masks, values, tr.gamma, tr.tau, None, last_v)
    adv = adv.reshape(-1, 1)
    fixed, out = None, []
    for _ in range(n_ep):
        means = pol.unroll(batch.states, batch.episode_start, batch.hx0)
        lp = pol.log_prob(means.reshape(N * T, -1), batch.actions.reshape(N * T, -1))
        if fixed is None:
            fixed = lp.detach()
        surr = ppo_surrogate(lp, fixed, adv, tr.clip_epsilon)
        ratio = torch.exp(lp.detach() - fixed)
        tr.opt_p.zero_grad(); surr.backward()
        gn = float(torch.nn.utils.clip_grad_norm_([p for g in tr.opt_p.param_groups for p in g["params"]], tr.policy_grad_clip))
        tr.opt_p.step()
        out.append((float(surr), float(ratio.mean()), float((ratio - 1).abs().gt(tr.clip_epsilon).float().mean()), gn))
    return out


def main():
    from kinpoly_amd import dataset as D
    from kinpoly_amd import sim as kpsim
    from kinpoly_amd.agent import AgentAR
    from kinpoly_amd.model_compiler import read_kpm
    std = np.load(os.path.join(ROOT, "tests", "golden", "standing_neutral.npz"))
    n = 4096
    fk_sim = kpsim.KpSim(kpsim.KpModel(kpsim.STEP_KPM), n, 0)
    takes = D.synthetic_takes(fk_sim, std["qpos"], n_per_action=4, T_range=(110, 160), body_mass=read_kpm(kpsim.STEP_KPM)["body_mass"], seed=4, amp_max=0.1)
    ds = D.StateARDataset(takes, fr_num=100, seed=4, device=fk_sim.device)
    agent = AgentAR(n, dataset=ds, device=0, horizon=24)
    agent.train_init(40, 4, 2000, 256)
    batch = agent.sampler.sample(24)
    for tag in ("fresh optimiser", "after 6 training iterations"):
        if tag != "fresh optimiser":
            for it in range(6):
                agent.optimize_policy(it)
            batch = agent.sampler.sample(24)
        rows = epochs(agent, batch, 12)
        print(tag, "| surrogate, mean ratio, share of samples outside the clip, gradient norm before the clip, per epoch:")
        for i, r in enumerate(rows):
            print(f"  epoch {i}: surr {r[0]:+.5f} ratio {r[1]:.4f} outside {r[2]:.3f} |g| {r[3]:.2f}")


if __name__ == "__main__":
    main()
