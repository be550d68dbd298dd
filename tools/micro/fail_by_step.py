#!/usr/bin/env python
"""Where inside a sample() call do episodes fail?  (diagnostic of the learning demo's plateau: mean episode length ~ horizon + 2.)
Builds the demo's working configuration (UHC checkpoint given, short warm start), runs a few iterations at two horizons and prints the number of failed /
ended envs per step index of the call, and the distribution of cur_t at failure."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    from kinpoly_amd import dataset as D
    from kinpoly_amd import sim as kpsim
    from kinpoly_amd.agent import AgentAR
    from kinpoly_amd.model_compiler import read_kpm
    cc = sys.argv[1]
    std = np.load(os.path.join(ROOT, "tests", "golden", "standing_neutral.npz"))
    n = 4096
    fk_sim = kpsim.KpSim(kpsim.KpModel(kpsim.STEP_KPM), n, 0)
    takes = D.synthetic_takes(fk_sim, std["qpos"], n_per_action=4, T_range=(110, 160), body_mass=read_kpm(kpsim.STEP_KPM)["body_mass"], seed=4)
    for hz in (24, 48):
        ds = D.StateARDataset(takes, fr_num=100, seed=4, device=fk_sim.device)
        agent = AgentAR(n, dataset=ds, device=0, horizon=hz, cc_checkpoint=cc)
        agent.train_init(60, 6, 2000, 256)
        for it in range(6):
            agent.trainer.per_epoch_update()
            batch = agent.sampler.sample(hz)
            done = (batch.masks == 0)
            fails = batch.fails
            if it >= 3:
                print(f"horizon {hz} iter {it}: episodes {int(done.sum())} fails {int(fails.sum())} | done per step index:", done.sum(0).int().tolist(), flush=True)
            agent.trainer.update(batch)
            from kinpoly_amd.supervised import update_supervised_step
            from kinpoly_amd.rollout import _allreduce_grads
            update_supervised_step(agent.policy_net, agent.opt_sup, agent.fk, batch, 20, _allreduce_grads)
        pc = np.asarray(batch.episodes["percent"])
        print(f"horizon {hz}: percent of the finished episodes: mean {pc.mean():.3f}, quantiles", np.quantile(pc, [0.1, 0.5, 0.9]).round(3).tolist(), "share == 1:", float((pc == 1).mean()), flush=True)
        del agent
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
