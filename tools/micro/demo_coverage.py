#!/usr/bin/env python
"""After the learning demo's working configuration (UHC checkpoint given, short warm start, a few PPO iterations): AgentAR.eval_policy('train') --
every take played WHOLE (110 - 160 frames) with mean actions of both policies.  Tells whether the sampled episodes' 20 - 45 frame lives come from the
exploration noise on the kinematic targets (then mean-action sequences run much longer) or from the controller itself."""
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
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    std = np.load(os.path.join(ROOT, "tests", "golden", "standing_neutral.npz"))
    n = 4096
    fk_sim = kpsim.KpSim(kpsim.KpModel(kpsim.STEP_KPM), n, 0)
    takes = D.synthetic_takes(fk_sim, std["qpos"], n_per_action=4, T_range=(110, 160), body_mass=read_kpm(kpsim.STEP_KPM)["body_mass"], seed=4)
    ds = D.StateARDataset(takes, fr_num=100, seed=4, device=fk_sim.device)
    agent = AgentAR(n, dataset=ds, device=0, horizon=24, cc_checkpoint=cc, eval_envs=16)
    agent.train_init(150, 12, 2000, 256)
    for tag in ("after the warm start", f"after {iters} PPO iterations"):
        res = agent.eval_policy("train")
        env, builder = agent._eval_engine()
        from kinpoly_amd.evaluate import eval_dataset
        full = eval_dataset(env, agent.policy_net, builder, ds)
        lens = {k: len(v["pred"]) for k, v in full.items()}
        pcs = {k: round(v["percent"], 3) for k, v in full.items()}
        print(tag, "| coverage", res[0], "| frames played per take", lens, "| percent", pcs, flush=True)
        if tag.startswith("after the warm"):
            for it in range(iters):
                info = agent.optimize_policy(it)
            print(f"sampled episodes at iteration {iters - 1}: fail_rate {info['fail_rate']:.3f} eps_len {info['log'].avg_episode_len:.1f}", flush=True)


if __name__ == "__main__":
    main()
