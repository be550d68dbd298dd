#!/usr/bin/env python
"""Time of AgentAR.train_init's two phases at kin_poly.yml's sizes (num_sample 2000, batch_size 256, fr_num 100) on the synthetic feature set:
seconds per epoch of update_init_supervised (x 500 in the reference) and of train_full_supervised (x 50).   python tools/warm_start_time.py [epochs]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from kinpoly_amd import dataset as D
    from kinpoly_amd import pretrain as P
    from kinpoly_amd import sim as kpsim
    from kinpoly_amd.agent import AgentAR
    from kinpoly_amd.model_compiler import read_kpm
    ep = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    std = np.load(os.path.join(ROOT, "tests", "golden", "standing_neutral.npz"))
    fk_sim = kpsim.KpSim(kpsim.KpModel(kpsim.STEP_KPM), 256, 0)
    takes = D.synthetic_takes(fk_sim, std["qpos"], n_per_action=4, T_range=(110, 160), body_mass=read_kpm(kpsim.STEP_KPM)["body_mass"], seed=4)
    ds = D.StateARDataset(takes, fr_num=100, seed=4, device=fk_sim.device)
    agent = AgentAR(256, dataset=ds, device=0, horizon=4)
    for name, fn in (("update_init_supervised", lambda n: P.update_init_supervised(agent.policy_net, agent.opt_sup, agent.fk, ds, n, 2000, 256)),
                     ("train_full_supervised", lambda n: P.train_full_supervised(agent.policy_net, agent.opt_sup, agent.fk, ds, n, 0.3, 2000, 256, noise_std=0.01))):
        first = fn(1)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        last = fn(ep)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / ep
        print(f"{name}: {dt:.2f} s per epoch (8 batches of 256 clips x 100 frames); loss after 1 epoch {first:.4f}, after {ep + 1} epochs {last:.4f}", flush=True)


if __name__ == "__main__":
    main()
