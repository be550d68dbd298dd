#!/usr/bin/env python
"""Counterpart of the reference's scripts/train_ar_policy.py on the batched MI355X engine.

    python scripts/train_ar_policy.py --num_envs 4096 --iters 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 scripts/train_ar_policy.py

The reference's MoCap dataset and trained UHC weights are not part of its repository (downlaod_data.sh), so this
driver builds synthetic takes in the reference's feature-file schema (all four action classes with their objects, SURVEY.md
section 8(d) config 4) unless --data points at a real feature file, and trains seeded random-init networks on them.
"""
import argparse
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--num_envs", type=int, default=4096)
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--horizon", type=int, default=99)
    ap.add_argument("--clip_len", type=int, default=100)
    ap.add_argument("--num_optim_epoch", type=int, default=10)
    ap.add_argument("--num_step_update", type=int, default=20)
    ap.add_argument("--pool_depth", type=int, default=4, help="clips kept queued behind every env's current one (one host read per pool_depth steps)")
    ap.add_argument("--cache_init_context", action="store_true", help="look init_qpos / init_qvel of a window up once it has been computed under the same context-network parameters")
    ap.add_argument("--save", type=str, default="")
    ap.add_argument("--data", type=str, default="", help="feature file in the reference's schema (<data_dir>/features/<data_file>.p)")
    args = ap.parse_args()
    world, rank, local = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    from kinpoly_amd.agent import AgentAR
    std = np.load(os.path.join(ROOT, "tests", "golden", "standing_neutral.npz"))
    from kinpoly_amd import dataset as D
    from kinpoly_amd import sim as kpsim
    from kinpoly_amd.model_compiler import read_kpm
    # the agent's kinematic twin sim doubles as the FK engine of the feature construction
    fk_sim = kpsim.KpSim(kpsim.KpModel(kpsim.STEP_KPM), args.num_envs, local)
    if args.data:
        ds = D.StateARDataset(args.data, fr_num=args.clip_len, seed=4 + rank, device=fk_sim.device)
    else:       # the reference's MoCap features are not in its repository: same schema, synthetic takes.  ONE
        # data set for the whole job (take seed independent of the rank): the job-wide freq_dict is keyed by take name, so a name must
        # mean the same motion on every rank; only the draw stream (dataset seed) differs per rank
        takes = D.synthetic_takes(fk_sim, std["qpos"], n_per_action=4, T_range=(args.clip_len + 10, args.clip_len + 60),
                                  body_mass=read_kpm(kpsim.STEP_KPM)["body_mass"], seed=4)
        ds = D.StateARDataset(takes, fr_num=args.clip_len, seed=4 + rank, device=fk_sim.device)
    if rank == 0:
        print(f"dataset: {ds.get_len()} takes, {len(ds.freq_indices)} windows of {args.clip_len} frames", flush=True)

    # every episode draws its clip through data_loader.sample_seq(freq_dict, sampling_temp, sampling_freq) (agent_ar.py:519-523): the
    # agent keeps the freq_dict and feeds each finished episode's [percent, fr_start] back (random window starts, adaptive takes)
    agent = AgentAR(args.num_envs, dataset=ds, device=local, horizon=args.horizon, num_optim_epoch=args.num_optim_epoch,
                    num_step_update=args.num_step_update, sampling_temp=0.3, sampling_freq=0.5, pool_depth=args.pool_depth, cache_init_context=args.cache_init_context)
    for it in range(args.iters):
        info = agent.optimize_policy(it)
        if rank == 0:
            print(json.dumps({"iter": it, **{k: (round(v, 5) if isinstance(v, float) else v) for k, v in info.items()}}), flush=True)
    if args.save and rank == 0:
        agent.save_checkpoint(args.save)
    if world > 1:
        dist.barrier(); dist.destroy_process_group()


if __name__ == "__main__":
    main()
