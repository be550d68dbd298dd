#!/usr/bin/env python
"""Counterpart of the reference's scripts/train_uhc.py on the batched MI355X engine: PPO training of the UHC (PolicyMCP) on
the imitation env.  The AMASS clips of the reference are not part of its repository, so the expert is the standing clip of
`sample_data/standing_neutral.pkl` (tests/golden/standing_neutral.npz) with small seeded joint-space sinusoids.

    python scripts/train_uhc.py --num_envs 4096 --iters 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 scripts/train_uhc.py
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
    ap.add_argument("--horizon", type=int, default=32)
    ap.add_argument("--clip_len", type=int, default=64)
    ap.add_argument("--num_optim_epoch", type=int, default=10)
    ap.add_argument("--save", type=str, default="")
    args = ap.parse_args()
    world, rank, local = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    from kinpoly_amd import checkpoint as ck
    from kinpoly_amd.uhc_env import BatchedHumanoidEnv, CopycatAgent
    std = np.load(os.path.join(ROOT, "tests", "golden", "standing_neutral.npz"))
    rng = np.random.default_rng(1 + rank)
    n, T = args.num_envs, args.clip_len
    clips = np.tile(std["qpos"], (n, T, 1))
    amp, freq, ph = rng.uniform(0, 0.15, (n, 1, 69)), rng.uniform(0.2, 1.0, (n, 1, 69)), rng.uniform(0, 2 * np.pi, (n, 1, 69))
    tt = np.arange(T)[None, :, None] / 30.0
    clips[:, :, 7:] += amp * (np.sin(2 * np.pi * freq * tt + ph) - np.sin(ph))
    torch.manual_seed(1 + rank)
    env = BatchedHumanoidEnv(n, local, env_init_noise=0.0, seed=1 + rank)
    env.load_expert(torch.tensor(clips, dtype=torch.float32))
    agent = CopycatAgent(env, num_optim_epoch=args.num_optim_epoch)
    if world > 1:
        for p in list(agent.policy.parameters()) + list(agent.value.parameters()):
            dist.broadcast(p.data, 0)
    for it in range(args.iters):
        stats = agent.optimize_policy(args.horizon)
        if rank == 0:
            print(json.dumps({"iter": it, **{k: (round(v, 5) if isinstance(v, float) else v) for k, v in stats.items()}}), flush=True)
    if args.save and rank == 0:
        rs = ck.ZFilter((784,), clip=agent.running_state.clip); rs.rs._n = agent.running_state.count      # the clip the controller was trained under (uhc.yml: 5)
        rs.rs._M = agent.running_state._mean64.cpu().numpy(); rs.rs._S = agent.running_state._m2.cpu().numpy()
        import pickle
        with ck._RefModulePath(), open(args.save, "wb") as f:
            pickle.dump({"policy_dict": {k: v.cpu() for k, v in agent.policy.state_dict().items()},
                         "value_dict": {k: v.cpu() for k, v in agent.value.state_dict().items()}, "running_state": rs}, f)
    if world > 1:
        dist.barrier(); dist.destroy_process_group()


if __name__ == "__main__":
    main()
