#!/usr/bin/env python
"""Counterpart of the reference's scripts/train_ar_policy.py on the batched MI355X engine.

    python scripts/train_ar_policy.py --num_envs 4096 --iters 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 scripts/train_ar_policy.py

The reference's MoCap dataset and trained UHC weights are not part of its repository (downlaod_data.sh), so this
driver runs the synthetic standing-clip configuration of SURVEY.md section 8(d) (config 3) with seeded random-init networks.
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
    ap.add_argument("--save", type=str, default="")
    args = ap.parse_args()
    world, rank, local = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    from kinpoly_amd.agent import AgentAR
    from kinpoly_amd.env import standing_context
    std = np.load(os.path.join(ROOT, "tests", "golden", "standing_neutral.npz"))
    holder = {}

    def context_fn(n):
        g = torch.Generator().manual_seed(4 + rank + 1000 * holder.get("calls", 0)); holder["calls"] = holder.get("calls", 0) + 1
        headings = (torch.rand(n, generator=g) * 2 - 1) * np.pi
        ctx = standing_context(n, args.clip_len, std["qpos"], std["qvel"], holder["agent_sim"], headings)
        ctx["obj_pose"] = torch.tensor([0.0, 0, 0, 1, 0, 0, 0], device=ctx["qpos"].device).repeat(n, args.clip_len, 1)
        return ctx

    # the context builder needs a sim for FK before the agent exists: use a throw-away one
    from kinpoly_amd import sim as kpsim
    holder["agent_sim"] = kpsim.KpSim(kpsim.KpModel(), args.num_envs, local)
    agent = AgentAR(args.num_envs, context_fn, device=local, horizon=args.horizon, num_optim_epoch=args.num_optim_epoch,
                    num_step_update=args.num_step_update)
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
