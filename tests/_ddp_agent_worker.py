"""Worker of tests/test_gpu_sampler.py::test_two_rank_agent_keeps_parameters_identical (launched by torch.distributed.run)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)                      # plumbing test on a 1-GPU box: every rank on device 0, gloo collectives
    dist.init_process_group("gloo")
    from kinpoly_amd import sim as kpsim
    from kinpoly_amd.agent import AgentAR
    from kinpoly_amd.env import standing_context
    std = np.load(os.path.join(ROOT, "tests", "golden", "standing_neutral.npz"))
    n, T = 32, 6
    fk_sim = kpsim.KpSim(kpsim.KpModel(), n, 0)

    def context_fn(m):
        ctx = standing_context(m, T + 2, std["qpos"], std["qvel"], fk_sim, torch.full((m,), 0.3 * rank))
        ctx["obj_pose"] = torch.tensor([0.0, 0, 0, 1, 0, 0, 0], device=ctx["qpos"].device).repeat(m, T + 2, 1)
        return ctx
    agent = AgentAR(n, context_fn, device=0, horizon=T, num_optim_epoch=2, num_step_update=2, use_init_context=False, pool_depth=T)
    before = torch.cat([p.detach().reshape(-1) for p in agent.policy_net.parameters()]).clone()
    info = agent.optimize_policy(0)
    flat = torch.cat([p.detach().reshape(-1) for p in list(agent.policy_net.parameters()) + list(agent.value_net.parameters())]).cpu()
    gathered = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    rew = torch.tensor([info["avg_reward"]]); allr = [torch.empty_like(rew) for _ in range(world)]
    dist.all_gather(allr, rew)
    ok = all(torch.equal(g, gathered[0]) for g in gathered) and float((flat[:before.numel()] - before.cpu()).abs().max()) > 0
    differ = abs(float(allr[0]) - float(allr[1])) > 0       # the ranks really sampled different shards (seed 4 + rank, other headings)
    if rank == 0:
        print("DDP_AGENT_OK" if (ok and differ) else f"DDP_AGENT_FAIL identical={ok} shards_differ={differ}", flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
