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
    # a dataset (synthetic takes in the reference's feature-file schema) so that the adaptive take sampling and its freq_dict run: ONE
    # job-wide dict, merged across ranks once per sample() (agent_ar.py:664-673)
    from kinpoly_amd import dataset as D
    from kinpoly_amd.model_compiler import read_kpm
    takes = D.synthetic_takes(fk_sim, std["qpos"], n_per_action=1, T_range=(T + 6, T + 12), body_mass=read_kpm(kpsim.STEP_KPM)["body_mass"], seed=3)
    ds = D.StateARDataset(takes, fr_num=T + 2, seed=3 + rank, device=fk_sim.device)
    agent = AgentAR(n, None, device=0, horizon=T, num_optim_epoch=2, num_step_update=2, use_init_context=False, pool_depth=T, dataset=ds)
    before = torch.cat([p.detach().reshape(-1) for p in agent.policy_net.parameters()]).clone()
    info = agent.optimize_policy(0)
    dicts = [None] * world
    dist.all_gather_object(dicts, agent.freq_dict)
    n_rec = sum(len(v) for v in dicts[0].values())
    same_freq = all(d == dicts[0] for d in dicts) and n_rec > 0
    eps = [None] * world
    dist.all_gather_object(eps, int(info["episodes"]))
    same_freq = same_freq and n_rec == sum(eps)            # every rank's finished episodes are in everybody's dict
    flat = torch.cat([p.detach().reshape(-1) for p in list(agent.policy_net.parameters()) + list(agent.value_net.parameters())]).cpu()
    gathered = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    rew = torch.tensor([info["avg_reward"]]); allr = [torch.empty_like(rew) for _ in range(world)]
    dist.all_gather(allr, rew)
    ok = all(torch.equal(g, gathered[0]) for g in gathered) and float((flat[:before.numel()] - before.cpu()).abs().max()) > 0
    differ = abs(float(allr[0]) - float(allr[1])) > 0       # the ranks really sampled different shards (seed 4 + rank, other headings)
    if rank == 0:
        print("DDP_AGENT_OK" if (ok and differ and same_freq) else f"DDP_AGENT_FAIL identical={ok} shards_differ={differ} freq_dict_shared={same_freq}", flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
