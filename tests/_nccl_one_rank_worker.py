"""Worker of tests/test_gpu_round3.py::test_one_rank_nccl_group_moves_device_tensors: a 1-rank `nccl` (= RCCL) process group on the box's
GPU, with the exchange step of the PPO iteration forced through it (rollout.FORCE_COLLECTIVES): the all-gather of advantages / returns
(normalize_advantages_global, uhc/khrylib/rl/core/common.py:22 made job-wide) and the gradient all-reduce of the data-parallel update."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29547")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    from kinpoly_amd import rollout as R
    R.FORCE_COLLECTIVES = True
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(5)
    adv = torch.randn((64, 10), device=dev, generator=g); ret = torch.randn((64, 10), device=dev, generator=g)
    calls = {"all_gather": 0, "all_reduce": 0}
    ag, ar = dist.all_gather, dist.all_reduce

    def count_ag(*a, **k):
        calls["all_gather"] += 1
        return ag(*a, **k)

    def count_ar(*a, **k):
        calls["all_reduce"] += 1
        return ar(*a, **k)
    dist.all_gather, dist.all_reduce = count_ag, count_ar
    nadv, _, all_ret = R.normalize_advantages_global(adv, ret)
    want = (adv - adv.mean()) / adv.std()
    ok1 = torch.allclose(nadv, want, atol=1e-6) and torch.equal(all_ret, ret.reshape(-1)) and all_ret.is_cuda
    lin = torch.nn.Linear(16, 8).to(dev)
    lin(torch.randn((4, 16), device=dev, generator=g)).sum().backward()
    before = [p.grad.clone() for p in lin.parameters()]
    R._allreduce_grads(list(lin.parameters()))
    ok2 = all(torch.allclose(p.grad, b) for p, b in zip(lin.parameters(), before))
    one = torch.ones(1, device=dev)
    ar(one)
    torch.cuda.synchronize()
    ok3 = calls["all_gather"] == 1 and calls["all_reduce"] == 1 and float(one) == 1.0 and dist.get_backend() == "nccl"
    # the job-wide freq_dict exchange (EpisodeSource.record: all_gather_object of the finished episodes, agent_ar.py:664-673) through RCCL
    class _DS:
        takes = ["a", "b"]
    src = R.EpisodeSource.__new__(R.EpisodeSource)
    src.dataset, src.freq_dict, src._probs = _DS(), {"a": [], "b": []}, None
    ago, n_ago = dist.all_gather_object, [0]

    def count_ago(*a, **k):
        n_ago[0] += 1
        return ago(*a, **k)
    dist.all_gather_object = count_ago
    src.record([0, 1, 1], [3, 5, 7], [1.0, 0.25, 0.5])
    ok4 = n_ago[0] == 1 and src.freq_dict == {"a": [[1.0, 3]], "b": [[0.25, 5], [0.5, 7]]}
    ok5 = R._agree_status(0, dev) == 0
    print("NCCL_ONE_RANK_OK" if (ok1 and ok2 and ok3 and ok4 and ok5) else f"NCCL_ONE_RANK_FAIL {ok1} {ok2} {ok3} {ok4} {ok5} {calls}", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
