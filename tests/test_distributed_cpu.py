"""world_size-2 gloo tests (CPU) of the multi-GPU exchange step: the all-gather of advantages/returns for
the global normalisation and the gradient all-reduce (kinpoly_amd/rollout.py)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from kinpoly_amd.rollout import _allreduce_grads, env_shard, normalize_advantages_global
    g = torch.Generator().manual_seed(123)
    full_adv = torch.randn(2 * 64 * 10, generator=g); full_ret = torch.randn(2 * 64 * 10, generator=g)
    n = full_adv.numel() // world
    adv, ret = full_adv[rank * n:(rank + 1) * n].view(64, 10), full_ret[rank * n:(rank + 1) * n].view(64, 10)
    nadv, _, all_ret = normalize_advantages_global(adv, ret)
    want = ((full_adv - full_adv.mean()) / full_adv.std())[rank * n:(rank + 1) * n].view(64, 10)
    ok1 = torch.allclose(nadv, want, atol=1e-6) and torch.allclose(all_ret, full_ret)
    lin = torch.nn.Linear(4, 3)
    torch.manual_seed(rank)
    lin(torch.randn(8, 4)).sum().backward()
    local = [p.grad.clone() for p in lin.parameters()]
    gathered = [None] * world
    dist.all_gather_object(gathered, [x.numpy() for x in local])
    _allreduce_grads(list(lin.parameters()))
    ok2 = all(np.allclose(p.grad.numpy(), np.mean([gathered[r][i] for r in range(world)], 0), atol=1e-6) for i, p in enumerate(lin.parameters()))
    ids, seed = env_shard(rank, world, 4096)
    ok3 = ids[0] == rank * 4096 and len(ids) == 4096 and seed == 4 + rank
    q.put((rank, bool(ok1), bool(ok2), bool(ok3)))
    dist.destroy_process_group()


def test_allgather_normalisation_and_grad_allreduce_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(r[0] for r in res) == [0, 1]
    for r in res:
        assert r[1] and r[2] and r[3], r


def _freq_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import types
    from kinpoly_amd.rollout import EpisodeSource
    ds = types.SimpleNamespace(takes=["sit-a", "push-b", "step-c"])
    src = EpisodeSource(dataset=ds)
    # rank r finished (take, fr_start, percent) episodes of its own shard; two sample() calls
    for call in range(2):
        ti = [(rank + call) % 3, 2, rank]
        src.record(ti, [10 * rank + call, 5, 7], [1.0, 0.25 * (rank + 1), 0.5])
    q.put((rank, src.freq_dict))
    dist.destroy_process_group()


def test_freq_dict_is_one_job_wide_dict_world2():
    """agent_ar.py:664-673: every worker's finished episodes are merged into ONE freq_dict before the next draws.  Sharded over ranks,
    EpisodeSource.record exchanges them (all_gather_object) and appends in rank order: both ranks end with the same dict, which holds
    both shards' evidence."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_freq_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
    assert res[0] == res[1], "ranks hold different freq_dicts"
    fd = res[0]
    assert sum(len(v) for v in fd.values()) == 2 * 2 * 3
    # rank order inside a call, call order across calls: take 'step-c' (index 2) gets rank 0's then rank 1's rows of each call
    assert fd["step-c"][:2] == [[0.25, 5], [0.5, 5]] or fd["step-c"][0] == [0.25, 5]
    assert [1.0, 10] in fd["push-b"] and [1.0, 0] in fd["sit-a"] and [1.0, 1] in fd["push-b"] and [1.0, 11] in fd["step-c"]


def test_single_process_normalisation_matches_reference_golden(golden):
    from kinpoly_amd.rollout import normalize_advantages_global
    from oracle import np_oracle as O
    g = golden("gae_zfilter")
    adv_raw, ret = O.estimate_advantages(g["rewards"], g["masks"], g["values"], 0.95, 0.95)
    # undo the oracle's normalisation to feed raw advantages: recompute raw via returns - values
    raw = torch.tensor(g["ret"] - g["values"])
    nadv, _, _ = normalize_advantages_global(raw, torch.tensor(g["ret"]))
    np.testing.assert_allclose(nadv.numpy(), g["adv"], atol=1e-10)


def _logger_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from kinpoly_amd.rollout import LoggerRL, _agree_status, episode_log
    g = torch.Generator().manual_seed(50 + rank)
    R, D, CI = torch.rand(6, 9, generator=g), torch.rand(6, 9, generator=g) < 0.3, torch.rand(6, 9, 6, generator=g)
    st, _ = episode_log(R, D, CI, torch.zeros(6, dtype=torch.float64))
    st = st.tolist()
    mine = LoggerRL(num_steps=int(st[0]), num_episodes=int(st[1]), total_reward=st[2], min_episode_reward=st[3], max_episode_reward=st[4],
                    total_c_reward=st[5], min_c_reward=st[6], max_c_reward=st[7], total_c_info=np.asarray(st[8:14]))
    every = [None] * world
    dist.all_gather_object(every, mine)                        # what AgentAR.optimize_policy does with the ranks' loggers
    m = LoggerRL.merge(every)
    status = _agree_status(7 if rank == 1 else 0, "cpu")       # one rank's stalled queue is every rank's status
    q.put((rank, m.num_steps, m.num_episodes, m.total_c_reward, m.min_c_reward, m.max_episode_reward, float(m.avg_c_info.sum()), mine.num_episodes, mine.total_c_reward, status))
    dist.destroy_process_group()


def test_logger_statistics_merge_over_ranks_world2():
    """LoggerRL.merge over the ranks' loggers (the reference merges its workers', logger_rl.py:44-70): every rank ends up with the same job-wide record."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_logger_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
    a, b = res
    assert a[1:7] == b[1:7] and a[1] == 2 * 54                                   # identical merged records, 2 x 6 x 9 steps
    assert a[2] == a[7] + b[7] and a[3] == pytest.approx(a[8] + b[8])            # episodes and reward sums add up
    assert a[9] == b[9] == 7


# ---------------------------------------------------------------------------------------------- round 5: None-grad sets, warm start, world 4 / 8
def _spawn(fn, world, *args, timeout=240):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=fn, args=(r, world, port, q, *args)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=timeout) for _ in procs]
    for p in procs:
        p.join(60)
    return sorted(res, key=lambda r: r[0])


def _none_grad_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from kinpoly_amd.rollout import _allreduce_grads
    torch.manual_seed(0)
    a, b, c = torch.nn.Linear(3, 2), torch.nn.Linear(3, 2), torch.nn.Linear(3, 2)
    x = torch.ones(4, 3) * (rank + 1)
    loss = a(x).sum() + (b(x).sum() if rank == 0 else 0.0)          # b gets a gradient on rank 0 only, c on no rank
    loss.backward()
    params = list(a.parameters()) + list(b.parameters()) + list(c.parameters())
    _allreduce_grads(params)
    out = [None if p.grad is None else p.grad.clone().numpy() for p in params]
    q.put((rank, out))
    dist.destroy_process_group()


def test_grad_allreduce_over_a_fixed_parameter_list_with_different_none_sets_world2():
    """ranks whose backward reached different parameters used to all_reduce flat buffers of different lengths.  Now every rank sends
    the whole list (zeros where it has no gradient) plus one flag per parameter: a parameter some rank has a gradient for gets the mean everywhere,
    a parameter NO rank has one for keeps grad None (its Adam state must not start counting)."""
    res = _spawn(_none_grad_worker, 2)
    g0, g1 = res[0][1], res[1][1]
    for x, y in zip(g0, g1):
        assert (x is None) == (y is None)
        if x is not None:
            np.testing.assert_array_equal(x, y)
    np.testing.assert_allclose(g0[0], np.full((2, 3), 4 * (1 + 2) / 2.0))          # a.weight: mean of the two ranks' gradients
    np.testing.assert_allclose(g0[2], np.full((2, 3), 4 * 1 / 2.0))                # b.weight: rank 0's gradient, rank 1 contributed zeros
    assert g0[4] is None and g0[5] is None                                          # c: untouched on every rank


def _warm_start_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from kinpoly_amd import dataset as D
    from kinpoly_amd import pretrain as P
    from kinpoly_amd.context import TrajARNet
    from kinpoly_amd.model_compiler import DEFAULT_KPM, read_kpm
    from kinpoly_amd.rollout import _allreduce_grads
    from kinpoly_amd.supervised import TorchFK
    kpm = read_kpm(DEFAULT_KPM)
    fk = TorchFK(kpm["body_pos"], kpm["body_parent"], "cpu", dtype=torch.float64)
    rng = np.random.default_rng(1)                                   # ONE data set for the job ...
    feats = {}
    for i, T in enumerate((14, 19, 16)):
        qp = np.zeros((T, 76)); qp[:, 2] = 0.9; qp[:, 3] = 1.0; qp[:, 7:] = 0.1 * np.sin(np.arange(T)[:, None] * 0.3 + rng.uniform(0, 6, 69))
        wb = fk.wbpos(torch.tensor(qp)).reshape(T, 72).numpy()
        hp = np.concatenate([wb[:, 39:42], np.tile([1.0, 0, 0, 0], (T, 1))], 1)
        feats[f"sit-{i}"] = dict(qpos=qp, qvel=np.zeros((T, 75)), head_pose=hp, head_vels=np.zeros((T, 6)), action_one_hot=np.tile([1.0, 0, 0, 0], (T, 1)),
                                 obj_head_relative_poses=np.tile([0.5, 0, 0, 1.0, 0, 0, 0], (T, 1)), obj_pose=np.tile([0.5, 0, 0.4, 1.0, 0, 0, 0], (T, 1)),
                                 wbpos=wb, wbquat=np.zeros((T, 96)), bquat=np.zeros((T, 96)), of_files=["x"] * T)
    ds = D.StateARDataset(feats, fr_num=8, seed=3 + rank)            # ... drawn from with a per-rank stream
    ds.data = {k: [x.double() for x in v] for k, v in ds.data.items()}
    torch.manual_seed(0)
    net = TrajARNet(rnn_hdim=32, mlp_hsize=(32, 16)).double()
    opt = torch.optim.Adam(net.parameters(), lr=5e-3)
    # scheduled sampling 0.5 over 6 batches: with per-rank coins the first coin (which decides whether the context network gets a gradient at all)
    # would differ between the ranks in half of the batches
    P.train_full_supervised(net, opt, fk, ds, num_epoch=3, scheduled_sampling=0.5, num_sample=8, batch_size=4, grad_allreduce=_allreduce_grads, rng=P.job_wide_rng(-1))
    flat = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    steps = sorted({int(s["step"]) for s in opt.state.values()})
    q.put((rank, flat.numpy(), steps, float(flat.abs().sum())))
    dist.destroy_process_group()


def test_warm_start_keeps_replicas_identical_world2():
    """train_full_supervised with gradient all-reduce on two ranks (different clips per rank, ONE job-wide stream of scheduled-sampling coins): context
    and action parameters stay bit-identical across the replicas (per-rank coins made the ranks disagree on which parameters had a gradient)."""
    res = _spawn(_warm_start_worker, 2, timeout=400)
    np.testing.assert_array_equal(res[0][1], res[1][1])
    assert res[0][2] == res[1][2] and np.isfinite(res[0][3])


def _world_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import types
    from kinpoly_amd.context import TrajARNet
    from kinpoly_amd.model_compiler import DEFAULT_KPM, read_kpm
    from kinpoly_amd.nets import MLP, Value
    from kinpoly_amd.rollout import EpisodeSource, RolloutBatch, env_shard
    from kinpoly_amd.update import ParamUpdate
    ids, seed = env_shard(rank, world, 4096)
    torch.manual_seed(0)                                             # identical replicas (AgentAR broadcasts rank 0's parameters)
    net = TrajARNet(rnn_hdim=16, mlp_hsize=(16, 8)).double().refresh_log_std()
    val = Value(MLP(105, (16, 8), "relu")).double()
    kpm = read_kpm(DEFAULT_KPM)
    upd = ParamUpdate(net, val, kpm["body_pos"], kpm["body_parent"], num_optim_epoch=2, num_step_update=2, policy_lr=1e-4)
    g = torch.Generator().manual_seed(seed)                          # this rank's shard of the experience
    N, T = 6, 5
    r = lambda *s: torch.randn(*s, generator=g, dtype=torch.float64)  # noqa: E731
    qpos = torch.zeros(N, T, 76, dtype=torch.float64); qpos[..., 2] = 0.9; qpos[..., 3] = 1.0; qpos[..., 7:] = 0.1 * r(N, T, 69)
    masks = torch.ones(N, T, dtype=torch.float64); masks[:, -1] = 0; masks[rank % N, 1] = 0
    starts = torch.cat([torch.ones(N, 1, dtype=torch.bool), masks[:, :-1] == 0], 1)
    batch = RolloutBatch(states=0.5 * r(N, T, 105), actions=0.1 * r(N, T, 80), rewards=torch.rand(N, T, generator=g, dtype=torch.float64), masks=masks, episode_start=starts,
                         fails=torch.zeros(N, T, dtype=torch.bool), curr_qpos=qpos, gt_target_qpos=qpos + 0.01 * r(N, T, 76), exps=torch.ones(N, T, dtype=torch.float64))
    upd.per_epoch_update()
    info = upd.update_params(batch)
    flat = torch.cat([p.detach().reshape(-1) for p in list(net.parameters()) + list(val.parameters())])
    # the normalised advantages of the whole job have mean 0 / std 1 over ALL ranks' rows, not per rank
    adv = upd.trainer.last_adv.reshape(-1)
    every = [torch.empty_like(adv) for _ in range(world)]
    dist.all_gather(every, adv)
    alladv = torch.cat(every)
    # one job-wide freq_dict
    src = EpisodeSource(dataset=types.SimpleNamespace(takes=[f"t{i}" for i in range(3)]))
    src.record([rank % 3], [rank], [0.5])
    q.put((rank, ids[0], ids[-1], len(ids), seed, flat.numpy(), float(alladv.mean()), float(alladv.std()), float(adv.mean()), src.freq_dict, float(info["surr_loss"])))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [4, 8])
def test_shards_update_and_freq_dict_world_4_and_8(world):
    """BASELINE configs[3]'s 8 x 4096 split without the node: under gloo, rank r owns envs [r * 4096, (r + 1) * 4096) -- disjoint, covering [0, world * 4096) --,
    one ParamUpdate.update_params on per-rank batches (all-gathered advantages, all-reduced gradients) leaves identical parameters on every rank,
    and the finished episodes of all ranks end up in one job-wide freq_dict (agent_ar.py:651-680; common.py:22)."""
    res = _spawn(_world_worker, world, timeout=600)
    assert [r[0] for r in res] == list(range(world))
    cover = []
    for r in res:
        assert r[3] == 4096 and r[4] == 4 + r[0]
        cover.append((r[1], r[2]))
    assert cover == [(k * 4096, (k + 1) * 4096 - 1) for k in range(world)]                      # disjoint and covering [0, world * 4096)
    for r in res[1:]:
        np.testing.assert_array_equal(r[5], res[0][5])                                          # replicas identical after the update
        assert r[9] == res[0][9]                                                                # one freq_dict
    assert sum(len(v) for v in res[0][9].values()) == world
    assert abs(res[0][6]) < 1e-12 and abs(res[0][7] - 1.0) < 1e-12                              # job-wide mean 0 / std 1 ...
    assert max(abs(r[8]) for r in res) > 1e-3                                                   # ... which no single rank's rows have
