"""GPU tests of the vectorised sampler's episode semantics (sample_worker, kin_poly/core/agent_ar.py:518-606) and of the update paths
that consume its TrajBatchEgo fields.  The reference's own memory rows cannot be generated here (they need MuJoCo), so the rows are
checked against what sample_worker defines them to be, field by field, and against the dataset they were drawn from."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STD = np.load(os.path.join(ROOT, "tests", "golden", "standing_neutral.npz"))


def _dataset(n_envs, fr_num=12, seed=3):
    from kinpoly_amd import dataset as D
    from kinpoly_amd import sim as kpsim
    from kinpoly_amd.model_compiler import read_kpm
    fk_sim = kpsim.KpSim(kpsim.KpModel(kpsim.STEP_KPM), n_envs, 0)
    takes = D.synthetic_takes(fk_sim, STD["qpos"], n_per_action=1, T_range=(fr_num + 4, fr_num + 12), body_mass=read_kpm(kpsim.STEP_KPM)["body_mass"], seed=seed)
    return D.StateARDataset(takes, fr_num=fr_num, seed=seed, device=fk_sim.device), fk_sim


def _tracking_policy_actions(env, obs):
    """the kinematic action that reproduces the current pose (step_ar encoding): a stand-in for a trained policy"""
    cur = env.sim.get("qpos")
    a = torch.zeros((env.n, 80), device=env.device)
    a[:, :74] = torch.cat([cur[:, 2:3], obs[:, 1:5], cur[:, 7:]], 1)
    return a


def test_every_episode_draws_a_new_clip_and_feeds_freq_dict():
    from kinpoly_amd.env import BatchedHumanoidAREnv
    from kinpoly_amd.nets import KinPolicy
    from kinpoly_amd.rollout import EpisodeSource, VectorSampler
    n, T, fr = 32, 10, 12
    ds, _ = _dataset(n, fr)
    torch.manual_seed(0)
    env = BatchedHumanoidAREnv(n, 0, mode="train", seed=0)
    src2 = EpisodeSource(dataset=ds, sampling_temp=0.3, sampling_freq=0.5)     # no context network: episodes start on the clip's first frame
    pol = KinPolicy().to(env.device)                                  # random init: episodes end within a step or two -> many turnovers
    sampler = VectorSampler(env, pol, source=src2, pool_depth=T, record_full=True)
    b = sampler.sample(T)
    assert sampler.pool_exhausted == 0
    done = (b.masks == 0)
    n_done = int(done.sum())
    assert n_done > n, "random-init policies should end many episodes"
    # freq_dict got one [percent, fr_start] entry per finished episode, under the take the episode ran on
    assert sum(len(v) for v in src2.freq_dict.values()) == n_done == len(b.episodes["percent"])
    for k, v in src2.freq_dict.items():
        for pc, fs in v:
            assert 0 < pc <= 1.0 and 0 <= fs <= ds.get_seq_len(ds.takes.index(k)) - fr
    # v_metas: (take, fr_start, fr_num) of the clip each row was sampled on; constant inside an episode, re-drawn after a done
    vm = b.v_metas.cpu().numpy(); es = b.episode_start.cpu().numpy(); dn = done.cpu().numpy()
    assert (vm[..., 2] == fr).all()
    same = (vm[:, 1:, :2] == vm[:, :-1, :2]).all(-1)
    assert same[~dn[:, :-1]].all(), "the clip changed in the middle of an episode"
    assert (es[:, 1:] == dn[:, :-1]).all() and es[:, 0].all()
    changed = (~same)[dn[:, :-1]].mean()
    assert changed > 0.5, "finished envs must move to freshly drawn clips"
    # gt_target_qpos = ar_context['qpos'][cur_t + 1] of THAT clip: look it up in the dataset
    gt = b.gt_target_qpos.cpu().numpy()
    t_in_ep = np.zeros((n, T), int)
    for t in range(1, T):
        t_in_ep[:, t] = np.where(es[:, t], 0, t_in_ep[:, t - 1] + 1)
    for e in range(0, n, 5):
        for t in range(T):
            ti, fs = int(vm[e, t, 0]), int(vm[e, t, 1])
            np.testing.assert_allclose(gt[e, t], ds.data["qpos"][ti][fs + t_in_ep[e, t] + 1].numpy(), atol=1e-6)
    # chaining of the recorded rows (agent_ar.py:556-598): res_qpos[t] == curr_qpos[t + 1] and next_state[t] == state[t + 1] inside an episode
    inside = ~dn[:, :-1]
    np.testing.assert_array_equal(b.res_qpos.cpu().numpy()[:, :-1][inside], b.curr_qpos.cpu().numpy()[:, 1:][inside])
    np.testing.assert_array_equal(b.next_states.cpu().numpy()[:, :-1][inside], b.states.cpu().numpy()[:, 1:][inside])
    assert b.cc_action.shape == (n, T, 75) and b.cc_state.shape == (n, T, 784) and torch.isfinite(b.cc_state).all() and (b.exps == 1).all()
    # a new episode starts on its clip's init pose
    st = es.copy(); st[:, 0] = False
    e_idx, t_idx = np.nonzero(st)
    for e, t in list(zip(e_idx, t_idx))[:20]:
        ti, fs = int(vm[e, t, 0]), int(vm[e, t, 1])
        np.testing.assert_allclose(b.curr_qpos[e, t].cpu().numpy()[7:], ds.data["qpos"][ti][fs].numpy()[7:], atol=1e-6)


@pytest.mark.parametrize("lagged", [True, False])
def test_every_episode_runs_on_a_fresh_clip_at_fail_rate_one(lagged):
    """random-init networks fail (almost) every episode after one step; every one of those episodes must still run on its own
    freshly drawn clip through init_context (agent_ar.py:518-535), the pool must not run dry (it cannot, by construction), every done flag is one
    episode in freq_dict, and unused queued clips survive the sample() calls instead of being re-drawn."""
    from kinpoly_amd import sim as kpsim
    from kinpoly_amd.context import PolicyARContext, TrajARNet
    from kinpoly_amd.env import BatchedHumanoidAREnv
    from kinpoly_amd.rollout import EpisodeSource, VectorSampler
    n, T, fr, depth = 64, 12, 12, 3
    ds, fk_sim = _dataset(n, fr, seed=5)
    assert ds.has_objects
    torch.manual_seed(3)
    env = BatchedHumanoidAREnv(n, 0, mode="train", seed=3)
    net = TrajARNet().to(env.device)
    src = EpisodeSource(dataset=ds, ctx_builder=PolicyARContext(net, fk_sim, need_rollout=False, keep_context_feat=False), sampling_temp=0.3, sampling_freq=0.5)
    sampler = VectorSampler(env, net, source=src, pool_depth=depth, record_full=True, lagged=lagged)
    sampler.start()
    D = (2 * depth if lagged else depth) + 1          # lagged: the host read of the ring is taken one period late, the ring holds two periods' worth of clips
    assert src.n_drawn == D * n and env.ctx["qpos"].shape[0] == D * n and "ar_qpos" not in env.ctx
    n_done_total, pairs, first_meta = 0, [], None
    for call in range(3):
        b = sampler.sample(T)
        dn = (b.masks == 0).cpu().numpy(); vm = b.v_metas.cpu().numpy()
        n_done = int(dn.sum()); n_done_total += n_done
        assert float(b.fails.float().mean()) > 0.9, "random-init networks should fail nearly every step"
        assert sampler.pool_exhausted == 0 and int(sampler.ahead.min()) >= 0
        assert len(b.episodes["percent"]) == n_done, "every done flag is one recorded episode (no replays to leave out)"
        # T is a multiple of pool_depth: the last top-up of the call has replaced every clip used so far (lagged: used up to one period ago), and nothing more
        lag_done = int(dn[:, -depth:].sum()) if lagged else 0
        assert src.n_drawn == D * n + n_done_total - lag_done and int(sampler.ahead.min()) == depth
        meta = vm[..., 0].astype(np.int64) * 100000 + vm[..., 1].astype(np.int64)          # (take, fr_start) of the clip every row ran on
        if first_meta is not None:      # the envs' clips continue across calls (same episode unless the last row of the previous call ended it)
            cont = ~last_done
            assert (meta[cont, 0] == first_meta[cont]).all()
        es = b.episode_start.cpu().numpy()
        for e in range(n):
            seq = [meta[e, 0]] + [meta[e, t] for t in range(1, T) if es[e, t]]
            pairs += list(zip(seq[:-1], seq[1:]))
        first_meta, last_done = meta[:, -1], dn[:, -1]
    assert sum(len(v) for v in src.freq_dict.values()) == n_done_total
    assert sampler.top_ups == 3 * T // depth
    # consecutive episodes of an env share (take_ind, fr_start) no more often than independent draws do
    allm = np.array([p[1] for p in pairs]); _, counts = np.unique(allm, return_counts=True)
    chance = float(((counts / counts.sum()) ** 2).sum())
    repeats = float(np.mean([a == b_ for a, b_ in pairs]))
    assert len(pairs) > 1500 and repeats < 3 * chance + 0.01, (repeats, chance)


def test_lazy_init_context_gives_the_training_episode_the_same_start():
    """need_rollout=False / keep_context_feat=False skip work no training episode reads (humanoid_ar_v1.py:88, 339-343): init_qpos / init_qvel
    must be what the full init_context computes."""
    from kinpoly_amd.context import PolicyARContext, TrajARNet
    n, fr = 24, 12
    ds, fk_sim = _dataset(n, fr, seed=2)
    torch.manual_seed(11)
    net = TrajARNet().to(fk_sim.device)
    data = {k: (v.to(fk_sim.device) if torch.is_tensor(v) else v) for k, v in ds.sample_batch(n, use_freq=False).items()}
    full = PolicyARContext(net, fk_sim).init_context(data)
    lazy = PolicyARContext(net, fk_sim, need_rollout=False, keep_context_feat=False).init_context(data)
    assert "ar_qpos" in full and "ar_qpos" not in lazy and "context_feat_rnn" not in lazy
    np.testing.assert_allclose(lazy["init_qpos"].cpu().numpy(), full["init_qpos"].cpu().numpy(), atol=2e-6)
    np.testing.assert_allclose(lazy["init_qvel"].cpu().numpy(), full["init_qvel"].cpu().numpy(), atol=2e-6)
    # any batch size goes through the kinematic twin in chunks
    m = 2 * n + 5
    data2 = {k: (v.to(fk_sim.device) if torch.is_tensor(v) else v) for k, v in ds.sample_batch(m, use_freq=False).items()}
    big = PolicyARContext(net, fk_sim).init_context(data2)
    assert big["ar_qpos"].shape == (m, fr, 76) and torch.isfinite(big["ar_qpos"]).all()
    one = PolicyARContext(net, fk_sim).init_context({k: (v[n:2 * n] if torch.is_tensor(v) and v.shape[:1] == (m,) else v) for k, v in data2.items()})
    np.testing.assert_allclose(big["ar_qpos"][n:2 * n].cpu().numpy(), one["ar_qpos"].cpu().numpy(), atol=1e-6)


def test_init_context_memo_returns_what_init_context_computes():
    """EpisodeSource(cache_init_context=True): init_qpos / init_qvel of a window (take, fr_start) are looked up once they have been computed under
    the same context-network parameters; the values are those of a plain init_context call, the memo is emptied when the parameters change."""
    from kinpoly_amd.context import PolicyARContext, TrajARNet
    from kinpoly_amd.rollout import EpisodeSource
    n, fr = 48, 12
    ds, fk_sim = _dataset(n, fr, seed=6)
    ds2, _ = _dataset(n, fr, seed=6)
    torch.manual_seed(5)
    net = TrajARNet().to(fk_sim.device)
    builder = PolicyARContext(net, fk_sim, need_rollout=False, keep_context_feat=False)
    plain = EpisodeSource(dataset=ds, ctx_builder=builder)
    memo = EpisodeSource(dataset=ds2, ctx_builder=builder, cache_init_context=True)
    assert memo.cache_init_context
    for _ in range(3):                              # the same draw stream on both sides (same dataset seed)
        a, b = plain.draw(n, fk_sim.device), memo.draw(n, fk_sim.device)
        assert torch.equal(a["take_ind"], b["take_ind"]) and torch.equal(a["fr_start"], b["fr_start"])
        np.testing.assert_allclose(b["init_qpos"].cpu().numpy(), a["init_qpos"].cpu().numpy(), atol=2e-6)
        np.testing.assert_allclose(b["init_qvel"].cpu().numpy(), a["init_qvel"].cpu().numpy(), atol=2e-6)
    assert memo.n_memo_hits > n, "4 takes x a handful of window starts: most of 3 x 48 draws repeat a window"
    with torch.no_grad():
        net.context_fc.bias.add_(0.01)              # the parameters moved: everything is recomputed
    hits = memo.n_memo_hits
    a, b = plain.draw(n, fk_sim.device), memo.draw(n, fk_sim.device)
    np.testing.assert_allclose(b["init_qpos"].cpu().numpy(), a["init_qpos"].cpu().numpy(), atol=2e-6)
    assert memo.n_memo_hits - hits < n and int(memo._memo["have"].sum()) <= n


def test_object_pose_of_the_observation_follows_the_simulated_object():
    """env.py's object bookkeeping lives behind the C ABI now (kp_sim_reset_rows / kp_sim_post_step): obj7 = get_obj_qpos(action_one_hot)
    (humanoid_ar_v1.py:466-477) is the action's slice of the simulator's data.qpos[76:111] after every reset and step, [0,0,0,1,0,0,0] for a
    clip without action, and the object block of a reset is convert_obj_qpos of the env's CURRENT row (:479-496)."""
    from kinpoly_amd.env import ACTION_INDEX_MAP, BatchedHumanoidAREnv, convert_obj_qpos
    n, fr = 32, 12
    ds, _ = _dataset(n, fr, seed=9)
    env = BatchedHumanoidAREnv(n, 0, mode="train", seed=9)
    ctx = {k: (v.to(env.device) if torch.is_tensor(v) else v) for k, v in ds.sample_batch(2 * n, use_freq=False).items()}
    ctx["action_one_hot"][:5] = 0                                   # a few clips without action
    ctx["init_qpos"], ctx["init_qvel"] = ctx["qpos"][:, 0].contiguous(), ctx["qvel"][:, 0].contiguous()
    env.load_context(ctx, row=torch.arange(n, dtype=torch.int32))   # 2 rows per env; start on the first n
    one_hot = ctx["action_one_hot"][:, 0] if ctx["action_one_hot"].dim() == 3 else ctx["action_one_hot"]
    blk_all, _ = convert_obj_qpos(one_hot, ctx["obj_pose"][:, 0])

    def expect(rows):
        sim35 = env.sim.get("obj_qpos")
        st = torch.tensor(ACTION_INDEX_MAP, device=env.device)[one_hot[rows].argmax(1)]
        got = torch.gather(sim35, 1, st[:, None] + torch.arange(7, device=env.device)[None])
        none = torch.tensor([0.0, 0, 0, 1, 0, 0, 0], device=env.device).expand(n, 7)
        return torch.where(one_hot[rows].sum(1, keepdim=True) > 0, got, none), sim35
    env.reset()
    rows = env.row.long()
    want, sim35 = expect(rows)
    assert torch.equal(env.obj7, want) and torch.equal(sim35, blk_all[rows])
    a = torch.zeros((n, 80), device=env.device); a[:, :74] = torch.cat([ctx["qpos"][:n, 0, 2:3], env._obs[:, 1:5], ctx["qpos"][:n, 0, 7:]], 1)
    for _ in range(2):
        env.step(a.contiguous())
        want, _ = expect(rows)
        assert torch.equal(env.obj7[5:], want[5:])
        assert torch.equal(env.obj7[:5], torch.tensor([0.0, 0, 0, 1, 0, 0, 0], device=env.device).expand(5, 7))
    # half of the envs move to their second row and are reset: their objects are that row's, the others keep their simulated state
    mask = torch.arange(n, device=env.device) % 2 == 0
    before = env.sim.get("obj_qpos").clone()
    env.set_rows(torch.arange(n, 2 * n, device=env.device), mask)
    env.reset(mask)
    rows2 = env.row.long()
    want, sim35 = expect(rows2)
    assert torch.equal(env.obj7, want)
    assert torch.equal(sim35[mask], blk_all[rows2][mask]) and torch.equal(sim35[~mask], before[~mask])


def test_hidden_state_carries_across_calls_and_ppo_ratio_starts_at_one():
    """episodes that continue from the previous sample() call keep their GRU state (RolloutBatch.hx0), so the means the
    update recomputes are the behaviour policy's: with mean actions the recorded actions ARE those means."""
    from kinpoly_amd.env import BatchedHumanoidAREnv, standing_context
    from kinpoly_amd.nets import KinPolicy
    from kinpoly_amd.rollout import VectorSampler
    n, T = 16, 6
    torch.manual_seed(1)
    env = BatchedHumanoidAREnv(n, 0, mode="test", seed=1)
    env.reward_cfg.body_diff_thresh = 1e9                          # nothing fails: episodes span both calls
    env.load_context(standing_context(n, 40, STD["qpos"], STD["qvel"], env.sim))
    pol = KinPolicy().to(env.device)
    with torch.no_grad():
        pol.action_fc.weight.mul_(0.01); pol.action_fc.bias.zero_()
    sampler = VectorSampler(env, pol, mean_action=True)
    b1 = sampler.sample(T)
    b2 = sampler.sample(T)
    assert not bool(b2.episode_start.any()) and float(b2.hx0.abs().max()) > 0
    assert torch.equal(b2.states[:, 0], b1.last_states)
    with torch.no_grad():
        m2 = pol.unroll(b2.states, b2.episode_start, b2.hx0)
        m2_wrong = pol.unroll(b2.states, b2.episode_start)
    err, err_wrong = float((m2 - b2.actions).abs().max()), float((m2_wrong - b2.actions).abs().max())
    assert err < 1e-6 and err_wrong > 20 * max(err, 1e-8), (err, err_wrong)
    # horizon cut: the last row keeps mask 1 and GAE bootstraps with V(last_states) instead of treating it as terminal
    from kinpoly_amd import sim as kpsim
    assert float(b2.masks[:, -1].min()) == 1.0
    v = torch.rand((n, T), device=env.device); lv = torch.rand(n, device=env.device)
    adv_b, _ = kpsim.gae(b2.rewards.contiguous(), b2.masks.contiguous(), v, 0.95, 0.95, lv)
    adv_0, _ = kpsim.gae(b2.rewards.contiguous(), b2.masks.contiguous(), v, 0.95, 0.95)
    want_last = b2.rewards[:, -1] + 0.95 * lv - v[:, -1]
    assert torch.allclose(adv_b[:, -1], want_last, atol=1e-6) and torch.allclose(adv_0[:, -1], b2.rewards[:, -1] - v[:, -1], atol=1e-6)


def test_lr_schedules_and_controller_update():
    """LambdaLR of agent_ar.py:215-225 (nepoch_fix, nepoch) stepped per iteration.  joint_controller: the reference's optimiser holds
    policy_net only (agent_ar.py:184-199), so its update_controller (:774-794) never moves the UHC -- the default here; `train_uhc`
    is the opt-in extension with the UHC's own optimiser and clip."""
    from kinpoly_amd.agent import AgentAR
    from kinpoly_amd.env import standing_context
    from kinpoly_amd import sim as kpsim
    n, T = 32, 8
    fk_sim = kpsim.KpSim(kpsim.KpModel(), n, 0)

    def context_fn(m):
        ctx = standing_context(m, T + 2, STD["qpos"], STD["qvel"], fk_sim, torch.zeros(m))
        ctx["obj_pose"] = torch.tensor([0.0, 0, 0, 1, 0, 0, 0], device=ctx["qpos"].device).repeat(m, T + 2, 1)
        return ctx
    for train_uhc in (False, True):
        agent = AgentAR(n, context_fn, device=0, horizon=T, num_optim_epoch=2, num_step_update=1, use_init_context=False,
                        num_epoch_fix=1, num_epoch=4, joint_controller=True, pool_depth=T, train_uhc=train_uhc)
        held = {id(p) for g in agent.trainer.opt_p.param_groups for p in g["params"]}
        assert held == {id(p) for p in agent.policy_net.parameters() if p.requires_grad}, "optimizer_policy = Adam(policy_net.parameters())"
        cc_before = [p.detach().clone() for p in agent.env.cc_policy.parameters()]
        lrs = []
        for it in range(4):
            info = agent.optimize_policy(it)
            lrs.append(info["policy_lr"])
            assert np.isfinite(info["surr_loss"]) and np.isfinite(info["cc_surr_loss"]) and np.isfinite(info["step_loss"])
        rule = [1e-5 * (1.0 - max(0, e - 1) / float(4 - 1 + 1)) for e in (1, 2, 3, 4)]      # epoch counter after per_epoch_update
        np.testing.assert_allclose(lrs, rule, rtol=1e-6)
        moved = sum(float((p.detach() - q).abs().max()) for p, q in zip(agent.env.cc_policy.parameters(), cc_before))
        assert (moved > 0) == train_uhc, f"train_uhc={train_uhc}: UHC parameters moved by {moved}"
        del agent


def test_joint_policy_update_runs_both_forms():
    """update_policy_joint (agent_ar.py:796-850, `grad_joint`): loss = 10 * supervised step loss + PPO surrogate in one policy step, and the
    `grad_alternate` form (odd epochs surrogate, even epochs supervised)."""
    from kinpoly_amd.agent import AgentAR
    from kinpoly_amd.env import standing_context
    from kinpoly_amd import sim as kpsim
    n, T = 32, 6
    fk_sim = kpsim.KpSim(kpsim.KpModel(), n, 0)

    def context_fn(m):
        ctx = standing_context(m, T + 2, STD["qpos"], STD["qvel"], fk_sim, torch.zeros(m))
        ctx["obj_pose"] = torch.tensor([0.0, 0, 0, 1, 0, 0, 0], device=ctx["qpos"].device).repeat(m, T + 2, 1)
        return ctx
    for alt in (False, True):
        agent = AgentAR(n, context_fn, device=0, horizon=T, num_optim_epoch=2, use_init_context=False, pool_depth=T, grad_joint=True, grad_alternate=alt)
        before = torch.cat([p.detach().reshape(-1) for p in agent.policy_net.parameters()]).clone()
        for it in range(2):
            info = agent.optimize_policy(it)
            assert np.isfinite(info["surr_loss"]) and np.isfinite(info["step_loss"]) and np.isfinite(info["value_loss"])
        after = torch.cat([p.detach().reshape(-1) for p in agent.policy_net.parameters()])
        assert float((after - before).abs().max()) > 0


def test_two_rank_agent_keeps_parameters_identical():
    """SURVEY 8(e) on one device: two ranks (gloo, both on cuda:0) shard the envs, all-gather advantages / returns, all-reduce
    gradients; after optimize_policy the policy / value parameters are bit-identical across ranks and differ from the initial ones."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", KP_SHARED_DEVICE="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29541",
           os.path.join(ROOT, "tests", "_ddp_agent_worker.py")]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "DDP_AGENT_OK" in out.stdout, out.stdout[-2000:]


def test_sampler_rows_match_the_cpu_episode_loop():
    """a1 as a PARITY test (VERDICT r2 next #3a): VectorSampler's TrajBatchEgo rows against oracle/episode.py, the CPU restatement of
    sample_worker (agent_ar.py:538-606) composed of the pinned np_oracle pieces, the fp64 C physics and fp64 copies of both policies.  Two
    envs x 14 steps with mean actions: env 0 tracks its 7-frame clip and ends it twice ('end' at cur_t = 6, reset, again); env 1's clip has
    its GT a metre above the humanoid, so the GT-diff termination (train mode, humanoid_ar_v1.py:303-306) fails it on every step.  All
    twelve memory fields, field by field, plus the GRU state being zeroed at every episode start."""
    import copy
    from kinpoly_amd.env import BatchedHumanoidAREnv, standing_context
    from kinpoly_amd.model_compiler import STEP_KPM, read_kpm
    from kinpoly_amd.nets import KinPolicy
    from kinpoly_amd.rollout import VectorSampler
    from oracle.episode import EpisodeOracle
    n, L, T = 2, 7, 14
    torch.manual_seed(7)
    env = BatchedHumanoidAREnv(n, 0, mode="train", joint_controller=True, seed=7)        # joint_controller: the UHC acts with its mean (humanoid_ar_v1.py:267-268)
    ctx = standing_context(n, L, STD["qpos"], STD["qvel"], env.sim, torch.tensor([0.4, -1.1]))
    ctx["qpos"][1, :, 2] += 1.0                                      # env 1: the GT clip floats a metre above the state the episode starts in
    env.load_context(ctx)
    pol = KinPolicy().to(env.device)
    obs0 = env.reset().clone()
    with torch.no_grad():                                            # a policy whose mean tracks the standing pose, modulated a little by its GRU / MLP path
        pol.action_fc.weight.mul_(0.02); pol.action_fc.bias.zero_()
        q0 = ctx["init_qpos"]
        pol.action_fc.bias[:74] = torch.cat([q0[0, 2:3], torch.tensor([1.0, 0, 0, 0], device=env.device), q0[0, 7:]])
    sampler = VectorSampler(env, pol, mean_action=True, record_full=True)
    b = sampler.sample(T)
    kpm = read_kpm(STEP_KPM)
    ep = EpisodeOracle(kpm, copy.deepcopy(pol).double().cpu(), copy.deepcopy(env.cc_policy).double().cpu())
    c = {k: v.double().cpu().numpy() for k, v in ctx.items()}
    saw_end = saw_fail = False
    for e in range(n):
        one = {k: (c[k][e] if c[k].ndim > 1 else c[k]) for k in ("qpos", "head_pose", "head_vels", "obj_head_relative_poses", "action_one_hot", "init_qpos", "init_qvel")}
        want = ep.rollout(one, T)
        g = lambda x: x[e].double().cpu().numpy()      # noqa: E731
        np.testing.assert_array_equal(g(b.masks), want["mask"])
        np.testing.assert_array_equal(g(b.episode_start).astype(bool), want["episode_start"].astype(bool))
        np.testing.assert_array_equal(g(b.fails).astype(bool), want["fail"].astype(bool))
        np.testing.assert_array_equal(g(b.exps), want["exp"])
        np.testing.assert_array_equal(g(b.v_metas), want["v_meta"])
        np.testing.assert_allclose(g(b.gt_target_qpos), want["gt_target_qpos"], atol=1e-6)
        np.testing.assert_allclose(g(b.states), want["state"], atol=5e-06)        # measured 3.9e-07
        np.testing.assert_allclose(g(b.actions), want["action"], atol=1e-06)        # measured 5.8e-08
        np.testing.assert_allclose(g(b.curr_qpos), want["curr_qpos"], atol=5e-06)        # measured 3.9e-07
        np.testing.assert_allclose(g(b.res_qpos), want["res_qpos"], atol=5e-06)        # measured 3.9e-07
        np.testing.assert_allclose(g(b.next_states), want["next_state"], atol=5e-06)        # measured 3.9e-07
        np.testing.assert_allclose(g(b.rewards), want["reward"], atol=1e-06)        # measured 4.3e-08
        np.testing.assert_allclose(g(b.cc_state), want["cc_state"], atol=0.0002)        # measured 1.0e-05
        np.testing.assert_allclose(g(b.cc_action), want["cc_action"], atol=1e-06)        # measured 5.8e-09
        # the first rows (no accumulated fp32 / fp64 drift yet) at kernel accuracy
        np.testing.assert_allclose(g(b.states)[0], want["state"][0], atol=3e-06)        # measured 2.0e-07
        np.testing.assert_allclose(g(b.res_qpos)[0], want["res_qpos"][0], atol=2e-06)        # measured 2.0e-07
        saw_end |= bool((want["done"] & ~want["fail"]).any()); saw_fail |= bool(want["fail"].any())
    assert saw_end and saw_fail
    assert (b.masks[0] == 0).sum() == 2 and (b.masks[1] == 0).all()


def test_sampler_rows_match_the_cpu_episode_loop_with_action_objects():
    """a1 parity with the scene's free objects (round 4): two envs whose clips carry an action -- `step` (the step box 0.55 m ahead of the humanoid)
    and `push` (the box on the table, table 0.75 m ahead) -- against oracle/episode.py, which places the objects by convert_obj_qpos at every
    episode start, simulates them as free bodies of the fp64 C physics and feeds get_obj_qpos(action_one_hot) (the SIMULATED pose of the
    action's first object) into get_ar_obs_v1.  On the device that pose reaches the observation through kp_sim_reset_rows / kp_sim_post_step
    (no torch bookkeeping any more): states and next_states carry it in columns 81:88."""
    import copy
    from kinpoly_amd.env import BatchedHumanoidAREnv, standing_context
    from kinpoly_amd.model_compiler import STEP_KPM, read_kpm
    from kinpoly_amd.nets import KinPolicy
    from kinpoly_amd.rollout import VectorSampler
    from oracle.episode import EpisodeOracle
    n, L, T = 2, 6, 11
    torch.manual_seed(9)
    env = BatchedHumanoidAREnv(n, 0, mode="train", joint_controller=True, seed=9)
    ctx = standing_context(n, L, STD["qpos"], STD["qvel"], env.sim, torch.tensor([0.0, 0.0]))
    x0, y0 = float(STD["qpos"][0]), float(STD["qpos"][1])
    obj = torch.zeros((n, L, 14), device=env.device)
    obj[0, :, :7] = torch.tensor([x0, y0 + 0.95, 0.3705, 1, 0, 0, 0.0], device=env.device)                    # step box ahead (+y is forward)
    obj[1, :, :7] = torch.tensor([x0, y0 + 0.9, 0.921, 1, 0, 0, 0.0], device=env.device)                      # box on ...
    obj[1, :, 7:] = torch.tensor([x0, y0 + 0.9, 0.7905, 1, 0, 0, 0.0], device=env.device)                     # ... the table
    ctx["obj_pose"] = obj
    ctx["action_one_hot"] = torch.tensor([[0.0, 0, 0, 1], [0.0, 1, 0, 0]], device=env.device)
    env.load_context(ctx)
    assert env.obj7 is not None
    pol = KinPolicy().to(env.device)
    env.reset()
    with torch.no_grad():
        pol.action_fc.weight.mul_(0.02); pol.action_fc.bias.zero_()
        q0 = ctx["init_qpos"]
        pol.action_fc.bias[:74] = torch.cat([q0[0, 2:3], torch.tensor([1.0, 0, 0, 0], device=env.device), q0[0, 7:]])
    sampler = VectorSampler(env, pol, mean_action=True, record_full=True)
    b = sampler.sample(T)
    kpm = read_kpm(STEP_KPM)
    ep = EpisodeOracle(kpm, copy.deepcopy(pol).double().cpu(), copy.deepcopy(env.cc_policy).double().cpu(), kpm_path=STEP_KPM)
    c = {k: v.double().cpu().numpy() for k, v in ctx.items()}
    for e in range(n):
        one = {k: (c[k][e] if c[k].ndim > 1 else c[k]) for k in ("qpos", "head_pose", "head_vels", "obj_head_relative_poses", "action_one_hot", "init_qpos", "init_qvel", "obj_pose")}
        want = ep.rollout(one, T)
        g = lambda x: x[e].double().cpu().numpy()      # noqa: E731
        np.testing.assert_array_equal(g(b.masks), want["mask"])
        np.testing.assert_array_equal(g(b.fails).astype(bool), want["fail"].astype(bool))
        np.testing.assert_allclose(g(b.states), want["state"], atol=5e-6)
        np.testing.assert_allclose(g(b.next_states), want["next_state"], atol=5e-6)
        np.testing.assert_allclose(g(b.res_qpos), want["res_qpos"], atol=5e-6)
        np.testing.assert_allclose(g(b.rewards), want["reward"], atol=1e-6)
        np.testing.assert_allclose(g(b.cc_action), want["cc_action"], atol=1e-6)
        # the object block of the observation moves with the simulated object (it settles on the floor / the table by ~0.1 mm): not the clip's constant pose
        assert np.abs(want["next_state"][:, 81:84] - want["state"][0, 81:84]).max() > 1e-5
        assert (want["mask"] == 0).sum() >= 1, "the clip must end (and the objects be re-placed) inside the window"
