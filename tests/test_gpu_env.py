"""GPU: the batched HumanoidAREnv (kinpoly_amd/env.py) end to end against the composed oracle, the single-env
numpy facade, the vectorised sampler and one PPO update."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from kinpoly_amd.model_compiler import DEFAULT_KPM, read_kpm  # noqa: E402
from oracle import np_oracle as O  # noqa: E402
from oracle.kpo import OracleSim  # noqa: E402

KPM = read_kpm(DEFAULT_KPM)
BP, BI, PAR = KPM["body_pos"].reshape(24, 3), KPM["body_ipos"].reshape(24, 3), KPM["body_parent"]
STD = np.load(os.path.join(os.path.dirname(__file__), "golden", "standing_neutral.npz"))


def _mk_env(n, T=12, mode="test", seed=0):
    from kinpoly_amd.env import BatchedHumanoidAREnv, standing_context
    torch.manual_seed(seed)
    env = BatchedHumanoidAREnv(n, 0, mode=mode, seed=seed)
    rng = np.random.default_rng(seed)
    headings = torch.tensor(rng.uniform(-np.pi, np.pi, size=n), dtype=torch.float32)
    ctx = standing_context(n, T, STD["qpos"], STD["qvel"], env.sim, headings)
    env.load_context(ctx)
    return env, ctx


def test_env_step_matches_composed_oracle():
    """One full HumanoidAREnv.step (test mode => mean UHC action) vs step_ar + qpos_fk + obs_cc + PolicyMCP (fp64) +
    do_simulation (C oracle) + obs_ar + reward, all on the CPU in fp64."""
    n = 8
    env, ctx = _mk_env(n)
    obs0 = env.reset().double().cpu().numpy()
    rng = np.random.default_rng(3)
    q0 = ctx["init_qpos"].double().cpu().numpy(); v0 = ctx["init_qvel"].double().cpu().numpy()
    head_pose = ctx["head_pose"].double().cpu().numpy(); head_vels = ctx["head_vels"].double().cpu().numpy()
    obj_rel = ctx["obj_head_relative_poses"].double().cpu().numpy()
    gt_qpos = ctx["qpos"].double().cpu().numpy()
    # kinematic action close to the current pose (what a trained policy emits)
    a = np.zeros((n, 80))
    for i in range(n):
        cur = q0[i].copy(); cur[3:7] = O.de_heading(cur[3:7])
        a[i, :74] = cur[2:] + rng.normal(size=74) * 0.02
        a[i, 74:] = rng.normal(size=6) * 0.1
    obs, _, done, info = env.step(torch.tensor(a, dtype=torch.float32, device=env.device))
    obs = obs.double().cpu().numpy(); cc_action = info["cc_action"].double().cpu().numpy(); cc_state = info["cc_state"].double().cpu().numpy()
    rew = info["custom_reward"].double().cpu().numpy()
    import copy
    mcp = copy.deepcopy(env.cc_policy).double().cpu()
    sim = OracleSim()
    for i in range(n):
        sim.reset(q0[i], v0[i])
        x = {k: sim.get(k) for k in ("qpos", "qvel", "xpos", "xquat", "xipos")}
        xp, xq, xi = x["xpos"].reshape(24, 3), x["xquat"].reshape(24, 4), x["xipos"].reshape(24, 3)
        want0 = O.obs_ar(x["qpos"], xp, xq, head_pose[i, 0], head_vels[i, 0], obj_rel[i, 0], np.zeros(4), None)
        np.testing.assert_allclose(obs0[i], want0, atol=2e-6)                 # measured 2.6e-7
        prev_bquat = O.get_body_quat(x["qpos"]); prev_h = np.concatenate([xp[13], xq[13]])
        tgt = O.qpos_fk(O.step_ar(x["qpos"], a[i]), BP, BI, PAR)
        cco = O.zfilter(O.obs_cc(x["qpos"], x["qvel"], xp, xq, xi, tgt), 0.0, 1.0, 5.0)
        np.testing.assert_allclose(cc_state[i], cco, atol=3e-6)               # measured 4.6e-7
        with torch.no_grad():
            cca = mcp.action_mean(torch.tensor(cco)[None])[0].numpy()
        np.testing.assert_allclose(cc_action[i], cca, atol=1e-6)              # fp32 GEMMs vs fp64: measured 2e-9 (output layer x 0.1)
        sim.do_simulation(cc_action[i], tgt["qpos"], 15)               # same action => isolates the physics
        x = {k: sim.get(k) for k in ("qpos", "qvel", "xpos", "xquat", "xipos")}
        xp, xq = x["xpos"].reshape(24, 3), x["xquat"].reshape(24, 4)
        np.testing.assert_allclose(env.sim.get("qpos")[i].double().cpu().numpy(), x["qpos"], atol=3e-6)      # measured 3.1e-7
        want = O.obs_ar(x["qpos"], xp, xq, head_pose[i, 1], head_vels[i, 1], obj_rel[i, 1], np.zeros(4), None)
        np.testing.assert_allclose(obs[i], want, atol=3e-6)
        gt = O.qpos_fk(gt_qpos[i, 1], BP, BI, PAR); gtp = O.qpos_fk(gt_qpos[i, 0], BP, BI, PAR)
        r, _ = O.dynamic_supervision_v1(np.concatenate([xp[13], xq[13]]), prev_h, O.get_body_quat(x["qpos"]), prev_bquat, xp, tgt, head_pose[i, 1],
                                        gt["bquat"].reshape(-1), gtp["bquat"].reshape(-1), 1 / 30, O.REWARD_WEIGHTS)
        assert abs(rew[i] - r) < 2e-6                                         # measured 1.2e-7
    assert not bool(done.any()) and int(env.cur_t[0]) == 1


def test_single_env_facade_numpy_surface():
    from kinpoly_amd.env import HumanoidAREnv, standing_context
    import types
    cfg = types.SimpleNamespace(policy_specs={"body_diff_thresh": 10, "body_diff_gt_thresh": 12}, joint_controller=False)
    env = HumanoidAREnv(cfg, types.SimpleNamespace(env_episode_len=100000), None, mode="test")
    ctx = standing_context(1, 10, STD["qpos"], STD["qvel"], env.b.sim)
    env.load_context({k: v.cpu() for k, v in ctx.items()} | {"action_one_hot": ctx["action_one_hot"].cpu()[:, None].repeat(1, 10, 1)})
    env.seed(4)
    obs = env.reset()
    assert obs.shape == (105,) and obs.dtype == np.float64
    cur = env.get_humanoid_qpos(); cur[3:7] = O.de_heading(cur[3:7])
    a = np.concatenate([cur[2:], np.zeros(6)])
    obs2, r, done, info = env.step(a)
    assert obs2.shape == (105,) and r == 1.0 and isinstance(done, bool)
    assert info["cc_action"].shape == (75,) and info["cc_state"].shape == (784,) and 0 < info["percent"] <= 1
    assert env.get_humanoid_qpos().shape == (76,) and env.get_head().shape == (7,) and env.get_body_quat().shape == (96,)
    assert env.target["wbpos"].shape == (72,) and env.cur_t == 1 and abs(env.dt - 1 / 30) < 1e-6
    assert env.get_obj_qpos().shape == (35,) and env.get_obj_qvel().shape == (30,)
    np.testing.assert_allclose(env.get_obj_qpos(np.zeros(4)), [0, 0, 0, 1, 0, 0, 0])
    # the rest of the reference surface (humanoid_ar_v1.py:28-112)
    assert env.action_space.shape == (75,) and env.observation_space.shape == (105,) and env.action_space.contains(env.action_space.sample(env.np_random))
    assert env.start_ind == 0 and env.render() is None and env.num_obj == 5 and env.action_names == ["sit", "push", "avoid", "step"]
    assert len(env.model.actuator_names) == 69 and env.model.actuator_names[0] == "L_Hip_z" and env.model.actuator_names[-1] == "R_Hand_x"
    assert env.model._body_name2id["Head"] == 14 and env.model.nu == 69
    gt = env.gt_targets
    assert gt["wbpos"].shape == (10, 24, 3) and gt["wbquat"].shape == (10, 24, 4) and gt["bquat"].shape == (10, 96)
    want = O.qpos_fk(STD["qpos"], BP, BI, PAR)
    np.testing.assert_allclose(gt["wbpos"][3], want["wbpos"], atol=2e-06)        # measured 1.4e-07
    assert isinstance(env.np_random.uniform(), float) and env.bquat.shape == (96,)


def test_wild_mode_env_runs_without_gt_termination(golden):
    """BASELINE configs[4] (`--wild`, eval_ar_policy.py:265-395): wild=True builds the env on ..._mesh_all.xml (no step box), mode
    'test' = mean actions, and the GT-based termination of humanoid_ar_v1.py:303-306 is off: an env whose GT clip is far from
    the simulated pose keeps running where the train-mode env fails, while the target-based test (:307-309) still applies."""
    from kinpoly_amd.env import BatchedHumanoidAREnv, standing_context
    n, T = 16, 12
    torch.manual_seed(0)
    envs = {}
    for name, kw in (("wild", dict(mode="test", wild=True)), ("train", dict(mode="train", wild=False))):
        env = BatchedHumanoidAREnv(n, 0, seed=0, **kw)
        ctx = standing_context(n, T, STD["qpos"], STD["qvel"], env.sim)
        # GT clip displaced by 1 m: sum_j |xpos_j - gt_j| = 24 m > body_diff_gt_thresh = 12
        far = ctx["qpos"].clone(); far[:, :, 0] += 1.0
        ctx["qpos"] = far
        env.load_context(ctx)
        envs[name] = env
    assert envs["wild"].model.get_option("timestep") == envs["train"].model.get_option("timestep")
    assert envs["wild"].reward_cfg.use_gt_term == 0 and envs["train"].reward_cfg.use_gt_term == 1
    acts = {}
    for name, env in envs.items():
        obs = env.reset().clone()
        cur = env.sim.get("qpos")
        a = torch.zeros((n, 80), device=env.device)
        a[:, :74] = torch.cat([cur[:, 2:3], obs[:, 1:5], cur[:, 7:]], 1)      # track the current pose (step_ar encoding)
        obs2, _, done, info = env.step(a)
        acts[name] = (done.clone(), info["fail"].clone(), info["cc_action"].clone(), info["body_diff"].clone())
        assert torch.isfinite(obs2).all()
    assert not bool(acts["wild"][1].any()), "wild mode must not terminate on the GT clip"
    assert bool(acts["train"][1].all()), "train mode terminates when the GT clip is > body_diff_gt_thresh away"
    assert float(acts["train"][3][:, 1].min()) > 12.0
    # test mode = mean UHC actions: deterministic, identical across the batch rows that share a state
    assert torch.allclose(acts["wild"][2][0], acts["wild"][2][1], atol=1e-6)
    assert int(envs["wild"].sim.diag()[:, 2].max()) == 0


def test_ar_mode_and_fail_safe():
    """ar_mode (eval_ar_policy.py --ar_mode): reset from ar_qpos[0], the UHC tracks ar_qpos[t + 1] instead of the policy's
    step_ar output; ar_fail_safe puts the humanoid back on the kinematic roll-out (humanoid_ar_v1.py:263-264, 327-331, 339-341)."""
    from kinpoly_amd.env import HumanoidAREnv, standing_context
    import types
    cfg = types.SimpleNamespace(policy_specs={"body_diff_thresh": 10, "body_diff_gt_thresh": 12}, joint_controller=False)
    env = HumanoidAREnv(cfg, types.SimpleNamespace(env_episode_len=100000), None, mode="test", ar_mode=True)
    T = 8
    ctx = standing_context(1, T, STD["qpos"], STD["qvel"], env.b.sim)
    ar_qpos = ctx["qpos"].clone(); ar_qpos[0, :, 7 + 3 * 15 + 1] += torch.linspace(0, 0.4, T, device=ar_qpos.device)   # the roll-out raises an arm
    ar_qvel = torch.zeros((1, T, 75), device=ar_qpos.device)
    d = {k: v.cpu() for k, v in ctx.items()} | {"action_one_hot": ctx["action_one_hot"].cpu()[:, None].repeat(1, T, 1), "ar_qpos": ar_qpos.cpu(), "ar_qvel": ar_qvel.cpu()}
    del d["init_qpos"], d["init_qvel"]
    env.load_context(d)
    env.reset()
    np.testing.assert_allclose(env.get_humanoid_qpos(), ar_qpos[0, 0].cpu().numpy(), atol=1e-6)
    junk = np.zeros(80)                                            # the kinematic action is ignored for the target in ar_mode
    env.step(junk)
    np.testing.assert_allclose(env.target["qpos"], ar_qpos[0, 1].cpu().numpy(), atol=1e-6)
    env.step(junk)
    np.testing.assert_allclose(env.target["qpos"], ar_qpos[0, 2].cpu().numpy(), atol=1e-6)
    env.ar_fail_safe()
    np.testing.assert_allclose(env.get_humanoid_qpos(), ar_qpos[0, 3].cpu().numpy(), atol=1e-6)
    np.testing.assert_allclose(env.get_humanoid_qvel(), 0.0, atol=1e-7)


def test_sampler_autoreset_and_ppo_update():
    from kinpoly_amd.nets import KinPolicy, MLP, Value
    from kinpoly_amd.rollout import PPOTrainer, VectorSampler
    n, T = 64, 12
    env, _ = _mk_env(n, T=8, mode="train", seed=1)      # clip length 8 => episodes end every 7 steps
    torch.manual_seed(1)
    policy = KinPolicy().to(env.device); value = Value(MLP(105, (512, 256), "relu")).to(env.device)
    sampler = VectorSampler(env, policy, record_qpos=True)
    batch = sampler.sample(T)
    assert batch.states.shape == (n, T, 105) and torch.isfinite(batch.states).all() and torch.isfinite(batch.rewards).all()
    m = batch.masks.cpu().numpy()
    assert (m[:, -1] == 0).all() and (m == 0).sum() >= n          # horizon cut + at least one episode end per env
    es = batch.episode_start.cpu().numpy()
    assert es[:, 0].all() and (es[:, 1:] == (m[:, :-1] == 0)).all()
    assert (batch.rewards >= 0).all() and (batch.rewards <= 1.0001).all()
    # the step right after an auto-reset starts from the clip's initial state
    assert torch.allclose(batch.curr_qpos[:, 0], batch.curr_qpos[torch.arange(n), (torch.tensor(m[:, :-1] == 0).float().argmax(1) + 1).to(env.device)], atol=1e-5)
    tr = PPOTrainer(policy, value, num_optim_epoch=2)
    before = [p.detach().clone() for p in policy.parameters() if p.requires_grad]
    stats = tr.update(batch)
    assert np.isfinite(stats["value_loss"]) and np.isfinite(stats["surr_loss"])
    assert any((a - b).abs().max() > 0 for a, b in zip(before, [p for p in policy.parameters() if p.requires_grad]))
    # supervised one-step update (step_update: true): the kinematic one-step loss goes down
    from kinpoly_amd.supervised import TorchFK, compute_loss_lite, kinematic_step, update_supervised_step
    fk = TorchFK(KPM["body_pos"], KPM["body_parent"], env.device)
    opt = torch.optim.Adam([p for p in policy.parameters() if p.requires_grad], lr=5e-4)
    def cur_loss():
        with torch.no_grad():
            m = policy.unroll(batch.states, batch.episode_start).reshape(n * T, -1)
            return float(compute_loss_lite(fk, kinematic_step(batch.curr_qpos.reshape(n * T, 76), m), batch.gt_target_qpos.reshape(n * T, 76))[0])
    l0 = cur_loss()
    update_supervised_step(policy, opt, fk, batch, num_epoch=5)
    assert cur_loss() < l0


def test_env_with_step_object():
    """action_one_hot = 'step': the 40 kg step box is a free body of the env -- the humanoid stands on it, its simulated
    pose enters the AR observation every step (get_obj_qpos reads data.qpos)."""
    from kinpoly_amd.env import BatchedHumanoidAREnv, standing_context
    n, T = 4, 6
    env = BatchedHumanoidAREnv(n, 0, mode="test", seed=0)
    ctx = standing_context(n, T, STD["qpos"], STD["qvel"], env.sim)
    one_hot = torch.zeros((n, 4), device=env.device); one_hot[:2, 3] = 1.0          # envs 0, 1: "step"
    pose = torch.tensor([STD["qpos"][0], STD["qpos"][1], 0.3705, 1.0, 0, 0, 0], device=env.device).repeat(n, 1)  # box resting on the floor, top at 0.3405
    pose[1, 0] += 1.0                                                             # env 1: the box stands 1 m ahead, untouched
    ctx["action_one_hot"] = one_hot
    ctx["obj_pose"] = pose.unsqueeze(1).repeat(1, T, 1).contiguous()
    ctx["init_qpos"][0, 2] += 0.341; ctx["qpos"][0, :, 2] += 0.341                 # env 0 starts on top of the box
    env.load_context(ctx)
    obs = env.reset().double().cpu().numpy()
    rd = {k: env.sim.get(k).double().cpu().numpy() for k in ("qpos", "xpos", "xquat", "obj_qpos")}
    assert np.allclose(rd["obj_qpos"][0, 28:35], pose[0].cpu().numpy(), atol=1e-6) and rd["obj_qpos"][2, 28] == 500.0
    c = {k: v.double().cpu().numpy() for k, v in env.ctx.items()}
    for i in range(n):
        want = O.obs_ar(rd["qpos"][i], rd["xpos"][i].reshape(24, 3), rd["xquat"][i].reshape(24, 4), c["head_pose"][i, 0], c["head_vels"][i, 0],
                        c["obj_head_relative_poses"][i, 0], c["action_one_hot"][i], rd["obj_qpos"][i, 28:35])
        np.testing.assert_allclose(obs[i], want, atol=1e-06)        # measured 6.6e-08
    a = torch.zeros((n, 80), device=env.device)
    q0 = env.sim.get("qpos")
    for i in range(n):
        cur = q0[i].double().cpu().numpy(); cur[3:7] = O.de_heading(cur[3:7]); a[i, :74] = torch.tensor(cur[2:], dtype=torch.float32)
    for _ in range(3):
        obs_t, _, _, _ = env.step(a)
    dg = env.sim.diag()
    assert dg[:, 2].max() == 0
    rd = {k: env.sim.get(k).double().cpu().numpy() for k in ("qpos", "xpos", "xquat", "obj_qpos")}
    z = rd["qpos"][:, 2]
    assert abs(z[0] - z[2] - 0.341) < 0.01        # env 0 still stands on the box, env 2 (no object) on the floor
    assert np.abs(rd["obj_qpos"][:2, 28:35] - pose[:2].cpu().numpy()).max() < 5e-3      # the boxes rest (settle by < 5 mm)
    assert np.abs(rd["obj_qpos"][0, 28:35] - pose[0].cpu().numpy()).max() > 1e-6         # ... but they are simulated, not frozen
    # the observation of step t uses the simulated object pose of step t
    for i in range(2):
        want = O.obs_ar(rd["qpos"][i], rd["xpos"][i].reshape(24, 3), rd["xquat"][i].reshape(24, 4), c["head_pose"][i, 3], c["head_vels"][i, 3],
                        c["obj_head_relative_poses"][i, 3], c["action_one_hot"][i], rd["obj_qpos"][i, 28:35])
        np.testing.assert_allclose(obs_t[i].double().cpu().numpy(), want, atol=3e-06)        # measured 2.6e-07


def test_context_rollout_matches_reference_fixture(golden):
    """Batched PolicyAR.init_context: kinematic roll-out of TrajARNet on the HIP kernels vs the reference's own
    TrajARNet.forward output (tests/golden/traj_ar_net.npz)."""
    from kinpoly_amd import sim as kpsim
    from kinpoly_amd.context import PolicyARContext
    from tests.test_context_cpu import _data, _net_from_fixture
    g = golden("traj_ar_net")
    net = _net_from_fixture(g, torch.float32).cuda()
    data = _data(g, torch.float32, "cuda")
    B, T = data["qpos"].shape[:2]
    kin_sim = kpsim.KpSim(kpsim.KpModel(), B)
    with torch.no_grad():
        init_qpos, init_qvel, _ = net.init_states(data)
    np.testing.assert_allclose(init_qpos.double().cpu().numpy(), g["init_qpos"], atol=3e-06)        # measured 2.6e-07
    q, v, a = net.rollout(data, kin_sim, init_qpos, init_qvel)
    np.testing.assert_allclose(a.double().cpu().numpy(), g["action"], atol=1e-06)        # measured 1.4e-08
    np.testing.assert_allclose(q.double().cpu().numpy(), g["ar_qpos"], atol=3e-06)        # measured 2.7e-07
    np.testing.assert_allclose(v.double().cpu().numpy(), g["ar_qvel"], atol=3e-05)        # measured 2.6e-06
    ctx = PolicyARContext(net, kin_sim, smooth=True).init_context(data)
    assert ctx["ar_qpos"].shape == (B, T, 76) and ctx["ar_wbpos"].shape == (B, T, 72) and torch.isfinite(ctx["ar_bquat"]).all()
    # cfg.smooth in the reference leaves ar_qpos as rolled out (its filter call is a no-op: smooth_effective.npz)
    np.testing.assert_allclose(ctx["ar_qpos"].double().cpu().numpy(), g["ar_qpos"], atol=3e-06)        # measured 2.7e-07
    assert torch.equal(ctx["ar_qpos"], q)
    # the documented deviation: real time-axis smoothing, opt-in
    from scipy.ndimage import gaussian_filter1d
    ctx_s = PolicyARContext(net, kin_sim, smooth=True, smooth_time_axis=True).init_context(data)
    want = gaussian_filter1d(g["ar_qpos"][:, :, 7:], 1, axis=1)
    np.testing.assert_allclose(ctx_s["ar_qpos"][:, :, 7:].double().cpu().numpy(), want, atol=1e-06)        # measured 8.8e-09
    np.testing.assert_allclose(ctx_s["ar_qpos"][:, :, :7].double().cpu().numpy(), g["ar_qpos"][:, :, :7], atol=3e-06)        # measured 2.7e-07


def test_agent_ar_iteration_and_checkpoint(tmp_path):
    """AgentAR.optimize_policy (sample + PPO + supervised step update + re-initialised contexts) runs two iterations
    on the batched engine, and its checkpoint round-trips through the reference pickle layout."""
    from kinpoly_amd import sim as kpsim
    from kinpoly_amd.agent import AgentAR
    from kinpoly_amd.env import standing_context
    n, T = 64, 9
    fk_sim = kpsim.KpSim(kpsim.KpModel(), n, 0)

    def context_fn(m):
        ctx = standing_context(m, T, STD["qpos"], STD["qvel"], fk_sim, torch.zeros(m))
        ctx["obj_pose"] = torch.tensor([0.0, 0, 0, 1, 0, 0, 0], device=ctx["qpos"].device).repeat(m, T, 1)
        return ctx

    agent = AgentAR(n, context_fn, device=0, horizon=8, num_optim_epoch=2, num_step_update=2, use_init_context=True)
    infos = [agent.optimize_policy(i) for i in range(2)]
    for info in infos:
        assert info["num_steps"] == n * 8 and np.isfinite(info["step_loss"]) and np.isfinite(info["value_loss"])
    path = str(tmp_path / "iter_0002.p")
    agent.save_checkpoint(path)
    before = {k: v.clone() for k, v in agent.policy_net.state_dict().items()}
    with torch.no_grad():
        for p in agent.policy_net.parameters():
            p.add_(1.0)
    agent.load_checkpoint(path)
    for k, v in agent.policy_net.state_dict().items():
        assert torch.equal(v, before[k]), k


def test_batched_evaluation_and_coverage_files(tmp_path):
    """run_seq / test_coverage (eval_ar_policy.py:178-262) batched: per-sequence records with the reference's keys, early
    termination handled per env, fail-safe continuation, coverage pickles readable with joblib."""
    import joblib
    from kinpoly_amd.context import TrajARNet
    from kinpoly_amd.env import BatchedHumanoidAREnv, standing_context
    from kinpoly_amd.evaluate import run_sequences, write_coverage
    n, T = 6, 10
    torch.manual_seed(3)
    env = BatchedHumanoidAREnv(n, 0, mode="test", seed=3)
    ctx = standing_context(n, T, STD["qpos"], STD["qvel"], env.sim)
    ctx["ar_qpos"] = ctx["qpos"].clone(); ctx["ar_qvel"] = torch.zeros((n, T, 75), device=env.device)
    env.load_context(ctx)
    env.reward_cfg.body_diff_thresh = 0.8              # a random-init policy drifts: some sequences terminate early
    net = TrajARNet().to(env.device)
    keys = [f"sit-{i:02d}" for i in range(n)]
    res = run_sequences(env, net, keys, fail_safe=False)
    assert set(res) == set(keys)
    for k, r in res.items():
        L = len(r["pred"])
        assert 1 <= L <= T - 1 and len(r["target"]) == L and len(r["obj_pose"]) == L and r["pred"][0].shape == (76,) and r["obj_pose"][0].shape == (35,)
        assert 0 < r["percent"] <= 1 and abs(r["percent"] - L / (T - 1)) < 1e-6 and r["fail_safe"] is False
    assert min(len(r["pred"]) for r in res.values()) < T - 1, "expected at least one early termination in this set-up"
    res_fs = run_sequences(env, net, keys, fail_safe=True)
    assert all(len(r["pred"]) == T - 1 and r["percent"] == 1.0 for r in res_fs.values()) and any(r["fail_safe"] for r in res_fs.values())
    cov = write_coverage(res_fs, str(tmp_path), 750, "features_test")
    full = joblib.load(str(tmp_path / "0750_features_test_coverage_full.pkl")); brief = joblib.load(str(tmp_path / "0750_features_test_coverage.pkl"))
    assert set(full) == set(keys) and set(brief[keys[0]]) == {"percent", "values", "fail_safe"}
    assert cov == sum(1 for r in res_fs.values() if not r["fail_safe"])


def test_ragged_episode_lengths():
    """Clips of different length in one batch: padded to the longest, ctx['len'] gives each env its own end
    (`end = cur_t + start_ind >= ar_context['len']`, humanoid_ar_v1.py:312) and `percent`."""
    from kinpoly_amd.env import BatchedHumanoidAREnv, standing_context
    n, T = 4, 10
    env = BatchedHumanoidAREnv(n, 0, mode="test", seed=0)
    ctx = standing_context(n, T, STD["qpos"], STD["qvel"], env.sim)
    lens = [10, 6, 4, 10]
    ctx["len"] = torch.tensor(lens)
    env.load_context(ctx)
    env.reward_cfg.body_diff_thresh = 1e9                      # no early termination: only the clip ends count
    env.reset()
    a = torch.zeros((n, 80), device=env.device)
    q0 = env.sim.get("qpos")
    for i in range(n):
        cur = q0[i].double().cpu().numpy(); cur[3:7] = O.de_heading(cur[3:7]); a[i, :74] = torch.tensor(cur[2:], dtype=torch.float32)
    ended_at = [None] * n
    for t in range(1, T):
        _, _, done, info = env.step(a)
        for i in range(n):
            if bool(info["end"][i]) and ended_at[i] is None:
                ended_at[i] = t
                assert abs(float(info["percent"][i]) - 1.0) < 1e-6
        assert not bool(info["fail"].any())
    assert ended_at == [L - 1 for L in lens]
    with pytest.raises(ValueError):
        bad = dict(ctx); bad["len"] = torch.tensor([11, 6, 4, 10]); env.load_context(bad)
    # masked reload of some envs' clips with their own lengths (a per-row list, as a caller that builds contexts on the host passes it)
    ctx2 = dict(ctx); ctx2["len"] = [7, 9, 5, 3]
    env.load_context(ctx2, env_mask=torch.tensor([False, True, False, True]))
    assert env.row_len.cpu().tolist() == [9, 8, 3, 2]            # ar_context['len'] = frames - 1; rows 0 and 2 keep theirs


def test_dataset_to_rollout_pipeline_with_objects(tmp_path):
    """The whole caller chain on the reference's feature-file schema: synthetic takes of all four action classes (with their
    objects) -> joblib feature file -> StateARDataset.sample_batch -> batched init_context -> env with free objects -> steps."""
    from kinpoly_amd import dataset as D
    from kinpoly_amd import sim as kpsim
    from kinpoly_amd.context import PolicyARContext, TrajARNet
    from kinpoly_amd.env import BatchedHumanoidAREnv
    n, fr = 16, 12
    torch.manual_seed(1)
    env = BatchedHumanoidAREnv(n, 0, mode="train", seed=1)
    takes = D.synthetic_takes(env.sim, STD["qpos"], n_per_action=1, T_range=(20, 30), body_mass=KPM["body_mass"], seed=2)
    assert sorted(t.split("-")[0] for t in takes) == ["avoid", "push", "sit", "step"]
    path = str(tmp_path / "features" / "synthetic.p")
    D.write_features(path, takes)
    ds = D.StateARDataset(path, fr_num=fr, seed=5, device=env.device)
    batch = ds.sample_batch(n)
    assert batch["qpos"].shape == (n, fr, 76) and batch["obj_pose"].shape == (n, fr, 14) and set(batch["action_one_hot"].sum(2).unique().tolist()) == {1.0}
    net = TrajARNet().to(env.device)
    ctx = PolicyARContext(net, kpsim.KpSim(env.model, n, 0), smooth=True).init_context(batch)
    ctx["init_qpos"] = batch["qpos"][:, 0].contiguous(); ctx["init_qvel"] = batch["qvel"][:, 0].contiguous()      # start on the clip (random-init net)
    env.load_context(ctx)
    obs = env.reset()
    assert torch.isfinite(obs).all()
    objq0 = env.sim.get("obj_qpos").clone()
    a_idx = batch["action_one_hot"][:, 0].argmax(1).cpu().numpy()
    for e in range(n):                                             # the action's object(s) sit at their clip pose, the others are parked
        st = (0, 7, 21, 28)[a_idx[e]]
        np.testing.assert_allclose(objq0[e, st:st + 7].cpu().numpy(), batch["obj_pose"][e, 0, :7].cpu().numpy(), atol=1e-5)
    hx = net.init_hidden(n, env.device)
    for _ in range(4):
        action, hx = net.select_action(obs, hx, True, env.gen)
        obs, _, done, info = env.step(action.contiguous())
        assert torch.isfinite(obs).all() and torch.isfinite(info["custom_reward"]).all()
    assert int(env.sim.diag()[:, 2].max()) == 0
    moved = (env.sim.get("obj_qpos") - objq0).abs().max(1).values
    assert float(moved.max()) < 0.05                               # objects resting on the floor / table stay put (mm-level settling)


@pytest.mark.timeout(240, method="thread")
def test_env_step_captures_into_a_hip_graph():
    """Every entry point of the C ABI only enqueues on the sim's stream, so a whole rollout step (kinematic policy, env.step
    through the library's kernels, masked reset) captures into one hipGraph; replaying it three times gives the same bits
    as three eager steps (test mode: mean actions, no random numbers).  n = 2304 > the wave slots, i.e. the job-queue launch."""
    from kinpoly_amd.nets import KinPolicy
    n = 2304

    def one_step(env, pol, st):
        a, st["hx"] = pol.select_action(st["obs"], st["hx"], True)
        a = a * 0.0 + st["a0"]                                  # keep the untrained policy's output near the current pose
        obs, _, done, _ = env.step(a.contiguous())
        st["obs"] = env.reset(done).clone()
        st["hx"] = st["hx"] * (~done).float().unsqueeze(1)

    def initial(env, ctx):
        obs0 = env.reset().clone()
        q0 = ctx["init_qpos"]
        a0 = torch.zeros((n, 80), device=env.device)
        a0[:, :74] = torch.cat([q0[:, 2:3], obs0[:, 1:5], q0[:, 7:]], 1)    # obs_ar[0:74] = qpos[2:] with the root de-headed
        return obs0, a0

    outs = []
    for use_graph in (False, True):
        side = torch.cuda.Stream()
        with torch.cuda.stream(side), torch.no_grad():
            env, ctx = _mk_env(n, T=12, mode="test", seed=5)
            torch.manual_seed(7)
            pol = KinPolicy().to(env.device).float()
            obs0, a0 = initial(env, ctx)
            st = {"obs": obs0, "hx": pol.init_hidden(n, env.device), "a0": a0}
            if not use_graph:
                for _ in range(3):
                    one_step(env, pol, st)
            else:
                side.synchronize()
                g = torch.cuda.CUDAGraph()
                s_obs, s_hx = st["obs"].clone(), st["hx"].clone()
                st["obs"], st["hx"] = s_obs, s_hx
                # warm the lazily built pieces (fused PolicyMCP weights, hipBLASLt workspaces) outside the capture, then restore
                saved = {k: env.sim.get(k).clone() for k in ("qpos", "qvel", "qpos_d", "qvel_d")}
                cur_t = env.cur_t.clone()
                one_step(env, pol, st)
                env.sim.set_full_state(saved["qpos"], saved["qvel"], saved["qpos_d"], saved["qvel_d"]); env.cur_t.copy_(cur_t)
                obs0b, _ = initial(env, ctx)
                s_obs.copy_(obs0b); s_hx.zero_(); st["obs"], st["hx"] = s_obs, s_hx
                side.synchronize()
                with torch.cuda.graph(g, stream=side):
                    st["obs"], st["hx"] = s_obs, s_hx
                    one_step(env, pol, st)
                    s_obs.copy_(st["obs"]); s_hx.copy_(st["hx"])
                # the capture itself executed nothing: state is still the initial one
                for _ in range(3):
                    g.replay()
                st["obs"], st["hx"] = s_obs, s_hx
            side.synchronize()
            outs.append((st["obs"].cpu().numpy(), env.sim.get("qpos").cpu().numpy(), env.sim.get("qvel").cpu().numpy(), env.cur_t.cpu().numpy()))
    for a_, b_ in zip(*outs):
        assert (a_ == b_).all()
    assert np.isfinite(outs[0][1]).all() and outs[0][3].max() == 3


def test_fk_backward_kernel_matches_autograd():
    """k_fk_wbpos_grad (analytic backward of qpos -> wbpos: subtree force / moment sums projected on the hinge axes, root
    quaternion through its normalisation) against torch autograd through the level-batched TorchFK, on random poses with a
    non-unit root quaternion; then one supervised update through either path gives the same loss trajectory."""
    from kinpoly_amd import sim as kpsim
    from kinpoly_amd.supervised import TorchFK
    dev = torch.device("cuda", 0)
    sim = kpsim.KpSim(kpsim.KpModel(), 8)
    fk = TorchFK(KPM["body_pos"], KPM["body_parent"], dev, sim=sim)
    g = torch.Generator(device="cpu").manual_seed(11)
    B = 777
    q = torch.tensor(STD["qpos"], dtype=torch.float32).repeat(B, 1)
    q[:, :3] += torch.randn(B, 3, generator=g) * 0.5
    q[:, 3:7] = torch.randn(B, 4, generator=g) * (0.5 + torch.rand(B, 1, generator=g))      # any orientation, |q| != 1
    q[:, 7:] = (torch.rand(B, 69, generator=g) * 2 - 1) * 3.0
    w = torch.randn(B, 24, 3, generator=g).to(dev)
    qa = q.to(dev).requires_grad_(True); qb = q.to(dev).double().requires_grad_(True)
    fk64 = TorchFK(KPM["body_pos"], KPM["body_parent"], dev, torch.float64)
    pa, pb = fk.wbpos(qa), fk64.wbpos_torch(qb)
    assert float((pa.double() - pb).detach().abs().max()) < 2e-5
    (pa * w).sum().backward(); (pb * w.double()).sum().backward()
    ga, gb = qa.grad.double(), qb.grad
    assert float((ga - gb).abs().max()) < 2e-4 * float(gb.abs().max())
    assert float((ga[:, 3:7] - gb[:, 3:7]).abs().max()) < 2e-4 * float(gb[:, 3:7].abs().max())


def test_step_outputs_live_as_long_as_the_docstring_says():
    """step() returns views into reused buffers.  done / fail / end / percent / rewards alternate between two sets (valid until the
    next-but-one step), obs and cc_state are single buffers (valid until the next step).  A caller holding them longer must clone."""
    from kinpoly_amd.env import BatchedHumanoidAREnv, standing_context
    n = 8
    env = BatchedHumanoidAREnv(n, 0, mode="train", seed=0)
    env.load_context(standing_context(n, 30, STD["qpos"], STD["qvel"], env.sim))
    obs = env.reset()
    a = torch.zeros((n, 80), device=env.device); a[:, :74] = torch.cat([env.sim.get("qpos")[:, 2:3], obs[:, 1:5], env.sim.get("qpos")[:, 7:]], 1)
    held = []
    for k in range(3):
        o, _, done, info = env.step(a.contiguous())
        held.append(dict(obs=o, done=done, reward=info["custom_reward"], percent=info["percent"], cc_state=info["cc_state"],
                         reward_copy=info["custom_reward"].clone(), percent_copy=info["percent"].clone()))
    # step 1's alternating outputs survived step 2 and were reused by step 3; step 2's are intact
    assert held[0]["reward"].data_ptr() == held[2]["reward"].data_ptr() != held[1]["reward"].data_ptr()
    assert held[0]["done"].data_ptr() == held[2]["done"].data_ptr() != held[1]["done"].data_ptr()
    assert torch.equal(held[1]["reward"], held[1]["reward_copy"]) and torch.equal(held[1]["percent"], held[1]["percent_copy"])
    assert not torch.equal(held[0]["percent"], held[0]["percent_copy"])          # cur_t / len moved on: the view shows step 3's values
    # single buffers
    assert held[0]["obs"].data_ptr() == held[1]["obs"].data_ptr() == held[2]["obs"].data_ptr()
    assert held[0]["cc_state"].data_ptr() == held[2]["cc_state"].data_ptr()
