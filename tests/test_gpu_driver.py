"""GPU tests of the driver plumbing around the rollout: LoggerRL statistics on the agent's record (agent_ar.py:243-262, logger_rl.py), periodic
`eval_policy` over whole takes (agent_ar.py:394-503), and the reference's command line (`--cfg`, `--iter`) of the two scripts end to end."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STD = np.load(os.path.join(ROOT, "tests", "golden", "standing_neutral.npz"))


def _takes(n_envs, fr_num=12, seed=3, n_per_action=1):
    from kinpoly_amd import dataset as D
    from kinpoly_amd import sim as kpsim
    from kinpoly_amd.model_compiler import read_kpm
    fk_sim = kpsim.KpSim(kpsim.KpModel(kpsim.STEP_KPM), n_envs, 0)
    return D.synthetic_takes(fk_sim, STD["qpos"], n_per_action=n_per_action, T_range=(fr_num + 4, fr_num + 14), body_mass=read_kpm(kpsim.STEP_KPM)["body_mass"], seed=seed), fk_sim


def test_logger_statistics_ride_on_the_agents_record(tmp_path):
    from kinpoly_amd import dataset as D
    from kinpoly_amd.agent import AgentAR
    n, T, fr = 32, 8, 12
    takes, fk_sim = _takes(n, fr)
    ds = D.StateARDataset(takes, fr_num=fr, seed=3, device=fk_sim.device)
    agent = AgentAR(n, dataset=ds, device=0, horizon=T, num_optim_epoch=1, num_step_update=1, result_dir=str(tmp_path))
    w = agent.env.reward_cfg
    weights = np.array([w.w_hp, w.w_hq, w.w_p, w.w_jp, w.w_act_p, w.w_act_v])
    total_eps = 0
    for it in range(2):
        info = agent.optimize_policy(it)
        log = info["log"]
        assert log.num_steps == n * T == info["num_steps"] and log.num_episodes == info["episodes"] > 0
        total_eps += log.num_episodes
        assert log.avg_episode_len == pytest.approx(n * T / log.num_episodes)
        assert log.avg_c_reward == pytest.approx(info["avg_reward"], rel=1e-5)
        # the reward IS the weighted sum of its six terms (reward_function.py:987-988): so are the averages
        assert float(weights @ log.avg_c_info) == pytest.approx(log.avg_c_reward, rel=1e-5)
        assert 0 <= log.min_c_reward <= log.avg_c_reward <= log.max_c_reward <= weights.sum() + 1e-6
        assert log.min_episode_reward <= log.avg_episode_reward <= log.max_episode_reward
        line = agent.log_train(info)
        assert "expert_R_avg" in line and "eps_len" in line and f"{log.avg_episode_len:.2f}" in line
    import joblib
    fd = joblib.load(str(tmp_path / "freq_dict.pt"))            # written after every iteration (agent_ar.py:297)
    assert set(fd) == set(ds.takes) and sum(len(v) for v in fd.values()) == total_eps
    # a second agent on the same result_dir resumes the sampling history (setup_logging, :228-234)
    agent2 = AgentAR(n, dataset=ds, device=0, horizon=T, num_optim_epoch=1, num_step_update=1, result_dir=str(tmp_path))
    assert agent2.freq_dict == fd


def test_eval_policy_plays_every_take_whole_and_feeds_the_history(tmp_path):
    import joblib
    from kinpoly_amd import dataset as D
    from kinpoly_amd import sim as kpsim
    from kinpoly_amd.agent import AgentAR
    from kinpoly_amd.context import PolicyARContext
    from kinpoly_amd.env import BatchedHumanoidAREnv
    from kinpoly_amd.evaluate import eval_dataset
    n, fr = 8, 12
    takes, fk_sim = _takes(n, fr)
    ds = D.StateARDataset(takes, fr_num=fr, seed=3, device=fk_sim.device)
    test_takes, _ = _takes(n, fr, seed=9)
    ds_test = D.StateARDataset(test_takes, data_mode="test", fr_num=fr, seed=9, device=fk_sim.device); ds_test.name = "synthetic_test"
    agent = AgentAR(n, dataset=ds, device=0, horizon=4, result_dir=str(tmp_path), eval_envs=3)          # 4 takes through 3 envs: one full chunk + a filled one
    agent.test_datasets = [ds_test]
    before = {k: len(v) for k, v in agent.freq_dict.items()}
    res = agent.eval_policy("train")
    cov = res[0]["coverage_train"]
    assert cov["all_coverage"] == ds.get_len() == 4 and 0 <= cov["num_coverage"] <= 4 and cov["mean_coverage"] == cov["num_coverage"] / 4
    hist = joblib.load(str(tmp_path / "eval_dict_train.pt"))
    assert set(hist[agent.epoch]) == set(ds.takes)
    for k, pc in hist[agent.epoch].items():
        added = agent.freq_dict[k][before[k]:]
        assert added == [[pc, 0]] * (1 if pc == 1 else 3)                        # :427-433
    res_t = agent.eval_policy("test")
    assert list(res_t[0]) == ["coverage_synthetic_test"] and os.path.exists(str(tmp_path / "eval_dict_test.pt"))
    assert {k: len(v) for k, v in agent.freq_dict.items()} == {k: before[k] + len(agent.freq_dict[k][before[k]:]) for k in before}      # test sets leave the history alone
    # a take played in a padded batch next to longer ones = the same take played alone (its own context mean, its own end)
    env, builder = agent._eval_engine()
    full = eval_dataset(env, agent.policy_net, builder, ds)
    lens = [ds.get_seq_len(i) for i in range(4)]
    short = int(np.argmin(lens))
    env1 = BatchedHumanoidAREnv(1, 0, mode="test", seed=0, cc_policy=agent.env.cc_policy, cc_running_state=agent.env.cc_running_state)
    b1 = PolicyARContext(agent.policy_net, kpsim.KpSim(env1.model, 1, 0), smooth=True, keep_context_feat=False)
    alone = eval_dataset(env1, agent.policy_net, b1, ds, inds=[short])
    key = ds.takes[short]
    assert set(alone) == {key} and full[key]["percent"] == pytest.approx(alone[key]["percent"], abs=1e-6)
    assert len(full[key]["pred"]) == len(alone[key]["pred"]) <= lens[short] - 1
    np.testing.assert_allclose(full[key]["pred"][0], alone[key]["pred"][0], atol=2e-5)              # init_qpos of the reset: the unpadded context mean
    np.testing.assert_allclose(full[key]["target"][0], alone[key]["target"][0], atol=2e-5)
    m = min(3, len(full[key]["pred"]))
    np.testing.assert_allclose(np.array(full[key]["pred"][:m]), np.array(alone[key]["pred"][:m]), atol=2e-3)


CFG = """
dataset_path: "{root}/sample_data/"
meta_id: mocap_meta
data_file: mocap_annotations
meta_wild_id: real_mocap
data_wild_file: real_annotations
seed: 4
fr_num: 12
use_of: false
use_context: false
smooth: True
root_deheading: true
obs_global: true
obs_quat: true
policy_specs:
  policy_v: 1
  log_std: -3.2
  fix_std: true
  gamma: 0.95
  tau: 0.95
  policy_lr: 1.e-5
  value_lr: 3.e-4
  clip_epsilon: 0.2
  min_batch_size: 256
  reward_id: dynamic_supervision_v1
  end_reward: false
  save_model_interval: 2
  max_iter_num: 4
  rl_update: true
  step_update: true
  sampling_temp: 0.3
  sampling_freq: 0.5
  num_step_update: 2
  num_optim_epoch: 2
  reward_weights: {{w_hp: 0.15, w_hq: 0.15, w_p: 0.2, w_jp: 0.2, w_act_p: 0.2, w_act_v: 0.1, k_hp: 45, k_hq: 45, k_p: 50, k_jp: 50, k_act_p: 5, k_act_v: 0.005}}
lr: 5.e-4
num_epoch: 4
num_epoch_fix: 0
save_model_interval: 2
"""


def test_reference_command_line_of_the_scripts(tmp_path):
    """scripts/train_ar_policy.py --cfg <id> [--iter N] and scripts/eval_ar_policy.py --cfg <id> --iter N on a feature file in the reference's
    schema: checkpoints under results/all/statear/<id>/models_policy, log.txt, freq_dict.pt, eval_dict_test.pt, resume, coverage pickles."""
    import joblib
    from kinpoly_amd import dataset as D
    root = tmp_path
    (root / "config" / "statear").mkdir(parents=True)
    (root / "config" / "statear" / "mini.yml").write_text(CFG.format(root=str(root)))
    takes, _ = _takes(4, 12)
    (root / "sample_data" / "features").mkdir(parents=True); (root / "sample_data" / "meta").mkdir(parents=True)
    D.write_features(str(root / "sample_data" / "features" / "mocap_annotations.p"), takes)
    names = sorted(takes)
    (root / "sample_data" / "meta" / "mocap_meta.yml").write_text(json.dumps({"train": names, "test": names[:2], "action_type": {k: k.split("-")[0] for k in names}, "object": {}}))
    env = dict(os.environ, PYTHONPATH=ROOT)
    run = lambda *a: subprocess.run([sys.executable, *a], env=env, capture_output=True, text=True, timeout=900)      # noqa: E731
    feat = str(root / "sample_data" / "features" / "mocap_annotations.p")
    r = run(os.path.join(ROOT, "scripts", "train_ar_policy.py"), "--cfg", "mini", "--config_root", str(root), "--num_envs", "64", "--iters", "2", "--test_data", feat)
    assert r.returncode == 0, r.stderr[-3000:]
    base = root / "results" / "all" / "statear" / "mini"
    recs = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert [x["iter"] for x in recs] == [0, 1] and recs[0]["num_steps"] == 64 * 4 and recs[0]["log"]["num_steps"] == 256       # horizon = min_batch_size / envs
    assert "log_eval" in recs[1] and "log_eval" not in recs[0]                                                                 # every save_model_interval iterations
    assert (base / "models_policy" / "iter_0002.p").exists() and not (base / "models_policy" / "iter_0001.p").exists()
    assert (base / "results" / "freq_dict.pt").exists() and (base / "results" / "eval_dict_test.pt").exists()
    log_lines = (base / "log" / "log.txt").read_text().splitlines()
    assert len(log_lines) == 2 and "expert_R_avg" in log_lines[0]
    # resume from the checkpoint: --iter 2 runs iterations 2, 3 and writes iter_0004.p
    r = run(os.path.join(ROOT, "scripts", "train_ar_policy.py"), "--cfg", "mini", "--config_root", str(root), "--num_envs", "64", "--iter", "2", "--iters", "2")
    assert r.returncode == 0, r.stderr[-3000:]
    recs = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert [x["iter"] for x in recs] == [2, 3] and (base / "models_policy" / "iter_0004.p").exists()
    assert len((base / "log" / "log.txt").read_text().splitlines()) == 4
    # the reference's resumed run builds fresh LambdaLR schedulers (agent_ar.py:215-225 run before load_checkpoint, nothing restores them): its decay
    # restarts -- reproduced by default; --no_reference_bugs continues the schedule (num_epoch_fix 0, num_epoch 4: factor 1 - epoch / 5)
    np.testing.assert_allclose([x["policy_lr"] for x in recs], [0.8e-5, 0.6e-5], rtol=1e-5)
    r = run(os.path.join(ROOT, "scripts", "train_ar_policy.py"), "--cfg", "mini", "--config_root", str(root), "--num_envs", "64", "--iter", "2", "--iters", "1", "--no_reference_bugs", "--update_dtype", "fp64")
    assert r.returncode == 0, r.stderr[-3000:]
    recs = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    np.testing.assert_allclose([x["policy_lr"] for x in recs], [0.4e-5], rtol=1e-5)
    assert len((base / "log" / "log.txt").read_text().splitlines()) == 5
    # evaluation of the test takes of the meta file with that checkpoint
    r = run(os.path.join(ROOT, "scripts", "eval_ar_policy.py"), "--cfg", "mini", "--config_root", str(root), "--iter", "4", "--num_seq", "3", "--data", "test")
    assert r.returncode == 0, r.stderr[-3000:]
    assert "Coverage of" in r.stdout and "out of 2" in r.stdout
    full = joblib.load(str(base / "results" / "0004_mocap_annotations_coverage_full.pkl"))
    assert set(full) == set(names[:2]) and all(len(v["pred"]) >= 1 and v["pred"][0].shape == (76,) for v in full.values())


def test_warm_start_trains_the_policy_and_its_rollout_equals_the_hip_rollout():
    """AgentAR.train_init (agent_ar.py:366-385) on the device in fp32: both supervised phases move the networks and end finite, the supervised optimiser is
    rebuilt afterwards (setup_optimizers); and the differentiable torch roll-out the warm start trains through (pretrain.forward_supervised) is the same
    computation as the tape-free HIP roll-out init_context uses (TrajARNet.rollout: kp_kin_advance / obs_ar / FK kernels)."""
    from kinpoly_amd import dataset as D
    from kinpoly_amd import pretrain as P
    from kinpoly_amd.agent import AgentAR
    n, fr = 8, 12
    takes, fk_sim = _takes(n, fr)
    ds = D.StateARDataset(takes, fr_num=fr, seed=3, device=fk_sim.device)
    agent = AgentAR(n, dataset=ds, device=0, horizon=4)
    net = agent.policy_net
    data = next(P.sampling_batches(ds, n, n, agent.device))
    with torch.no_grad():
        pred = P.forward_supervised(net, agent.fk, data)
        iq, iv, _ = net.init_states(data, keep_feat=False)
        Q, V, A = net.rollout(data, agent.kin_sim, iq.contiguous(), iv.contiguous())
    np.testing.assert_allclose(pred["qpos"].cpu().numpy(), Q.cpu().numpy(), atol=2e-4)
    np.testing.assert_allclose(pred["action"].cpu().numpy(), A.cpu().numpy(), atol=5e-4)
    np.testing.assert_allclose(pred["qvel"].cpu().numpy(), V.cpu().numpy(), atol=2e-2)            # finite differences over 1 / 30 s of fp32 poses
    before = {k: v.detach().clone() for k, v in net.named_parameters()}
    old_opt = agent.opt_sup
    l0, _ = P.compute_loss(P.forward_supervised(net, agent.fk, data), data)
    drawn = agent.source.n_drawn
    out = agent.train_init(warm_update_init=4, warm_update_full=6, num_sample=16, batch_size=8)
    # the clips queued before the warm start carried init_qpos / init_qvel of the UNTRAINED context network: the ring is drawn again
    assert agent.source.n_drawn == drawn + agent.sampler.n_slots * n and int(agent.sampler.head.abs().sum()) == 0 and bool(agent.sampler.fresh.all())
    assert np.isfinite(out["init_loss"]) and np.isfinite(out["full_loss"])
    moved = {k: float((v.detach() - before[k]).abs().max()) for k, v in net.named_parameters() if v.requires_grad}
    assert moved["context_fc.weight"] > 0 and moved["action_fc.weight"] > 0 and moved["action_rnn.rnn_f.weight_hh"] > 0
    assert agent.opt_sup is not old_opt and len(agent.opt_sup.state) == 0
    with torch.no_grad():
        l1, _ = P.compute_loss(P.forward_supervised(net, agent.fk, data), data)
    assert float(l1) < float(l0.detach())
    info = agent.optimize_policy(0)                      # and the RL iteration runs on the warm-started networks
    assert info["num_steps"] == n * 4


def test_optional_supervised_branches_of_update_params():
    """update_params' other supervised branches (agent_ar.py:711-745; off in kin_poly.yml, on in other statear configs): init_update,
    step_update_dyna (the one-step loss against the pose the SIMULATION reached: res_qpos is recorded for it) and full_update."""
    from kinpoly_amd import dataset as D
    from kinpoly_amd.agent import AgentAR
    n, fr = 8, 12
    takes, fk_sim = _takes(n, fr)
    ds = D.StateARDataset(takes, fr_num=fr, seed=3, device=fk_sim.device)
    agent = AgentAR(n, dataset=ds, device=0, horizon=4, num_optim_epoch=1, num_step_update=1, init_update=True, num_init_update=1, step_update_dyna=True,
                    num_step_dyna_update=2, full_update=True, num_sample=8, batch_size=8)
    before = agent.policy_net.context_fc.weight.detach().clone()
    info = agent.optimize_policy(0)
    for k in ("init_loss", "step_loss", "step_dyna_loss", "full_loss", "surr_loss"):
        assert k in info and np.isfinite(info[k]), k
    assert float((agent.policy_net.context_fc.weight.detach() - before).abs().max()) > 0          # only the init / full branches reach the context network
    # the dyna target is what the simulation produced, not the clip: the two one-step losses differ
    assert info["step_dyna_loss"] != info["step_loss"]
    # a --wild test set is evaluated on the wild engine (no GT termination, ..._mesh_all.xml), the training set on the training model
    ds_w = D.StateARDataset(takes, data_mode="test", fr_num=fr, wild=True, seed=5, device=fk_sim.device); ds_w.name = "wild"
    agent.test_datasets = [ds_w]
    agent.eval_envs = 4
    out = agent.eval_policy("test")
    assert list(out[0]) == ["coverage_wild"] and out[0]["coverage_wild"]["all_coverage"] == 4
    assert set(agent._eval) == {True} and agent._eval[True][0].wild is True


def test_env_loads_a_uhc_checkpoint_in_the_reference_layout(tmp_path):
    """humanoid_ar_v1.py:60-81: running_state always, policy_dict unless the controller is trained jointly; the pickle is what
    scripts/train_uhc.py --save writes (reference class paths: a ZFilter holding a RunningStat)."""
    import pickle
    from kinpoly_amd import checkpoint as ck
    from kinpoly_amd.agent import AgentAR
    from kinpoly_amd.env import BatchedHumanoidAREnv
    from kinpoly_amd.nets import PolicyMCP
    torch.manual_seed(11)
    src = PolicyMCP()
    rs = ck.ZFilter((784,)); rs.rs._n = 10
    rng = np.random.default_rng(0)
    rs.rs._M = rng.normal(size=784); rs.rs._S = rng.uniform(1.0, 4.0, size=784) * 9
    path = str(tmp_path / "iter_0100.p")
    with ck._RefModulePath(), open(path, "wb") as f:
        pickle.dump({"policy_dict": {k: v.cpu() for k, v in src.state_dict().items()}, "value_dict": {}, "running_state": rs}, f)
    env = BatchedHumanoidAREnv(4, 0, mode="train", seed=0)
    env.load_uhc_checkpoint(path)
    for k, v in src.state_dict().items():
        assert torch.equal(env.cc_policy.state_dict()[k].cpu(), v), k
    np.testing.assert_allclose(env.cc_running_state.mean.cpu().numpy(), rs.rs.mean, rtol=1e-6)
    np.testing.assert_allclose(env.cc_running_state.std.cpu().numpy(), rs.rs.std, rtol=1e-6)
    assert env.cc_running_state.clip == rs.clip
    # joint_controller: the controller's weights are the run's own, only the observation filter is taken (:79-81)
    from kinpoly_amd.env import standing_context
    agent = AgentAR(4, context_fn=lambda n: standing_context(n, 10, STD["qpos"], STD["qvel"], env.sim), device=0, horizon=2, use_init_context=False,
                    joint_controller=True, cc_checkpoint=path)
    assert not all(torch.equal(agent.env.cc_policy.state_dict()[k].cpu(), v) for k, v in src.state_dict().items() if v.dim() > 1)
    np.testing.assert_allclose(agent.env.cc_running_state.mean.cpu().numpy(), rs.rs.mean, rtol=1e-6)
