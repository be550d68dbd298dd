"""CPU: the driver-side plumbing around the rollout -- LoggerRL statistics (uhc/khrylib/rl/core/logger_rl.py), the statear yml reader
(kin_poly/utils/statear_smpl_config.py) and init_context over ragged whole-take batches (eval_seq, agent_ar.py:463-503)."""
import math
import os

import numpy as np
import pytest
import torch


def _reference_logger_loop(R, D, CI, ep):
    """LoggerRL.step / end_episode (logger_rl.py:28-42) applied row by row, the way sample_worker drives it (agent_ar.py:536-606);
    `ep` carries every env's running episode return from call to call."""
    st = dict(num_steps=0, num_episodes=0, total_reward=0.0, min_episode_reward=math.inf, max_episode_reward=-math.inf,
              total_c_reward=0.0, min_c_reward=math.inf, max_c_reward=-math.inf, total_c_info=np.zeros(6))
    N, T = R.shape
    for e in range(N):
        for t in range(T):
            r = float(R[e, t])
            ep[e] += r
            st["total_c_reward"] += r; st["total_c_info"] += CI[e, t].numpy().astype(np.float64)
            st["min_c_reward"], st["max_c_reward"] = min(st["min_c_reward"], r), max(st["max_c_reward"], r)
            st["num_steps"] += 1
            if D[e, t]:
                st["num_episodes"] += 1; st["total_reward"] += ep[e]
                st["min_episode_reward"], st["max_episode_reward"] = min(st["min_episode_reward"], ep[e]), max(st["max_episode_reward"], ep[e])
                ep[e] = 0.0
    return st


def test_episode_log_equals_the_row_by_row_logger_across_calls():
    from kinpoly_amd.rollout import LoggerRL, episode_log
    g = torch.Generator().manual_seed(5)
    N, T = 9, 17
    carry, ep, logs = torch.zeros(N, dtype=torch.float64), [0.0] * N, []
    for call in range(4):
        R = torch.rand(N, T, generator=g)
        D = torch.rand(N, T, generator=g) < (0.0 if call == 2 else 0.15)          # call 2: no episode ends (everything is carried into call 3)
        CI = torch.rand(N, T, 6, generator=g)
        want = _reference_logger_loop(R.double(), D, CI, ep)
        stats, carry = episode_log(R, D, CI, carry)
        st = stats.tolist()
        log = LoggerRL(num_steps=int(st[0]), num_episodes=int(st[1]), total_reward=st[2], min_episode_reward=st[3], max_episode_reward=st[4],
                       total_c_reward=st[5], min_c_reward=st[6], max_c_reward=st[7], total_c_info=np.asarray(st[8:14]))
        for k in LoggerRL.FIELDS:
            assert getattr(log, k) == pytest.approx(want[k], rel=1e-12, abs=1e-12), (call, k)
        np.testing.assert_allclose(log.total_c_info, want["total_c_info"], rtol=1e-12)
        np.testing.assert_allclose(carry.numpy(), np.array(ep), rtol=1e-12, atol=1e-12)
        if want["num_episodes"]:
            assert log.avg_episode_len == pytest.approx(N * T / want["num_episodes"]) and log.avg_episode_reward == pytest.approx(want["total_reward"] / want["num_episodes"])
        else:
            assert log.num_episodes == 0 and log.min_episode_reward == math.inf
        assert log.avg_c_reward == pytest.approx(want["total_c_reward"] / (N * T))
        logs.append(log)
    m = LoggerRL.merge(logs, reference_bugs=False)                               # LoggerRL.merge (:44-70): sums, min of mins, max of maxes (the corrected form)
    assert m.num_steps == 4 * N * T and m.num_episodes == sum(x.num_episodes for x in logs)
    assert m.min_c_reward == min(x.min_c_reward for x in logs) and m.max_episode_reward == max(x.max_episode_reward for x in logs)
    assert m.min_episode_reward == min(x.min_episode_reward for x in logs)
    assert LoggerRL.merge(logs).min_episode_reward == max(x.min_episode_reward for x in logs)      # the default reproduces the reference's max of mins (logger_rl.py:60)
    assert m.avg_c_reward == pytest.approx(sum(x.total_c_reward for x in logs) / m.num_steps)
    assert set(m.as_dict()) >= {"num_steps", "avg_episode_len", "avg_c_info", "avg_c_reward"}


YML = """
dataset_path: "{data}/"
meta_id: mocap_meta
data_file: mocap_annotations
meta_wild_id: real_mocap
data_wild_file: real_annotations
seed: 4
fr_num: 100
use_of: false
use_context: false
smooth: True
root_deheading: true
obs_global: true
obs_quat: true
model_specs: {{model_v: 1, rnn_hdim: 1024, mlp_hsize: [1024, 512, 256], mlp_htype: relu, rnn_type: gru}}
policy_specs:
  policy_v: 1
  log_std: -3.2
  fix_std: true
  gamma: 0.95
  tau: 0.9
  policy_lr: 2.e-5
  value_lr: 3.e-4
  policy_weightdecay: 0.0
  value_weightdecay: 0.0
  policy_optimizer: Adam
  value_optimizer: Adam
  clip_epsilon: 0.2
  min_batch_size: 10000
  reward_id: dynamic_supervision_v1
  end_reward: false
  save_model_interval: 50
  rl_update: true
  step_update: true
  sampling_temp: 0.3
  sampling_freq: 0.5
  num_step_update: 20
  num_optim_epoch: 10
  reward_weights: {{w_hp: 0.15, w_hq: 0.15, w_p: 0.2, w_jp: 0.2, w_act_p: 0.2, w_act_v: 0.1, k_hp: 45, k_hq: 45, k_p: 50, k_jp: 50, k_act_p: 5, k_act_v: 0.005}}
lr: 5.e-4
num_epoch: 10000
num_epoch_fix: 100
save_model_interval: 50
"""


def _write_cfg(tmp_path, text=None, name="kin_poly"):
    d = tmp_path / "config" / "statear"
    d.mkdir(parents=True, exist_ok=True)
    (d / f"{name}.yml").write_text((text or YML).format(data=str(tmp_path / "sample_data")))
    return tmp_path


def test_config_reads_the_statear_schema(tmp_path):
    from kinpoly_amd import sim as kpsim
    from kinpoly_amd.config import Config, ConfigError
    root = _write_cfg(tmp_path)
    meta = tmp_path / "sample_data" / "meta"
    meta.mkdir(parents=True)
    (meta / "mocap_meta.yml").write_text("train: [a-1, b-1]\ntest: [a-2]\naction_type: {a-1: sit, b-1: push, a-2: sit}\nobject: {}\n")
    cfg = Config("kin_poly", config_root=str(root), base_dir=str(tmp_path / "results"), create_dirs=True)          # found by id under config/**/
    assert cfg.id == "kin_poly" and cfg.fr_num == 100 and cfg.seed == 4 and cfg.smooth is True
    assert cfg.policy_model_dir == os.path.join(str(tmp_path / "results"), "all", "statear", "kin_poly", "models_policy") and os.path.isdir(cfg.result_dir) and os.path.isdir(cfg.log_dir)
    assert cfg.checkpoint_path(750).endswith("models_policy/iter_0750.p")
    assert cfg.feature_path().endswith("sample_data/features/mocap_annotations.p")
    assert cfg.takes == {"train": ["a-1", "b-1"], "test": ["a-2"]}
    assert Config("kin_poly", action="push", config_root=str(root)).takes["train"] == ["b-1"]
    assert Config("kin_poly", wild=True, config_root=str(root)).data_file == "real_annotations"
    # the UHC checkpoint the env is built around (humanoid_ar_v1.py:69-76): cc_iter -1 = the latest under results/motion_im/<cc_cfg>/models
    assert cfg.cc_checkpoint_path() is None
    mdir = tmp_path / "results" / "motion_im" / "uhc" / "models"
    mdir.mkdir(parents=True)
    for it in (5, 100, 20):
        (mdir / ("iter_%04d.p" % it)).write_bytes(b"x")
    assert cfg.cc_checkpoint_path().endswith("motion_im/uhc/models/iter_0100.p") and cfg.cc_checkpoint_path(20).endswith("iter_0020.p") and cfg.cc_checkpoint_path(7) is None
    kw = cfg.agent_kwargs()
    assert kw["policy_lr"] == 2e-5 and kw["tau"] == 0.9 and kw["supervised_lr"] == 5e-4 and kw["num_step_update"] == 20 and kw["rl_update"] and kw["step_update"]
    assert kw["sampling_temp"] == 0.3 and kw["sampling_freq"] == 0.5 and kw["num_epoch"] == 10000 and kw["num_epoch_fix"] == 100 and kw["log_std"] == -3.2 and kw["smooth"] is True
    assert cfg.horizon(4096) == 3 and cfg.horizon(4096, world_size=8) == 1 and cfg.horizon(64) == 157 and cfg.horizon(4096, floor=24) == 24
    class _Env:
        reward_cfg = kpsim.KpRewardCfg.default()
    rc = cfg.apply_reward_weights(_Env())
    assert rc.k_hp == 45.0 and rc.w_act_v == pytest.approx(0.1) and rc.k_act_v == pytest.approx(0.005)
    # reward_weights' own defaults (reward_function.py:936-939) when the file gives none
    cfg.reward_weights = {}
    rc = cfg.apply_reward_weights(_Env())
    assert rc.w_p == 1.0 and rc.k_jp == pytest.approx(0.1) and rc.k_hp == 1.0
    # a file path works as well as an id; what the kernels do not implement is refused, not silently run differently
    assert Config(str(tmp_path / "config" / "statear" / "kin_poly.yml")).id == "kin_poly"
    _write_cfg(tmp_path, YML.replace("use_of: false", "use_of: true"), "with_of")
    with pytest.raises(ConfigError, match="use_of"):
        Config("with_of", config_root=str(root))
    _write_cfg(tmp_path, YML.replace("use_context: false\n", ""), "default_context")          # the reference's default for a missing use_context is True
    with pytest.raises(ConfigError, match="use_context"):
        Config("default_context", config_root=str(root))
    _write_cfg(tmp_path, YML.replace("policy_optimizer: Adam", "policy_optimizer: SGD"), "sgd")
    with pytest.raises(ConfigError, match="policy_optimizer"):
        Config("sgd", config_root=str(root))
    _write_cfg(tmp_path, YML.replace("seed: 4\n", ""), "no_seed")
    with pytest.raises(ConfigError, match="seed"):
        Config("no_seed", config_root=str(root))
    with pytest.raises(ConfigError, match="exactly one"):
        Config("absent", config_root=str(root))


def test_init_states_of_a_ragged_batch_equal_the_unpadded_sequences():
    """eval_seq runs init_context on ONE whole take at a time; a batch of whole takes is padded to its longest, and the context mean of a
    row must be over its own frames only (both forms of the mean: the kept feature sequence and the running mean)."""
    from kinpoly_amd.context import TrajARNet
    torch.manual_seed(2)
    net = TrajARNet().double()
    lens, T = [7, 12, 4], 12
    g = torch.Generator().manual_seed(3)
    rows = []
    for L in lens:
        q = torch.randn(L, 76, generator=g, dtype=torch.float64); q[:, 3:7] /= q[:, 3:7].norm(dim=1, keepdim=True)
        rows.append(dict(qpos=q, head_vels=torch.randn(L, 6, generator=g, dtype=torch.float64), obj_head_relative_poses=torch.randn(L, 7, generator=g, dtype=torch.float64),
                         action_one_hot=torch.tensor([[0.0, 1, 0, 0]], dtype=torch.float64).repeat(L, 1)))
    pad = lambda x: torch.cat([x, x[-1:].repeat(T - x.shape[0], 1)], 0)      # noqa: E731  (StateARDataset.batch pads with the last frame)
    batch = {k: torch.stack([pad(r[k]) for r in rows], 0) for k in rows[0]}
    batch["len"], batch["ragged"] = torch.tensor(lens, dtype=torch.int32), True
    with torch.no_grad():
        for keep in (True, False):
            q, v, _ = net.init_states(batch, keep_feat=keep)
            for i, r in enumerate(rows):
                qi, vi, _ = net.init_states({k: x[None] for k, x in r.items()}, keep_feat=keep)
                np.testing.assert_allclose(q[i].numpy(), qi[0].numpy(), rtol=1e-10, atol=1e-12)
                np.testing.assert_allclose(v[i].numpy(), vi[0].numpy(), rtol=1e-10, atol=1e-12)
        # without the flag the padding is averaged in (a rectangular training batch: every frame counts)
        plain = dict(batch); plain["ragged"] = False
        q_plain, _, _ = net.init_states(plain, keep_feat=False)
        assert not np.allclose(q_plain[2].numpy(), q[2].numpy(), atol=1e-6)
