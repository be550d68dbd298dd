"""GPU: the batched UHC training env (kinpoly_amd/uhc_env.py): expert features against the reference-generated fixture,
one env step against the composed oracle, and the copycat training iteration."""
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


def test_expert_features_match_reference_fixture(golden):
    from kinpoly_amd.uhc_env import BatchedHumanoidEnv
    g = golden("uhc_expert_reward")
    env = BatchedHumanoidEnv(2, 0)
    clip = torch.tensor(g["clip"], dtype=torch.float32)[None].repeat(2, 1, 1)
    env.load_expert(clip)
    ex = env.expert
    tol = dict(qvel=5e-4, rlinv=2e-5, rlinv_local=2e-5, rangv=5e-4, rq_rmh=2e-6, com=2e-6, body_com=2e-6, head_pose=2e-6, ee_pos=2e-6, ee_wpos=2e-6,
               bquat=2e-6, bangvel=5e-4, wbpos=2e-6, wbquat=2e-6)      # finite differences divide fp32 round-off by dt = 1/30: measured <= 7.5e-5 on values up to 126 (velocities), 1.4e-6 (root linear)
    for k, a in tol.items():
        np.testing.assert_allclose(ex[k][1].double().cpu().numpy(), g["e_" + k], atol=a, rtol=1e-5, err_msg=k)
    assert abs(float(ex["height_lb"][0]) - float(g["e_height_lb"])) < 1e-6


def test_uhc_env_step_matches_composed_oracle():
    """reset + one step: observation (expert frame t + 1), physics, reward, termination vs the fp64 composition of
    get_expert / obs_cc / do_simulation / world_rfc_implicit_reward / calc_body_diff (mean)."""
    from kinpoly_amd.uhc_env import BatchedHumanoidEnv
    n, T = 4, 8
    rng = np.random.default_rng(5)
    clips = np.tile(STD["qpos"], (n, T, 1))
    for e in range(n):
        clips[e, :, 7:] += 0.05 * np.sin(0.4 * np.arange(T)[:, None] + np.arange(69)[None] + e)
    env = BatchedHumanoidEnv(n, 0)
    env.load_expert(torch.tensor(clips, dtype=torch.float32))
    obs0 = env.reset().double().cpu().numpy()
    a = rng.normal(size=(n, 75)) * 0.1
    obs1, _, done, info = env.step(torch.tensor(a, dtype=torch.float32, device=env.device))
    obs1 = obs1.double().cpu().numpy()
    for e in range(n):
        ex = O.get_expert(clips[e], BP, BI, PAR, KPM["body_mass"])
        o = OracleSim()
        o.reset(clips[e, 0], ex["qvel"][0])
        x = {k: o.get(k) for k in ("qpos", "qvel", "xpos", "xquat", "xipos")}
        tgt = O.qpos_fk(clips[e, 1], BP, BI, PAR); tgt["qpos"] = clips[e, 1]
        want0 = O.obs_cc(x["qpos"], x["qvel"], x["xpos"].reshape(24, 3), x["xquat"].reshape(24, 4), x["xipos"].reshape(24, 3), tgt)
        np.testing.assert_allclose(obs0[e], want0, atol=3e-05)        # measured 2.8e-06
        prev_bquat = O.get_body_quat(x["qpos"])
        o.do_simulation(a[e], clips[e, 0], 15)             # PD base pose = expert frame t (delta_t = 0), the observation looks at t + 1
        x = {k: o.get(k) for k in ("qpos", "qvel", "xpos", "xquat", "xipos")}
        np.testing.assert_allclose(env.sim.get("qpos")[e].double().cpu().numpy(), x["qpos"], atol=5e-06)        # measured 3.2e-07
        com = (KPM["body_mass"][:, None] * x["xipos"].reshape(24, 3)).sum(0) / KPM["body_mass"].sum()
        r, rinfo = O.world_rfc_implicit_reward(x["xpos"].reshape(24, 3), O.get_body_quat(x["qpos"]), prev_bquat, com, a[e], ex, 1, KPM["uhc_b_diffw"][1:])
        assert abs(float(info["custom_reward"][e]) - r) < 2e-4
        np.testing.assert_allclose(info["custom_info"][e].double().cpu().numpy(), rinfo, atol=5e-05)        # measured 3.0e-06
        bd = O.calc_body_diff_mean(x["xpos"].reshape(24, 3), ex["wbpos"][1], KPM["body_diffw"])
        assert abs(float(info["body_diff"][e]) - bd) < 1e-5 and bool(info["fail"][e]) == (bd > 0.5)
        tgt2 = O.qpos_fk(clips[e, 2], BP, BI, PAR); tgt2["qpos"] = clips[e, 2]
        want1 = O.obs_cc(x["qpos"], x["qvel"], x["xpos"].reshape(24, 3), x["xquat"].reshape(24, 4), x["xipos"].reshape(24, 3), tgt2)
        np.testing.assert_allclose(obs1[e], want1, atol=3e-4)


def test_copycat_agent_iterations():
    from kinpoly_amd.uhc_env import BatchedHumanoidEnv, CopycatAgent
    n, T = 128, 20
    clips = torch.tensor(np.tile(STD["qpos"], (n, T, 1)), dtype=torch.float32)
    env = BatchedHumanoidEnv(n, 0, env_init_noise=0.01)
    env.load_expert(clips)
    torch.manual_seed(0)
    agent = CopycatAgent(env, num_optim_epoch=3)
    stats = [agent.optimize_policy(horizon=8) for _ in range(2)]
    for s in stats:
        assert np.isfinite(s["value_loss"]) and np.isfinite(s["surr_loss"]) and 0 < s["avg_reward"] <= 1 and s["num_steps"] == n * 8
    assert agent.running_state.count == 2 * n * 9
    assert int(env.sim.diag()[:, 2].max()) == 0
