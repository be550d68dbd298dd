"""The whole update loop against the reference's own `AgentAR.update_params` (kin_poly/core/agent_ar.py:682-772, 852-870; policy_ar.py:277-287),
run by tools/make_golden.py::gen_update_params in this container in fp64 for TWO consecutive iterations on a recorded batch
(tests/golden/update_params.npz: 8 workers x 12 rows of whole episodes, kin_poly.yml's switches and rates, 10 PPO epochs + 10 value steps +
20 supervised step updates per iteration, Adam state / LambdaLR / generator-consumed clip all in the loop).

CPU, fp64: `ParamUpdate(update_dtype=float64)` replays both iterations and must land on the reference's parameters -- it does to 1e-15 (same
torch, same arithmetic: two things had to be found for that, the fp64 value of `action_log_std` and quat_mul_vec_batch's missing normalisation)."""
import numpy as np
import pytest
import torch

from oracle import np_oracle as O


def build(g, dtype=torch.float64, device="cpu", reference_bugs=True, master_from=None):
    from kinpoly_amd.context import TrajARNet
    from kinpoly_amd.model_compiler import DEFAULT_KPM, read_kpm
    from kinpoly_amd.nets import MLP, Value
    from kinpoly_amd.update import ParamUpdate
    base = dtype if master_from is None else master_from
    net = TrajARNet(state_dim=int(g["state_dim"]), context_dim=int(g["context_dim"])).to(base).refresh_log_std()
    shapes = [tuple(int(x) for x in row if x > 0) for row in g["shapes"]]
    sd = O.seeded_state_dict(list(zip([str(k) for k in g["keys"]], shapes)), int(g["seed_policy"]))
    for k in sd:
        if k.startswith(("action_fc", "context_fc")):
            sd[k] = sd[k] * 0.05
    missing = net.load_state_dict({k: torch.tensor(v, dtype=base) for k, v in sd.items()}, strict=False)
    assert not missing.unexpected_keys and set(missing.missing_keys) <= {"action_log_std"}, missing
    val = Value(MLP(int(g["state_dim"]), (512, 256), "relu")).to(base)
    vshapes = [tuple(int(x) for x in row if x > 0) for row in g["value_shapes"]]
    vsd = O.seeded_state_dict(list(zip([str(k) for k in g["value_keys"]], vshapes)), int(g["seed_value"]))
    val.load_state_dict({k: torch.tensor(v, dtype=base) for k, v in vsd.items()})
    net, val = net.to(device), val.to(device)
    kpm = read_kpm(DEFAULT_KPM)
    upd = ParamUpdate(net, val, kpm["body_pos"], kpm["body_parent"], update_dtype=dtype, reference_bugs=reference_bugs,
                      policy_lr=float(g["policy_lr"]), value_lr=float(g["value_lr"]), supervised_lr=float(g["sup_lr"]),
                      num_epoch_fix=int(g["num_epoch_fix"]), num_epoch=int(g["num_epoch"]))
    return net, val, upd


def batch_of(g, it, device="cpu", dtype=torch.float64):
    """the reference's flat batch (workers' rows back to back) as an env-major RolloutBatch: worker = env row, masks == 0 on every row's last step"""
    from kinpoly_amd.rollout import RolloutBatch
    N, T = int(g["N"]), int(g["T"])
    masks = g["masks"].reshape(N, T)
    assert (masks[:, -1] == 0).all()
    starts = np.concatenate([np.ones((N, 1), bool), masks[:, :-1] == 0], 1)
    t = lambda a, *s: torch.tensor(a, dtype=dtype, device=device).reshape(N, T, *s)  # noqa: E731
    tag = f"it{it}_"
    return RolloutBatch(states=t(g[tag + "states"], -1), actions=t(g[tag + "actions"], -1), rewards=t(g[tag + "rewards"]), masks=t(masks),
                        episode_start=torch.tensor(starts, device=device), fails=torch.zeros((N, T), dtype=torch.bool, device=device),
                        curr_qpos=t(g[tag + "curr_qpos"], 76), gt_target_qpos=t(g[tag + "gt_target_qpos"], 76), exps=torch.ones((N, T), dtype=dtype, device=device))


def replay(g, upd, device="cpu", dtype=torch.float64):
    out = []
    for it in range(2):
        upd.per_epoch_update()                                  # optimize_policy: the schedulers step before the iteration's update (agent_ar.py:264-275)
        lrs = [upd.trainer.opt_p.param_groups[0]["lr"], upd.trainer.opt_v.param_groups[0]["lr"], upd.opt_sup.param_groups[0]["lr"]]
        info = upd.update_params(batch_of(g, it, device, dtype), epoch=it)
        tr = upd.trainer
        out.append(dict(lr=lrs, info=info, adv=tr.last_adv.reshape(-1).double().cpu().numpy(), ret=tr.last_ret.reshape(-1).double().cpu().numpy(),
                        surr=np.array([float(x) for x in tr.surr_history]), vloss=np.array([float(x) for x in tr.vloss_history]),
                        step=np.array([float(x) for x in upd.step_history]),
                        params={k: v.detach().double().cpu().numpy().copy() for k, v in upd.policy.named_parameters()},
                        vparams={k: v.detach().double().cpu().numpy().copy() for k, v in upd.value.named_parameters()}))
    return out


def compare(g, out, tol_loss, tol_param, tol_adv):
    """tol_param is relative to the CHANGE the iteration made to the tensor (a parameter that moved by 1e-4 and is met to 1e-8 has its update right to 4 digits)"""
    worst = {}
    for it, o in enumerate(out):
        tag = f"it{it}_"
        np.testing.assert_allclose(o["lr"], g[tag + "lr"], rtol=1e-12)
        np.testing.assert_allclose(o["adv"], g[tag + "adv"].reshape(-1), rtol=0, atol=tol_adv)
        np.testing.assert_allclose(o["ret"], g[tag + "ret"].reshape(-1), rtol=0, atol=tol_adv)
        np.testing.assert_allclose(o["surr"], g[tag + "surr"], rtol=0, atol=tol_loss)
        np.testing.assert_allclose(o["vloss"], g[tag + "vloss"], rtol=tol_loss, atol=tol_loss)
        np.testing.assert_allclose(o["step"], g[tag + "step"], rtol=tol_loss, atol=tol_loss)
        for key in g.files:
            if key.startswith(tag + "p:") or key.startswith(tag + "v:"):
                name = key.split(":", 1)[1]
                got = (o["params"] if key[len(tag)] == "p" else o["vparams"])[name]
                want = g[key]
                if want.shape != got.shape:
                    got = got[:want.shape[0]]
                err = float(np.abs(got - want).max())
                worst[key] = err
                assert err <= tol_param, (key, err)
    return worst


def test_two_iterations_of_update_params_land_on_the_references_parameters(golden):
    g = golden("update_params")
    # the fixture really exercises what it is meant to pin: the first clip call saw a norm above 40, every later one saw no parameters
    assert g["it0_clip_norm"][0] > 40 and (g["it0_clip_norm"][1:] == 0).all() and (g["it1_clip_norm"] == 0).all()
    assert g["it0_lr"][0] == pytest.approx(1e-5 * 0.8) and g["it1_lr"][2] == pytest.approx(5e-4 * 0.8)      # LambdaLR in the loop, sup schedule one step behind
    assert g["it0_surr"][0] == 0 and g["it0_surr"][-1] < -0.1                                                 # epoch 0's ratio is 1; the surrogate falls
    net, val, upd = build(g)
    start = {k: v.detach().clone() for k, v in net.named_parameters()}
    out = replay(g, upd)
    worst = compare(g, out, tol_loss=1e-12, tol_param=1e-13, tol_adv=1e-13)        # measured: losses 6e-14, parameters 7e-16, advantages 0
    assert float(upd.trainer.clip_norms[0]) == pytest.approx(float(g["it0_clip_norm"][0]), rel=1e-9)
    # the parameters moved by far more than the tolerance they are met to
    moved = float((dict(net.named_parameters())["action_fc.bias"] - start["action_fc.bias"]).abs().max())
    assert moved > 1e-4 > 1e8 * max(worst.values()), (moved, worst)
    # nothing in the RL phase reaches the context network (its Adam entries never see a gradient)
    for k, v in net.named_parameters():
        if k.startswith("context_"):
            assert torch.equal(v, start[k]), k


def test_clipping_every_step_is_not_what_the_reference_does(golden):
    """reference_bugs=False (clip at every PPO step) parts from the fixture where the reference's later steps went unclipped -- the default has to be the bug"""
    g = golden("update_params")
    _, _, upd = build(g, reference_bugs=False)
    out = replay(g, upd)
    d = np.abs(out[0]["params"]["action_fc.bias"] - g["it0_p:action_fc.bias"]).max()
    assert d > 1e-7, d
    assert len(upd.trainer.clip_norms) == 1 and upd.trainer._clip_calls == 20


def test_fp64_master_copies_feed_fp32_rollout_modules(golden):
    """update_dtype=float64 over fp32 roll-out modules: the optimisers own fp64 copies, the fp32 modules receive the result after every update"""
    g = golden("update_params")
    net, val, upd = build(g, dtype=torch.float64, master_from=torch.float32)
    assert upd.has_master and next(upd.policy.parameters()).dtype == torch.float64 and next(net.parameters()).dtype == torch.float32
    before = net.action_fc.bias.detach().clone()
    upd.per_epoch_update()
    upd.update_params(batch_of(g, 0, dtype=torch.float32), epoch=0)
    assert not torch.equal(net.action_fc.bias, before)
    assert torch.equal(net.action_fc.bias, upd.policy.action_fc.bias.float())
    assert torch.equal(val.value_head.weight, upd.value.value_head.weight.float())
    # started from fp32-rounded weights and an fp32-rounded batch, it still follows the reference's fp64 run to fp32 rounding of the inputs
    assert np.abs(upd.policy.action_fc.bias.detach().numpy() - g["it0_p:action_fc.bias"]).max() < 2e-6


def test_gae_scan_equals_the_reference_recurrence(golden):
    from kinpoly_amd.rollout import gae_scan
    g = golden("gae_zfilter")
    n = g["rewards"].shape[0]
    r, m, v = (torch.tensor(np.asarray(g[k]).reshape(1, n)) for k in ("rewards", "masks", "values"))
    adv, ret = gae_scan(r, m, v, 0.95, 0.95)
    np.testing.assert_allclose(ret.reshape(-1).numpy(), g["ret"].reshape(-1), atol=1e-12)
    nadv = (adv - adv.mean()) / adv.std()
    np.testing.assert_allclose(nadv.reshape(-1).numpy(), g["adv"].reshape(-1), atol=1e-10)
    # a row cut by the horizon: the value behind the last row enters as kp_gae_bootstrap's rule says
    m2 = m.clone(); m2[0, -1] = 1
    adv_b, _ = gae_scan(r, m2, v, 0.95, 0.95, last_values=torch.tensor([0.7], dtype=torch.float64))
    want_last = r[0, -1] + 0.95 * 0.7 - v[0, -1]
    assert float(adv_b[0, -1]) == pytest.approx(float(want_last), abs=1e-12)


def test_logger_merge_reproduces_the_references_max_of_mins():
    """uhc/khrylib/rl/core/logger_rl.py:60: `min_episode_reward = max(...)` over the workers -- reproduced by default, min on request"""
    from kinpoly_amd.rollout import LoggerRL
    a, b = LoggerRL(num_steps=4, num_episodes=1, min_episode_reward=1.0, max_episode_reward=3.0), LoggerRL(num_steps=4, num_episodes=1, min_episode_reward=2.0, max_episode_reward=5.0)
    assert LoggerRL.merge([a, b]).min_episode_reward == 2.0 and LoggerRL.merge([a, b]).max_episode_reward == 5.0
    assert LoggerRL.merge([a, b], reference_bugs=False).min_episode_reward == 1.0
