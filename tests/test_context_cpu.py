"""CPU (fp64 torch): the context network and helpers of kinpoly_amd/context.py against fixtures generated from the
reference's TrajARNet (tools/make_golden.py::gen_traj_ar_net)."""
import numpy as np
import torch

from oracle import np_oracle as O


def _net_from_fixture(g, dtype=torch.float64):
    from kinpoly_amd.context import TrajARNet
    net = TrajARNet(state_dim=int(g["state_dim"]), context_dim=int(g["context_dim"])).to(dtype)
    shapes = [tuple(int(x) for x in row if x > 0) for row in g["shapes"]]
    sd = O.seeded_state_dict(list(zip([str(k) for k in g["keys"]], shapes)), int(g["seed"]))
    for k in sd:
        if k.startswith(("action_fc", "context_fc")):
            sd[k] = sd[k] * 0.05
    missing = net.load_state_dict({k: torch.tensor(v, dtype=dtype) for k, v in sd.items()}, strict=False)
    assert not missing.unexpected_keys and set(missing.missing_keys) <= {"action_log_std"}, missing   # reference module names load as they are
    return net


def _data(g, dtype=torch.float64, device="cpu"):
    return {k[3:]: torch.tensor(g[k], dtype=dtype, device=device) for k in g.files if k.startswith("in_")}


def test_context_gru_and_init_states_match_reference(golden):
    g = golden("traj_ar_net")
    net = _net_from_fixture(g)
    data = _data(g)
    with torch.no_grad():
        init_qpos, init_qvel, ctx = net.init_states(data)
    np.testing.assert_allclose(ctx.numpy(), g["context_feat_rnn"], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(init_qpos.numpy(), g["init_qpos"], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(init_qvel.numpy(), g["init_qvel"], rtol=1e-9, atol=1e-11)


def test_gaussian_smoothing_matches_scipy(golden):
    from kinpoly_amd.context import gaussian_filter1d_time
    g = golden("smooth")
    y = gaussian_filter1d_time(torch.tensor(g["x"])[None], 1.0)[0].numpy()
    np.testing.assert_allclose(y, g["y"], rtol=1e-12, atol=1e-13)


def test_reference_smoothing_statement_is_a_noop(golden):
    """PolicyAR.init_context's `ar_qpos[:, 7:] = gaussian_filter1d(ar_qpos[:, 7:], 1, axis=0)` on its [1, T, 76] tensor
    (policy_ar.py:150-152) filters a length-1 axis: the fixture holds that statement's input and output."""
    g = golden("smooth_effective")
    assert g["x"].shape[0] == 1 and g["x"].shape[2] == 76
    np.testing.assert_allclose(g["y"], g["x"], rtol=0, atol=1e-15)
    import inspect
    from kinpoly_amd.context import PolicyARContext
    assert inspect.signature(PolicyARContext.__init__).parameters["smooth_time_axis"].default is False


def test_qvel_fd_consistent_with_fixture_rollout(golden):
    """ar_qvel of the reference roll-out is get_qvel_fd_batch of consecutive ar_qpos (after fix_qvel)."""
    from kinpoly_amd.context import get_qvel_fd_batch
    g = golden("traj_ar_net")
    q = torch.tensor(g["ar_qpos"])
    for t in range(q.shape[1] - 1):
        v = get_qvel_fd_batch(q[:, t], q[:, t + 1], 1 / 30)
        np.testing.assert_allclose(v.numpy(), g["ar_qvel"][:, t], rtol=1e-8, atol=1e-9)


def test_kinematic_step_and_loss_lite_match_reference(golden):
    """differentiable TrajARNet.step + compute_loss_lite (kinpoly_amd/supervised.py) vs the reference's outputs."""
    from kinpoly_amd.model_compiler import DEFAULT_KPM, read_kpm
    from kinpoly_amd.supervised import TorchFK, compute_loss_lite, kinematic_step
    g = golden("step_loss")
    kpm = read_kpm(DEFAULT_KPM)
    fk = TorchFK(kpm["body_pos"], kpm["body_parent"], "cpu", torch.float64)
    cur, act, gt = (torch.tensor(g[k]) for k in ("cur", "act", "gt"))
    act.requires_grad_(True)
    nxt = kinematic_step(cur, act)
    np.testing.assert_allclose(nxt.detach().numpy(), g["next_qpos"], rtol=0, atol=1e-14)   # (2e-8 until quat_mul_vec_batch was followed to the letter: it does not normalise q)
    loss, idv = compute_loss_lite(fk, nxt, gt)
    np.testing.assert_allclose(float(loss.detach()), float(g["loss"]), rtol=1e-12)
    np.testing.assert_allclose([float(x.detach()) for x in idv], g["loss_idv"], rtol=1e-11)
    loss.backward()
    assert torch.isfinite(act.grad).all() and act.grad.abs().sum() > 0
    # FK agrees with the pinned numpy restatement
    fkn = O.qpos_fk(g["cur"][0], kpm["body_pos"].reshape(24, 3), kpm["body_ipos"].reshape(24, 3), kpm["body_parent"])
    np.testing.assert_allclose(fk.wbpos(cur[:1])[0].numpy(), fkn["wbpos"], atol=1e-12)


def test_reference_checkpoint_layout_roundtrip(golden, tmp_path):
    import os
    from kinpoly_amd import checkpoint as ck
    from kinpoly_amd.context import TrajARNet
    from kinpoly_amd.nets import MLP, Value
    here = os.path.join(os.path.dirname(__file__), "golden")
    cp = ck.load_checkpoint(os.path.join(here, "ref_checkpoint_small.p"))       # written by the reference's own classes
    e = golden("ref_checkpoint_small_expect")
    assert set(cp) == {"policy_dict", "value_dict", "running_state"}
    mean, std, clip = ck.running_state_arrays(cp["running_state"])
    np.testing.assert_allclose(mean, e["mean"]); np.testing.assert_allclose(std, e["std"]); assert clip == 5
    sd = ck.split_policy_dict(cp["policy_dict"])
    np.testing.assert_allclose(sd["action_fc.weight"].numpy(), e["w"]); assert "action_log_std" in sd
    # write our own nets in the reference layout and read them back through the reference-style unpickler
    net, val = TrajARNet(rnn_hdim=8, mlp_hsize=(8, 8)), Value(MLP(105, (8, 8), "relu"))
    rs = ck.ZFilter((105,), clip=5.0); rs.rs._n = 3; rs.rs._M[:] = 1.0; rs.rs._S[:] = 8.0
    ck.save_checkpoint(str(tmp_path / "iter_0001.p"), net, val, rs)
    raw = open(tmp_path / "iter_0001.p", "rb").read()
    assert b"uhc.khrylib.utils.zfilter" in raw                                    # class path the reference's CustomUnpickler expects
    cp2 = ck.load_checkpoint(raw)
    net2 = TrajARNet(rnn_hdim=8, mlp_hsize=(8, 8))
    net2.load_state_dict(ck.split_policy_dict(cp2["policy_dict"]))
    for a, b in zip(net.state_dict().values(), net2.state_dict().values()):
        assert torch.equal(a, b)
    assert all(k.startswith("traj_ar_net.") or k == "action_log_std" for k in cp2["policy_dict"])
    np.testing.assert_allclose(ck.running_state_arrays(cp2["running_state"])[1], 2.0)


def test_uhc_torch_features_and_reward_match_reference(golden):
    """The batched torch pieces of the UHC env (finite-difference velocities, body angular velocities, world_rfc_implicit
    reward) in fp64 on the CPU against the reference-generated fixture."""
    from kinpoly_amd.uhc_env import get_angvel_fd_t, get_qvel_fd_new_t, world_rfc_implicit_reward_t
    from kinpoly_amd.model_compiler import DEFAULT_KPM, read_kpm
    g = golden("uhc_expert_reward")
    t = lambda a: torch.tensor(np.asarray(a), dtype=torch.float64)  # noqa: E731
    clip = t(g["clip"])
    qvel = get_qvel_fd_new_t(clip[:-1], clip[1:], 1 / 30).clamp(-10, 10)
    np.testing.assert_allclose(qvel.numpy(), g["e_qvel"][1:], rtol=1e-8, atol=1e-8)
    bav = get_angvel_fd_t(t(g["e_bquat"][:-1]), t(g["e_bquat"][1:]), 1 / 30)
    np.testing.assert_allclose(bav.numpy(), g["e_bangvel"][1:], rtol=1e-8, atol=1e-8)
    kpm = read_kpm(DEFAULT_KPM)
    from oracle import np_oracle as O
    idx = [int(x) for x in g["r_t"]]
    bq = np.stack([O.get_body_quat(q) for q in g["r_qpos"]])
    r, info = world_rfc_implicit_reward_t(t(g["r_xpos"]).reshape(3, 72), t(bq), t(g["r_prev_bquat"]), t(g["r_com"]), t(g["r_action"]),
                                          t(g["e_bquat"][idx]), t(g["e_bangvel"][idx]), t(g["e_ee_wpos"][idx]), t(g["e_com"][idx]), t(kpm["uhc_b_diffw"]))
    np.testing.assert_allclose(r.numpy(), g["r_reward"], rtol=1e-7)
    np.testing.assert_allclose(info.numpy(), g["r_info"], rtol=1e-7)


def test_running_state_online_equals_sequential_pushes(golden):
    """RunningStateOnline.update on whole batches == ZFilter/RunningStat.push row by row (fixture from the reference's ZFilter)."""
    from kinpoly_amd.uhc_env import RunningStateOnline
    g = golden("gae_zfilter")
    # the 50 pushed rows are regenerated with the fixture generator's draw order (tools/make_golden.py gen_gae_zfilter, seed 104)
    rng = np.random.default_rng(104)
    B = 257
    rng.uniform(0, 1, size=(B, 1)); rng.normal(size=(B, 1)); rng.choice(B, 9, replace=False)
    xs = rng.normal(size=(50, 784)) * rng.uniform(0.1, 3, size=784) + rng.normal(size=784)
    rs = RunningStateOnline(784, 5.0, "cpu")
    for chunk in torch.split(torch.tensor(xs), 7):
        rs.update(chunk)
    np.testing.assert_allclose(rs._mean64.numpy(), g["zf_mean"], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(torch.sqrt(rs._m2 / (rs.count - 1)).numpy(), g["zf_std"], rtol=1e-10)
    y = rs(torch.tensor(g["zf_x"], dtype=torch.float32)[None], update=False)[0]
    np.testing.assert_allclose(y.numpy(), g["zf_y"], rtol=1e-4, atol=1e-5)


def test_dataset_features_and_sampling(golden):
    """kinpoly_amd.dataset: feature construction vs the reference fixture (fp64 torch on the CPU), then the sampler's
    contract: fixed-length windows inside their take, adaptive take probabilities, ragged full-sequence batches."""
    from kinpoly_amd import dataset as D
    g = golden("dataset_features")
    t = lambda a: torch.tensor(np.asarray(a), dtype=torch.float64)  # noqa: E731
    np.testing.assert_allclose(D.get_head_vel(t(g["head_pose"])).numpy(), g["head_vels"], rtol=1e-8, atol=1e-8)
    np.testing.assert_allclose(D.get_obj_relative_pose(t(g["obj_pose"]), t(g["head_pose"]), 2).numpy(), g["obj_head_relative_poses"], rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(D.get_traj_de_heading(t(g["clip"])).numpy(), g["traj_pos"], rtol=1e-9, atol=1e-10)
    rng = np.random.default_rng(0)
    feats = {}
    for i, T in enumerate((14, 30, 9, 22)):
        clip = np.tile(g["clip"][:1], (T, 1)); clip[:, :3] += 0.01 * np.arange(T)[:, None]
        feats[f"sit-{i}"] = dict(qpos=clip, qvel=rng.normal(size=(T, 75)), head_pose=np.tile(g["head_pose"][:1], (T, 1)), head_vels=rng.normal(size=(T, 6)),
                                 action_one_hot=np.tile([1.0, 0, 0, 0], (T, 1)), obj_head_relative_poses=rng.normal(size=(T, 14)), obj_pose=rng.normal(size=(T, 14)),
                                 wbpos=rng.normal(size=(T, 72)), wbquat=rng.normal(size=(T, 96)), bquat=rng.normal(size=(T, 96)), of_files=["x"] * T)
    ds = D.StateARDataset(feats, fr_num=10, seed=3)
    assert ds.takes == ["sit-0", "sit-1", "sit-3"] and ds.get_len() == 3        # the 9-frame take is shorter than fr_num: dropped in train mode
    assert sorted(set(ds.freq_indices.tolist())) == [0, 1, 2] and len(ds.freq_indices) == 2 + 3 + 3
    np.testing.assert_allclose(ds.data["target"][0][:, :74].numpy(), D.get_traj_de_heading(t(feats["sit-0"]["qpos"])).numpy(), atol=1e-6)
    freq = {k: [[1, 0]] * 8 if k == "sit-1" else [[0, 5]] * 8 for k in ds.takes}
    p = ds.take_probs(freq)
    assert p[1] < p[0] and abs(p.sum() - 1) < 1e-12                                # often-succeeding takes are sampled less
    hist = {k: [[float(v), 0] for v in np.random.RandomState(j).randint(0, 2, size=1 + 37 * j)] for j, k in enumerate(ds.takes)}      # the reference's recursion, literally
    def ewma(x, alpha=0.05):
        avg = x[0]
        for v in x[1:]:
            avg = alpha * v + (1 - alpha) * avg
        return avg
    want = np.exp(-np.array([ewma((np.array(hist[k])[:, 0] == 1).astype(float)) for k in hist]) / 0.5)
    np.testing.assert_allclose(ds.take_probs(hist, 0.5), want / want.sum(), rtol=1e-12)
    b = ds.sample_batch(64, freq_dict=freq)
    assert b["qpos"].shape == (64, 10, 76) and b["target"].shape == (64, 10, 80) and b["obj_head_relative_poses"].shape == (64, 10, 7)
    for r in range(64):
        i, s0 = int(b["take_ind"][r]), int(b["fr_start"][r])
        assert 0 <= s0 <= ds.get_seq_len(i) - 10 and torch.equal(b["qpos"][r], ds.data["qpos"][i][s0:s0 + 10])
    full = ds.batch([0, 1, 2])
    assert full["qpos"].shape == (3, 30, 76) and full["len"].tolist() == [14, 30, 22]
    assert torch.equal(full["qpos"][0, 13], full["qpos"][0, 29])                   # padded with the last frame
    one = ds.iter_seq(); two = ds.iter_seq()
    assert one["qpos"].shape[1] == 14 and two["qpos"].shape[1] == 30 and ds.curr_key == "sit-1"


def test_unroll_matches_reference_padded_rnn_forward(golden):
    """KinPolicy.unroll over an env-major batch with mid-batch episode starts vs the reference's own train-mode forward
    (PolicyAR.initialize_rnn + forward: scatter into the padded [T_max, n_episodes] layout, GRU from zeros per episode, gather back;
    policy_ar.py:104-122, 216-240), which wrote tests/golden/unroll.npz with the seeded TrajARNet of traj_ar_net.npz."""
    g, gt = golden("unroll"), golden("traj_ar_net")
    net = _net_from_fixture(gt)
    states, masks = torch.tensor(g["states"]), g["masks"]
    assert int(g["num_episode"]) == 4 and int(g["max_episode_len"]) == 6
    N, T = 2, 7                                                    # the flat batch is the concatenation of two workers' rows
    starts = np.concatenate([[True], masks[:-1] == 0]).reshape(N, T)
    starts[:, 0] = True
    assert masks.reshape(N, T)[:, -1].max() == 0                   # the reference's workers always finish their last episode
    with torch.no_grad():
        means = net.unroll(states.view(N, T, -1), torch.tensor(starts))
    np.testing.assert_allclose(means.reshape(N * T, -1).numpy(), g["action_mean"], rtol=1e-9, atol=1e-11)
    # a carried-in hidden state is used on rows that do not start an episode, ignored on rows that do
    hx0 = torch.randn(N, net.rnn_hdim, dtype=torch.float64)
    with torch.no_grad():
        m2 = net.unroll(states.view(N, T, -1), torch.tensor(starts), hx0)
        st2 = starts.copy(); st2[1, 0] = False
        m3 = net.unroll(states.view(N, T, -1), torch.tensor(st2), hx0)
    assert torch.equal(m2, means) and torch.equal(m3[0], means[0]) and not torch.allclose(m3[1, :1], means[1, :1])
