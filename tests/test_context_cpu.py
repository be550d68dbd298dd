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


def test_qvel_fd_consistent_with_fixture_rollout(golden):
    """ar_qvel of the reference roll-out is get_qvel_fd_batch of consecutive ar_qpos (after fix_qvel)."""
    from kinpoly_amd.context import get_qvel_fd_batch
    g = golden("traj_ar_net")
    q = torch.tensor(g["ar_qpos"])
    for t in range(q.shape[1] - 1):
        v = get_qvel_fd_batch(q[:, t], q[:, t + 1], 1 / 30)
        np.testing.assert_allclose(v.numpy(), g["ar_qvel"][:, t], rtol=1e-8, atol=1e-9)
