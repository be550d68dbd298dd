"""CPU: kinpoly_amd/metrics.py against tests/golden/metrics.npz, written by tools/make_golden.py::gen_metrics with the reference's own functions
(kin_poly/utils/metrics.py: get_root_matrix, get_frobenious_norm, get_joint_vels, get_joint_accels; compute_error_accel and the mpjpe lines of
scripts/eval_pose_all.py)."""
import numpy as np


def test_kinematic_metrics_match_reference(golden):
    from kinpoly_amd import metrics as M
    g = golden("metrics")
    dt = float(g["dt"])
    np.testing.assert_allclose(M.joint_vels(g["pred"], dt), g["vels_pred"], rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(M.joint_vels(g["gt"], dt), g["vels_gt"], rtol=1e-9, atol=1e-10)
    assert np.all(M.joint_vels(g["pred"], dt)[4, 3:6] == 0.0)                        # frame 5 repeats frame 4's root rotation: the `1 - w < 1e-6` branch
    m = M.sequence_metrics(g["pred"], g["gt"], g["jpos_pred"], g["jpos_gt"], g["head_pred"], g["head_gt"], dt)
    for k in ("root_dist", "head_dist", "vel_dist", "accel_dist", "mpjpe"):
        np.testing.assert_allclose(m[k], float(g[k]), rtol=1e-10, err_msg=k)
    np.testing.assert_allclose(np.abs(np.diff(M.joint_vels(g["pred"], dt), axis=0) / dt).mean(), float(g["accels_abs"]), rtol=1e-10)


def test_coverage_metrics_over_a_result_file():
    """the coverage_full layout end to end with the torch forward kinematics: a prediction equal to the clip scores zero everywhere, a shifted root shows in
    root_dist only, a roll-out one frame short of its clip is compared with the clip's first frames."""
    import torch
    from kinpoly_amd import metrics as M
    from kinpoly_amd.model_compiler import DEFAULT_KPM, read_kpm
    from kinpoly_amd.supervised import TorchFK
    kpm = read_kpm(DEFAULT_KPM)
    tfk = TorchFK(kpm["body_pos"], kpm["body_parent"], "cpu", dtype=torch.float64)
    fk = lambda q: tuple(x.numpy() for x in tfk.chain_torch(torch.as_tensor(q)))         # noqa: E731
    rng = np.random.default_rng(0)
    T = 9
    q = np.zeros((T, 76)); q[:, 2] = 0.9; q[:, 3] = 1.0; q[:, 7:] = 0.2 * np.sin(np.arange(T)[:, None] * 0.4 + rng.uniform(0, 6, 69))
    jp, qq = fk(q)
    gt = {"a": {"qpos": q, "head_pose": np.concatenate([jp[:, 13], qq[:, 13]], 1)}, "b": {"qpos": q, "head_pose": np.concatenate([jp[:, 13], qq[:, 13]], 1)}}
    shifted = q.copy(); shifted[:, 0] += 0.1
    res = {"a": {"pred": list(q[:-1]), "percent": 1.0, "fail_safe": False}, "b": {"pred": list(shifted), "percent": 0.5, "fail_safe": False}, "c": {"pred": list(q), "percent": 1.0}}
    out = M.coverage_metrics(res, gt, fk)
    a, b = out["per_take"]["a"], out["per_take"]["b"]
    assert set(out["per_take"]) == {"a", "b"} and out["succ"] == 0.5
    assert max(a["root_dist"], a["mpjpe"], a["accel_dist"], a["vel_dist"], a["head_dist"]) < 1e-9
    assert abs(b["root_dist"] - 0.1) < 1e-12 and b["mpjpe"] < 1e-9 and b["vel_dist"] < 1e-9 and abs(b["head_dist"] - 0.1) < 1e-9
