"""The kinematic part of the reference's paper-metric script (scripts/eval_pose_all.py:113-197, kin_poly/utils/metrics.py) on the records
`evaluate.write_coverage` stores (`*_coverage_full.pkl`: per take `pred`, `target`, `obj_pose`, `percent`, `fail_safe`):

    root_dist   mean Frobenius norm of I - T_pred T_gt^-1 over the root's 4 x 4 transforms      (get_root_matrix, get_frobenious_norm)
    mpjpe       mean root-relative joint position error, mm                                      (:170-172)
    accel_dist  mean norm of the second-difference error of the joint positions, mm / frame^2    (compute_error_accel, :45-74)
    vel_dist    mean norm of the difference of the finite-difference generalised velocities      (get_joint_vels -> get_qvel_fd(.., 'heading'), get_mean_dist)
    head_dist   root_dist's measure on the head's transform against the data set's head_pose

numpy, host side (an analysis step after the roll-outs, as in the reference).  Joint positions come from the forward kinematics of the poses (the
reference reads MuJoCo's body_xpos after sim.forward: the same quantities).  NOT here: penetration, foot sliding and the object-interaction
success rate of compute_physcis_metris (:205-292), which query MuJoCo's contact list frame by frame.
"""
from __future__ import annotations

import numpy as np


def _qmat(q):
    """Gohlke quaternion_matrix (rotation part), batched [T, 4] -> [T, 3, 3]; a near-zero quaternion gives the identity as there."""
    q = np.asarray(q, np.float64)
    n = (q * q).sum(-1)
    out = np.tile(np.eye(3), (q.shape[0], 1, 1))
    ok = n > np.finfo(float).eps * 4.0
    s = np.where(ok, np.sqrt(2.0 / np.where(ok, n, 1.0)), 0.0)
    w, x, y, z = (q * s[:, None]).T
    R = np.stack([np.stack([1 - y * y - z * z, x * y - z * w, x * z + y * w], -1),
                  np.stack([x * y + z * w, 1 - x * x - z * z, y * z - x * w], -1),
                  np.stack([x * z - y * w, y * z + x * w, 1 - x * x - y * y], -1)], -2)
    out[ok] = R[ok]
    return out


def _qmul(a, b):
    w1, x1, y1, z1 = a.T; w0, x0, y0, z0 = b.T
    return np.stack([-x1 * x0 - y1 * y0 - z1 * z0 + w1 * w0, x1 * w0 + y1 * z0 - z1 * y0 + w1 * x0,
                     -x1 * z0 + y1 * w0 + z1 * x0 + w1 * y0, x1 * y0 - y1 * x0 + z1 * w0 + w1 * z0], -1)


def root_matrices(poses7):
    """get_root_matrix (metrics.py:15-24): [T, 4, 4] from rows (x, y, z, qw, qx, qy, qz)."""
    p = np.asarray(poses7, np.float64)
    M = np.tile(np.eye(4), (p.shape[0], 1, 1))
    M[:, :3, :3] = _qmat(p[:, 3:7])
    M[:, :3, 3] = p[:, :3]
    return M


def frobenius_dist(A, B):
    """get_frobenious_norm (metrics.py:64-72): mean over frames of || I - A_t B_t^-1 ||_F."""
    return float(np.linalg.norm(np.eye(4)[None] - A @ np.linalg.inv(B), axis=(1, 2)).mean())


def joint_vels(qpos, dt):
    """get_joint_vels (metrics.py:38-44) = get_qvel_fd(p_t, p_t+1, dt, 'heading') of kin_poly/utils/math_utils.py:26-43 for every frame pair -> [T - 1, 75]."""
    q = np.asarray(qpos, np.float64)
    cur, nxt = q[:-1], q[1:]
    v = (nxt[:, :3] - cur[:, :3]) / dt
    cq = cur[:, 3:7]
    inv = np.concatenate([cq[:, :1], -cq[:, 1:]], 1) / (cq * cq).sum(1, keepdims=True)
    qrel = _qmul(nxt[:, 3:7], inv)
    w = qrel[:, 0]
    none = (np.abs(1.0 - w) < 1e-6) | (np.abs(1.0 + w) < 1e-6)                       # rotation_from_quaternion (transformation.py:362-372)
    angle = np.where(none, 0.0, 2.0 * np.arccos(np.clip(w, -1.0, 1.0)))
    ax = qrel[:, 1:] / np.where(none, 1.0, np.sin(angle / 2.0))[:, None]
    ax = ax / np.where(none, 1.0, np.linalg.norm(ax, axis=1))[:, None]
    ax[none] = [1.0, 0.0, 0.0]
    angle = np.where(angle > np.pi, angle - 2 * np.pi, angle)
    rv = np.einsum("tji,tj->ti", _qmat(cq), ax * angle[:, None] / dt)               # transform_vec(., cur, 'root'): R^T rv
    hq = cq.copy(); hq[:, 1:3] = 0.0; hq /= np.linalg.norm(hq, axis=1, keepdims=True)
    v = np.einsum("tji,tj->ti", _qmat(hq), v)                                        # root velocity into the heading frame
    return np.concatenate([v, rv, (nxt[:, 7:] - cur[:, 7:]) / dt], 1)


def accel_error(jpos_gt, jpos_pred):
    """compute_error_accel (eval_pose_all.py:45-74, vis=None): per frame triple, mean over joints of || (second difference of pred) - (of gt) ||."""
    a_gt = jpos_gt[:-2] - 2 * jpos_gt[1:-1] + jpos_gt[2:]
    a_pr = jpos_pred[:-2] - 2 * jpos_pred[1:-1] + jpos_pred[2:]
    return np.linalg.norm(a_pr - a_gt, axis=2).mean(1)


def sequence_metrics(pred_qpos, gt_qpos, jpos_pred, jpos_gt, head_pred=None, head_gt=None, dt=1.0 / 30.0) -> dict:
    """One take: pred / gt qpos [T, 76], their joint positions [T, 24, 3] (forward kinematics), optional head poses [T, 7]."""
    pred_qpos, gt_qpos = np.asarray(pred_qpos, np.float64), np.asarray(gt_qpos, np.float64)
    jp, jg = np.asarray(jpos_pred, np.float64).reshape(-1, 24, 3), np.asarray(jpos_gt, np.float64).reshape(-1, 24, 3)
    out = {"root_dist": frobenius_dist(root_matrices(pred_qpos[:, :7]), root_matrices(gt_qpos[:, :7])),
           "vel_dist": float(np.linalg.norm(joint_vels(pred_qpos, dt) - joint_vels(gt_qpos, dt), axis=1).mean()),
           "accel_dist": float(accel_error(jp, jg).mean() * 1000.0),                 # the reference passes (pred, gt) into (gt, pred): symmetric
           "mpjpe": float(np.linalg.norm((jp - jp[:, :1]) - (jg - jg[:, :1]), axis=2).mean() * 1000.0)}
    if head_pred is not None and head_gt is not None:
        out["head_dist"] = frobenius_dist(root_matrices(head_pred), root_matrices(head_gt))
    return out


def coverage_metrics(results: dict, gt: dict, fk, dt=1.0 / 30.0) -> dict:
    """compute_metrics (:113-197) over a `*_coverage_full.pkl` dict.  gt: {take: {'qpos' [T, 76], 'head_pose' [T, 7]}} (the feature file's entries);
    fk(qpos [T, 76]) -> (joint positions [T, 24, 3], body quaternions [T, 24, 4]) -- e.g. supervised.TorchFK.chain_torch.  A take whose roll-out is one
    frame shorter than its clip is compared with the clip's first T - 1 frames, as the reference does (:683-688).  Returns the means over the takes
    + 'succ' = share of takes played to their end without the fail-safe, + 'per_take'."""
    per = {}
    for k, r in results.items():
        if k not in gt:
            continue
        pred = np.asarray(r["pred"], np.float64)
        g, hp = np.asarray(gt[k]["qpos"], np.float64), np.asarray(gt[k]["head_pose"], np.float64)
        n = min(len(pred), len(g))
        if n < 3:
            continue
        pred, g, hp = pred[:n], g[:n], hp[:n]
        jp, qp = fk(pred); jg, _ = fk(g)
        jp, qp, jg = np.asarray(jp, np.float64), np.asarray(qp, np.float64), np.asarray(jg, np.float64)
        m = sequence_metrics(pred, g, jp, jg, np.concatenate([jp[:, 13], qp[:, 13]], 1), hp, dt)
        m["succ"] = float(r.get("percent", 0.0) == 1 and not r.get("fail_safe", False))
        per[k] = m
    keys = sorted({kk for m in per.values() for kk in m})
    return {**{kk: float(np.mean([m[kk] for m in per.values() if kk in m])) for kk in keys}, "per_take": per}
