"""CPU: physics invariants that anchor oracle/kp_oracle.c where no MuJoCo golden vectors exist
(parity unpinned at the MuJoCo boundary)."""
import os

import numpy as np
import pytest

from kinpoly_amd.model_compiler import DEFAULT_KPM, read_kpm
from oracle import np_oracle as O
from oracle.kpo import OracleSim

KPM = read_kpm(DEFAULT_KPM)
STD = np.load(os.path.join(os.path.dirname(__file__), "golden", "standing_neutral.npz"))
PARENT = KPM["body_parent"]; BODY_POS = KPM["body_pos"].reshape(24, 3); BODY_IPOS = KPM["body_ipos"].reshape(24, 3)
MASS = KPM["body_mass"]; INERTIA = KPM["body_inertia"].reshape(24, 6)
H = KPM["opt"][0]


def rand_state(rng, scale=0.5):
    q = STD["qpos"].copy(); q[2] += 5.0
    quat = rng.normal(size=4); q[3:7] = quat / np.linalg.norm(quat)
    q[7:] += rng.normal(size=69) * scale
    v = rng.normal(size=75) * 0.7
    return q, v


def numpy_mass_matrix(qpos):
    """Independent derivation: M = sum_b Jv^T m Jv + Jw^T (R I R^T) Jw with explicit world-frame Jacobians."""
    fk = O.qpos_fk(qpos, BODY_POS, BODY_IPOS, PARENT)
    R = [O.quaternion_matrix3(q) for q in fk["wbquat"]]
    axes, anchors, trans, body_of = [], [], [], []
    for k in range(3):
        axes.append(np.eye(3)[k]); anchors.append(np.zeros(3)); trans.append(True); body_of.append(0)
    for k in range(3):
        axes.append(R[0][:, k]); anchors.append(fk["wbpos"][0]); trans.append(False); body_of.append(0)
    for b in range(1, 24):
        tz, ty, tx = qpos[7 + 3 * (b - 1): 10 + 3 * (b - 1)]
        Rp = R[PARENT[b]]
        Rz = O.quaternion_matrix3([np.cos(tz / 2), 0, 0, np.sin(tz / 2)]); Ry = O.quaternion_matrix3([np.cos(ty / 2), 0, np.sin(ty / 2), 0])
        for ax in (Rp @ [0, 0, 1.0], Rp @ Rz @ [0, 1.0, 0], Rp @ Rz @ Ry @ [1.0, 0, 0]):
            axes.append(ax); anchors.append(fk["wbpos"][b]); trans.append(False); body_of.append(b)
    anc = np.zeros((24, 24), bool)
    for b in range(24):
        k = b
        while k >= 0:
            anc[b, k] = True; k = PARENT[k]
    M = np.zeros((75, 75))
    for b in range(24):
        Jv, Jw = np.zeros((3, 75)), np.zeros((3, 75))
        for d in range(75):
            if anc[b, body_of[d]]:
                if trans[d]:
                    Jv[:, d] = axes[d]
                else:
                    Jw[:, d] = axes[d]; Jv[:, d] = np.cross(axes[d], fk["body_com"][b] - anchors[d])
        Ib = INERTIA[b]; I3 = np.array([[Ib[0], Ib[3], Ib[4]], [Ib[3], Ib[1], Ib[5]], [Ib[4], Ib[5], Ib[2]]])
        M += MASS[b] * Jv.T @ Jv + Jw.T @ (R[b] @ I3 @ R[b].T) @ Jw
    return M + np.diag(KPM["dof_armature"]), fk


def test_mass_matrix_matches_independent_jacobian_form():
    rng = np.random.default_rng(0)
    o = OracleSim(contact=False)
    for _ in range(5):
        q, v = rand_state(rng)
        o.reset(q, v)
        M_ref, fk = numpy_mass_matrix(o.get("qpos"))
        np.testing.assert_allclose(o.fullM(), M_ref, atol=1e-11)
        # kinematics agree with the (reference-pinned) FK restatement
        np.testing.assert_allclose(o.get("xpos").reshape(24, 3), fk["wbpos"], atol=1e-12)
        np.testing.assert_allclose(o.get("xipos").reshape(24, 3), fk["body_com"], atol=1e-12)
        x = rng.normal(size=75)
        np.testing.assert_allclose(M_ref @ o.solveM(x), x, atol=1e-9)      # sparse L^T D L solve


def test_bias_force_equals_lagrangian_derivative():
    """C(q, v) from RNE vs d/dt(dT/dv) - dT/dq + dV/dq from the independent M(q) (root rotation held at rest so that
    qvel = d qpos / dt for every remaining coordinate)."""
    rng = np.random.default_rng(1)
    o = OracleSim(contact=False)
    q, v = rand_state(rng, 0.3)
    v[3:6] = 0.0
    o.reset(q, v)
    C = o.get("qfrc_bias")
    idx = [0, 1, 2] + list(range(6, 75))          # translation + hinges
    qidx = [0, 1, 2] + list(range(7, 76))
    eps = 1e-6

    def Mq(qq):
        return numpy_mass_matrix(qq)[0]

    def V(qq):
        fk = O.qpos_fk(qq, BODY_POS, BODY_IPOS, PARENT)
        return 9.81 * (MASS * fk["body_com"][:, 2]).sum()

    dM, dV = [], []
    for qi in qidx:
        qp, qm = q.copy(), q.copy(); qp[qi] += eps; qm[qi] -= eps
        dM.append((Mq(qp) - Mq(qm)) / (2 * eps)); dV.append((V(qp) - V(qm)) / (2 * eps))
    dM = np.array(dM)                               # [k, 75, 75]
    vv = v.copy()
    want = np.zeros(len(idx))
    for a, i in enumerate(idx):
        s1 = sum(dM[k][i, :] @ vv * vv[j] for k, j in enumerate(idx))            # sum_jk dM_ij/dq_k v_j v_k
        s2 = 0.5 * vv @ dM[a] @ vv                                               # 1/2 dM_jk/dq_i v_j v_k
        want[a] = s1 - s2 + dV[a]
    np.testing.assert_allclose(C[idx], want, rtol=2e-5, atol=2e-4)


def test_free_fall_com_and_momentum():
    rng = np.random.default_rng(2)
    o = OracleSim(contact=False)
    q, v = rand_state(rng, 0.3)
    o.reset(q, v)
    coms = []
    for _ in range(200):
        o.step(); o.forward()
        coms.append((MASS[:, None] * o.get("xipos").reshape(24, 3)).sum(0) / MASS.sum())
    acc = np.diff(np.array(coms), n=2, axis=0) / H ** 2
    # joint-space semi-implicit Euler conserves momentum only to O(h): tolerance 1e-2 m/s^2 on a tumbling body
    assert abs(acc[:, 2].mean() + 9.81) < 2e-3 and np.abs(acc[:, :2]).mean() < 1e-2


def test_contact_solution_satisfies_kkt():
    """At the Newton solution: M qacc = qfrc_smooth + J^T f, f_e = D_e * max(0, -(J qacc - aref)_e) >= 0,
    and the resultant floor force supports the body (pyramid rows keep tangential/normal <= mu)."""
    o = OracleSim()
    o.reset(STD["qpos"], STD["qvel"])
    for _ in range(30):
        o.do_simulation(np.zeros(75), STD["qpos"], 1)
    o.forward()
    f, D, aref, J = o.efc()
    assert o.nefc >= 12 and (f >= 0).all()
    qacc = o.get("qacc")
    jar = J @ qacc - aref
    np.testing.assert_allclose(f, D * np.maximum(0.0, -jar), rtol=1e-9, atol=1e-9)
    M = o.fullM()
    qs = M @ o.get("qacc_smooth")
    np.testing.assert_allclose(M @ qacc, qs + J.T @ f, rtol=1e-7, atol=1e-5)
    body, pos, dist = o.contacts()
    assert (dist < KPM["opt"][14]).all()
    # resultant contact force on the root translation dofs: upward, friction inside the cone
    F = (J.T @ f)[:3]
    assert F[2] > 0.5 * MASS.sum() * 9.81 and np.hypot(F[0], F[1]) <= KPM["opt"][11] * F[2] + 1e-9


def test_standing_is_held_by_contacts():
    o = OracleSim()
    o.reset(STD["qpos"], STD["qvel"])
    for _ in range(5):
        o.do_simulation(np.zeros(75), STD["qpos"], 15)
    assert o.get("qpos")[2] > 0.85 and np.isfinite(o.get("qvel")).all()
    assert o.get("xpos").reshape(24, 3)[:, 2].min() > -0.05
