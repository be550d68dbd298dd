"""CPU: physics invariants that anchor oracle/kp_oracle.c where no MuJoCo golden vectors exist
(parity unpinned at the MuJoCo boundary)."""
import os

import numpy as np
import pytest

from kinpoly_amd.model_compiler import DEFAULT_KPM, read_kpm
from oracle import np_oracle as O
from oracle.kpo import OracleSim

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

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


# ------------------------------------------------------------------ dynamic free objects (chair / box / table / Can / step)
def _no_armature(kpm):
    k = dict(kpm); oi = kpm["obj_inertial"].reshape(-1, 13).copy(); oi[:, 12] = 0.0; k["obj_inertial"] = oi.reshape(-1)
    return k


def test_object_torque_free_motion_conserves_momenta():
    """A spinning, translating box in zero gravity: linear momentum of its COM and angular momentum about it are
    conserved by the free-body equations (bias forces of kpo_obj_forward), up to the O(h) integrator drift."""
    kpm = _no_armature(KPM)
    o = OracleSim(contact=False, gravity=0.0)
    quat = np.array([0.8, 0.2, -0.4, 0.4]); quat /= np.linalg.norm(quat)
    o.set_object(0, kpm, 0, [0.3, -0.2, 2.0, *quat], [0.4, -0.3, 0.2, 1.5, -0.7, 0.9])     # the chair: COM offset + non-spherical inertia
    q = STD["qpos"].copy(); q[0] += 30
    o.reset(q, np.zeros(75))
    inert = kpm["obj_inertial"].reshape(-1, 13)[0]
    m, c = inert[0], inert[1:4]
    Ib = np.array([[inert[4], inert[7], inert[8]], [inert[7], inert[5], inert[9]], [inert[8], inert[9], inert[6]]])
    P, L, E = [], [], []
    for _ in range(300):
        oq, ov = o.get_object(0)
        R = O.quaternion_matrix3(oq[3:7]); w = R @ ov[3:]
        vc = ov[:3] + np.cross(w, R @ c)
        P.append(m * vc); L.append(R @ Ib @ R.T @ w); E.append(0.5 * m * vc @ vc + 0.5 * w @ (R @ Ib @ R.T) @ w)
        o.step()
    P, L, E = np.array(P), np.array(L), np.array(E)
    assert np.abs(P - P[0]).max() < 2e-3 * np.linalg.norm(P[0])
    assert np.abs(L - L[0]).max() < 2e-2 * np.linalg.norm(L[0]) and abs(E[-1] - E[0]) < 2e-2 * E[0]
    # halving the step halves the drift: it is integrator error, not a wrong bias term
    assert np.abs(L[150] - L[0]).max() < 0.6 * np.abs(L[-1] - L[0]).max() + 1e-9


def test_object_mass_matrix_and_free_fall():
    kpm = _no_armature(KPM)
    o = OracleSim(contact=False)
    quat = np.array([0.5, -0.5, 0.3, 0.6]); quat /= np.linalg.norm(quat)
    o.set_object(0, kpm, 3, [0.0, 0.0, 3.0, *quat], [0.1, 0.2, 0.0, 0.3, 0.5, -0.4])
    q = STD["qpos"].copy(); q[0] += 30
    o.reset(q, np.zeros(75))
    M, bias = o.object_dyn(0)
    assert np.allclose(M, M.T) and np.linalg.eigvalsh(M).min() > 0
    oq, ov = o.get_object(0)
    qa = o.qacc_full()[75:81]
    R = O.quaternion_matrix3(oq[3:7]); c = R @ kpm["obj_inertial"].reshape(-1, 13)[3, 1:4]
    w = R @ ov[3:]; alpha = R @ qa[3:]
    a_com = qa[:3] + np.cross(alpha, c) + np.cross(w, np.cross(w, c))
    np.testing.assert_allclose(a_com, [0, 0, -9.81], atol=1e-9)


def test_objects_rest_and_carry_load():
    """Push scene at rest: the floor carries table + box, the table carries the box; standing on the step: the floor under
    the step carries humanoid + step.  Also the KKT conditions of the coupled solve (all dofs)."""
    from kinpoly_amd.model_compiler import STEP_KPM
    kpm = read_kpm(STEP_KPM)
    inert = kpm["obj_inertial"].reshape(-1, 13)
    o = OracleSim(kpm=STEP_KPM)
    o.set_object(0, kpm, 1, [0, 0, 0.921, 1, 0, 0, 0]); o.set_object(1, kpm, 2, [0, 0, 0.7905, 1, 0, 0, 0])
    q = STD["qpos"].copy(); q[0] += 30
    o.reset(q, np.zeros(75))
    for _ in range(450):
        o.step()
    o.forward()
    f, D, aref, _ = o.efc(); J = o.efc_J_full(); b1, b2 = o.contact_pairs()
    rows = np.repeat(np.arange(len(b1)), 4)
    lim = o.nefc - 4 * len(b1)
    fz_floor_table = sum((J[lim + r, 75 + 6 + 2] * f[lim + r]) for r in range(4 * len(b1)) if b1[rows[r]] == 25 and b2[rows[r]] == -1)
    fz_table_box = sum((J[lim + r, 75 + 2] * f[lim + r]) for r in range(4 * len(b1)) if b1[rows[r]] == 24 and b2[rows[r]] == 25)
    assert abs(fz_floor_table - (inert[1, 0] + inert[2, 0]) * 9.81) < 0.01 * inert[2, 0] * 9.81
    assert abs(fz_table_box - inert[1, 0] * 9.81) < 0.02 * inert[1, 0] * 9.81
    bq, bv = o.get_object(0)
    assert abs(bq[2] - 0.921) < 2e-3 and np.abs(bv).max() < 1e-3
    # KKT of the coupled system: M (qacc - qacc_smooth) = J^T f on humanoid and object dofs
    M = np.zeros((87, 87)); M[:75, :75] = o.fullM(); M[75:81, 75:81] = o.object_dyn(0)[0]; M[81:87, 81:87] = o.object_dyn(1)[0]
    res = M @ (o.qacc_full() - o.qacc_smooth_full()) - J.T @ f
    assert np.abs(res).max() < 1e-4 * np.abs(J.T @ f).max()

    o = OracleSim(kpm=STEP_KPM)
    o.set_object(0, kpm, 4, [STD["qpos"][0], STD["qpos"][1], 0.3705, 1, 0, 0, 0])
    q = STD["qpos"].copy(); q[2] += 0.341
    o.reset(q, STD["qvel"])
    total = (MASS.sum() + inert[4, 0]) * 9.81
    fzs = []
    for it in range(10):          # an open-loop PD stance on four point contacts sways and tips over within a second: look at the first third
        o.do_simulation(np.zeros(75), q, 15)
        o.forward()
        f, D, aref, _ = o.efc(); J = o.efc_J_full(); b1, b2 = o.contact_pairs()
        # feet on the step: ONE contact per (hull, box) pair -- mjc_Convex / libccd returns a single point (ankles + toes = 4);
        # the step on the floor: mjc_PlaneBox's four bottom corners
        assert ((b1 < 24) & (b2 == 24)).sum() == 4 and ((b1 == 24) & (b2 == -1)).sum() == 4
        rows = np.repeat(np.arange(len(b1)), 4); lim = o.nefc - 4 * len(b1)
        fzs.append(sum((J[lim + r, 75 + 2] * f[lim + r]) for r in range(4 * len(b1)) if b1[rows[r]] == 24 and b2[rows[r]] == -1))
    # the floor under the step carries humanoid + step on average (the stance bounces a little on its soft contacts)
    assert abs(np.mean(fzs[2:]) - total) < 0.1 * total and o.get("qpos")[2] > q[2] - 0.05


def test_resting_box_sinks_to_the_closed_form_depth_of_the_soft_contact_model():
    """Known answer from MuJoCo's documented constraint model alone (computation.html#soft-constraint-model): a box at rest on the
    plane carries its weight on four corner contacts (mjc_PlaneBox).  With zero velocity and acceleration every pyramid row has
    residual -aref = k d(r) r, force f = -D k d(r) r per row, D = 1 / (2 mu^2 R), R = (1 - d) / d * (1 + mu^2) * invweight0, and the
    four rows of a contact add up to the normal force (their tangential parts cancel):  16 D(r) k d(r) |r| = m g  fixes r = dist - margin."""
    from kinpoly_amd.model_compiler import STEP_KPM
    from scipy.optimize import brentq
    kpm = read_kpm(STEP_KPM)
    opt = kpm["opt"]
    inert = kpm["obj_inertial"].reshape(-1, 13)[4]                # the step box: mass, ..., invweight0 (translational) at [10]
    mass, invw = inert[0], inert[10]
    tc, dr, solimp, mu, margin = max(opt[4], 2 * opt[0]), opt[5], opt[6:11], opt[11], opt[14]
    d0, dw, width, mid, power = solimp
    assert power == 2.0                                            # the closed form below is written for the model's impedance power

    def imp(r):
        x = min(abs(r) / width, 1.0)
        y = x * x / mid if x <= mid else 1.0 - (1.0 - x) ** 2 / (1.0 - mid)
        return d0 + y * (dw - d0)

    k = 1.0 / (dw * dw * tc * tc * dr * dr)

    def load(r):                                                  # normal force of the four contacts at depth r < 0, minus the weight
        d = imp(r)
        R = (1.0 - d) / d * (1.0 + mu * mu) * invw
        return 16.0 * k * d * abs(r) / (2.0 * mu * mu * R) - mass * 9.81
    r_star = -brentq(lambda x: load(-x), 1e-9, 0.5 * width)

    o = OracleSim(kpm=STEP_KPM)
    og = kpm["obj_geoms"].reshape(-1, 18)
    half_h = og[og[:, 0].astype(int) == 4][0, 4]                   # box half height (size[2])
    o.set_object(0, kpm, 4, [0.0, 0.0, half_h, 1, 0, 0, 0])
    q = STD["qpos"].copy(); q[0] += 30
    o.reset(q, np.zeros(75))
    for _ in range(1500):
        o.step()
    o.forward()
    c = o.contacts_full()
    mine = c["body"] == 24
    assert mine.sum() == 4 and np.all(c["b2"][mine] == -1)
    bq, bv = o.get_object(0)
    assert np.abs(bv).max() < 1e-6                                 # at rest
    np.testing.assert_allclose(c["dist"][mine], r_star + margin, rtol=2e-4)


# ------------------------------------------------------------------ known answers of MuJoCo's documented model (tests/known_answers.py); the
# HIP kernel's twins are in tests/test_gpu_round3.py.  Call site of everything below in the reference: sim.step(), uhc/envs/humanoid_im.py:527.
def _box_alone(z, gravity=None):
    """the free step box on the plane, the humanoid parked 30 m away and 50 m up (it free-falls for the length of these tests)"""
    import known_answers as K
    from kinpoly_amd.model_compiler import STEP_KPM
    o = OracleSim(kpm=STEP_KPM, gravity=gravity)
    o.set_object(0, K.KPM, K.STEP_OBJ, [0.0, 0.0, z, 1, 0, 0, 0])
    q = STD["qpos"].copy(); q[0] += 30; q[2] += 50
    o.reset(q, np.zeros(75))
    return o


def test_known_answer_drop_follows_the_scalar_soft_contact_recurrence():
    """(i) computation.html 'Soft constraint model' + modeling.html 'solref / solimp': a flat box dropped from 2 mm above the contact margin.
    Reference acceleration aref = -b v - k d(r) r with k = 1 / (dmax^2 tc^2 dr^2), b = 2 / (dmax tc), tc = max(solref[0], 2 h); impedance
    d(r) by the solimp sigmoid; pyramidal row regulariser R = (1 - d) / d (1 + mu^2) invweight0, weight 1 / (2 mu^2 R), sixteen rows that
    all see a_z - aref; semi-implicit Euler.  The scalar recurrence written from those statements (known_answers.drop_recurrence) is the
    oracle's trajectory substep by substep: free fall, impact, the critically damped approach to the resting depth."""
    import known_answers as K
    z0 = K.box_origin_height(K.MARGIN + 0.002)
    o = _box_alone(z0)
    zs = []
    for _ in range(600):
        o.step()
        zs.append(o.get_object(0)[0][2])
    ref = K.drop_recurrence(z0, 600)
    assert np.abs(np.array(zs) - ref).max() < 1e-12
    depth = K.box_origin_height(K.MARGIN) - ref            # penetration of the margin
    assert depth.max() > 1.1e-3 and abs(depth[-1] - depth[-50]) < 1e-12 and 0.05e-3 < depth[-1] < 0.2e-3     # overshoot on impact, then rest
    # critically damped: after the impact the depth approaches its resting value from one side, without ringing
    k0 = int(np.argmax(depth))
    tail = depth[k0 + 40:] - depth[-1]
    assert (tail > -1e-9).all() and (np.diff(tail) < 1e-12).all()


def test_known_answer_friction_creep_and_pyramid_cone_limits():
    """(ii) computation.html 'Friction cones' (pyramidal): gravity tilted by theta puts the tangential load m g sin(theta) on the box.
    Inside the pyramid the load is carried by the (n + mu t, n - mu t) row pair of every contact, whose force difference is the viscous
    -2 mu^2 D b v_t: the box creeps at v = m g sin(theta) / sum_c 2 mu^2 D_c b (closed form; D_c from each contact's depth) -- along the
    frame axes t1 = y, t2 = -x and along their diagonal alike.  The cone itself is the pyramid: at tan(theta) = 0.9 (mu = 1) the box holds
    along the axes (limit mu) and runs away along the diagonal (limit mu / sqrt 2); at tan(theta) = 1.3 it runs away along an axis too."""
    import known_answers as K
    z_rest = K.box_origin_height(0.00089)
    s2 = np.sqrt(0.5)

    def run(u, tan):
        th = np.arctan(tan)
        o = _box_alone(z_rest, gravity=[K.G * np.sin(th) * u[0], K.G * np.sin(th) * u[1], -K.G * np.cos(th)])
        for _ in range(270):
            o.step()
        v = o.get_object(0)[1]
        o.forward()
        c = o.contacts_full()
        return v[0] * u[0] + v[1] * u[1], c["dist"][c["body"] == 24], th
    for u in ((1.0, 0.0), (0.0, 1.0), (s2, s2)):
        v, dists, th = run(u, 0.3)
        assert len(dists) == 4
        assert v == pytest.approx(K.creep_velocity(K.BOX_MASS * K.G * np.sin(th), dists), rel=1e-6)
    hold_x, hold_y, slide_d, slide_x = run((1.0, 0.0), 0.9)[0], run((0.0, 1.0), 0.9)[0], run((s2, s2), 0.9)[0], run((1.0, 0.0), 1.3)[0]
    assert 0 < hold_x < 0.02 and 0 < hold_y < 0.02 and slide_d > 0.5 and slide_x > 0.5


@pytest.mark.parametrize("j", [6, 50])
def test_known_answer_hinge_limit_penetration(j):
    """(iii) joint limits are the same soft constraint with one row, R = (1 - d) / d * dof_invweight0: a hinge driven past its +180 degree
    limit by a saturated actuator (stable-PD asked for 6 rad more: torque = torque_lim) settles where D k d |r| = torque, i.e.
    |r| = torque (1 - dmax) invweight0 / (k dmax^2) beyond the solimp width.  No gravity, no contacts; the rest of the body is held by its PD."""
    import known_answers as K
    o = OracleSim(contact=False, gravity=0.0)
    q = STD["qpos"].copy(); q[2] += 2.0; q[7 + j] = 3.0
    o.reset(q, np.zeros(75))
    act = np.zeros(75); act[j] = 6.0
    for _ in range(60):
        o.do_simulation(act, q, 15)
    pred = K.limit_penetration(KPM["torque_lim"][j], KPM["dof_invweight0"][6 + j])
    assert o.get("qpos")[7 + j] - np.pi == pytest.approx(pred, rel=2e-4) and np.abs(o.get("qvel")).max() < 0.05


def _energy(o, qpos, qvel):
    o.set_state_raw(qpos, qvel); o.forward()
    M = o.fullM(); xi = o.get("xipos").reshape(24, 3)
    return 0.5 * qvel @ M @ qvel + 9.81 * float((MASS * xi[:, 2]).sum())


def test_known_answer_free_flight_energy_drift():
    """(iv) BASELINE configs[1]: torque-free flight, no contact, 1500 substeps.  Semi-implicit Euler (v += h a, q += h v) is symplectic: the
    energy of the falling centre of mass drifts by exactly -1/2 m g^2 h^2 per substep, the internal (rotational / joint) energy stays bounded."""
    rng = np.random.default_rng(5)
    o = OracleSim(contact=False, limits=False)
    q = STD["qpos"].copy(); q[2] += 10; q[7:] += rng.normal(size=69) * 0.2
    v = np.concatenate([rng.normal(size=3), rng.normal(size=72) * 0.5])
    e0 = _energy(o, q, v)
    ke0 = e0 - 9.81 * float((MASS * o.get("xipos").reshape(24, 3)[:, 2]).sum())
    o.reset(q, v)
    for _ in range(1500):
        o.step()
    e1 = _energy(o, o.get("qpos").copy(), o.get("qvel").copy())
    closed = -0.5 * MASS.sum() * 9.81 ** 2 * H * H * 1500
    assert abs((e1 - e0) - closed) < 0.05 * ke0, (e1 - e0, closed, ke0)


def test_known_answer_contact_frame_and_pyramid_rows():
    """(v) mju_makeFrame [MJ-ext] and the pyramidal basis (computation.html 'Friction cones'): for a contact with normal n the four rows of
    the constraint Jacobian are (n + mu t1, n - mu t1, n + mu t2, n - mu t2) . (point Jacobian), t1 = the part of y (or of z when n is
    within 60 degrees of y) orthogonal to n, t2 = n x t1.  Read from efc_J at the translational columns of a free box: tilted boxes on
    the plane keep n = z (t1 = y, t2 = -x); a box resting on a tilted box gives an oblique n."""
    import known_answers as K
    from kinpoly_amd.model_compiler import STEP_KPM
    rot = lambda ax, a: np.concatenate([[np.cos(a / 2)], np.sin(a / 2) * np.asarray(ax, float) / np.linalg.norm(ax)])      # noqa: E731
    o = OracleSim(kpm=STEP_KPM)
    # slot 0: the step box rotated 35 degrees about an oblique axis (one corner 0.1 mm inside the plane's margin); slot 1: the small box flat
    # on its upper face, 0.5 mm inside
    qa = rot([1.0, 2.0, 0.3], 0.6)
    R = O.quaternion_matrix3(qa)
    og = K.KPM["obj_geoms"].reshape(-1, 18)
    g4, g1 = og[og[:, 0].astype(int) == 4][0], og[og[:, 0].astype(int) == 1][0]
    corners = np.array([[sx * g4[2], sy * g4[3], sz * g4[4]] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)]) + g4[5:8]
    origin4 = np.array([0.0, 0.0, 0.0009 - (corners @ R.T)[:, 2].min()])
    face = origin4 + R @ (g4[5:8] + [0.0, 0.0, g4[4]])                   # centre of the upper face, normal R e_z
    centre1 = face + R @ [0.0, 0.0, g1[4] - 0.0005]
    o.set_object(0, K.KPM, 4, [*origin4, *qa])
    o.set_object(1, K.KPM, 1, [*(centre1 - R @ g1[5:8]), *qa])
    q = STD["qpos"].copy(); q[0] += 30; q[2] += 50
    o.reset(q, np.zeros(75))
    o.forward()
    c = o.contacts_full(); J = o.efc_J_full()
    assert len(c["body"]) > 0
    lim = o.nefc - 4 * len(c["body"])
    seen_oblique = False
    for i, (a, b, n) in enumerate(zip(c["body"], c["b2"], c["normal"])):
        rows = K.pyramid_rows(n)
        for e in range(4):
            jr = J[lim + 4 * i + e]
            if a >= 24:                                               # entity carrying the first geom's counterpart: moves along +row
                np.testing.assert_allclose(jr[75 + 6 * (a - 24): 78 + 6 * (a - 24)], rows[e], atol=1e-12)
            if b >= 24:
                np.testing.assert_allclose(jr[75 + 6 * (b - 24): 78 + 6 * (b - 24)], -rows[e], atol=1e-12)
        seen_oblique |= bool(a >= 24 and b >= 24 and abs(n[2]) < 0.95)
    assert seen_oblique, "the box-on-tilted-box contact was not generated"
    n, t1, t2 = K.make_frame([0.0, 0.0, 1.0])
    np.testing.assert_allclose(t1, [0, 1, 0]); np.testing.assert_allclose(t2, [-1, 0, 0])
    n, t1, t2 = K.make_frame([0.1, 0.9, 0.2])                           # within 60 degrees of y: reference axis z
    assert abs(t1 @ n) < 1e-15 and t1[2] > 0.9 and np.allclose(np.cross(n, t1), t2)


def test_known_answer_box_resting_on_the_table_follows_the_two_body_recurrence():
    """(vi) round 4: the push scene's resting pair -- the table on its four upright legs (mjc_PlaneCylinder: 3 contacts per leg), the box flat on the
    table top (mjc_BoxBox: 4 contacts), 64 pyramid rows in all -- released 1 mm above their contact margins.  Vertical motion of two bodies
    under the documented soft-contact model is the 2 x 2 piecewise-quadratic problem of known_answers.stack_recurrence; the oracle's dense
    Newton solve on 75 + 12 dofs must trace it substep by substep, with 16 contacts once both pairs touch."""
    import known_answers as K
    from kinpoly_amd.model_compiler import STEP_KPM
    zt0 = -K.TABLE_FEET + K.MARGIN + 0.001
    zb0 = zt0 + K.TABLE_TOP - K.PUSH_BOX_BOTTOM + K.MARGIN + 0.001
    o = OracleSim(kpm=STEP_KPM)
    o.set_object(0, K.KPM, K.BOX_OBJ, [0.0, 0.0, zb0, 1, 0, 0, 0])
    o.set_object(1, K.KPM, K.TABLE_OBJ, [0.0, 0.0, zt0, 1, 0, 0, 0])
    q = STD["qpos"].copy(); q[0] += 30; q[2] += 50
    o.reset(q, np.zeros(75))
    zs, ncon = [], []
    for _ in range(500):
        o.step()
        zs.append((o.get_object(0)[0][2], o.get_object(1)[0][2]))
        ncon.append(len(o.contacts()[0]))
    ref = K.stack_recurrence(zb0, zt0, 500)
    assert np.abs(np.array(zs) - ref).max() < 1e-11
    assert ncon[-1] == K.N_LEG_CONTACTS + K.N_BOX_CONTACTS and max(ncon) == 16
    # nothing but vertical motion happened, and both pairs came to rest inside their margins
    qb, qt = o.get_object(0)[0], o.get_object(1)[0]
    assert np.abs(qb[:2]).max() < 1e-12 and np.abs(qt[:2]).max() < 1e-12 and abs(qb[3] - 1) < 1e-12 and abs(qt[3] - 1) < 1e-12
    gap = (ref[-1, 0] + K.PUSH_BOX_BOTTOM) - (ref[-1, 1] + K.TABLE_TOP)
    feet = ref[-1, 1] + K.TABLE_FEET
    assert 0 < gap < K.MARGIN and 0 < feet < K.MARGIN and abs(ref[-1, 0] - ref[-40, 0]) < 1e-10
    # resting depths of the documented model: n rows x D(r) x k d(r) |r| carry the weight above them (aref = -k d(r) r at rest)
    for n_rows, invw, load, r in ((4 * K.N_LEG_CONTACTS, K.TABLE_INVW, (K.TABLE_MASS + K.PUSH_BOX_MASS) * K.G, feet - K.MARGIN),
                                  (4 * K.N_BOX_CONTACTS, K.PUSH_BOX_INVW + K.TABLE_INVW, K.PUSH_BOX_MASS * K.G, gap - K.MARGIN)):
        assert n_rows * K.row_weight(r, invw) * K.K_REF * K.impedance(r) * abs(r) == pytest.approx(load, rel=1e-6)


def test_known_answer_set0_constants_of_the_compiled_model():
    """(vii) round 4: body_invweight0 / dof_invweight0 / meaninertia scale every regulariser R, every joint-limit row and the solver's
    termination test.  The compiled blob's values against an independent evaluation of engine_setconst.c's set0 definitions from the blob's
    raw body parameters (explicit centre-of-mass Jacobians at qpos0, dense inverse) and, for meaninertia, the objects' diagonal inertia
    from their XML geoms by the textbook box / cylinder formulas."""
    import known_answers as K
    kpm = K.KPM
    bw, dw, diagM = K.set0_constants(BODY_POS, BODY_IPOS, PARENT, MASS, INERTIA, kpm["dof_armature"], O.qpos_fk, O.quaternion_matrix3)
    np.testing.assert_allclose(kpm["body_invweight0"].reshape(24, 2), bw, rtol=1e-9)
    np.testing.assert_allclose(kpm["dof_invweight0"], dw, rtol=1e-9)
    obj_diag = np.concatenate([K.object_diag_inertia(i) for i in range(5)])
    mean_inertia = (diagM.sum() + obj_diag.sum()) / (75 + 30)
    assert float(kpm["opt"][16]) == pytest.approx(mean_inertia, rel=1e-9)
    # the objects' own invweight0 (the regulariser of every object contact): the free body's 6 x 6 inertia about its origin and its
    # centre-of-mass Jacobian -- NOT 1 / (m + armature): the armature sits on the origin's dofs and the centre of mass is 0.1 ... 0.44 m below it
    for i in range(5):
        inert = kpm["obj_inertial"].reshape(-1, 13)[i]
        wt, wr = K.object_invweight(i)
        assert float(inert[10]) == pytest.approx(wt, rel=1e-9) and float(inert[11]) == pytest.approx(wr, rel=1e-9)


# ---------------------------------------------------------------------------------------------- the dormant MuJoCo pin harness (tests/mj_pin.py)
def test_mjcf_written_from_the_blob_is_the_reference_scene():
    """mjcf_from_kpm: local-coordinate MJCF of the reference scene from the compiled blob (what a MuJoCo binding that no longer reads
    coordinate="global" would be handed).  Checked structurally here; MuJoCo's own reading of it is the live pin."""
    import xml.etree.ElementTree as ET
    import mj_pin as MP
    from kinpoly_amd.model_compiler import DEFAULT_KPM, read_kpm
    from kinpoly_amd import sim as kpsim
    kpm = read_kpm(DEFAULT_KPM)
    root = ET.fromstring(MP.mjcf_from_kpm(kpm))
    assert root.find("compiler").attrib.get("coordinate") is None and root.find("compiler").attrib["angle"] == "radian"
    assert float(root.find("option").attrib["timestep"]) == float(kpm["opt"][0])
    bodies = {}

    def walk(el, origin):
        for b in el.findall("body"):
            p = origin + np.array([float(x) for x in b.attrib["pos"].split()])
            bodies[b.attrib["name"]] = (p, b)
            walk(b, p)
    walk(root.find("worldbody"), np.zeros(3))
    assert list(bodies) == MP.NAMES                                   # depth-first order = the blob's body order
    got = np.stack([bodies[n][0] for n in MP.NAMES])
    np.testing.assert_allclose(got, kpm["body_gpos0"].reshape(24, 3), atol=1e-12)      # local offsets add up to the XML's global positions
    hinges = [j for n in MP.NAMES for j in bodies[n][1].findall("joint") if j.attrib["type"] == "hinge"]
    assert len(hinges) == 69 and [j.attrib["axis"] for j in hinges[:3]] == ["0 0 1", "0 1 0", "1 0 0"]
    np.testing.assert_allclose([[float(x) for x in j.attrib["range"].split()] for j in hinges], kpm["jnt_range"].reshape(-1, 2))
    free = [j for j in bodies["Pelvis"][1].findall("joint")]
    assert len(free) == 1 and free[0].attrib["type"] == "free" and free[0].attrib["armature"] == "0"
    motors = root.find("actuator").findall("motor")
    assert [m.attrib["joint"] for m in motors] == [j.attrib["name"] for j in hinges] and all(m.attrib["gear"] == "1" for m in motors)
    meshes = root.find("asset").findall("mesh")
    assert [len(m.attrib["vertex"].split()) // 3 for m in meshes] == list(np.diff(kpm["vert_adr"]))
    floor = root.find("worldbody").find("geom")
    assert floor.attrib["type"] == "plane" and floor.attrib["condim"] == "3" and floor.attrib["friction"].split()[0] in ("1.", "1", "1.0")
    # free fall: contacts disabled by flag; with the free objects of ..._all_step.xml
    assert 'contact="disable"' in MP.mjcf_from_kpm(kpm, contact=False)
    k2 = read_kpm(kpsim.STEP_KPM)
    r2 = ET.fromstring(MP.mjcf_from_kpm(k2, objects=True))
    objs = [b for b in r2.find("worldbody").findall("body") if b.attrib["name"] != "Pelvis"]
    assert len(objs) == int(k2["dims"][6]) == 5 and sum(len(b.findall("geom")) for b in objs) == int(k2["dims"][7])
    assert all(b.find("joint").attrib["type"] == "free" for b in objs)
    # the model comparison of the pin, fed with the blob's own arrays, reports zeros
    same = MP.compare_model({k: kpm[k] for k in ("body_mass", "body_ipos", "body_inertia", "body_invweight0", "dof_invweight0")}, kpm)
    assert max(same.values()) == 0.0


def test_pin_harness_runs_end_to_end_with_the_oracle_in_mujocos_seat():
    """The harness's stepping logic (do_simulation's loop around a backend's qM / qfrc_bias, the free-fall and contact protocols, the report)
    exercised with OracleBackend where MuJoCo will sit: the Python control loop around the backend equals the C oracle's own do_simulation,
    two oracle backends never part, and without a MuJoCo binding the report is None (bench.py prints "mujoco_pin": null)."""
    import mj_pin as MP
    from kinpoly_amd.model_compiler import DEFAULT_KPM, read_kpm
    kpm = read_kpm(DEFAULT_KPM)
    std = np.load(os.path.join(ROOT, "tests", "golden", "standing_neutral.npz"))
    b = MP.OracleBackend()
    b.set_state(std["qpos"], std["qvel"])
    o = OracleSim()
    o.reset(std["qpos"], std["qvel"])
    rng = np.random.default_rng(5)
    for t in range(3):
        a = rng.normal(size=75) * 0.1
        for _ in range(15):
            MP.control_substep(b, a, std["qpos"], kpm)
        o.do_simulation(a, std["qpos"], 15)
        np.testing.assert_allclose(b.qpos(), o.get("qpos"), rtol=0, atol=1e-10)      # scipy's Cholesky vs the C oracle's: 1e-12
    assert len(b.contacts()) > 0 and all(0 <= body < 24 for body, _ in b.contacts())
    rep = MP.run_pin("contact", MP.OracleBackend(), {"oracle": MP.OracleBackend()}, kpm, std["qpos"], std["qvel"], n_steps=2,
                     hip=lambda q, v, acts, tgt: np.tile(q, (len(acts), 1)))
    assert rep["oracle"]["max_dqpos"] == 0.0 and rep["oracle"]["contact_set_diffs"] == 0 and rep["oracle"]["first_step_above_1e-3"] is None
    assert rep["hip"]["max_dqpos"] > 0                                                 # the stand-in "product" (frozen pose) is seen to part
    ff = MP.run_pin("free_fall", MP.OracleBackend(contact=False), {"oracle": MP.OracleBackend(contact=False)}, kpm, std["qpos"], std["qvel"], n_steps=20)
    assert ff["oracle"]["max_dqpos"] == 0.0
    q, v = MP.free_fall_state(std["qpos"])
    assert q[2] > 10 and abs(np.linalg.norm(q[3:7]) - 1) < 1e-6 and np.abs(q[7:]).max() <= np.pi
    if MP.find_mujoco() is None:
        assert MP.pin_report() is None


def test_model_table_names_the_entry_that_differs():
    """tools/mujoco_pin.py --report: the model-array comparison entry by entry.  With one body's mass and one dof's invweight perturbed the table names exactly
    those (array, body / dof, both values), every other row reads ok."""
    import mj_pin as MP
    from kinpoly_amd.model_compiler import DEFAULT_KPM, read_kpm
    kpm = read_kpm(DEFAULT_KPM)
    arrays = {k: np.array(kpm[k], float).copy() for k in ("body_mass", "body_ipos", "body_inertia", "body_invweight0", "dof_invweight0")}
    arrays["body_mass"][7] *= 1.0 + 3e-5
    arrays["dof_invweight0"][40] *= 1.0 - 2e-4
    rows = {r["array"]: r for r in MP.model_table(arrays, kpm)}
    assert rows["body_mass"]["status"] == "DIFFERS" and rows["body_mass"]["worst_item"] == "body 7 component 0" and rows["body_mass"]["beyond_tolerance"] == 1
    assert rows["dof_invweight0"]["status"] == "DIFFERS" and rows["dof_invweight0"]["worst_item"] == "dof 40 component 0"
    assert abs(rows["dof_invweight0"]["worst"] - 2e-4) < 1e-8
    assert rows["body_ipos"]["status"] == "ok" and rows["body_inertia"]["status"] == "ok" and rows["body_pos"]["status"].startswith("not provided")
    text = MP.format_model_table(list(rows.values()))
    assert "body 7 component 0" in text and text.count("\n") == len(rows) + 1
