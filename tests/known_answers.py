"""Known answers of MuJoCo's DOCUMENTED constraint model (computation.html: soft constraint model, solref / solimp, pyramidal friction
cones; modeling.html: solver parameters), written down independently of oracle/kp_oracle.c and of the HIP kernels and used by both
tests/test_physics_oracle.py (fp64 oracle, CPU) and tests/test_gpu_round3.py (HIP kernel).  The physics side of the oracle cannot be
pinned against MuJoCo itself (not installable here or on the GPU box); these closed forms and scalar recurrences shrink what is left
unverified to MuJoCo's implementation details.

Scene for the contact tests: the reference's free `step` box (assets/mujoco_models/humanoid_smpl_neutral_mesh_all_step.xml:190-214,
0.8 x 0.8 x 0.34 m, 40 kg) alone on the floor plane, the humanoid parked 30 m away."""
import numpy as np

from kinpoly_amd.model_compiler import STEP_KPM, read_kpm

KPM = read_kpm(STEP_KPM)
OPT = KPM["opt"]
H = float(OPT[0])
G = 9.81
STEP_OBJ = 4
_inert = KPM["obj_inertial"].reshape(-1, 13)[STEP_OBJ]
BOX_MASS, BOX_INVW, BOX_ARM = float(_inert[0]), float(_inert[10]), float(_inert[12])
_og = KPM["obj_geoms"].reshape(-1, 18)
_g = _og[_og[:, 0].astype(int) == STEP_OBJ][0]
BOX_HALF = _g[2:5].copy()                      # half extents
BOX_ORIGIN_ABOVE_CENTRE = -float(_g[7])        # the body origin sits this far above the geom centre (geom pos z = -0.2)
SOLREF_TC, SOLREF_DR = max(float(OPT[4]), 2 * H), float(OPT[5])       # refsafe: timeconst >= 2 h
D0, DW, WIDTH, MID, POWER = [float(x) for x in OPT[6:11]]
MU, MARGIN = float(OPT[11]), float(OPT[14])
K_REF = 1.0 / (DW * DW * SOLREF_TC * SOLREF_TC * SOLREF_DR * SOLREF_DR)   # stiffness of the reference acceleration
B_REF = 2.0 / (DW * SOLREF_TC)                                            # damping


def impedance(r):
    """solimp = (d0, dwidth, width, midpoint, power): d(r) rises from d0 at r = 0 to dwidth at |r| = width (power-2 sigmoid)."""
    x = min(abs(r) / WIDTH, 1.0)
    assert POWER == 2.0
    y = x * x / MID if x <= MID else 1.0 - (1.0 - x) ** 2 / (1.0 - MID)
    return D0 + y * (DW - D0)


def row_weight(r, invw):
    """efc_D of one pyramid row of a frictional contact: regulariser R = (1 - d) / d * (1 + mu^2) * invweight0 scaled by 2 mu^2."""
    d = impedance(r)
    R = max(1e-15, (1.0 - d) / d * (1.0 + MU * MU) * invw)
    return 1.0 / (2.0 * MU * MU * R)


def aref(r, v):
    """reference acceleration of a constraint row with position residual r and velocity v"""
    return -B_REF * v - K_REF * impedance(r) * r


def box_origin_height(corner_dist):
    """z of the box's body origin when its bottom corners are `corner_dist` above the plane"""
    return corner_dist + float(BOX_HALF[2]) + BOX_ORIGIN_ABOVE_CENTRE


def drop_recurrence(z0, n_steps):
    """A flat box dropped on the plane, vertical motion only: all 4 corner contacts x 4 pyramid rows see the same residual a_z - aref
    (their tangential parts do not move), so the solver's primal problem  min 1/2 M (a - a0)^2 + 1/2 sum_rows D min(0, a - aref)^2
    is scalar with the closed-form minimiser below; integrated with MuJoCo's semi-implicit Euler (v += h a, z += h v).  Contacts exist
    while dist <= margin.  M = mass + free-joint armature (the armature adds inertia, not weight: the unconstrained acceleration is
    -g m / M).  Returns z of the body origin after each substep."""
    M = BOX_MASS + BOX_ARM
    a0 = -G * BOX_MASS / M
    z, v, out = z0, 0.0, []
    for _ in range(n_steps):
        dist = z - float(BOX_HALF[2]) - BOX_ORIGIN_ABOVE_CENTRE
        a = a0
        if dist <= MARGIN:
            r = dist - MARGIN
            ar, D = aref(r, v), row_weight(r, BOX_INVW)
            if a0 - ar < 0.0:                                  # the unconstrained acceleration violates the rows: they are active
                a = (M * a0 + 16.0 * D * ar) / (M + 16.0 * D)
        v += H * a
        z += H * v
        out.append(z)
    return np.array(out)


def creep_velocity(tilt_force, contact_dists):
    """Inside the friction pyramid a tangential load is carried by the (+, -) row pair of every contact along that tangent:
    f+- = -D (k d r +- b mu v_t), tangential force per contact mu (f+ - f-) = -2 mu^2 D b v_t: viscous creep.  contact_dists: the
    measured distances of the contacts in steady state (D depends on the depth through the impedance)."""
    c = sum(2.0 * MU * MU * row_weight(d - MARGIN, BOX_INVW) * B_REF for d in contact_dists)
    return tilt_force / c


def limit_penetration(torque, invw):
    """A hinge pushed into its limit by a constant torque settles where the limit row's force D k d |r| equals the torque; for
    |r| > solimp width d = dwidth:  |r| = torque * (1 - d) * invweight0 / (k d^2)."""
    r = torque * (1.0 - DW) * invw / (K_REF * DW * DW)
    assert r > WIDTH
    return r


def make_frame(n):
    """mju_makeFrame: complete the contact normal to a right-handed frame (n, t1, t2): the reference axis is y unless n is within
    60 degrees of it (|n.y| >= 0.5), then z; t1 = its part orthogonal to n, normalised; t2 = n x t1."""
    n = np.asarray(n, float) / np.linalg.norm(n)
    ref = np.array([0.0, 1.0, 0.0]) if abs(n[1]) < 0.5 else np.array([0.0, 0.0, 1.0])
    t1 = ref - n.dot(ref) * n
    t1 /= np.linalg.norm(t1)
    return n, t1, np.cross(n, t1)


def pyramid_rows(n):
    """the four rows of a condim-3 pyramidal contact: n + mu t1, n - mu t1, n + mu t2, n - mu t2"""
    n, t1, t2 = make_frame(n)
    return np.stack([n + MU * t1, n - MU * t1, n + MU * t2, n - MU * t2])


# ------------------------------------------------------------------ round 4: a box resting on the table (the data set's only box - box pair) and set0's constants
def _obj(idx):
    inert = KPM["obj_inertial"].reshape(-1, 13)[idx]
    return float(inert[0]), float(inert[10]), float(inert[12])          # mass, translational invweight0, free-joint armature


BOX_OBJ, TABLE_OBJ = 1, 2
PUSH_BOX_MASS, PUSH_BOX_INVW, PUSH_BOX_ARM = _obj(BOX_OBJ)
TABLE_MASS, TABLE_INVW, TABLE_ARM = _obj(TABLE_OBJ)
_gb = _og[_og[:, 0].astype(int) == BOX_OBJ][0]
_gt = _og[_og[:, 0].astype(int) == TABLE_OBJ]
PUSH_BOX_BOTTOM = float(_gb[7]) - float(_gb[4])                         # z of the box's bottom face relative to its body origin (-0.22)
TABLE_TOP = float(_gt[0][7]) + float(_gt[0][4])                         # top face of the table top (-0.09)
TABLE_FEET = float(_gt[1][7]) - float(_gt[1][3])                        # bottom caps of the four upright leg cylinders (-0.79)
N_LEG_CONTACTS = 12     # mjc_PlaneCylinder on an upright cylinder: the rim point along its x axis and the two points 120 degrees either side, x 4 legs
N_BOX_CONTACTS = 4      # mjc_BoxBox, face on face: the four corners of the smaller face


def stack_recurrence(zb0, zt0, n_steps):
    """The push scene at rest, vertical motion only: the table on its four legs on the plane, the box lying flat on the table top.  Two unknown
    accelerations (a_b, a_t); N_LEG_CONTACTS x 4 pyramid rows see a_t - aref(r_t, v_t) with the table's row weight (the plane has no weight),
    N_BOX_CONTACTS x 4 rows see (a_b - a_t) - aref(r_bt, v_b - v_t) with the weight of invweight0(box) + invweight0(table)  (a contact's
    regulariser uses the sum of the two bodies' invweight0).  The primal problem
        min  1/2 M_b (a_b - a0_b)^2 + 1/2 M_t (a_t - a0_t)^2 + 1/2 n_t D_t min(0, a_t - ar_t)^2 + 1/2 n_bt D_bt min(0, a_b - a_t - ar_bt)^2
    is strictly convex and piecewise quadratic: the minimiser is the solution of the one active-set case that is consistent with itself.
    Returns (z_box, z_table) of the body origins after each substep."""
    Mb, Mt = PUSH_BOX_MASS + PUSH_BOX_ARM, TABLE_MASS + TABLE_ARM
    a0b, a0t = -G * PUSH_BOX_MASS / Mb, -G * TABLE_MASS / Mt
    zb, vb, zt, vt, out = zb0, 0.0, zt0, 0.0, []
    for _ in range(n_steps):
        dist_t = zt + TABLE_FEET
        dist_bt = (zb + PUSH_BOX_BOTTOM) - (zt + TABLE_TOP)
        has_t, has_bt = dist_t <= MARGIN, dist_bt <= MARGIN
        ct = cbt = art = arbt = 0.0
        if has_t:
            r = dist_t - MARGIN
            ct, art = 4.0 * N_LEG_CONTACTS * row_weight(r, TABLE_INVW), aref(r, vt)
        if has_bt:
            r = dist_bt - MARGIN
            cbt, arbt = 4.0 * N_BOX_CONTACTS * row_weight(r, PUSH_BOX_INVW + TABLE_INVW), aref(r, vb - vt)
        sol = None
        for act_t in ((True, False) if has_t else (False,)):
            for act_bt in ((True, False) if has_bt else (False,)):
                kt, kbt = (ct if act_t else 0.0), (cbt if act_bt else 0.0)
                # gradient = 0:  Mb (ab - a0b) + kbt (ab - at - arbt) = 0 ;  Mt (at - a0t) + kt (at - art) - kbt (ab - at - arbt) = 0
                A = np.array([[Mb + kbt, -kbt], [-kbt, Mt + kt + kbt]])
                rhs = np.array([Mb * a0b + kbt * arbt, Mt * a0t + kt * art - kbt * arbt])
                ab, at = np.linalg.solve(A, rhs)
                if (not has_t or ((at - art < 0.0) == act_t)) and (not has_bt or ((ab - at - arbt < 0.0) == act_bt)):
                    sol = (ab, at)
        assert sol is not None
        ab, at = sol
        vb += H * ab; zb += H * vb
        vt += H * at; zt += H * vt
        out.append((zb, zt))
    return np.array(out)


def set0_constants(body_pos, body_ipos, parent, mass, inertia, armature, qpos_fk, quaternion_matrix3):
    """mjModel.body_invweight0 / dof_invweight0 and the humanoid's share of stat.meaninertia as engine_setconst.c's set0 defines them, evaluated
    from scratch at qpos0 (root at its XML position with the identity quaternion, every hinge at zero):  M = sum_b m_b Jv_b^T Jv_b +
    Jw_b^T (R_b I_b R_b^T) Jw_b + diag(armature) with the explicit world-frame Jacobians of every body's centre of mass;
    body_invweight0[b] = (tr(Jv M^-1 Jv^T) / 3, tr(Jw M^-1 Jw^T) / 3); dof_invweight0 = diag(M^-1), averaged over the three translational and
    over the three rotational dofs of the free joint; meaninertia = mean of diag(M) (over ALL dofs of the scene: the caller adds the objects).
    qpos_fk / quaternion_matrix3: the reference-pinned FK restatement (oracle/np_oracle.py)."""
    nb = len(parent)
    qpos0 = np.zeros(76); qpos0[:3] = body_pos[0]; qpos0[3] = 1.0
    fk = qpos_fk(qpos0, body_pos, body_ipos, parent)
    R = [quaternion_matrix3(q) for q in fk["wbquat"]]
    axes, anchors, trans, body_of = [], [], [], []
    for k in range(3):
        axes.append(np.eye(3)[k]); anchors.append(np.zeros(3)); trans.append(True); body_of.append(0)
    for k in range(3):
        axes.append(R[0][:, k]); anchors.append(fk["wbpos"][0]); trans.append(False); body_of.append(0)
    for b in range(1, nb):                                     # hinges z, y, x at the body origin; all angles are zero at qpos0
        for k in (2, 1, 0):
            axes.append(R[parent[b]][:, k]); anchors.append(fk["wbpos"][b]); trans.append(False); body_of.append(b)
    nv = len(axes)
    anc = np.zeros((nb, nb), bool)
    for b in range(nb):
        k = b
        while k >= 0:
            anc[b, k] = True; k = parent[k]
    M = np.diag(np.asarray(armature, float).copy())
    Js = []
    for b in range(nb):
        Jv, Jw = np.zeros((3, nv)), np.zeros((3, nv))
        for d in range(nv):
            if anc[b, body_of[d]]:
                if trans[d]:
                    Jv[:, d] = axes[d]
                else:
                    Jw[:, d] = axes[d]; Jv[:, d] = np.cross(axes[d], fk["body_com"][b] - anchors[d])
        Ib = inertia[b]; I3 = np.array([[Ib[0], Ib[3], Ib[4]], [Ib[3], Ib[1], Ib[5]], [Ib[4], Ib[5], Ib[2]]])
        M += mass[b] * Jv.T @ Jv + Jw.T @ (R[b] @ I3 @ R[b].T) @ Jw
        Js.append((Jv, Jw))
    Minv = np.linalg.inv(M)
    body_invw = np.array([[np.trace(Jv @ Minv @ Jv.T) / 3.0, np.trace(Jw @ Minv @ Jw.T) / 3.0] for Jv, Jw in Js])
    dof_invw = np.diag(Minv).copy()
    dof_invw[0:3] = dof_invw[0:3].mean(); dof_invw[3:6] = dof_invw[3:6].mean()
    return body_invw, dof_invw, np.diag(M).copy()


def object_diag_inertia(obj_index):
    """diag of qM for one free object at qpos0 from its XML geoms alone: 3 x (mass + armature) and the body-frame inertia about the BODY ORIGIN
    (a free joint's rotational dofs are body-frame axes through the body origin) of the explicit-mass boxes / z-cylinders, + armature."""
    gs = _og[_og[:, 0].astype(int) == obj_index]
    arm = float(KPM["obj_inertial"].reshape(-1, 13)[obj_index][12])
    m_tot, I = 0.0, np.zeros((3, 3))
    for g in gs:
        typ, size, pos, R, m = int(g[1]), g[2:5], g[5:8], g[8:17].reshape(3, 3), float(g[17])
        if typ == 0:
            a, b, c = size
            Il = np.diag([m / 3.0 * (b * b + c * c), m / 3.0 * (a * a + c * c), m / 3.0 * (a * a + b * b)])
        else:
            r, hh = size[0], size[1]
            Il = np.diag([m * (3 * r * r + 4 * hh * hh) / 12.0, m * (3 * r * r + 4 * hh * hh) / 12.0, m * r * r / 2.0])
        I += R @ Il @ R.T + m * (pos @ pos * np.eye(3) - np.outer(pos, pos))
        m_tot += m
    return np.concatenate([np.full(3, m_tot + arm), np.diag(I) + arm])


def object_invweight(obj_index):
    """body_invweight0 (translational, rotational) of one free object by set0's definition: its 6 x 6 joint-space inertia at qpos0 (translation
    of the BODY ORIGIN along world axes, rotation about body axes through the origin; the centre of mass sits at c, so the two couple through
    m [c]x), + armature on the diagonal, and the centre-of-mass Jacobian Jv = [1, -[c]x], Jw = [0, 1]."""
    gs = _og[_og[:, 0].astype(int) == obj_index]
    arm = float(KPM["obj_inertial"].reshape(-1, 13)[obj_index][12])
    m = float(gs[:, 17].sum())
    c = (gs[:, 17:18] * gs[:, 5:8]).sum(0) / m
    Io = np.zeros((3, 3))                      # inertia about the body origin
    for g in gs:
        typ, size, pos, R, mg = int(g[1]), g[2:5], g[5:8], g[8:17].reshape(3, 3), float(g[17])
        if typ == 0:
            a, b, cc = size
            Il = np.diag([mg / 3.0 * (b * b + cc * cc), mg / 3.0 * (a * a + cc * cc), mg / 3.0 * (a * a + b * b)])
        else:
            r, hh = size[0], size[1]
            Il = np.diag([mg * (3 * r * r + 4 * hh * hh) / 12.0, mg * (3 * r * r + 4 * hh * hh) / 12.0, mg * r * r / 2.0])
        Io += R @ Il @ R.T + mg * (pos @ pos * np.eye(3) - np.outer(pos, pos))
    cx = np.array([[0, -c[2], c[1]], [c[2], 0, -c[0]], [-c[1], c[0], 0]])
    M = np.block([[m * np.eye(3), -m * cx], [m * cx, Io]]) + arm * np.eye(6)      # v_com = v - c x w = v - [c]x w
    Minv = np.linalg.inv(M)
    Jv, Jw = np.hstack([np.eye(3), -cx]), np.hstack([np.zeros((3, 3)), np.eye(3)])
    return np.trace(Jv @ Minv @ Jv.T) / 3.0, np.trace(Jw @ Minv @ Jw.T) / 3.0
