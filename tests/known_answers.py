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
