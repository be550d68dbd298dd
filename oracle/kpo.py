"""ctypes wrapper around oracle/_build/libkp_oracle.so (CPU fp64 oracle).

TEST INFRASTRUCTURE ONLY -- see the header of kp_oracle.c.  Imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg; never by kinpoly_amd/.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libkp_oracle.so")
DEFAULT_KPM = os.path.join(os.path.dirname(_HERE), "kinpoly_amd", "assets", "smpl_humanoid.kpm")
NQ, NV, NU, NB, NM = 76, 75, 69, 24, 1221


def build(force: bool = False):
    srcs = [os.path.join(_HERE, f) for f in ("kp_oracle.c", "kp_collide.h", "Makefile")]
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        P, D, I = C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int)
        L.kpo_model_load.restype = P; L.kpo_model_load.argtypes = [C.c_char_p]
        L.kpo_model_free.argtypes = [P]
        L.kpo_model_set_flags.argtypes = [P, C.c_int, C.c_int]
        L.kpo_model_set_gravity.argtypes = [P, C.c_double]
        L.kpo_model_set_gravity3.argtypes = [P, C.c_double, C.c_double, C.c_double]
        L.kpo_model_set_ls_exact.argtypes = [P, C.c_int]
        L.kpo_model_set_planemesh.argtypes = [P, C.c_int, C.c_double]
        L.kpo_data_new.restype = P
        L.kpo_data_free.argtypes = [P]
        for f in ("kpo_forward", "kpo_step"):
            getattr(L, f).argtypes = [P, P]
        L.kpo_reset.argtypes = [P, P, D, D]
        L.kpo_fullM.argtypes = [P, P, D]
        L.kpo_compute_torque.argtypes = [P, P, D, D, D]
        L.kpo_rfc_implicit.argtypes = [P, P, D]
        L.kpo_do_simulation.argtypes = [P, P, D, D, C.c_int]
        L.kpo_set_qpos_qvel.argtypes = [P, D, D]
        L.kpo_set_ctrl.argtypes = [P, D, D]
        L.kpo_solveM.argtypes = [P, P, D]
        L.kpo_get_contacts.argtypes = [P, I, D, D]
        L.kpo_get_contact_normals.argtypes = [P, D]
        L.kpo_flops_get.restype = C.c_double
        L.kpo_set_geoms.argtypes = [P, C.c_int, D]
        L.kpo_set_object.argtypes = [P, C.c_int, D, C.c_int, D, D, D]
        L.kpo_clear_objects.argtypes = [P]
        L.kpo_get_object.argtypes = [P, C.c_int, D, D]
        L.kpo_get_object_dyn.argtypes = [P, C.c_int, D, D]
        L.kpo_get_qacc_full.argtypes = [P, D]
        L.kpo_get_qacc_smooth_full.argtypes = [P, D]
        L.kpo_get_efc_J_full.argtypes = [P, D]
        L.kpo_get_contact_pairs.argtypes = [P, I, I]
        L.kpo_get_efc.argtypes = [P, D, D, D]
        L.kpo_get_efc_J.argtypes = [P, D]
        L.kpo_rollout_batch.argtypes = [P, C.c_int, D, D, D, D, C.c_int, C.c_int]
        L.kpo_narrowphase.argtypes = [C.c_int, D, D, C.c_int, D, D, C.c_int, I, I, C.c_double, C.c_double, C.c_int, D]; L.kpo_narrowphase.restype = C.c_int
        for g in ("ncon", "nefc", "niter"):
            getattr(L, "kpo_get_" + g).argtypes = [P]; getattr(L, "kpo_get_" + g).restype = C.c_int
        for f in _FIELDS:
            getattr(L, "kpo_get_" + f).argtypes = [P, D]
        _lib = L
    return _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


_FIELDS = dict(qpos=NQ, qvel=NV, xpos=72, xquat=96, xipos=72, qM=NM, qfrc_bias=NV, qacc=NV, qacc_smooth=NV,
               subtree_com=3, ctrl=NU, qfrc_applied=NV, qfrc_constraint=NV, cvel=144)


class OracleSim:
    """One scalar fp64 environment."""

    def __init__(self, kpm: str = DEFAULT_KPM, contact=True, limits=True, gravity=None, ls_exact=False, planemesh=None):
        L = lib()
        self.L = L
        self.m = L.kpo_model_load(kpm.encode())
        assert self.m, f"cannot load {kpm}"
        L.kpo_model_set_flags(self.m, int(contact), int(limits))
        if gravity is not None:                              # scalar: gravity z; 3-vector: the whole mjOption.gravity (tilted-plane tests)
            if np.ndim(gravity) == 0:
                L.kpo_model_set_gravity(self.m, float(gravity))
            else:
                L.kpo_model_set_gravity3(self.m, float(gravity[0]), float(gravity[1]), float(gravity[2]))
        L.kpo_model_set_ls_exact(self.m, int(ls_exact))     # default: MuJoCo's PrimalSearch; True: the exact minimiser (as the HIP kernel)
        if planemesh is not None:                           # (maxplanemesh, tolplanemesh) of mjc_PlaneConvex; default: the blob's (3, 0.3)
            L.kpo_model_set_planemesh(self.m, int(planemesh[0]), float(planemesh[1]))
        self.d = L.kpo_data_new()

    def __del__(self):
        try:
            self.L.kpo_data_free(self.d); self.L.kpo_model_free(self.m)
        except Exception:
            pass

    def reset(self, qpos, qvel):
        qpos = np.ascontiguousarray(qpos, np.float64); qvel = np.ascontiguousarray(qvel, np.float64)
        self.L.kpo_reset(self.m, self.d, _dp(qpos), _dp(qvel))

    def set_state_raw(self, qpos, qvel):
        qpos = np.ascontiguousarray(qpos, np.float64); qvel = np.ascontiguousarray(qvel, np.float64)
        self.L.kpo_set_qpos_qvel(self.d, _dp(qpos), _dp(qvel))

    def set_ctrl(self, ctrl, applied6=None):
        ctrl = np.ascontiguousarray(ctrl, np.float64)
        a = None if applied6 is None else np.ascontiguousarray(applied6, np.float64)
        self.L.kpo_set_ctrl(self.d, _dp(ctrl), None if a is None else _dp(a))

    def set_geoms(self, packed):
        """packed [n,17]: type, size3, pos3, R9 (world), invweight (see object_geoms())."""
        g = np.ascontiguousarray(packed, np.float64).reshape(-1, 17)
        self.L.kpo_set_geoms(self.d, g.shape[0], _dp(g))

    def set_object(self, slot, kpm: dict, obj_index, qpos7, qvel6=None):
        """Make object `obj_index` of the compiled model (chair, box, table, Can, step) the dynamic free body in `slot` (0/1)."""
        inert = np.ascontiguousarray(kpm["obj_inertial"].reshape(-1, 13)[obj_index], np.float64)
        og = kpm["obj_geoms"].reshape(-1, 18)
        g = np.ascontiguousarray(og[og[:, 0].astype(int) == obj_index][:, 1:17], np.float64)
        q = np.ascontiguousarray(qpos7, np.float64); v = np.zeros(6) if qvel6 is None else np.ascontiguousarray(qvel6, np.float64)
        self.L.kpo_set_object(self.d, slot, _dp(inert), g.shape[0], _dp(g), _dp(q), _dp(v))

    def clear_objects(self):
        self.L.kpo_clear_objects(self.d)

    def get_object(self, slot):
        q, v = np.zeros(7), np.zeros(6); self.L.kpo_get_object(self.d, slot, _dp(q), _dp(v)); return q, v

    def object_dyn(self, slot):
        M, b = np.zeros((6, 6)), np.zeros(6); self.L.kpo_get_object_dyn(self.d, slot, _dp(M), _dp(b)); return M, b

    def qacc_full(self):
        out = np.zeros(NV + 12); self.L.kpo_get_qacc_full(self.d, _dp(out)); return out

    def qacc_smooth_full(self):
        out = np.zeros(NV + 12); self.L.kpo_get_qacc_smooth_full(self.d, _dp(out)); return out

    def efc_J_full(self):
        n = self.nefc; J = np.zeros((max(n, 1), NV + 12)); self.L.kpo_get_efc_J_full(self.d, _dp(J)); return J[:n]

    def contact_pairs(self):
        n = self.L.kpo_get_ncon(self.d)
        b1 = np.zeros(64, np.int32); b2 = np.zeros(64, np.int32)
        self.L.kpo_get_contact_pairs(self.d, b1.ctypes.data_as(C.POINTER(C.c_int)), b2.ctypes.data_as(C.POINTER(C.c_int)))
        return b1[:n], b2[:n]

    def forward(self):
        self.L.kpo_forward(self.m, self.d)

    def step(self):
        self.L.kpo_step(self.m, self.d)

    def get(self, name):
        out = np.zeros(_FIELDS[name])
        getattr(self.L, "kpo_get_" + name)(self.d, _dp(out))
        return out

    def fullM(self):
        M = np.zeros((NV, NV)); self.L.kpo_fullM(self.m, self.d, _dp(M)); return M

    def solveM(self, x):
        x = np.array(x, np.float64); self.L.kpo_solveM(self.m, self.d, _dp(x)); return x

    def compute_torque(self, ctrl, target_qpos):
        ctrl = np.ascontiguousarray(ctrl, np.float64); tq = np.ascontiguousarray(target_qpos, np.float64)
        out = np.zeros(NU); self.L.kpo_compute_torque(self.m, self.d, _dp(ctrl), _dp(tq), _dp(out)); return out

    def rfc_implicit(self, vf):
        vf = np.ascontiguousarray(vf, np.float64); self.L.kpo_rfc_implicit(self.m, self.d, _dp(vf))

    def do_simulation(self, action, target_qpos, n_frames=15):
        a = np.ascontiguousarray(action, np.float64); tq = np.ascontiguousarray(target_qpos, np.float64)
        self.L.kpo_do_simulation(self.m, self.d, _dp(a), _dp(tq), n_frames)

    def contacts(self):
        n = self.L.kpo_get_ncon(self.d)
        body = np.zeros(64, np.int32); pos = np.zeros((64, 3)); dist = np.zeros(64)
        self.L.kpo_get_contacts(self.d, body.ctypes.data_as(C.POINTER(C.c_int)), _dp(pos), _dp(dist))
        return body[:n], pos[:n], dist[:n]

    def contacts_full(self):
        """dict(body, b2, dist, pos, normal) of the last collision pass (same layout as KpSim.contacts())."""
        body, pos, dist = self.contacts()
        b1, b2 = self.contact_pairs()
        nrm = np.zeros((64, 3)); self.L.kpo_get_contact_normals(self.d, _dp(nrm))
        return dict(body=body.astype(int), b2=b2.astype(int), dist=dist, pos=pos, normal=nrm[:len(body)])

    def efc(self):
        n = self.nefc
        f, D, a, J = np.zeros(max(n, 1)), np.zeros(max(n, 1)), np.zeros(max(n, 1)), np.zeros((max(n, 1), NV))
        self.L.kpo_get_efc(self.d, _dp(f), _dp(D), _dp(a)); self.L.kpo_get_efc_J(self.d, _dp(J))
        return f[:n], D[:n], a[:n], J[:n]

    @property
    def nefc(self):
        return self.L.kpo_get_nefc(self.d)

    @property
    def niter(self):
        return self.L.kpo_get_niter(self.d)

    def rollout_batch(self, qpos, qvel, action, target, n_steps, n_frames=15):
        """CPU-baseline driver: n envs sequentially on ONE core (the reference is one env per process)."""
        qpos = np.ascontiguousarray(qpos, np.float64).copy(); qvel = np.ascontiguousarray(qvel, np.float64).copy()
        a = np.ascontiguousarray(action, np.float64); t = np.ascontiguousarray(target, np.float64)
        self.L.kpo_rollout_batch(self.m, qpos.shape[0], _dp(qpos), _dp(qvel), _dp(a), _dp(t), n_steps, n_frames)
        return qpos, qvel


def object_geoms(kpm: dict, obj_qpos35, max_dist=50.0):
    """World-frame collision geoms [n,17] of the objects that are not parked (convert_obj_qpos parks inactive
    objects at [(i+1)*100, 100, 0] with a zero quaternion, kin_poly/envs/humanoid_ar_v1.py:479-496)."""
    og = kpm["obj_geoms"].reshape(-1, 18); mass = kpm["obj_mass"]
    out = []
    for gi in range(og.shape[0]):
        oi = int(og[gi, 0])
        pose = np.asarray(obj_qpos35[7 * oi: 7 * oi + 7], float)
        if np.linalg.norm(pose[:3]) > max_dist:
            continue
        q = pose[3:7]; n = np.linalg.norm(q)
        q = np.array([1.0, 0, 0, 0]) if n < 1e-15 else q / n
        w, x, y, z = q
        R = np.array([[w * w + x * x - y * y - z * z, 2 * (x * y - w * z), 2 * (x * z + w * y)],
                      [2 * (x * y + w * z), w * w - x * x + y * y - z * z, 2 * (y * z - w * x)],
                      [2 * (x * z - w * y), 2 * (y * z + w * x), w * w - x * x - y * y + z * z]])
        Rg = R @ og[gi, 8:17].reshape(3, 3)
        pos = pose[:3] + R @ og[gi, 5:8]
        out.append(np.concatenate([[og[gi, 1]], og[gi, 2:5], pos, Rg.reshape(-1), [1.0 / mass[oi]]]))
    return np.array(out).reshape(-1, 17)


def shape_record(kind, size=(0, 0, 0), pos=(0, 0, 0), mat=None, center=None):
    """[type, size3 | hull centre, pos3, mat9] for narrowphase(): kind in ('box', 'cylinder', 'hull')."""
    t = {"box": 0, "cylinder": 1, "hull": 2}[kind]
    m = np.eye(3) if mat is None else np.asarray(mat, float)
    s = np.asarray(center if t == 2 else size, float)
    return np.concatenate([[t], np.resize(s, 3) if len(s) == 3 else np.concatenate([s, np.zeros(3 - len(s))]), np.asarray(pos, float), m.reshape(-1)])


def narrowphase(kind, a=None, b=None, verts_a=None, verts_b=None, graph=None, margin=0.001, tol_rbound=0.0, maxcon=3):
    """One geom pair through the oracle's restatement of the MuJoCo narrow phase (kp_collide.h):
    kind = 'convex' (a, b) | 'plane_box' (a) | 'plane_cylinder' (a) | 'box_box' (a, b) | 'plane_mesh' (b + graph = neighbour lists;
    tol_rbound = tolplanemesh * geom_rbound, maxcon = maxplanemesh).
    Returns [n, 7] rows (dist, pos3, normal3 from geom 1 to geom 2)."""
    L = lib()
    k = {"convex": 0, "plane_box": 1, "plane_cylinder": 2, "box_box": 3, "plane_mesh": 4}[kind]
    ip = lambda x: x.ctypes.data_as(C.POINTER(C.c_int))  # noqa: E731
    va = None if verts_a is None else np.ascontiguousarray(verts_a, np.float64)
    vb = None if verts_b is None else np.ascontiguousarray(verts_b, np.float64)
    adr = nbr = None
    if graph is not None:
        adr = np.zeros(len(graph) + 1, np.int32); adr[1:] = np.cumsum([len(g) for g in graph]); nbr = np.array([j for g in graph for j in g] or [0], np.int32)
    out = np.zeros((8, 7))
    ar = None if a is None else np.ascontiguousarray(a, np.float64); br = None if b is None else np.ascontiguousarray(b, np.float64)
    n = L.kpo_narrowphase(k, None if ar is None else _dp(ar), None if va is None else _dp(va), 0 if va is None else len(va),
                          None if br is None else _dp(br), None if vb is None else _dp(vb), 0 if vb is None else len(vb),
                          None if adr is None else ip(adr), None if nbr is None else ip(nbr), float(margin), float(tol_rbound), int(maxcon), _dp(out))
    return out[:n].copy()


def flops_reset():
    lib().kpo_flops_reset()


def flops():
    """floating-point operations the oracle executed since flops_reset() (counted at its loop bodies, see kp_oracle.c)."""
    return float(lib().kpo_flops_get())
