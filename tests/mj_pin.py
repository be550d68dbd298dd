"""Live pin of the physics against MuJoCo itself -- dormant until a MuJoCo binary is importable (TEST INFRASTRUCTURE; SURVEY.md 8(c) "optional live pin").

The reference's arithmetic at `uhc/envs/humanoid_im.py:527` (`self.sim.step()`), `:423-426` (`mj_fullM`, `qfrc_bias`) and
`uhc/khrylib/rl/envs/common/mujoco_env.py:23-24, 99-103` is the MuJoCo 2.1.0 binary (mujoco-py < 2.2), which exists neither in the build
container nor on the GPU box.  The day `import mujoco` (DeepMind bindings, any version) or `import mujoco_py` succeeds, this module

  1. writes the scene as MJCF **from the compiled blob** (`mjcf_from_kpm`: local coordinates -- `coordinate="global"` was removed from newer
     schemas --, the 24 hulls as inline `<mesh vertex=...>` assets, hinges z / y / x, the XML's defaults, floor, motors; optionally the free
     objects), or takes the reference's own XML when KP_REFERENCE_ROOT points at a checkout and the binding still reads it;
  2. compares what MuJoCo's compiler made of it with the blob (`compare_model`: body_mass / body_ipos / body_inertia / invweight0 / meaninertia);
  3. steps BASELINE `configs[1]` (free fall, contacts off, 1500 substeps) and `configs[2]` (150 control steps from `standing_neutral` with the
     reference's stable-PD + RFC loop of `do_simulation`, humanoid_im.py:506-533, restated around the backend's `qM / qfrc_bias`) on MuJoCo,
     on the fp64 oracle and -- when a HIP simulator is passed in -- on the product, and reports per step `max |dqpos|` and the contact-set
     differences (`run_pin`).

Everything above the thin binding adapters (`MujocoBackend`, `MujocoPyBackend`) runs against the `Backend` interface; `OracleBackend` plays
MuJoCo's part in the CPU tests so that the harness itself is exercised end to end today (tests/test_physics_oracle.py).
"""
from __future__ import annotations

import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = ["Pelvis", "L_Hip", "L_Knee", "L_Ankle", "L_Toe", "R_Hip", "R_Knee", "R_Ankle", "R_Toe", "Torso", "Spine", "Chest",
         "Neck", "Head", "L_Thorax", "L_Shoulder", "L_Elbow", "L_Wrist", "L_Hand", "R_Thorax", "R_Shoulder", "R_Elbow", "R_Wrist", "R_Hand"]
OBJ_NAMES = ["chair", "box", "table", "Can", "step"]


def find_mujoco():
    """('mujoco', module) for DeepMind's bindings, ('mujoco_py', module) for the reference's, None when neither imports."""
    try:
        import mujoco
        return "mujoco", mujoco
    except Exception:
        pass
    try:
        import mujoco_py
        return "mujoco_py", mujoco_py
    except Exception:
        return None


# ---------------------------------------------------------------------------------------------- the scene as MJCF, from the blob
def _fmt(a):
    return " ".join(repr(float(x)) for x in np.asarray(a, float).reshape(-1))


def _mat2quat(R):
    """rotation matrix -> (w, x, y, z)"""
    R = np.asarray(R, float).reshape(3, 3)
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        q = [0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s]
    else:
        i = int(np.argmax(np.diag(R))); j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0) * 2
        q = [0.0] * 4
        q[0] = (R[k, j] - R[j, k]) / s; q[1 + i] = 0.25 * s; q[1 + j] = (R[j, i] + R[i, j]) / s; q[1 + k] = (R[k, i] + R[i, k]) / s
    return np.array(q)


def mjcf_from_kpm(kpm: dict, objects: bool = False, contact: bool = True, gravity=(0.0, 0.0, -9.81), obj_qpos=None) -> str:
    """The reference scene (`assets/mujoco_models/humanoid_smpl_neutral_mesh_all[_step].xml`) in LOCAL coordinates, rebuilt from the compiled
    blob: the defaults of the XML (`:11-14`: joint damping 0 / armature 0.01 / stiffness 0 / limited, geom conaffinity 7 condim 1 contype 7
    margin 0.001), the floor (`:47`: condim 3, friction 1 .1 .1), one mesh geom `contype 0 conaffinity 1` per body, free root with armature 0,
    three hinges per body (z, y, x; ranges from `jnt_range`), gear-1 motors in dof order.  Body frames are world-aligned at qpos0, so a body's
    local position is `body_pos` as the blob holds it and its hull's vertices are body-frame coordinates."""
    nb = int(kpm["dims"][0])
    parent, pos = kpm["body_parent"], kpm["body_pos"].reshape(nb, 3)
    vadr, verts = kpm["vert_adr"], kpm["verts"].reshape(-1, 3)
    rng, lim = kpm["jnt_range"].reshape(-1, 2), kpm["jnt_limited"]
    opt = kpm["opt"]
    out = ['<mujoco model="humanoid">', '  <compiler angle="radian" inertiafromgeom="true"/>',
           f'  <option timestep="{float(opt[0])!r}" gravity="{_fmt(gravity)}">' + ('' if contact else '<flag contact="disable"/>') + '</option>',
           '  <default>', '    <joint damping="0.0" armature="0.01" stiffness="0.0" limited="true"/>',
           '    <geom conaffinity="7" condim="1" contype="7" margin="0.001"/>', '  </default>', '  <asset>']
    for b in range(nb):
        out.append(f'    <mesh name="{NAMES[b]}" vertex="{_fmt(verts[vadr[b]:vadr[b + 1]])}"/>')
    out += ['  </asset>', '  <worldbody>',
            '    <geom condim="3" friction="1. .1 .1" name="floor" pos="0 0 0" size="100 100 .2" type="plane"/>']
    children = {b: [c for c in range(nb) if parent[c] == b] for b in range(-1, nb)}
    gpos0 = kpm["body_gpos0"].reshape(nb, 3)

    def emit(b, ind):
        p = gpos0[b] if parent[b] < 0 else pos[b]
        out.append(f'{ind}<body name="{NAMES[b]}" pos="{_fmt(p)}">')
        if parent[b] < 0:
            out.append(f'{ind}  <joint name="{NAMES[b]}" limited="false" type="free" armature="0" damping="0" stiffness="0"/>')
        else:
            for k, (ax, axis) in enumerate((("z", "0 0 1"), ("y", "0 1 0"), ("x", "1 0 0"))):
                j = 3 * (b - 1) + k
                out.append(f'{ind}  <joint name="{NAMES[b]}_{ax}" type="hinge" pos="0 0 0" axis="{axis}" range="{_fmt(rng[j])}" limited="{"true" if lim[j] else "false"}"/>')
        out.append(f'{ind}  <geom type="mesh" mesh="{NAMES[b]}" contype="0" conaffinity="1"/>')
        for c in children[b]:
            emit(c, ind + "  ")
        out.append(f'{ind}</body>')
    for r in children[-1]:
        emit(r, "    ")
    if objects:
        og, oadr = kpm["obj_geoms"].reshape(-1, 18), kpm["obj_geom_adr"]
        nobj = int(kpm["dims"][6])
        oq = np.tile([0.0, 0, 0, 1, 0, 0, 0], (nobj, 1)) if obj_qpos is None else np.asarray(obj_qpos, float).reshape(nobj, 7)
        for o in range(nobj):
            out.append(f'    <body name="{OBJ_NAMES[o] if o < len(OBJ_NAMES) else "obj%d" % o}" pos="{_fmt(oq[o, :3])}" quat="{_fmt(oq[o, 3:])}">')
            out.append(f'      <joint name="{OBJ_NAMES[o] if o < len(OBJ_NAMES) else "obj%d" % o}" type="free" limited="false"/>')
            for gi in range(int(oadr[o]), int(oadr[o + 1])):
                rec = og[gi]
                typ = "box" if int(rec[1]) == 0 else "cylinder"
                size = rec[2:5] if typ == "box" else rec[2:4]
                out.append(f'      <geom contype="1" conaffinity="1" type="{typ}" size="{_fmt(size)}" pos="{_fmt(rec[5:8])}" quat="{_fmt(_mat2quat(rec[8:17]))}" condim="3" mass="{float(rec[17])!r}"/>')
            out.append('    </body>')
    out += ['  </worldbody>', '  <actuator>']
    for b in range(1, nb):
        for ax in "zyx":
            out.append(f'    <motor name="{NAMES[b]}_{ax}" joint="{NAMES[b]}_{ax}" gear="1"/>')
    out += ['  </actuator>', '</mujoco>']
    return "\n".join(out)


# ---------------------------------------------------------------------------------------------- backends
class Backend:
    """What the pin needs from a simulator of ONE env (humanoid, nq 76 / nv 75 / nu 69)."""
    name = "?"

    def set_state(self, qpos, qvel): raise NotImplementedError          # = sim.set_state + sim.forward (mujoco_env.py:99-103)
    def set_ctrl(self, ctrl, applied6): raise NotImplementedError       # data.ctrl[:], data.qfrc_applied[:6]
    def step(self): raise NotImplementedError                           # mj_step
    def qpos(self): raise NotImplementedError
    def qvel(self): raise NotImplementedError
    def fullM(self): raise NotImplementedError                          # mj_fullM(model, M, data.qM)[:75, :75] -- of the last mj_step / mj_forward
    def qfrc_bias(self): raise NotImplementedError
    def contacts(self): raise NotImplementedError                       # sorted list of (body, dist) of the last collision pass (floor contacts)
    def model_arrays(self): return {}                                   # body_mass, body_ipos, body_inertia6, body_invweight0, dof_invweight0, meaninertia


class OracleBackend(Backend):
    """this repo's fp64 restatement (oracle/kp_oracle.c) behind the same interface: the comparison partner of a real backend, and MuJoCo's stand-in
    for the harness's own CPU test"""
    name = "oracle"

    def __init__(self, kpm_path=None, contact=True, gravity=None):
        from oracle.kpo import DEFAULT_KPM, OracleSim
        self.o = OracleSim(kpm_path or DEFAULT_KPM, contact=contact, gravity=gravity)

    def set_state(self, qpos, qvel): self.o.reset(qpos, qvel)
    def set_ctrl(self, ctrl, applied6): self.o.set_ctrl(ctrl, applied6)
    def step(self): self.o.step()
    def qpos(self): return self.o.get("qpos")
    def qvel(self): return self.o.get("qvel")
    def fullM(self): return self.o.fullM()
    def qfrc_bias(self): return self.o.get("qfrc_bias")

    def contacts(self):
        body, _, dist = self.o.contacts()
        return sorted((int(b), float(d)) for b, d in zip(body, dist))


class MujocoBackend(Backend):
    """DeepMind's `mujoco` bindings (any version that reads inline mesh vertices)."""
    name = "mujoco"

    def __init__(self, mj, xml: str):
        self.mj = mj
        self.m = mj.MjModel.from_xml_string(xml)
        self.d = mj.MjData(self.m)
        self.nv = 75
        self.floor = mj.mj_name2id(self.m, mj.mjtObj.mjOBJ_GEOM, "floor")

    def set_state(self, qpos, qvel):
        self.mj.mj_resetData(self.m, self.d)
        self.d.qpos[:76] = qpos; self.d.qvel[:75] = qvel
        self.mj.mj_forward(self.m, self.d)

    def set_ctrl(self, ctrl, applied6):
        self.d.ctrl[:] = ctrl
        self.d.qfrc_applied[:6] = 0.0 if applied6 is None else applied6

    def step(self): self.mj.mj_step(self.m, self.d)
    def qpos(self): return np.array(self.d.qpos[:76])
    def qvel(self): return np.array(self.d.qvel[:75])

    def fullM(self):
        M = np.zeros((self.m.nv, self.m.nv))
        self.mj.mj_fullM(self.m, M, self.d.qM)
        return M[:75, :75]

    def qfrc_bias(self): return np.array(self.d.qfrc_bias[:75])

    def contacts(self):
        out = []
        for i in range(self.d.ncon):
            c = self.d.contact[i]
            g = c.geom2 if c.geom1 == self.floor else c.geom1
            out.append((int(self.m.geom_bodyid[g]) - 1, float(c.dist)))
        return sorted(out)

    def model_arrays(self):
        m = self.m
        nb = 24
        I6 = np.zeros((nb, 6))
        for b in range(nb):                      # principal inertia + body_iquat -> tensor about the COM in body axes
            R = np.zeros(9); self.mj.mju_quat2Mat(R, m.body_iquat[b + 1]); R = R.reshape(3, 3)
            T = R @ np.diag(m.body_inertia[b + 1]) @ R.T
            I6[b] = [T[0, 0], T[1, 1], T[2, 2], T[0, 1], T[0, 2], T[1, 2]]
        return dict(body_mass=np.array(m.body_mass[1:nb + 1]), body_ipos=np.array(m.body_ipos[1:nb + 1]).reshape(-1), body_inertia=I6.reshape(-1),
                    body_invweight0=np.array(m.body_invweight0[1:nb + 1]).reshape(-1), dof_invweight0=np.array(m.dof_invweight0[:75]),
                    meaninertia=float(m.stat.meaninertia), body_pos=np.array(m.body_pos[1:nb + 1]).reshape(-1))


class MujocoPyBackend(MujocoBackend):
    """mujoco-py 2.1 (the reference's binding).  MuJoCo 2.1.0 does not read inline mesh vertices: pass the reference's own XML path
    (KP_REFERENCE_ROOT/assets/mujoco_models/...), which this binding still reads with `coordinate="global"`."""
    name = "mujoco_py"

    def __init__(self, mjpy, xml_path: str):
        self.mjpy = mjpy
        self.m = mjpy.load_model_from_path(xml_path)
        self.sim = mjpy.MjSim(self.m)
        self.d = self.sim.data
        self.floor = self.m.geom_name2id("floor")

    def set_state(self, qpos, qvel):
        self.sim.reset()
        st = self.sim.get_state()
        q = st.qpos.copy(); v = st.qvel.copy()
        q[:76] = qpos; v[:75] = qvel
        self.sim.set_state(self.mjpy.MjSimState(st.time, q, v, st.act, st.udd_state))
        self.sim.forward()

    def step(self): self.sim.step()

    def fullM(self):
        nv = self.m.nv
        M = np.zeros(nv * nv)
        self.mjpy.functions.mj_fullM(self.m, M, self.d.qM)
        return M.reshape(nv, nv)[:75, :75]

    def contacts(self):
        out = []
        for i in range(self.d.ncon):
            c = self.d.contact[i]
            g = c.geom2 if c.geom1 == self.floor else c.geom1
            out.append((int(self.m.geom_bodyid[g]) - 1, float(c.dist)))
        return sorted(out)

    def model_arrays(self):
        m = self.m
        return dict(body_mass=np.array(m.body_mass[1:25]), body_ipos=np.array(m.body_ipos[1:25]).reshape(-1), body_pos=np.array(m.body_pos[1:25]).reshape(-1),
                    body_invweight0=np.array(m.body_invweight0[1:25]).reshape(-1), dof_invweight0=np.array(m.dof_invweight0[:75]), meaninertia=float(m.stat.meaninertia))


def open_backend(kpm: dict, contact=True, gravity=(0.0, 0.0, -9.81)):
    """the real backend when one is importable, else None"""
    found = find_mujoco()
    if found is None:
        return None
    kind, mod = found
    ref = os.environ.get("KP_REFERENCE_ROOT", "")
    if kind == "mujoco":
        return MujocoBackend(mod, mjcf_from_kpm(kpm, contact=contact, gravity=gravity))
    xml = os.path.join(ref, "assets", "mujoco_models", "humanoid_smpl_neutral_mesh.xml")
    if not os.path.exists(xml):
        raise RuntimeError("mujoco_py (MuJoCo 2.1.0) needs the reference's XML + STL files: set KP_REFERENCE_ROOT to a KinPoly checkout")
    return MujocoPyBackend(mod, xml)


# ---------------------------------------------------------------------------------------------- what is compared
def compare_model(arrays: dict, kpm: dict) -> dict:
    """max |difference| between what the backend's compiler produced and the blob, per array (relative for masses / inertias)"""
    out = {}
    nb = int(kpm["dims"][0])
    for k, rel in (("body_mass", True), ("body_ipos", False), ("body_inertia", True), ("body_invweight0", True), ("dof_invweight0", True), ("body_pos", False)):
        if k not in arrays:
            continue
        want = np.asarray(kpm[k], float).reshape(-1)
        got = np.asarray(arrays[k], float).reshape(-1)
        if k == "body_pos":                      # the root's entry is its world position at qpos0 in MuJoCo
            want = want.copy(); want[:3] = kpm["body_gpos0"].reshape(nb, 3)[0]
        d = np.abs(got - want)
        out[k] = float((d / np.maximum(np.abs(want), 1e-12)).max()) if rel else float(d.max())
    if "meaninertia" in arrays:
        out["meaninertia_humanoid_only"] = float(arrays["meaninertia"])        # the blob's opt[16] is the 105-dof scene's; reported, not differenced
    return out


MODEL_ARRAY_TOL = 1e-6          # what the live pin asserts on every entry of compare_model (tests/test_gpu_round5.py)


def model_table(arrays: dict, kpm: dict, tol: float = MODEL_ARRAY_TOL) -> list:
    """The model-array comparison the pin asserts on, entry by entry instead of as one maximum per array: for every array the number of entries, how many are
    beyond `tol`, and the worst entry with its index, both values and the (relative / absolute) difference -- so that the first run against a real binding
    says WHERE the compiled blob and MuJoCo's compiler part (a body, a dof), not just that they do.  Rows are dicts; `format_model_table` prints them."""
    rows = []
    nb = int(kpm["dims"][0])
    per = {"body_mass": 1, "body_ipos": 3, "body_inertia": 6, "body_invweight0": 2, "dof_invweight0": 1, "body_pos": 3}
    for k, rel in (("body_mass", True), ("body_ipos", False), ("body_inertia", True), ("body_invweight0", True), ("dof_invweight0", True), ("body_pos", False)):
        if k not in arrays:
            rows.append({"array": k, "status": "not provided by this backend"})
            continue
        want = np.asarray(kpm[k], float).reshape(-1)
        got = np.asarray(arrays[k], float).reshape(-1)
        if k == "body_pos":
            want = want.copy(); want[:3] = kpm["body_gpos0"].reshape(nb, 3)[0]
        if got.shape != want.shape:
            rows.append({"array": k, "status": f"shape {got.shape} vs {want.shape}"})
            continue
        d = np.abs(got - want)
        e = d / np.maximum(np.abs(want), 1e-12) if rel else d
        i = int(e.argmax())
        rows.append({"array": k, "entries": int(e.size), "measure": "relative" if rel else "absolute", "tolerance": tol, "beyond_tolerance": int((e > tol).sum()),
                     "worst": float(e[i]), "worst_index": i, "worst_item": f"{'dof' if k.startswith('dof') else 'body'} {i // per[k]} component {i % per[k]}",
                     "backend_value": float(got[i]), "blob_value": float(want[i]), "status": "ok" if e[i] <= tol else "DIFFERS"})
    if "meaninertia" in arrays:
        rows.append({"array": "meaninertia (humanoid only)", "backend_value": float(arrays["meaninertia"]), "status": "reported (the blob's is the 105-dof scene's)"})
    return rows


def format_model_table(rows: list) -> str:
    out = ["| array | entries | measure | tolerance | beyond | worst | at | backend | blob | status |", "|---|---|---|---|---|---|---|---|---|---|"]
    for r in rows:
        out.append("| {} | {} | {} | {} | {} | {} | {} | {} | {} | {} |".format(
            r["array"], r.get("entries", ""), r.get("measure", ""), r.get("tolerance", ""), r.get("beyond_tolerance", ""),
            "%.3g" % r["worst"] if "worst" in r else "", r.get("worst_item", ""), "%.9g" % r["backend_value"] if "backend_value" in r else "",
            "%.9g" % r["blob_value"] if "blob_value" in r else "", r["status"]))
    return "\n".join(out)


def control_substep(backend: Backend, action, target_qpos, kpm: dict):
    """one pass of do_simulation's loop body (humanoid_im.py:509-529) on `backend`: stable-PD torque from the backend's CURRENT qpos / qvel and its
    qM / qfrc_bias as they stand (= of the previous mj_step's state, the staleness the reference lives with), clip, RFC, mj_step"""
    from oracle.np_spd import compute_torque_np, rfc_implicit_np
    qpos, qvel = backend.qpos(), backend.qvel()
    torque = compute_torque_np(qpos, qvel, backend.fullM(), backend.qfrc_bias(), action, target_qpos, kpm)
    torque = np.clip(torque, -kpm["torque_lim"], kpm["torque_lim"])
    vf = rfc_implicit_np(qpos, np.array(action[69:75], float), kpm)
    backend.set_ctrl(torque, vf)
    backend.step()


def free_fall_state(std_qpos, seed=1234):
    """BASELINE configs[1]: standing pose 10 m up, random heading, joint angles + N(0, 0.2^2), velocities N(0, 1) / N(0, 0.5^2)"""
    rng = np.random.default_rng(seed)
    q = np.array(std_qpos, float).copy()
    q[2] += 10.0
    h = rng.uniform(-np.pi, np.pi)
    w, x, y, z = q[3:7]
    c, s = np.cos(h / 2), np.sin(h / 2)                 # heading (x) root
    q[3:7] = [c * w - s * z, c * x - s * y, c * y + s * x, c * z + s * w]
    q[7:] = np.clip(q[7:] + rng.normal(size=69) * 0.2, -np.pi, np.pi)
    v = np.concatenate([rng.normal(size=3), rng.normal(size=72) * 0.5])
    return q, v


def run_pin(kind: str, ref: Backend, others: dict, kpm: dict, std_qpos, std_qvel, n_steps=None, seed=1234, hip=None):
    """kind 'free_fall': n_steps substeps (default 1500) with zero control, contacts off, every backend from the same state.
    kind 'contact': n_steps control steps (default 150) of 15 substeps from standing_neutral, tracking the standing pose with a small seeded
    action per control step.  `ref` is the backend everything is compared with (MuJoCo; the oracle in the self-test), `others` {name: Backend},
    `hip` an optional callable (qpos0, qvel0, actions [n, 75], target) -> qpos trajectory [n, 76] of the product after every control step.
    Returns {name: {max_dqpos_per_step: [...], first_step_above_1e-3, contact_set_diffs}}."""
    sims = {"ref": ref, **others}
    if kind == "free_fall":
        n = int(n_steps or 1500)
        q0, v0 = free_fall_state(std_qpos, seed)
        for s in sims.values():
            s.set_state(q0, v0)
        rep = {k: dict(max_dqpos_per_step=[], contact_set_diffs=0) for k in others}
        for _ in range(n):
            for s in sims.values():
                s.set_ctrl(np.zeros(69), np.zeros(6)); s.step()
            rq = ref.qpos()
            for k, s in others.items():
                rep[k]["max_dqpos_per_step"].append(float(np.abs(s.qpos() - rq).max()))
    else:
        n = int(n_steps or 150)
        rng = np.random.default_rng(seed)
        actions = rng.normal(size=(n, 75)) * 0.05
        target = np.array(std_qpos, float)
        for s in sims.values():
            s.set_state(std_qpos, std_qvel)
        rep = {k: dict(max_dqpos_per_step=[], contact_set_diffs=0) for k in others}
        ref_traj = []
        for t in range(n):
            for i in range(15):
                for s in sims.values():
                    control_substep(s, actions[t], target, kpm)
                rc = [b for b, _ in ref.contacts()]
                for k, s in others.items():
                    rep[k]["contact_set_diffs"] += int([b for b, _ in s.contacts()] != rc)
            rq = ref.qpos()
            ref_traj.append(rq)
            for k, s in others.items():
                rep[k]["max_dqpos_per_step"].append(float(np.abs(s.qpos() - rq).max()))
        if hip is not None:
            traj = np.asarray(hip(np.array(std_qpos, float), np.array(std_qvel, float), actions, target), float)
            rep["hip"] = dict(max_dqpos_per_step=[float(np.abs(traj[t] - ref_traj[t]).max()) for t in range(n)], contact_set_diffs=None)
    for k, r in rep.items():
        e = np.array(r["max_dqpos_per_step"])
        above = np.nonzero(e > 1e-3)[0]
        r["first_step_above_1e-3"] = int(above[0]) if len(above) else None
        r["max_dqpos"] = float(e.max()) if len(e) else 0.0
    return rep


def pin_report(kpm_path=None, n_free_fall=None, n_contact=None, hip=None) -> dict | None:
    """Everything, or None when no MuJoCo binding imports.  bench.py prints this as `"mujoco_pin"`."""
    found = find_mujoco()
    if found is None:
        return None
    from kinpoly_amd.model_compiler import DEFAULT_KPM, read_kpm
    kpm = read_kpm(kpm_path or DEFAULT_KPM)
    std = np.load(os.path.join(ROOT, "tests", "golden", "standing_neutral.npz"))
    out = {"binding": found[0], "version": getattr(found[1], "__version__", "?")}
    b = open_backend(kpm)
    out["model"] = compare_model(b.model_arrays(), kpm)
    out["model_table"] = model_table(b.model_arrays(), kpm)
    ff = open_backend(kpm, contact=False)
    out["free_fall"] = {k: {kk: vv for kk, vv in v.items() if kk != "max_dqpos_per_step"} for k, v in
                        run_pin("free_fall", ff, {"oracle": OracleBackend(kpm_path, contact=False)}, kpm, std["qpos"], std["qvel"], n_free_fall).items()}
    out["contact"] = {k: {kk: vv for kk, vv in v.items() if kk != "max_dqpos_per_step"} for k, v in
                      run_pin("contact", b, {"oracle": OracleBackend(kpm_path)}, kpm, std["qpos"], std["qvel"], n_contact, hip=hip).items()}
    return out
