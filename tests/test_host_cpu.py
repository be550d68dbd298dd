"""CPU tests of the host logic: model compiler, KPM blob, C-ABI surface (no compute calls without a GPU)."""
import ctypes
import os
import re
import sys

import numpy as np
import pytest

from kinpoly_amd import build as kpbuild
from kinpoly_amd.model_compiler import DEFAULT_KPM, read_kpm, write_kpm

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_kpm_blob_roundtrip(tmp_path):
    m = read_kpm(DEFAULT_KPM)
    assert list(m["dims"][:6]) == [24, 75, 76, 69, 1221, 1199]
    p = tmp_path / "copy.kpm"
    write_kpm({k: v for k, v in m.items() if not k.startswith("_")}, str(p))
    m2 = read_kpm(str(p))
    for k in m:
        if not k.startswith("_"):
            np.testing.assert_array_equal(m[k], m2[k])


def test_model_tables_consistent():
    m = read_kpm(DEFAULT_KPM)
    parent = m["body_parent"]
    assert list(parent) == [-1, 0, 1, 2, 3, 0, 5, 6, 7, 0, 9, 10, 11, 12, 11, 14, 15, 16, 17, 11, 19, 20, 21, 22]
    assert abs(m["body_mass"].sum() - 80.29) < 0.01
    assert m["dof_madr"][-1] == 1221 and m["dof_depth"].max() == 29
    # subtree sizes from DFS order
    st = np.ones(24, int)
    for b in range(23, 0, -1):
        st[parent[b]] += st[b]
    np.testing.assert_array_equal(st, m["body_subtree"])
    # PD gains table (uhc.yml:88-156): hips 500/50/200, knees 500/50/150, toes 200/20/100
    assert (m["kp"][:3] == 500).all() and (m["kd"][:3] == 50).all() and (m["torque_lim"][:3] == 200).all()
    assert (m["torque_lim"][3:6] == 150).all() and (m["kp"][9:12] == 200).all()
    M0 = m["M0"].reshape(75, 75)
    assert np.allclose(M0, M0.T) and np.linalg.eigvalsh(M0).min() > 0


def test_model_compiler_reproduces_blob():
    """Only where the reference assets exist (build container): recompiling gives the committed blob."""
    xml = "/root/reference/assets/mujoco_models/humanoid_smpl_neutral_mesh_all.xml"
    yml = "/root/reference/config/uhc/uhc.yml"
    if not os.path.exists(xml):
        pytest.skip("reference assets not present (GPU box)")
    from kinpoly_amd.model_compiler import compile_model
    m = compile_model(xml, yml)
    ref = read_kpm(DEFAULT_KPM)
    for k in ("body_mass", "body_ipos", "body_inertia", "verts", "kp", "opt", "dof_invweight0", "body_invweight0"):
        np.testing.assert_allclose(np.asarray(m[k], float).ravel(), ref[k], rtol=1e-12, atol=1e-14, err_msg=k)


def test_abi_library_exports_every_declared_symbol():
    lib = kpbuild.LIB
    if not os.path.exists(lib):
        kpbuild.build_native()
    L = ctypes.CDLL(lib)
    hdr = open(os.path.join(ROOT, "include", "kinpoly_sim.h")).read()
    declared = set(re.findall(r"\b(kp_[a-z_0-9]+)\s*\(", hdr))
    assert len(declared) >= 20
    for sym in sorted(declared):
        assert hasattr(L, sym), f"{sym} declared in include/kinpoly_sim.h but not exported"
    from kinpoly_amd.sim import ABI_SYMBOLS
    assert set(ABI_SYMBOLS) == declared


def test_no_gpu_fails_loudly():
    """The product path has no CPU fallback: creating a simulator without a HIP device raises."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from kinpoly_amd.sim import KinPolyNativeError, KpModel, KpSim
    m = KpModel()
    assert m.get_option("lds_bytes_per_env") > 1000
    with pytest.raises(KinPolyNativeError):
        KpSim(m, 4)


def test_product_never_imports_oracle():
    """oracle/ is test infrastructure: nothing under kinpoly_amd/ may import, include, dlopen or link it."""
    pat = re.compile(r"^\s*(from\s+oracle|import\s+oracle|from\s+\.\.?oracle)|#include\s+[\"<][^\">]*oracle|libkp_oracle|CDLL\([^)]*oracle", re.M)
    for dirpath, _, files in os.walk(os.path.join(ROOT, "kinpoly_amd")):
        for f in files:
            if f.endswith((".py", ".hpp", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not pat.search(src), f"{f} references the oracle"


def test_ppo_surrogate_and_log_prob_match_reference(golden):
    """KinPolicy.log_prob (DiagGaussian) + rollout.ppo_surrogate against AgentPPO.ppo_loss run in the reference (fixture)."""
    import torch
    from kinpoly_amd.nets import KinPolicy
    from kinpoly_amd.rollout import ppo_surrogate
    g = golden("ppo_loss")
    pol = KinPolicy(log_std=float(g["log_std"])).double()
    pol.action_log_std.data.fill_(float(g["log_std"]))      # the fp32-constructed parameter carries -3.2 rounded to float
    t = lambda k: torch.tensor(g[k], dtype=torch.float64)  # noqa: E731
    fixed = pol.log_prob(t("mean_old"), t("actions"))
    new = pol.log_prob(t("mean_new"), t("actions"))
    np.testing.assert_allclose(fixed.detach().numpy(), g["fixed_log_probs"], rtol=1e-12, atol=1e-9)
    np.testing.assert_allclose(new.detach().numpy(), g["new_log_probs"], rtol=1e-12, atol=1e-9)
    ind = torch.tensor(g["exps"]).nonzero(as_tuple=False).squeeze(1)
    loss = ppo_surrogate(new, fixed, t("adv"), float(g["clip_epsilon"]), ind)
    assert abs(float(loss) - float(g["loss"])) < 1e-10


def test_job_schedule_host_logic():
    """kp_job_schedule (host arithmetic of the queue-scheduled control step): sizes sum to the control step, at most 16 jobs, the
    last job is substeps_per_job long unless the whole step is shorter, tapered sizes never grow towards the end."""
    from kinpoly_amd import sim as kpsim
    assert kpsim.job_schedule(15) == [6, 5, 4]            # the default schedule of a control step
    assert kpsim.job_schedule(15, 3, 2) == [7, 5, 3]
    assert kpsim.job_schedule(15, 3, 0) == [3, 3, 3, 3, 3]
    assert kpsim.job_schedule(15, 16, 1) == [15]
    for nsub in (1, 2, 7, 15, 16, 30, 100, 255):
        for spj in (1, 2, 3, 5, 8):
            for taper in (0, 1, 2):
                sz = kpsim.job_schedule(nsub, spj, taper)
                assert sum(sz) == nsub and 1 <= len(sz) <= 16 and min(sz) >= 1
                assert sz[-1] == spj or len(sz) == 1 or len(sz) == 16
                if taper and len(sz) > 2:
                    assert all(a >= b for a, b in zip(sz[1:], sz[2:]))
    with pytest.raises(kpsim.KinPolyNativeError):
        kpsim.job_schedule(0, 3, 1)


# ------------------------------------------------------------------ the model compiler behind the C ABI (kp_model_compile; needs no GPU)
def _write_stl(path, verts, faces):
    import struct
    with open(path, "wb") as f:
        f.write(b"\0" * 80 + struct.pack("<I", len(faces)))
        for a, b, c in faces:
            f.write(struct.pack("<12fH", 0.0, 0.0, 0.0, *verts[a], *verts[b], *verts[c], 0))


def _toy_scene(tmp):
    """a three-body tree (free root + two 3-hinge children) on irregular octahedron hulls, a plane and one free box + cylinder object: the
    constructs of the reference's XML (assets/mujoco_models/humanoid_smpl_neutral_mesh_all_step.xml) in miniature, written by the test"""
    rng = np.random.default_rng(7)
    faces = [(0, 2, 4), (2, 1, 4), (1, 3, 4), (3, 0, 4), (2, 0, 5), (1, 2, 5), (3, 1, 5), (0, 3, 5)]
    os.makedirs(os.path.join(tmp, "geom"))
    pos = {"A": (0.1, -0.2, 0.9), "B": (0.2, -0.1, 0.5), "C": (0.25, -0.15, 0.1)}
    for nm, p in pos.items():
        base = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]], float) * rng.uniform(0.05, 0.12, size=(6, 1))
        v = (base + rng.normal(size=(6, 3)) * 0.005 + np.array(p)).astype(np.float32)
        _write_stl(os.path.join(tmp, "geom", nm + ".stl"), v.tolist(), faces)
    hinges = lambda n, p: "".join(f'<joint name="{n}_{a}" type="hinge" pos="{p[0]} {p[1]} {p[2]}" axis="{ax}" range="-180 180"/>'     # noqa: E731
                                  for a, ax in (("z", "0 0 1"), ("y", "0 1 0"), ("x", "1 0 0")))
    xml = f"""<mujoco model="toy">
  <compiler angle="degree" inertiafromgeom="true" coordinate="global"/>
  <!-- a comment with <tags/> inside -->
  <default><joint damping="0.0" armature="0.01" limited="true"/><geom condim="1" margin="0.001" conaffinity="7"/></default>
  <option timestep="0.002"/>
  <asset><mesh file="./geom/A.stl"/><mesh file="./geom/B.stl"/><mesh file="./geom/C.stl"/></asset>
  <worldbody>
    <geom condim="3" friction="1. .1 .1" name="floor" pos="0 0 0" size="100 100 .2" type="plane"/>
    <body name="A" pos="0.1 -0.2 0.9"><joint name="A" pos="0.1 -0.2 0.9" limited="false" type="free" armature="0"/><geom type="mesh" mesh="A"/>
      <body name="B" pos="0.2 -0.1 0.5">{hinges("B", pos["B"])}<geom type="mesh" mesh="B"/>
        <body name="C" pos="0.25 -0.15 0.1">{hinges("C", pos["C"])}<geom type="mesh" mesh="C"/></body>
      </body>
    </body>
    <body name="thing" pos="0 0 0"><joint limited="false" name="thing" type="free"/>
      <geom type="box" size="0.2 0.1 0.05" pos="0 0.1 -0.2" euler="14 0 30" condim="3" mass="3.5"/>
      <geom type="cylinder" size="0.03 0.2" pos="0.1 0 0.1" condim="3" mass="1.25"/></body>
  </worldbody>
</mujoco>
"""
    open(os.path.join(tmp, "toy.xml"), "w").write(xml)
    yml = ("residual_force_scale: 150.0\njoint_params:\n  # name, kp, kd, a_ref, a_scale, torque_limit\n" +
           "".join(f'  - ["{b}_{a}" , {100.0 + 10 * i}, {10.0 + i}, 0.0, 1.0, {50.0 + i}]\n' for i, (b, a) in enumerate((b, a) for b in "BC" for a in "zyx")) +
           'body_params:\n- ["B" , 1.0]\n- ["C" , 0.0]\ndata_specs:\n  dataset_name: toy\n  base_rot: [0.5, 0.5, 0.5, 0.5]\n')
    open(os.path.join(tmp, "toy.yml"), "w").write(yml)
    return os.path.join(tmp, "toy.xml"), os.path.join(tmp, "toy.yml")


def test_native_model_compiler_writes_the_python_compilers_bytes(tmp_path):
    """kp_model_compile (kinpoly_amd/csrc/kp_compile.hpp: XML subset reader, binary STL, mesh inertia, hull graph, M(qpos0) and its
    inverse, uhc.yml gains, free objects) against kinpoly_amd/model_compiler.py, the compiler the shipped blobs and every fixture came
    from: the two blobs are equal BYTE FOR BYTE on a scene the test writes itself and -- where the reference tree is present -- on both of
    the reference's scenes; the shipped blobs are what either compiler writes."""
    from kinpoly_amd import model_compiler as mc
    from kinpoly_amd import sim
    xml, yml = _toy_scene(str(tmp_path))
    a, b = str(tmp_path / "native.kpm"), str(tmp_path / "python.kpm")
    sim.compile_model_native(xml, yml, a)
    mc.write_kpm(mc.compile_model(xml, yml), b)
    ka, kb = mc.read_kpm(a), mc.read_kpm(b)
    assert list(ka) == list(kb)
    for k in kb:
        if not k.startswith("_"):
            assert ka[k].tobytes() == kb[k].tobytes(), k
    assert open(a, "rb").read() == open(b, "rb").read()
    assert ka["dims"][0] == 3 and ka["dims"][1] == 12 and ka["dims"][6] == 1 and ka["dims"][7] == 2 and ka["opt"][17] == 150.0 and list(ka["uhc_b_diffw"]) == [1.0, 1.0, 0.0]
    # errors come back as return codes with a text, not as crashes
    with pytest.raises(sim.KinPolyNativeError, match="cannot read"):
        sim.compile_model_native(str(tmp_path / "missing.xml"), None, a)
    open(tmp_path / "bad.xml", "w").write('<mujoco><compiler angle="radian"/></mujoco>')
    with pytest.raises(sim.KinPolyNativeError, match="compiler element"):
        sim.compile_model_native(str(tmp_path / "bad.xml"), None, a)
    ref = "/root/reference/assets/mujoco_models/"
    if os.path.exists(ref + "humanoid_smpl_neutral_mesh_all.xml"):
        for name, shipped in (("humanoid_smpl_neutral_mesh_all.xml", mc.DEFAULT_KPM), ("humanoid_smpl_neutral_mesh_all_step.xml", mc.STEP_KPM)):
            sim.compile_model_native(ref + name, "/root/reference/config/uhc/uhc.yml", a)
            assert open(a, "rb").read() == open(shipped, "rb").read(), f"{shipped} is not what kp_model_compile writes from {name}"


@pytest.mark.skipif(not os.path.exists("/root/reference/assets/mujoco_models/humanoid_smpl_neutral_mesh_all.xml"), reason="reference tree not present")
def test_python_model_compiler_and_hull_graph_against_qhull():
    """(build container only: needs the reference's XML + STL) the Python compiler writes the shipped blob byte for byte, and the hull
    graph's EDGE SET -- built by the compilers' own rule -- is qhull's ("Qt", what MuJoCo's mesh compiler runs) on all 24 hulls; only the
    order of a vertex's neighbours differs (ascending vertex number here, qhull's facet order there; MuJoCo's own cannot be known)."""
    from kinpoly_amd import model_compiler as mc
    xml = "/root/reference/assets/mujoco_models/humanoid_smpl_neutral_mesh_all.xml"
    m = mc.compile_model(xml, "/root/reference/config/uhc/uhc.yml")
    import tempfile
    with tempfile.TemporaryDirectory() as t:
        mc.write_kpm(m, os.path.join(t, "m.kpm"))
        assert open(os.path.join(t, "m.kpm"), "rb").read() == open(mc.DEFAULT_KPM, "rb").read()
    adr, va, nb_ = m["vert_adr"], m["vert_nbr_adr"], m["vert_nbr"]
    verts = m["verts"].reshape(-1, 3)
    for b in range(24):
        v = verts[adr[b]:adr[b + 1]]
        mine = {(i, int(j)) for i in range(len(v)) for j in nb_[va[adr[b] + i]:va[adr[b] + i + 1]]}
        qh = mc.hull_graph_qhull(v)
        assert mine == {(i, j) for i in range(len(v)) for j in qh[i]}, f"hull {b}"
        assert len(mine) == 2 * (3 * len(v) - 6)          # a triangulated convex polyhedron: E = 3 V - 6


def test_parity_tool_scene_generators_are_deterministic_and_fp32_valued():
    """tools/_scenes.py feeds the parity sweeps (substep_parity.py, obj_fuzz_trace.py): same seed -> same scenes, every number already an fp32 value
    (both sides of a sweep must get identical inputs), unit root quaternions, objects only where the scene's action class puts them."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import _scenes
    a, b = _scenes.object_scenes(16, 3), _scenes.object_scenes(16, 3)
    for k in ("qpos", "qvel", "action", "blk"):
        assert np.array_equal(a[k], b[k]) and np.array_equal(a[k], a[k].astype(np.float32).astype(np.float64)), k
    assert not np.array_equal(a["qpos"], _scenes.object_scenes(16, 4)["qpos"])
    for e, objs in enumerate(a["objects"]):
        assert sorted(objs) == _scenes.OBJ_OF_ACTION[int(a["kind"][e])]
        for oi, pose in objs.items():
            assert np.array_equal(pose, a["blk"][e, 7 * oi: 7 * oi + 7]) and abs(np.linalg.norm(pose[3:7]) - 1) < 1e-6
        parked = [oi for oi in range(5) if oi not in objs]
        assert all(a["blk"][e, 7 * oi] >= 100 for oi in parked)
    f = _scenes.floor_scenes(20)
    assert f["qpos"].shape == (20, 76) and f["target"].shape == (20, 76) and set(f["kind"]) == {0, 1, 2, 3, 4}
    assert np.abs(np.linalg.norm(f["qpos"][:, 3:7], axis=1) - 1).max() < 1e-6 and f["blk"] is None and all(o == {} for o in f["objects"])


def test_profile_stamps_null_stale_figures(tmp_path, monkeypatch):
    """bench.py reads PMC / parity summaries from profiles/ only when they were taken on THIS device code: kernel_source_sha256 (csrc/* + flags) is
    stable, changes with the flag list, and a committed parity log with another stamp comes back as {"stale": true} instead of as figures."""
    import importlib
    from kinpoly_amd import build as kpbuild
    a = kpbuild.kernel_source_sha256()
    assert a == kpbuild.kernel_source_sha256() and len(a) == 16
    monkeypatch.setenv("KP_HIPCC_FLAGS", "-DKP_SOMETHING=1")
    assert kpbuild.kernel_source_sha256() != a
    monkeypatch.delenv("KP_HIPCC_FLAGS")
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    monkeypatch.setattr(bench, "PROFILE_DIR", str(tmp_path))
    body = ("bench:tracked: 2048 scenes (seed 4) x 45 substeps, every substep from a common fp32-rounded state: one-substep |dqpos| median 4.9e-08 p99 1.1e-07 max 1.0e-03\n"
            "   substeps whose contact sets differ between the two sides at the same state: 22 of 92160; same entities but another vertex of a hull at the same height (to 1e-7): 13; "
            "with the same contact points: max |dqpos| 4.3e-07, above 1e-6: 0\n")
    (tmp_path / "substep_parity_bench.log").write_text("kernel_source_sha256 0000000000000000\n" + body)
    got = bench.parity_summary("tracked")
    assert got["stale"] is True and "substeps" not in got
    (tmp_path / "substep_parity_bench.log").write_text(f"kernel_source_sha256 {a}\n" + body)
    got = bench.parity_summary("tracked")
    assert got["substeps"] == 92160 and got["contact_set_diffs"] == 22 and got["same_entities_other_hull_vertex"] == 13 and got["max_same_set_dqpos"] == 4.3e-07
    (tmp_path / "substep_parity_bench.log").write_text(body)             # no stamp at all (a log of an earlier round): stale
    assert bench.parity_summary("tracked")["stale"] is True
