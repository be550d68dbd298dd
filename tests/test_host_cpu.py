"""CPU tests of the host logic: model compiler, KPM blob, C-ABI surface (no compute calls without a GPU)."""
import ctypes
import os
import re

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
    assert kpsim.job_schedule(15, 3, True) == [7, 5, 3]
    assert kpsim.job_schedule(15, 3, False) == [3, 3, 3, 3, 3]
    assert kpsim.job_schedule(15, 16, True) == [15]
    for nsub in (1, 2, 7, 15, 16, 30, 100, 255):
        for spj in (1, 2, 3, 5, 8):
            for taper in (False, True):
                sz = kpsim.job_schedule(nsub, spj, taper)
                assert sum(sz) == nsub and 1 <= len(sz) <= 16 and min(sz) >= 1
                assert sz[-1] == spj or len(sz) == 1 or len(sz) == 16
                if taper and len(sz) > 2:
                    assert all(a >= b for a, b in zip(sz[1:], sz[2:]))
    with pytest.raises(kpsim.KinPolyNativeError):
        kpsim.job_schedule(0, 3, True)
