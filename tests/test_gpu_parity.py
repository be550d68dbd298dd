"""GPU parity tests (run on a real MI355X with -m gpu): the HIP path, called through the C ABI of
libkinpoly_sim.so, against the fp64 oracle and the committed golden fixtures.

Tolerances (fp32 device vs fp64 oracle; north_star: per-step pose error <= 1e-3 rad):
  * one control step (15 substeps): |dqpos| <= 2e-5, well inside the 1e-3 rad budget;
  * pure arithmetic kernels (FK, obs, step_ar): 2e-5 absolute on O(1) quantities.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from kinpoly_amd.model_compiler import DEFAULT_KPM, read_kpm  # noqa: E402
from oracle import np_oracle as O  # noqa: E402
from oracle.kpo import OracleSim  # noqa: E402

KPM = read_kpm(DEFAULT_KPM)
BODY_POS, BODY_IPOS, PARENT = KPM["body_pos"].reshape(24, 3), KPM["body_ipos"].reshape(24, 3), KPM["body_parent"]
DIFFW = KPM["body_diffw"]
STD = np.load(os.path.join(os.path.dirname(__file__), "golden", "standing_neutral.npz"))


@pytest.fixture(scope="module")
def kp():
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible")
    from kinpoly_amd import sim as kpsim
    return kpsim


def dev(a):
    return torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device="cuda")


def make_states(n, seed, lift=0.0, vel=0.5, noise=0.2):
    rng = np.random.default_rng(seed)
    qpos = np.tile(STD["qpos"], (n, 1))
    qpos[:, 2] += lift
    qpos[:, 7:] += np.clip(rng.normal(size=(n, 69)) * noise, -np.pi, np.pi)
    qvel = rng.normal(size=(n, 75)) * vel
    return qpos, qvel


def run_pair(kp, n, nsub, steps, contact, lift, act_scale, threads=64, seed=1):
    qpos, qvel = make_states(n, seed, lift=lift)
    rng = np.random.default_rng(seed + 1)
    action = rng.normal(size=(n, 75)) * act_scale
    target = np.tile(STD["qpos"], (n, 1)); target[:, 7:] += rng.normal(size=(n, 69)) * 0.1
    model = kp.KpModel(contact=contact, threads_per_env=threads)
    sim = kp.KpSim(model, n)
    sim.set_state(dev(qpos), dev(qvel)); sim.set_target(dev(target))
    a = dev(action)
    for _ in range(steps):
        sim.step_ctrl(a, nsub)
    out = {k: sim.get(k).cpu().numpy().astype(np.float64) for k in ("qpos", "qvel", "xpos", "xquat", "xipos", "qpos_d")}
    out["diag"] = sim.diag()
    ref = {k: [] for k in ("qpos", "qvel", "xpos", "xquat", "xipos")}
    o = OracleSim(contact=bool(contact))
    for e in range(n):
        o.reset(qpos[e], qvel[e])
        for _ in range(steps):
            o.do_simulation(action[e], target[e], nsub)
        for k in ref:
            ref[k].append(o.get(k))
    return out, {k: np.stack(v) for k, v in ref.items()}


def test_abi_loads_and_errors(kp):
    L = kp.load_library()
    assert b"kinpoly_sim" in L.kp_version()
    m = kp.KpModel()
    assert m.get_option("timestep") == pytest.approx(0.00222222222)
    with pytest.raises(kp.KinPolyNativeError):
        m.set_option("no_such_option", 1)
    with pytest.raises(kp.KinPolyNativeError):
        kp.KpModel("/nonexistent.kpm")
    # error behaviour of the C ABI: negative return + kp_last_error text, never an exception / crash across the boundary
    import ctypes as C
    sim = kp.KpSim(m, 4)
    assert L.kp_sim_set_state(sim.h, None, None, None) == -1 and b"null" in L.kp_last_error()
    assert L.kp_sim_get(sim.h, 999, C.c_void_p(sim.get("qpos").data_ptr())) == -1 and b"unknown field" in L.kp_last_error()
    assert L.kp_sim_step_ctrl(None, None, 15, None) != 0
    assert not L.kp_sim_create(m.h, 0, 0, None) and b"bad arguments" in L.kp_last_error()
    assert L.kp_model_set_option(m.h, b"threads_per_env", C.c_double(96.0)) == -1
    m2 = kp.KpModel(kp.STEP_KPM, threads_per_env=128)
    s2 = kp.KpSim(m2, 2)
    blk = np.zeros((2, 35), np.float32); blk[:, 28:35] = [0, 0, 0.37, 1, 0, 0, 0]
    s2.set_objects(dev(blk))
    with pytest.raises(kp.KinPolyNativeError):                      # the object kernel is one wavefront per env
        s2.set_state(dev(np.tile(STD["qpos"], (2, 1))), dev(np.zeros((2, 75))))


@pytest.mark.parametrize("threads", [64, 128, 256])
def test_freefall_matches_oracle(kp, threads):
    """BASELINE.json configs[1]: free fall, no contact (ABA/CRBA correctness)."""
    out, ref = run_pair(kp, 32, 15, 2, contact=0, lift=10.0, act_scale=0.0, threads=threads)
    assert out["diag"][:, 2].max() == 0
    assert np.abs(out["qpos"] - ref["qpos"]).max() < 2e-5      # measured 3.5e-6 after two control steps of tumbling at 10 m
    assert np.abs(out["qvel"] - ref["qvel"]).max() < 4e-5      # measured 4.3e-6
    assert np.abs(out["xpos"] - ref["xpos"]).max() < 2e-5      # stale (x_14) kinematics
    assert np.abs(out["xipos"] - ref["xipos"]).max() < 2e-5
    assert np.abs(np.abs((out["xquat"].reshape(-1, 4) * ref["xquat"].reshape(-1, 4)).sum(1)) - 1).max() < 1e-6


def test_spd_rfc_control_matches_oracle(kp):
    out, ref = run_pair(kp, 32, 15, 3, contact=0, lift=10.0, act_scale=0.5)
    assert np.abs(out["qpos"] - ref["qpos"]).max() < 3e-5      # measured 3.4e-6 after three control steps
    assert np.abs(out["qvel"] - ref["qvel"]).max() < 2e-4      # measured 1.6e-5


@pytest.mark.parametrize("threads", [64, 128, 256])
def test_contact_matches_oracle(kp, threads):
    out, ref = run_pair(kp, 32, 15, 1, contact=1, lift=0.0, act_scale=0.3, threads=threads)
    assert (out["diag"][:, 3] & 255).max() >= 6          # feet are on the floor
    assert np.abs(out["qpos"] - ref["qpos"]).max() < 8e-6      # measured 6e-7 .. 7.5e-7 after one control step on the floor
    assert np.abs(out["qvel"] - ref["qvel"]).max() < 1e-3      # measured 5e-5 .. 1.1e-4 (contact forces of 1e3 N: a rounding of the penetration depth is a velocity)


def test_contact_ten_control_steps(kp):
    """1/3 s of standing-with-contact; trajectories stay within the 1e-3 rad budget of north_star."""
    out, ref = run_pair(kp, 16, 15, 10, contact=1, lift=0.0, act_scale=0.2)
    err = np.abs(out["qpos"] - ref["qpos"]).max(axis=1)
    assert np.median(err) < 1e-5          # measured 8.2e-7
    assert err.max() < 5e-5               # measured 5.0e-6 after ten control steps


def test_freefall_invariants_4096(kp):
    """Size-independent properties at BASELINE scale: COM acceleration = g, finite state, unit quaternions."""
    n = 4096
    qpos, qvel = make_states(n, 7, lift=10.0)
    model = kp.KpModel(contact=0)
    sim = kp.KpSim(model, n)
    sim.set_state(dev(qpos), dev(qvel)); sim.set_target(dev(qpos))
    a = torch.zeros((n, 75), device="cuda")
    mass = torch.tensor(KPM["body_mass"], dtype=torch.float32, device="cuda")
    coms = []
    for _ in range(4):
        sim.step_ctrl(a, 15)
        xi = sim.get("xipos").view(n, 24, 3)
        coms.append(((xi * mass[None, :, None]).sum(1) / mass.sum()).double().cpu().numpy())
    assert sim.diag()[:, 2].max() == 0
    dt = 15 * KPM["opt"][0]
    acc = (coms[3] - 2 * coms[2] + coms[1]) / dt ** 2
    assert np.abs(acc[:, 2].mean() + 9.81) < 2e-2
    assert np.abs(acc[:, :2]).mean() < 2e-2
    q = sim.get("qpos")[:, 3:7]
    assert (q.norm(dim=1) - 1).abs().max().item() < 1e-5


def test_contact_invariants_4096(kp):
    """BASELINE scale with contact (4096 envs = 64 distinct states x 64 copies, 10 control steps): bit-identical results
    for identical environments wherever they run (determinism across workgroups / CUs / launch rounds), and run-to-run;
    finite states, no hull sinking through the floor, supported bodies do not fall."""
    n, m = 4096, 64
    q64, v64 = make_states(m, 11, lift=0.0, vel=0.3, noise=0.05)
    qpos, qvel = np.tile(q64, (n // m, 1)), np.tile(v64, (n // m, 1))
    rng = np.random.default_rng(12)
    act = np.tile(rng.normal(size=(m, 75)) * 0.1, (n // m, 1))
    outs = []
    for rep in range(2):
        sim = kp.KpSim(kp.KpModel(), n)
        sim.set_state(dev(qpos), dev(qvel)); sim.set_target(dev(np.tile(STD["qpos"], (n, 1))))
        a = dev(act)
        for _ in range(10):
            sim.step_ctrl(a, 15)
        dg = sim.diag()
        assert dg[:, 2].max() == 0 and dg[:, 0].max() > 0
        outs.append((sim.get("qpos").cpu().numpy(), sim.get("qvel").cpu().numpy(), sim.get("xpos").cpu().numpy()))
    q, v, x = outs[0]
    assert np.isfinite(q).all() and np.isfinite(v).all()
    qt = q.reshape(n // m, m, 76)
    assert (qt == qt[0:1]).all(), "identical environments must give bit-identical states"
    assert (outs[0][0] == outs[1][0]).all() and (outs[0][1] == outs[1][1]).all(), "run-to-run determinism"
    assert x.reshape(n, 24, 3)[:, :, 2].min() > -0.03          # joint origins stay above the floor (soft contact penetration is mm)
    assert q[:, 2].min() > 0.6                                  # PD-held standing poses have not collapsed after 1/3 s


def test_launch_order_does_not_change_results(kp):
    """Model option lpt_order (workgroup i simulates env order[i], longest env of the previous launch first) is a schedule,
    not arithmetic: states are bit-identical with it on and off; kp_sim_launch_cost reports non-zero cycles for every env."""
    n = 1000                                                    # not a multiple of anything in the launch geometry
    qpos, qvel = make_states(n, 21, lift=0.0, vel=0.5, noise=0.2)
    act = np.random.default_rng(22).normal(size=(n, 75)) * 0.2
    res = []
    for lpt in (0, 1):
        sim = kp.KpSim(kp.KpModel(lpt_order=lpt), n)
        sim.set_state(dev(qpos), dev(qvel)); sim.set_target(dev(np.tile(STD["qpos"], (n, 1))))
        a = dev(act)
        for _ in range(4):
            sim.step_ctrl(a, 15)
        cost = sim.launch_cost()
        assert cost.shape == (n,) and cost.min() > 100_000 and cost.max() < 100_000_000
        res.append((sim.get("qpos").cpu().numpy(), sim.get("qvel").cpu().numpy(), sim.diag()))
    assert (res[0][0] == res[1][0]).all() and (res[0][1] == res[1][1]).all() and (res[0][2] == res[1][2]).all()


def _run_sched(kp, model, n, qpos, qvel, act, blk=None, steps=3, split=None, mask=None):
    sim = kp.KpSim(model, n)
    m8 = None if mask is None else torch.tensor(mask, dtype=torch.uint8, device="cuda")
    if blk is not None:
        sim.set_objects(dev(blk))
    sim.set_state(dev(qpos), dev(qvel)); sim.set_target(dev(np.tile(STD["qpos"], (n, 1))))
    a = dev(act)
    for _ in range(steps):
        for ns in (split or [15]):
            sim.step_ctrl(a, ns, m8)
    out = [sim.get(k).cpu().numpy() for k in ("qpos", "qvel", "xpos", "xquat", "qpos_d")]
    if blk is not None:
        out += [sim.get("obj_qpos").cpu().numpy(), sim.get("obj_qvel").cpu().numpy()]
    return out, sim.diag()


@pytest.mark.timeout(300, method="thread")
def test_job_queue_schedule_is_bit_identical(kp):
    """kp_step_queue_kernel (a control step = jobs of `substeps_per_job` substeps pulled from a FIFO by resident waves, the env's
    state handed from wave to wave through HBM) against kp_step_kernel (one workgroup per env): same bits in every state and
    diagnostic, (i) forced onto 48 / 7 slots so that every env migrates and the queue runs many rounds deep, with 5-, 4- (ragged
    last job) and 1-substep jobs, (ii) at the device's real slot count with 4096 envs, (iii) with free objects; and a control
    step launched as 5 + 5 + 5 substeps equals one launch of 15 (the hand-over the jobs rely on)."""
    n = 300
    qpos, qvel = make_states(n, 41, lift=0.0, vel=0.5, noise=0.2)
    act = np.random.default_rng(42).normal(size=(n, 75)) * 0.2
    ref, dref = _run_sched(kp, kp.KpModel(substeps_per_job=0), n, qpos, qvel, act)
    for spj, slots, fence in ((5, 48, 0), (4, 7, 0), (1, 48, 0), (5, 48, 1)):     # fence = 1: the release / acquire variant of the hand-over
        got, dg = _run_sched(kp, kp.KpModel(substeps_per_job=spj, queue_slots=slots, queue_fence=fence), n, qpos, qvel, act)
        for a_, b_ in zip(ref, got):
            assert (a_ == b_).all(), f"substeps_per_job={spj} slots={slots} queue_fence={fence}"
        assert (dref == dg).all()
    mask = (np.arange(n) % 3 != 1).astype(np.uint8)             # masked envs still pass through the queue (their jobs are empty)
    refm, drefm = _run_sched(kp, kp.KpModel(substeps_per_job=0), n, qpos, qvel, act, mask=mask)
    gotm, dgm = _run_sched(kp, kp.KpModel(substeps_per_job=3, queue_slots=16), n, qpos, qvel, act, mask=mask)
    for a_, b_ in zip(refm, gotm):
        assert (a_ == b_).all()
    assert (drefm == dgm).all() and (refm[0][mask == 0] == np.float32(qpos)[mask == 0]).all()
    # a control step launched as 5 + 5 + 5 substeps equals one launch of 15 -- with MuJoCo's plain warm start: the extrapolated start (warm_extrap, round 5)
    # begins anew with every kp_sim_step_ctrl call, so three calls extrapolate on other substeps than one call does (same minimisers, other rounding)
    ref0, _ = _run_sched(kp, kp.KpModel(substeps_per_job=0, warm_extrap=0), n, qpos, qvel, act)
    split, dsp = _run_sched(kp, kp.KpModel(substeps_per_job=0, warm_extrap=0), n, qpos, qvel, act, split=[5, 5, 5])
    for a_, b_ in zip(ref0[:5], split):
        assert (a_ == b_).all(), "5+5+5 substeps must equal one launch of 15"
    one_e, _ = _run_sched(kp, kp.KpModel(substeps_per_job=0, warm_extrap=0.75), n, qpos, qvel, act)
    split_e, _ = _run_sched(kp, kp.KpModel(substeps_per_job=0, warm_extrap=0.75), n, qpos, qvel, act, split=[5, 5, 5])
    assert np.abs(split_e[0] - one_e[0]).max() < 5e-5
    # real slot count
    n = 4096
    qpos, qvel = make_states(n, 43, lift=0.0, vel=0.5, noise=0.2)
    act = np.random.default_rng(44).normal(size=(n, 75)) * 0.2
    ref, dref = _run_sched(kp, kp.KpModel(substeps_per_job=0), n, qpos, qvel, act, steps=2)
    # queue_heavy: a wave that finds its env heavy runs the env's next job itself instead of queueing it (0 = never, 105 = a third of the jobs,
    # default 160): who runs a job must not change its result, and the shortened queue must still drain
    for heavy in (None, 0, 105):
        got, dg = _run_sched(kp, kp.KpModel(substeps_per_job=5, **({} if heavy is None else {"queue_heavy": heavy})), n, qpos, qvel, act, steps=2)
        for a_, b_ in zip(ref, got):
            assert (a_ == b_).all(), f"queue_heavy={heavy}"
        assert (dref == dg).all() and dg[:, 2].max() == 0
    # objects (6 envs/CU kernel): push scene and standing on the step box, replicated
    from kinpoly_amd.model_compiler import STEP_KPM
    n = 200
    x0, y0 = STD["qpos"][0], STD["qpos"][1]
    cases = [{1: [x0 + 1.2, y0, 0.921, 1, 0, 0, 0], 2: [x0 + 1.2, y0, 0.7905, 1, 0, 0, 0]}, {4: [x0, y0, 0.3705, 1, 0, 0, 0]}]
    qpos, qvel = make_states(n, 45, lift=0.0, vel=0.2, noise=0.05)
    qpos[1::2, 2] += 0.341
    act = np.random.default_rng(46).normal(size=(n, 75)) * 0.1
    blk = _obj_block(n, [cases[e % 2] for e in range(n)])
    ref, dref = _run_sched(kp, kp.KpModel(STEP_KPM, substeps_per_job=0), n, qpos, qvel, act, blk=blk)
    got, dg = _run_sched(kp, kp.KpModel(STEP_KPM, substeps_per_job=5, queue_slots=24), n, qpos, qvel, act, blk=blk)
    for a_, b_ in zip(ref, got):
        assert (a_ == b_).all()
    assert (dref == dg).all()


@pytest.mark.timeout(180, method="thread")
def test_absurd_targets_do_not_spin_the_kernel(kp):
    """The stable-PD target is unwrapped by multiples of 2 pi towards the joint angle (humanoid_im.py:447-452, a data-dependent
    loop in the reference).  Non-finite or absurd targets (a diverged kinematic policy) must neither hang the launch nor touch
    the other environments."""
    n = 8
    qpos, qvel = make_states(n, 31, lift=0.0, vel=0.2, noise=0.05)
    act = np.random.default_rng(32).normal(size=(n, 75)) * 0.1
    clean = np.tile(STD["qpos"], (n, 1))
    bad = clean.copy()
    bad[0, 17] = np.inf; bad[1, 30] = 1e30; bad[2, 44] = np.nan; bad[3, 9] = -np.inf; bad[3, 60] = -3e38
    outs = []
    for tgt in (clean, bad):
        sim = kp.KpSim(kp.KpModel(), n)
        sim.set_state(dev(qpos), dev(qvel)); sim.set_target(dev(tgt))
        for _ in range(2):
            sim.step_ctrl(dev(act), 15)
        outs.append((sim.get("qpos").cpu().numpy(), sim.diag()))
    assert (outs[0][0][4:] == outs[1][0][4:]).all()
    assert np.isfinite(outs[0][0]).all() and outs[0][1][:, 2].max() == 0
    bad_rows = ~np.isfinite(outs[1][0]).all(1)
    assert (outs[1][1][bad_rows, 2] != 0).all()                # whatever went non-finite is flagged


def test_target_fk_matches_golden(kp, golden):
    g = golden("fk")
    n = len(g["qpos_in"])
    sim = kp.KpSim(kp.KpModel(), n)
    sim.set_target(dev(g["qpos_in"]))
    for field, key in (("target_qpos", "qpos"), ("target_wbpos", "wbpos"), ("target_wbquat", "wbquat"), ("target_bquat", "bquat"), ("target_com", "body_com")):
        got = sim.get(field).cpu().numpy().astype(np.float64)
        np.testing.assert_allclose(got, g[key].reshape(n, -1), atol=5e-06, rtol=0, err_msg=field)        # measured 3.7e-07


def test_step_kin_and_bquat_match_golden(kp, golden):
    g = golden("env_funcs")
    n = len(g["qpos"])
    sim = kp.KpSim(kp.KpModel(), n)
    sim.set_state(dev(g["qpos"]), dev(g["qvel"]))
    nxt = sim.step_kin(dev(g["kin_action"])).cpu().numpy().astype(np.float64)
    np.testing.assert_allclose(nxt, g["next_qpos"], atol=1e-06, rtol=0)        # measured 9.1e-08
    bq = sim.get("bquat").cpu().numpy().astype(np.float64)
    # set_state normalises the root quaternion in place like mj_kinematics; the fixture's qpos are unit already
    np.testing.assert_allclose(bq, g["bquat"], atol=2e-06, rtol=0)        # measured 1.2e-07


def test_obs_cc_matches_pinned_oracle(kp, golden):
    """784-d UHC observation: HIP kernel vs oracle/np_oracle.obs_cc (pinned to the reference by
    tests/test_oracle_golden.py) on the simulator's own fresh qpos/qvel and stale kinematics."""
    g = golden("env_funcs")
    n = len(g["qpos"])
    rng = np.random.default_rng(5)
    sim = kp.KpSim(kp.KpModel(), n)
    sim.set_state(dev(g["qpos"]), dev(g["qvel"])); sim.set_target(dev(g["target_qpos"]))
    sim.step_ctrl(dev(rng.normal(size=(n, 75)) * 0.2), 15)
    obs = sim.obs_cc().cpu().numpy().astype(np.float64)
    rd = {k: sim.get(k).cpu().numpy().astype(np.float64) for k in ("qpos", "qvel", "xpos", "xquat", "xipos", "target_qpos")}
    for i in range(n):
        t = O.qpos_fk(rd["target_qpos"][i], BODY_POS, BODY_IPOS, PARENT)
        want = O.obs_cc(rd["qpos"][i], rd["qvel"][i], rd["xpos"][i].reshape(24, 3), rd["xquat"][i].reshape(24, 4), rd["xipos"][i].reshape(24, 3), t)
        np.testing.assert_allclose(obs[i], want, atol=5e-6, rtol=0)        # measured 6e-7 (tools/obs_reward_errors.py)
    # ZFilter + clip path
    zg = golden("gae_zfilter")
    ob2 = sim.obs_cc(zf_mean=dev(zg["zf_mean"]), zf_std=dev(zg["zf_std"]), clip=5.0).cpu().numpy().astype(np.float64)
    np.testing.assert_allclose(ob2, np.clip((obs - zg["zf_mean"]) / (zg["zf_std"] + 1e-8), -5, 5), atol=1e-05, rtol=1e-6)        # measured 8.3e-07


def test_obs_cc_small_headings_keep_their_digits(kp):
    """rel_heading = get_heading(target) - get_heading(current) (humanoid_im.py:185-190) when both headings are tiny: 2 acos(w) next to w = 1 would
    leave 1e-4 rad of fp32 noise where the reference (fp64) has none; the kernel takes the angle of (w, z) instead.  Feature 301 of the 784."""
    rng = np.random.default_rng(21)
    n = 40
    qpos = np.tile(np.asarray(STD["qpos"], np.float64), (n, 1))
    h0 = O.get_heading(qpos[0, 3:7])                 # the standing pose faces 3.0 rad: turn it back so that the headings below are the tiny ones
    tq = qpos.copy()
    hc, ht = 10.0 ** rng.uniform(-5, -2, n) * rng.choice([-1, 1], n), 10.0 ** rng.uniform(-5, -2, n) * rng.choice([-1, 1], n)

    def yawed(q, h):          # heading rotation about z in front of the stored root quaternion
        w1, z1 = np.cos(h / 2), np.sin(h / 2)
        w2, x2, y2, z2 = q
        return np.array([w1 * w2 - z1 * z2, w1 * x2 - z1 * y2, w1 * y2 + z1 * x2, w1 * z2 + z1 * w2])
    for i in range(n):
        qpos[i, 3:7] = yawed(qpos[i, 3:7], hc[i] - h0); tq[i, 3:7] = yawed(tq[i, 3:7], ht[i] - h0)
    sim = kp.KpSim(kp.KpModel(), n)
    sim.set_state(dev(qpos), dev(np.zeros((n, 75)))); sim.set_target(dev(tq))
    obs = sim.obs_cc().cpu().numpy().astype(np.float64)
    rd = {k: sim.get(k).cpu().numpy().astype(np.float64) for k in ("qpos", "qvel", "xpos", "xquat", "xipos", "target_qpos")}
    worst = 0.0
    for i in range(n):
        t = O.qpos_fk(rd["target_qpos"][i], BODY_POS, BODY_IPOS, PARENT)
        want = O.obs_cc(rd["qpos"][i], rd["qvel"][i], rd["xpos"][i].reshape(24, 3), rd["xquat"][i].reshape(24, 4), rd["xipos"][i].reshape(24, 3), t)
        rel = O.get_heading(rd["target_qpos"][i, 3:7]) - O.get_heading(rd["qpos"][i, 3:7])
        assert abs((want[301] - rel + np.pi) % (2 * np.pi) - np.pi) < 1e-8       # feature 301 is rel_heading, wrapped to (-pi, pi] (acos next to 1 costs even fp64 six digits)
        worst = max(worst, abs(obs[i, 301] - want[301]))
    assert worst < 2e-6, worst


def test_env_mask_and_reset(kp):
    n = 8
    qpos, qvel = make_states(n, 11, lift=10.0)
    sim = kp.KpSim(kp.KpModel(contact=0), n)
    sim.set_state(dev(qpos), dev(qvel)); sim.set_target(dev(qpos))
    mask = torch.tensor([1, 0, 1, 0, 1, 0, 1, 0], dtype=torch.uint8, device="cuda")
    sim.step_ctrl(torch.zeros((n, 75), device="cuda"), 15, env_mask=mask)
    q = sim.get("qpos").cpu().numpy()
    assert np.allclose(q[1::2], qpos[1::2].astype(np.float32), atol=1e-6)      # masked-out envs untouched
    assert np.abs(q[0::2, 2] - qpos[0::2, 2]).min() > 1e-4                    # stepped envs moved
    sim.set_state(dev(qpos), dev(qvel), env_mask=mask)
    q2 = sim.get("qpos").cpu().numpy()
    assert np.allclose(q2[0::2], qpos[0::2].astype(np.float32), atol=1e-6)


def _ctx_from_golden(kp, sim, g, n, T=6):
    """Per-env context of length T whose row cur_t carries the fixture's rows."""
    rng = np.random.default_rng(9)
    t = g["t"].astype(np.int32)
    head_pose = rng.normal(size=(n, T, 7)); head_vels = rng.normal(size=(n, T, 6)); obj_rel = rng.normal(size=(n, T, 7))
    gt_bquat = np.tile(np.array([1.0, 0, 0, 0]), (n, T, 24)); gt_wbpos = rng.normal(size=(n, T, 72))
    for i in range(n):
        head_pose[i, t[i]] = g["head_pose"][i]; head_vels[i, t[i]] = g["head_vels"][i]; obj_rel[i, t[i]] = g["obj_rel"][i]
        gt_bquat[i, t[i]] = g["gt_bquat"][i]; gt_bquat[i, t[i] - 1] = g["gt_prev_bquat"][i]; gt_wbpos[i, t[i]] = g["gt_wbpos"][i].reshape(-1)
    cur_t = torch.tensor(t, dtype=torch.int32, device="cuda")
    ctx = sim.make_ctx(T, dev(head_pose), dev(head_vels), dev(obj_rel), dev(g["action_one_hot"]), dev(gt_bquat), dev(gt_wbpos), cur_t,
                       obj_qpos=dev(g["obj_qpos"][:, :7]))
    return ctx, dict(head_pose=head_pose, head_vels=head_vels, obj_rel=obj_rel, gt_bquat=gt_bquat, gt_wbpos=gt_wbpos, t=t)


def test_obs_ar_and_reward_match_pinned_oracle(kp, golden):
    g = golden("ar_obs_reward")
    n = len(g["qpos"])
    rng = np.random.default_rng(6)
    sim = kp.KpSim(kp.KpModel(), n)
    sim.set_state(dev(g["qpos"]), dev(g["qvel"])); sim.set_target(dev(g["target_qpos"]))
    sim.step_begin()
    sim.step_ctrl(dev(rng.normal(size=(n, 75)) * 0.2), 15)
    ctx, c = _ctx_from_golden(kp, sim, g, n)
    obs = sim.obs_ar(ctx).cpu().numpy().astype(np.float64)
    rew, info, fail, diffs = sim.term_reward(ctx, kp.KpRewardCfg.default())
    rew, info, fail, diffs = rew.cpu().numpy(), info.cpu().numpy(), fail.cpu().numpy(), diffs.cpu().numpy()
    rd = {k: sim.get(k).cpu().numpy().astype(np.float64) for k in ("qpos", "xpos", "xquat", "target_qpos", "prev_bquat", "prev_hpos")}
    # prev snapshots are the pre-step body quats / head pose
    np.testing.assert_allclose(rd["prev_bquat"], np.stack([O.get_body_quat(q) for q in g["qpos"]]), atol=2e-06)        # measured 1.3e-07
    for i in range(n):
        t = c["t"][i]
        xpos, xquat = rd["xpos"][i].reshape(24, 3), rd["xquat"][i].reshape(24, 4)
        want = O.obs_ar(rd["qpos"][i], xpos, xquat, c["head_pose"][i, t], c["head_vels"][i, t], c["obj_rel"][i, t], g["action_one_hot"][i], g["obj_qpos"][i][:7])
        np.testing.assert_allclose(obs[i], want, atol=5e-6, rtol=0)        # measured 3e-7 .. 7e-7 (build flags move the last bit)
        tgt = O.qpos_fk(rd["target_qpos"][i], BODY_POS, BODY_IPOS, PARENT)
        head = np.concatenate([xpos[13], xquat[13]])
        r, inf = O.dynamic_supervision_v1(head, rd["prev_hpos"][i], O.get_body_quat(rd["qpos"][i]), rd["prev_bquat"][i], xpos, tgt, c["head_pose"][i, t],
                                          c["gt_bquat"][i, t], c["gt_bquat"][i, t - 1], 1.0 / 30.0, O.REWARD_WEIGHTS)
        np.testing.assert_allclose(info[i], inf, atol=2e-6, rtol=0)        # measured 6e-8
        assert abs(rew[i] - r) < 2e-6
        bd = O.calc_body_diff(xpos, tgt["wbpos"], DIFFW); bgd = O.calc_body_diff(xpos, c["gt_wbpos"][i, t].reshape(24, 3), DIFFW)
        np.testing.assert_allclose(diffs[i], [bd, bgd], rtol=1e-6, atol=3e-05)        # measured 2.7e-06
        assert bool(fail[i]) == bool(bd > 10 or bgd > 12)


def test_gae_matches_golden(kp, golden):
    g = golden("gae_zfilter")
    # the fixture is one flat batch; as an env-major layout it is a single env with T = B rows
    r, m, v = (dev(g[k].reshape(1, -1)) for k in ("rewards", "masks", "values"))
    adv, ret = kp.gae(r, m, v, 0.95, 0.95)
    adv = adv.double().cpu().numpy().reshape(-1, 1); ret = ret.double().cpu().numpy().reshape(-1, 1)
    np.testing.assert_allclose(ret, g["ret"], atol=2e-5)
    advn = (adv - adv.mean()) / adv.std(ddof=1)
    np.testing.assert_allclose(advn, g["adv"], atol=1e-05)        # measured 7.0e-07


def _obj_block(n, active):
    """data.qpos[76:111] as convert_obj_qpos builds it: all parked, `active` = {obj index: pose7} overrides."""
    blk = np.zeros((n, 35))
    for i in range(5):
        blk[:, 7 * i: 7 * i + 3] = [(i + 1) * 100, 100, 0]
    for e, d in enumerate(active):
        for oi, pose in d.items():
            blk[e, 7 * oi: 7 * oi + 7] = pose
    return blk


def test_object_contact_matches_oracle(kp):
    """Hull-vs-box / hull-vs-cylinder contacts of the active object (step box under the feet, Can against the legs,
    chair seat under the pelvis, table slab + legs) against the oracle on the same geoms."""
    from kinpoly_amd.model_compiler import STEP_KPM
    from oracle.kpo import object_geoms
    kpm = read_kpm(STEP_KPM)
    x0, y0 = STD["qpos"][0], STD["qpos"][1]
    yaw = lambda a: [np.cos(a / 2), 0, 0, np.sin(a / 2)]  # noqa: E731
    fk = O.qpos_fk(STD["qpos"], BODY_POS, BODY_IPOS, PARENT)
    verts, vadr = kpm["verts"].reshape(-1, 3), kpm["vert_adr"]
    tip = max((fk["wbpos"][b] + verts[vadr[b]:vadr[b + 1]] @ O.quaternion_matrix3(fk["wbquat"][b]).T)[:, 0].max() for b in (4, 8))
    cases = [
        ({4: [x0, y0, 0.23, 1, 0, 0, 0]}, 0.21),                       # standing on the step box (top at z = 0.20)
        ({4: [x0 + 0.3, y0, 0.23, *yaw(0.4)]}, 0.0),                    # half on / half beside a rotated step box
        ({3: [x0 + 0.36, y0 + 0.05, 0.69, 1, 0, 0, 0]}, 0.0),           # Can (cylinder r = 0.279) touching the legs
        ({0: [tip + 0.209 - 0.004, y0, 0.38, 1, 0, 0, 0]}, 0.0),          # chair seat block 4 mm into the toe tips
        ({2: [x0 + 0.55, y0, 0.95, 1, 0, 0, 0], 1: [x0 + 0.4, y0, 1.2, 1, 0, 0, 0]}, 0.0),   # table slab at hand height + box
        ({}, 0.0),                                                      # no active object: must equal the floor-only path
    ]
    n = len(cases)
    rng = np.random.default_rng(21)
    qpos = np.tile(STD["qpos"], (n, 1)); qvel = rng.normal(size=(n, 75)) * 0.1
    for e, (_, lift) in enumerate(cases):
        qpos[e, 2] += lift
    action = rng.normal(size=(n, 75)) * 0.1
    blk = _obj_block(n, [c[0] for c in cases])
    model = kp.KpModel(STEP_KPM)
    model.set_option("dynamic_objects", 0)          # this test: objects frozen as static obstacles
    sim = kp.KpSim(model, n)
    sim.set_objects(dev(blk))
    sim.set_state(dev(qpos), dev(qvel)); sim.set_target(dev(qpos))
    np.testing.assert_allclose(sim.get("obj_qpos").cpu().numpy(), blk, atol=1e-06)        # measured 4.8e-08
    a = dev(action)
    for _ in range(3):
        sim.step_ctrl(a, 15)
    got = sim.get("qpos").double().cpu().numpy(); gv = sim.get("qvel").double().cpu().numpy()
    dg = sim.diag()
    assert dg[:, 2].max() == 0
    errs, verrs, ncs = [], [], []
    for e in range(n):
        o = OracleSim(kpm=STEP_KPM)
        o.set_geoms(object_geoms(kpm, blk[e]))
        o.reset(qpos[e], qvel[e])
        ncon_max = 0
        for _ in range(3):
            o.do_simulation(action[e], qpos[e], 15)
            ncon_max = max(ncon_max, len(o.contacts()[0]))
        errs.append(np.abs(o.get("qpos") - got[e]).max())
        verrs.append(np.abs(o.get("qvel") - gv[e]).max()); ncs.append((int(dg[e, 0]), len(o.contacts()[0])))
    print("object-contact |dqpos| per case:", ["%.2e" % x for x in errs], "|dqvel|", ["%.2e" % x for x in verrs], "ncon (hip, oracle)", ncs)
    # near-ties between equally deep hull vertices on a flat face can pick a different 3-vertex contact set in fp32;
    # the trajectories still agree far inside north_star's 1e-3 rad budget
    assert max(errs) < 2e-5 and np.median(errs) < 5e-6           # measured 1.7e-6 / 4e-7
    # the object cases really produced object contacts (more than the floor alone) or changed the motion
    floor_only = got[-1]
    assert np.abs(got[0] - floor_only).max() > 0.05 and np.abs(got[2] - floor_only).max() > 1e-3


def test_dynamic_objects_match_oracle(kp):
    """Free-body dynamics of the active objects (object-floor, box-on-table, hull-object contacts coupled through the
    Newton solve) against the oracle: humanoid and object trajectories after 3 control steps."""
    from kinpoly_amd.model_compiler import STEP_KPM
    kpm = read_kpm(STEP_KPM)
    x0, y0 = STD["qpos"][0], STD["qpos"][1]
    fk = O.qpos_fk(STD["qpos"], BODY_POS, BODY_IPOS, PARENT)
    verts, vadr = kpm["verts"].reshape(-1, 3), kpm["vert_adr"]
    tip = max((fk["wbpos"][b] + verts[vadr[b]:vadr[b + 1]] @ O.quaternion_matrix3(fk["wbquat"][b]).T)[:, 0].max() for b in (4, 8))
    tilt = np.array([np.cos(0.3), np.sin(0.3) * 0.6, np.sin(0.3) * 0.8, 0.0])
    cases = [
        ({1: [x0 + 1.5, y0, 0.30, *tilt]}, 0.0),                                  # tilted box dropped next to the humanoid: impact + tumble
        ({1: [x0 + 1.2, y0, 0.921, 1, 0, 0, 0], 2: [x0 + 1.2, y0, 0.7905, 1, 0, 0, 0]}, 0.0),  # push scene: box resting on the table
        ({4: [x0, y0, 0.3705, 1, 0, 0, 0]}, 0.341),                               # standing on the (40 kg) step (top at 0.3405)
        ({3: [x0 + 0.36, y0 + 0.05, 0.69, 1, 0, 0, 0]}, 0.0),                     # Can against the legs
        ({1: [tip + 0.15 - 0.004, y0, 0.2205, 1, 0, 0, 0]}, 0.0),                 # 1 kg box on the floor, 4 mm into the toe tips
        ({0: [tip + 0.209 - 0.004, y0, 0.3805, 1, 0, 0, 0]}, 0.0),                # chair (100 t) at the toe tips
    ]
    n = len(cases)
    rng = np.random.default_rng(22)
    qpos = np.tile(STD["qpos"], (n, 1)); qvel = rng.normal(size=(n, 75)) * 0.1
    for e, (_, lift) in enumerate(cases):
        qpos[e, 2] += lift
    action = rng.normal(size=(n, 75)) * 0.1
    blk = _obj_block(n, [c[0] for c in cases])
    model = kp.KpModel(STEP_KPM)
    assert model.get_option("dynamic_objects") == 1
    sim = kp.KpSim(model, n)
    sim.set_objects(dev(blk))
    sim.set_state(dev(qpos), dev(qvel)); sim.set_target(dev(qpos))
    a = dev(action)
    nstep = 3
    for _ in range(nstep):
        sim.step_ctrl(a, 15)
    got = sim.get("qpos").double().cpu().numpy()
    gobj = sim.get("obj_qpos").double().cpu().numpy(); gobjv = sim.get("obj_qvel").double().cpu().numpy()
    dg = sim.diag()
    assert dg[:, 2].max() == 0
    errs, oerrs, moved = [], [], []
    for e in range(n):
        o = OracleSim(kpm=STEP_KPM)
        for slot, oi in enumerate(sorted(cases[e][0])):
            o.set_object(slot, kpm, oi, cases[e][0][oi])
        o.reset(qpos[e], qvel[e])
        for _ in range(nstep):
            o.do_simulation(action[e], qpos[e], 15)
        errs.append(np.abs(o.get("qpos") - got[e]).max())
        oe, mv = 0.0, 0.0
        for slot, oi in enumerate(sorted(cases[e][0])):
            oq, ov = o.get_object(slot)
            oe = max(oe, np.abs(oq - gobj[e, 7 * oi: 7 * oi + 7]).max(), 0.1 * np.abs(ov - gobjv[e, 6 * oi: 6 * oi + 6]).max())
            mv = max(mv, np.abs(oq - np.asarray(cases[e][0][oi], float)).max())
        oerrs.append(oe); moved.append(mv)
        # parked objects are untouched
        for oi in range(5):
            if oi not in cases[e][0]:
                np.testing.assert_allclose(gobj[e, 7 * oi: 7 * oi + 7], blk[e, 7 * oi: 7 * oi + 7], atol=1e-4)
    print("dynamic objects |dqpos| humanoid:", ["%.2e" % x for x in errs], "object:", ["%.2e" % x for x in oerrs], "object moved:", ["%.3f" % x for x in moved],
          "newton iters", dg[:, 1].tolist(), "ncon", dg[:, 0].tolist())
    assert max(errs) < 1e-5 and np.median(errs) < 4e-6           # measured 7.3e-7 / 3e-7
    assert max(oerrs) < 1e-5 and np.median(oerrs) < 2e-6         # measured 5.2e-7 / 6e-8
    assert moved[0] > 0.02 and moved[4] > 1e-4          # the dropped box fell; the light box was pushed by the toes


def test_save_restore_with_objects(kp):
    """MjSimState-style save / restore including the object block: replaying from a restored state follows the same trajectory
    (up to the solver warm start, which mujoco-py does not restore either)."""
    from kinpoly_amd.model_compiler import STEP_KPM
    n = 4
    x0, y0 = STD["qpos"][0], STD["qpos"][1]
    blk = _obj_block(n, [{4: [x0, y0, 0.3705, 1, 0, 0, 0]}] * n)
    qpos = np.tile(STD["qpos"], (n, 1)); qpos[:, 2] += 0.341
    rng = np.random.default_rng(5)
    qvel = rng.normal(size=(n, 75)) * 0.1
    sim = kp.KpSim(kp.KpModel(STEP_KPM), n)
    sim.set_objects(dev(blk)); sim.set_state(dev(qpos), dev(qvel)); sim.set_target(dev(qpos))
    a = dev(rng.normal(size=(n, 75)) * 0.1)
    for _ in range(2):
        sim.step_ctrl(a, 15)
    saved = {k: sim.get(k).clone() for k in ("qpos", "qvel", "qpos_d", "qvel_d", "obj_qpos", "obj_qvel")}
    for _ in range(2):
        sim.step_ctrl(a, 15)
    ref = {k: sim.get(k).clone() for k in ("qpos", "obj_qpos")}
    sim.set_obj_state(saved["obj_qpos"], saved["obj_qvel"])
    sim.set_full_state(saved["qpos"], saved["qvel"], saved["qpos_d"], saved["qvel_d"])
    for _ in range(2):
        sim.step_ctrl(a, 15)
    assert (sim.get("qpos") - ref["qpos"]).abs().max().item() < 2e-5
    assert (sim.get("obj_qpos") - ref["obj_qpos"]).abs().max().item() < 2e-5
    s2 = kp.KpSim(kp.KpModel(STEP_KPM), n)
    with pytest.raises(kp.KinPolyNativeError):
        s2.set_obj_state(saved["obj_qpos"], saved["obj_qvel"])


def test_lying_many_contacts_and_joint_limits(kp):
    """Hard cases of the constraint solve against the oracle: the humanoid lying on the floor in random orientations (30+
    simultaneous contacts, the 64-contact cap in reach) with joints wound past their +-180 degree limits (active limit rows)."""
    n = 12
    rng = np.random.default_rng(31)
    qpos = np.tile(STD["qpos"], (n, 1)); qvel = rng.normal(size=(n, 75)) * 0.05
    for e in range(n):
        ax = rng.normal(size=3); ax /= np.linalg.norm(ax)
        ang = rng.uniform(1.2, 1.9)                                   # tipped over by 70-110 degrees about a random axis
        tip = np.concatenate([[np.cos(ang / 2)], np.sin(ang / 2) * ax])
        qpos[e, 3:7] = O.quaternion_multiply(tip, STD["qpos"][3:7])
        qpos[e, 2] = 0.16 + 0.04 * rng.uniform()
        qpos[e, 7:] += rng.normal(size=69) * 0.15
    for e in range(n):                                                # three joints per env just past +-180 degrees
        jj = rng.choice(69, 3, replace=False)
        qpos[e, 7 + jj] = rng.choice([-1.0, 1.0], 3) * (np.pi + rng.uniform(0.02, 0.15, 3))
    action = rng.normal(size=(n, 75)) * 0.2
    sim = kp.KpSim(kp.KpModel(), n)
    sim.set_state(dev(qpos), dev(qvel)); sim.set_target(dev(qpos))
    a = dev(action)
    nstep = 4
    maxc = np.zeros(n, int)
    for _ in range(nstep):
        sim.step_ctrl(a, 15)
        dg = sim.diag()
        assert dg[:, 2].max() == 0
        maxc = np.maximum(maxc, dg[:, 3] & 255)
    got = sim.get("qpos").double().cpu().numpy()
    errs, ncs, nlims = [], [], []
    for e in range(n):
        o = OracleSim()
        o.reset(qpos[e], qvel[e])
        nl = 0
        for _ in range(nstep):
            o.do_simulation(action[e], qpos[e], 15)
            nl = max(nl, o.nefc - 4 * len(o.contacts()[0]))
        errs.append(np.abs(o.get("qpos") - got[e]).max()); ncs.append((int(maxc[e]), len(o.contacts()[0]))); nlims.append(nl)
    print("lying |dqpos|:", ["%.1e" % x for x in errs], "max contacts (hip) / last (oracle):", ncs, "limit rows:", nlims)
    assert max(c[0] for c in ncs) >= 30 and max(nlims) >= 1
    # equally deep vertices on flat hull faces can swap between fp32 and fp64 (different 3-vertex set): bounded, not bit-level
    assert max(errs) < 3e-5 and np.median(errs) < 5e-6           # measured 2.7e-6 / 4e-7


def test_c_abi_from_plain_cpp(kp):
    """examples/c_abi_demo (C++, no Python, no torch: kp_model_load .. kp_sim_step_ctrl .. kp_sim_get through the shared library)
    gives the same state as the ctypes binding on the same seeded inputs."""
    import re
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "examples", "c_abi_demo")
    assert os.path.exists(exe), "examples/c_abi_demo is built by __graft_entry__.build()"
    n, steps = 256, 3
    out = subprocess.run([exe, DEFAULT_KPM, os.path.join(root, "tests", "golden", "standing_neutral_qpos.f32"), str(n), str(steps)],
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    m = re.search(r"qpos checksum ([-0-9.]+), lowest root height ([-0-9.]+), contacts/env ([0-9.]+), non-finite (\d+)", out.stdout)
    assert m, out.stdout
    # the same inputs through the Python binding: the demo's LCG, in numpy
    q0 = np.fromfile(os.path.join(root, "tests", "golden", "standing_neutral_qpos.f32"), np.float32)
    qpos = np.tile(q0, (n, 1)).astype(np.float32)
    rng = np.uint32(12345)
    with np.errstate(over="ignore"):
        for e in range(n):
            for i in range(76):
                rng = np.uint32(rng * np.uint32(1664525) + np.uint32(1013904223))
                u = np.float32(np.float32(rng >> np.uint32(8)) * np.float32(1.0 / 16777216.0) - np.float32(0.5))
                if i >= 7:
                    qpos[e, i] = np.float32(q0[i] + np.float32(0.05) * u)
    sim = kp.KpSim(kp.KpModel(), n)
    q = torch.tensor(qpos, device="cuda")
    sim.set_state(q, torch.zeros((n, 75), device="cuda")); sim.set_target(q.clone())
    a = torch.zeros((n, 75), device="cuda")
    for _ in range(steps):
        sim.step_ctrl(a, 15)
    got = sim.get("qpos").double().cpu().numpy()
    assert abs(float(m.group(1)) - got.sum()) < 2e-2 and abs(float(m.group(2)) - got[:, 2].min()) < 1e-4 and int(m.group(4)) == 0
