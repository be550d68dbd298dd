"""GPU tests added in round 2 (run with -m gpu on an MI355X), all through the C ABI of libkinpoly_sim.so:

  * PolicyMCP / Value: the fused fp32 device path against the REFERENCE's own forward (tests/golden/policies.npz was written by
    uhc/core/policy_mcp.py and uhc/khrylib/rl/core/critic.py with seeded weights that the test regenerates);
  * KP_M / KP_BIAS read-outs (mj_fullM / data.qfrc_bias, uhc/envs/humanoid_im.py:422-426) against the fp64 oracle;
  * the Newton solver's iteration cap: default = the model's mjOption.iterations (100), cap hits counted in kp_sim_diag.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from oracle import np_oracle as O  # noqa: E402
from oracle.kpo import OracleSim  # noqa: E402

STD = np.load(os.path.join(os.path.dirname(__file__), "golden", "standing_neutral.npz"))


@pytest.fixture(scope="module")
def kp():
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible")
    from kinpoly_amd import sim as kpsim
    return kpsim


def dev(a):
    return torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device="cuda")


def _mcp_shapes():
    shapes = {}
    for k in range(8):
        shapes[f"nets.{k}.0.affine_layers.0.weight"] = (512, 784); shapes[f"nets.{k}.0.affine_layers.0.bias"] = (512,)
        shapes[f"nets.{k}.0.affine_layers.1.weight"] = (256, 512); shapes[f"nets.{k}.0.affine_layers.1.bias"] = (256,)
        shapes[f"nets.{k}.1.weight"] = (75, 256); shapes[f"nets.{k}.1.bias"] = (75,)
    dims = [784, 300, 200, 8]
    for i in range(3):
        shapes[f"composer.0.affine_layers.{i}.weight"] = (dims[i + 1], dims[i]); shapes[f"composer.0.affine_layers.{i}.bias"] = (dims[i + 1],)
    shapes["action_log_std"] = (1, 75)
    return shapes


def test_policy_mcp_fused_path_matches_reference_forward(kp, golden):
    """nets.PolicyMCP (3 batched GEMMs, fp32, MFMA) loaded with the fixture's seeded weights vs the output the
    reference's PolicyMCP.forward produced for them (uhc/core/policy_mcp.py:30-38)."""
    from kinpoly_amd.nets import PolicyMCP
    g = golden("policies")
    shapes = _mcp_shapes()
    sd = O.seeded_state_dict([(str(k), shapes[str(k)]) for k in g["mcp_keys"]], int(g["mcp_seed"]))
    pol = PolicyMCP()
    missing = pol.load_state_dict({k: torch.tensor(v, dtype=torch.float32) for k, v in sd.items()}, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    pol = pol.cuda().float()
    x = dev(g["x"])
    with torch.no_grad():
        fused = pol.action_mean(x)                       # rollout path: fused GEMMs (no grad)
        w = pol.composer(x)
    np.testing.assert_allclose(w.double().cpu().numpy(), g["mcp_weights"], atol=2e-06)        # measured 1.7e-07
    np.testing.assert_allclose(fused.double().cpu().numpy(), g["mcp_mean"], atol=1e-05, rtol=1e-6)        # measured 7.0e-07
    # the training path (plain modules, autograd on) computes the same thing
    for p in pol.parameters():
        p.requires_grad_(True)
    plain = pol.action_mean(x)
    assert plain.requires_grad
    np.testing.assert_allclose(plain.detach().double().cpu().numpy(), g["mcp_mean"], atol=1e-05, rtol=1e-6)        # measured 5.9e-07
    # select_action(mean_action=True) is what the env calls in test mode
    with torch.no_grad():
        np.testing.assert_allclose(pol.select_action(x, True).double().cpu().numpy(), g["mcp_mean"], atol=1e-05, rtol=1e-6)        # measured 7.0e-07


def test_value_net_matches_reference_forward(kp, golden):
    from kinpoly_amd.nets import MLP, Value
    g = golden("policies")
    vshapes = {"net.affine_layers.0.weight": (512, 105), "net.affine_layers.0.bias": (512,), "net.affine_layers.1.weight": (256, 512),
               "net.affine_layers.1.bias": (256,), "value_head.weight": (1, 256), "value_head.bias": (1,)}
    vsd = O.seeded_state_dict([(str(k), vshapes[str(k)]) for k in g["value_keys"]], int(g["value_seed"]))
    val = Value(MLP(105, (512, 256), "relu"))
    val.load_state_dict({k: torch.tensor(v, dtype=torch.float32) for k, v in vsd.items()}, strict=True)
    val = val.cuda()
    with torch.no_grad():
        got = val(dev(g["s"]))
    np.testing.assert_allclose(got.double().cpu().numpy(), g["value"], atol=5e-06, rtol=1e-6)        # measured 3.5e-07


def test_mass_matrix_and_bias_readouts_match_oracle(kp):
    """KP_M / KP_BIAS = mj_fullM(model, M, data.qM)[:75,:75] / data.qfrc_bias[:75] (humanoid_im.py:422-426) of the state the
    derived quantities belong to: after set_state that is the state itself, after a control step it is x_14 (qpos_d)."""
    n = 8
    rng = np.random.default_rng(5)
    qpos = np.tile(STD["qpos"], (n, 1)); qpos[:, 2] += 0.3
    qpos[:, 7:] += rng.normal(size=(n, 69)) * 0.4
    q = rng.normal(size=(n, 4)); qpos[:, 3:7] = q / np.linalg.norm(q, axis=1, keepdims=True)
    qvel = rng.normal(size=(n, 75)) * 1.5
    sim = kp.KpSim(kp.KpModel(), n)
    sim.set_state(dev(qpos), dev(qvel))
    M, b = sim.mass_matrix()
    M, b = M.double().cpu().numpy(), b.double().cpu().numpy()
    q32, v32 = dev(qpos).double().cpu().numpy(), dev(qvel).double().cpu().numpy()
    o = OracleSim()
    for e in range(n):
        o.reset(q32[e], v32[e])
        Mo, bo = o.fullM(), o.get("qfrc_bias")
        np.testing.assert_allclose(M[e], Mo, atol=2e-6 * np.abs(Mo).max())        # measured 2e-7 of the largest entry
        np.testing.assert_allclose(M[e], M[e].T, atol=5e-7 * np.abs(Mo).max())        # measured 3e-8 of the largest entry
        np.testing.assert_allclose(b[e], bo, atol=2e-6 * max(1.0, np.abs(bo).max()))        # measured 1.7e-7 of the largest entry
    # the generic getter returns the same data, flattened
    np.testing.assert_array_equal(sim.get("M").view(n, 75, 75).double().cpu().numpy(), M)
    np.testing.assert_array_equal(sim.get("bias").double().cpu().numpy(), b)
    # after a control step the read-outs belong to x_14 (stale derived quantities, like mujoco-py's data.qM)
    act = dev(rng.normal(size=(n, 75)) * 0.2)
    sim.set_target(dev(qpos))
    sim.step_ctrl(act, 15)
    M2 = sim.get("M").view(n, 75, 75).double().cpu().numpy()
    qd, vd = sim.get("qpos_d").double().cpu().numpy(), sim.get("qvel_d").double().cpu().numpy()
    for e in range(2):
        o.reset(qd[e], vd[e])
        np.testing.assert_allclose(M2[e], o.fullM(), atol=2e-6 * np.abs(M2[e]).max())        # measured 2e-7 of the largest entry


def _buried_states(n, seed):
    rng = np.random.default_rng(seed)
    qpos = np.tile(STD["qpos"], (n, 1)); qvel = rng.normal(size=(n, 75)) * 0.5
    for e in range(n):
        if e % 2:
            q = rng.normal(size=4); qpos[e, 3:7] = q / np.linalg.norm(q)
        qpos[e, 2] = rng.uniform(-0.1, 0.35); qpos[e, 7:] += rng.normal(size=69) * 0.3
    return qpos, qvel, rng.normal(size=(n, 75)) * 0.3


def test_newton_iteration_cap_is_the_models_and_cap_hits_are_counted(kp):
    """the product used to cap Newton at 12 iterations silently.  Now the default is the blob's
    mjOption.iterations (100, what the oracle and MuJoCo run with), diag counts the substeps that ended at the cap, and on
    half-buried starts (dozens of deep contacts, the worst case seen in training) no env hits it and HIP == oracle."""
    model = kp.KpModel()
    assert model.get_option("solver_iter") == 100
    n = 32
    qpos, qvel, act = _buried_states(n, 11)
    sim = kp.KpSim(model, n)
    sim.set_state(dev(qpos), dev(qvel)); sim.set_target(dev(np.tile(STD["qpos"], (n, 1))))
    a = dev(act)
    sim.step_ctrl(a, 15)
    dg = sim.diag()
    assert int((dg[:, 2] >> 8).max()) == 0, "a Newton solve ended at the 100-iteration cap"
    assert int((dg[:, 2] & 1).max()) == 0
    assert int(dg[:, 3].max() & 255) >= 20          # the scene family really is contact-heavy
    got = sim.get("qpos").double().cpu().numpy()
    q32, v32, a32 = dev(qpos).double().cpu().numpy(), dev(qvel).double().cpu().numpy(), a.double().cpu().numpy()
    o = OracleSim()
    for e in range(8):
        o.reset(q32[e], v32[e]); o.do_simulation(a32[e], STD["qpos"], 15)
        assert np.abs(o.get("qpos") - got[e]).max() < 1e-4
    # a deliberately small cap is reported, not hidden
    m2 = kp.KpModel(solver_iter=2)
    s2 = kp.KpSim(m2, n)
    s2.set_state(dev(qpos), dev(qvel)); s2.set_target(dev(np.tile(STD["qpos"], (n, 1))))
    s2.step_ctrl(a, 15)
    d2 = s2.diag()
    assert int((d2[:, 2] >> 8).sum()) > 0 and int((d2[:, 2] >> 8).max()) <= 15
    assert (d2[:, 1] <= 2 * 15).all()


def test_sim_follows_an_explicit_stream_rebind(kp):
    """KpSim enqueues on the stream it was created on; use_current_stream() rebinds it (kp_sim_set_stream)."""
    n = 4
    sim = kp.KpSim(kp.KpModel(), n)
    q, v = dev(np.tile(STD["qpos"], (n, 1))), dev(np.tile(STD["qvel"], (n, 1)))
    sim.set_state(q, v); sim.set_target(q)
    a = torch.zeros((n, 75), device="cuda")
    sim.step_ctrl(a, 15)
    want = sim.get("qpos").clone()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        sim.use_current_stream()
        sim.set_state(q, v); sim.set_target(q)
        sim.step_ctrl(a, 15)
        got = sim.get("qpos")
    side.synchronize()
    sim.use_current_stream()
    assert torch.equal(got, want)
    st = sim.status_tensor()
    assert st.shape == (4,) and int(st[2]) == 0


def _object_scenes(n, seed):
    """Randomised scenes of all four action classes (tools/obj_fuzz.py generator): objects dropped, tilted and shifted into the humanoid."""
    from kinpoly_amd.model_compiler import STEP_KPM, read_kpm
    kpm = read_kpm(STEP_KPM)
    rng = np.random.default_rng(seed)
    x0, y0 = STD["qpos"][0], STD["qpos"][1]
    nominal = {0: [[0.0, -0.45, 0.3805]], 1: [[0.0, 0.55, 0.921], [0.0, 0.55, 0.7905]], 2: [[0.0, 0.45, 0.69]], 3: [[0.0, 0.0, 0.3705]]}
    obj_of_action = {0: [0], 1: [1, 2], 2: [3], 3: [4]}
    blk = np.zeros((n, 35))
    for i in range(5):
        blk[:, 7 * i: 7 * i + 3] = [(i + 1) * 100, 100, 0]
    qpos = np.tile(STD["qpos"], (n, 1)); qvel = rng.normal(size=(n, 75)) * 0.2
    scenes = []
    for e in range(n):
        a = e % 4
        shift = rng.normal(size=2) * 0.15
        lift = rng.uniform(0, 0.25) if rng.uniform() < 0.5 else 0.0
        ax = rng.normal(size=3); ax /= np.linalg.norm(ax)
        ang = rng.normal() * (0.25 if rng.uniform() < 0.5 else 0.0)
        tilt = np.concatenate([[np.cos(ang / 2)], np.sin(ang / 2) * ax])
        objs = {}
        for oi, (lx, ly, lz) in zip(obj_of_action[a], nominal[a]):
            objs[oi] = [x0 + lx + shift[0], y0 + ly + shift[1], lz + lift + 0.0003, *tilt]
            blk[e, 7 * oi: 7 * oi + 7] = objs[oi]
        if a == 3:
            qpos[e, 2] += 0.341 + lift + 0.02
        qpos[e, 7:] += rng.normal(size=69) * 0.1
        scenes.append(objs)
    return kpm, blk, qpos, qvel, scenes


def test_contact_sets_match_oracle_contact_by_contact(kp):
    """Narrow-phase parity below the trajectory level: for one state, the contact list the kernel builds (mjc_PlaneConvex walk of the
    hull graph, mjc_PlaneBox / mjc_PlaneCylinder, libccd MPR for hull - box / cylinder, box - box clipping) equals the oracle's
    (oracle/kp_collide.h) contact by contact: same entity pairs in the same order, dist / position / normal to fp32 accuracy."""
    from kinpoly_amd.model_compiler import STEP_KPM
    n = 48
    kpm, blk, qpos, qvel, scenes = _object_scenes(n, 5)
    sim = kp.KpSim(kp.KpModel(STEP_KPM), n)
    sim.record_contacts()
    sim.set_objects(dev(blk)); sim.set_state(dev(qpos), dev(qvel)); sim.set_target(dev(qpos))
    sim.step_ctrl(dev(np.zeros((n, 75))), 1)
    hip = sim.contacts()
    q32, v32, b32 = dev(qpos).double().cpu().numpy(), dev(qvel).double().cpu().numpy(), dev(blk).double().cpu().numpy()
    kinds = set()
    n_contacts = 0
    for e in range(n):
        o = OracleSim(kpm=STEP_KPM)
        for slot, oi in enumerate(sorted(scenes[e])):
            o.set_object(slot, kpm, oi, b32[e, 7 * oi:7 * oi + 7])
        o.reset(q32[e], v32[e])
        c, h = o.contacts_full(), hip[e]
        assert list(c["body"]) == list(h["body"]) and list(c["b2"]) == list(h["b2"]), f"scene {e}: entity lists differ"
        if len(c["body"]) == 0:
            continue
        n_contacts += len(c["body"])

        def canon(x):      # the order of the contacts inside one geom pair carries no meaning (box - box clipping may start its polygon elsewhere in fp32)
            key = np.lexsort((np.round(x["pos"][:, 2], 4), np.round(x["pos"][:, 1], 4), np.round(x["pos"][:, 0], 4), x["b2"], x["body"]))
            return {k: v[key] for k, v in x.items()}
        c, h = canon(c), canon(h)
        np.testing.assert_allclose(h["dist"], c["dist"], atol=2e-06, err_msg=f"scene {e}")        # measured 1.7e-07
        np.testing.assert_allclose(h["pos"], c["pos"], atol=2e-06, err_msg=f"scene {e}")        # measured 1.8e-07
        np.testing.assert_allclose(h["normal"], c["normal"], atol=2e-05, err_msg=f"scene {e}")        # measured 1.6e-06
        for a, b in zip(c["body"], c["b2"]):
            kinds.add(("hull" if a < 24 else "obj", "floor" if b < 0 else "obj"))
    assert kinds == {("hull", "floor"), ("hull", "obj"), ("obj", "floor"), ("obj", "obj")} and n_contacts > 300


def test_floor_contacts_follow_the_hull_graph_rule(kp):
    """mjc_PlaneConvex on the device: per hull the deepest vertex plus hull-graph neighbours not within 0.3 rbound of it -- never more
    than maxplanemesh = 3 contacts per hull, and the same set as the oracle on feet-flat, lying and half-buried states."""
    n = 24
    qpos, qvel, act = _buried_states(n, 3)
    qpos[:8] = STD["qpos"]; qpos[:8, 2] -= np.linspace(0.0, 0.03, 8)          # feet pressed flat into the floor
    sim = kp.KpSim(kp.KpModel(), n)
    sim.record_contacts()
    sim.set_state(dev(qpos), dev(qvel)); sim.set_target(dev(np.tile(STD["qpos"], (n, 1))))
    sim.step_ctrl(dev(act), 1)
    hip = sim.contacts()
    q32, v32 = dev(qpos).double().cpu().numpy(), dev(qvel).double().cpu().numpy()
    o = OracleSim()
    for e in range(n):
        o.reset(q32[e], v32[e])
        c, h = o.contacts_full(), hip[e]
        assert list(c["body"]) == list(h["body"])
        assert len(h["body"]) == 0 or np.bincount(h["body"]).max() <= 3
        np.testing.assert_allclose(h["dist"], c["dist"], atol=2e-6)
        np.testing.assert_allclose(h["pos"], c["pos"], atol=2e-6)
    assert max(len(h["body"]) for h in hip) >= 10


def test_fused_gru_unroll_matches_gru_cell_loop_forward_and_backward(kp):
    """the fused recurrence (kinpoly_amd/gru_unroll.py: HIP gate kernels, one recurrent GEMM per step, weight gradients as
    one GEMM) vs torch.nn.GRUCell stepped in a Python loop: hidden states and the gradients of W_ih, W_hh, b_ih, b_hh and hx0 under a
    smooth loss, with mid-batch episode starts.  (The MLP on top is the same torch module on both paths; its relu makes parameter
    gradients jump with 1-ulp changes of its input, so the recurrence is compared on its own.)"""
    from kinpoly_amd.gru_unroll import gru_unroll
    from kinpoly_amd.nets import KinPolicy
    torch.manual_seed(3)
    pol = KinPolicy().cuda()
    cell = pol.action_rnn.rnn_f
    N, T = 48, 9
    states = torch.randn((N, T, 105), device="cuda") * 0.5
    starts = torch.rand((N, T), device="cuda") < 0.25
    starts[:, 0] = torch.rand(N, device="cuda") < 0.5
    hx0 = (torch.randn((N, 1024), device="cuda") * 0.3).requires_grad_(True)
    w = torch.randn((N, T, 1024), device="cuda")

    def loop():
        hx, outs = hx0, []
        for t in range(T):
            hx = cell(states[:, t], hx * (~starts[:, t]).float().unsqueeze(1))
            outs.append(hx)
        return torch.stack(outs, 1)
    res = []
    for fn in (lambda: gru_unroll(cell, states, starts, hx0), loop):
        for p in cell.parameters():
            p.grad = None
        hx0.grad = None
        h = fn()
        (h * w).sum().backward()
        res.append((h.detach().clone(), [p.grad.detach().clone() for p in cell.parameters()] + [hx0.grad.detach().clone()]))
    assert float((res[0][0] - res[1][0]).abs().max()) < 1e-6
    for a, b in zip(res[0][1], res[1][1]):
        assert float((a - b).abs().max()) <= 2e-5 * max(1.0, float(b.abs().max())), (tuple(a.shape), float((a - b).abs().max()), float(b.abs().max()))
    # the whole policy head on top: same means as the per-step loop, and episode starts cut the recurrence
    with torch.no_grad():
        m_f, m_l = pol.unroll(states, starts, hx0.detach()), pol.unroll_reference(states, starts, hx0.detach())
        m_z = pol.unroll(states, starts, torch.zeros_like(hx0))
    assert float((m_f - m_l).abs().max()) < 2e-5
    cut = starts[:, 0]
    assert torch.equal(m_f[cut], m_z[cut]) and not torch.allclose(m_f[~cut], m_z[~cut])


def test_fused_unroll_matches_reference_padded_forward_fp32(kp, golden):
    from tests.test_context_cpu import _net_from_fixture
    g, gt = golden("unroll"), golden("traj_ar_net")
    net = _net_from_fixture(gt, torch.float32).cuda()
    masks = g["masks"]
    N, T = 2, 7
    starts = np.concatenate([[True], masks[:-1] == 0]).reshape(N, T); starts[:, 0] = True
    with torch.no_grad():
        means = net.unroll(dev(g["states"]).view(N, T, -1), torch.tensor(starts, device="cuda"))
    np.testing.assert_allclose(means.reshape(N * T, -1).double().cpu().numpy(), g["action_mean"], atol=1e-06)        # measured 8.3e-09


def test_resting_box_depth_matches_the_closed_form_on_the_device(kp):
    """The known answer of tests/test_physics_oracle.py (soft-contact model in closed form: 16 D(r) k d(r) |r| = m g) for the HIP
    kernel's own coupled solve: the free step box alone on the floor settles at dist = r* + margin on its four mjc_PlaneBox corners."""
    from scipy.optimize import brentq
    from kinpoly_amd.model_compiler import STEP_KPM, read_kpm
    kpm = read_kpm(STEP_KPM)
    opt = kpm["opt"]
    inert = kpm["obj_inertial"].reshape(-1, 13)[4]
    mass, invw = inert[0], inert[10]
    tc, dr, (d0, dw, width, mid, power), mu, margin = max(opt[4], 2 * opt[0]), opt[5], opt[6:11], opt[11], opt[14]

    def load(r):
        x = min(abs(r) / width, 1.0)
        y = x * x / mid if x <= mid else 1.0 - (1.0 - x) ** 2 / (1.0 - mid)
        d = d0 + y * (dw - d0)
        R = (1.0 - d) / d * (1.0 + mu * mu) * invw
        return 16.0 * d * abs(r) / (dw * dw * tc * tc * dr * dr) / (2.0 * mu * mu * R) - mass * 9.81
    r_star = -brentq(lambda x: load(-x), 1e-9, 0.5 * width)

    n = 4
    blk = np.zeros((n, 35))
    for i in range(5):
        blk[:, 7 * i: 7 * i + 3] = [(i + 1) * 100, 100, 0]
    blk[:, 28:35] = [0.0, 0.0, 0.3705, 1, 0, 0, 0]                  # the step box on the floor, 30 m from the humanoid
    qpos = np.tile(STD["qpos"], (n, 1)); qpos[:, 0] += 30
    sim = kp.KpSim(kp.KpModel(STEP_KPM), n)
    sim.record_contacts()
    sim.set_objects(dev(blk)); sim.set_state(dev(qpos), dev(np.zeros((n, 75)))); sim.set_target(dev(qpos))
    for _ in range(100):
        sim.step_ctrl(dev(np.zeros((n, 75))), 15)
    c = sim.contacts()[0]
    mine = c["body"] == 24
    assert mine.sum() == 4 and np.all(c["b2"][mine] == -1)
    assert float(sim.get("obj_qvel")[0, 24:30].abs().max()) < 1e-4
    np.testing.assert_allclose(c["dist"][mine], r_star + margin, rtol=2e-3)      # fp32 positions at z = 0.37 m resolve 3e-8 m of a 9e-4 m gap
