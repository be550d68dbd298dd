"""Round 6, on the device: the floor scenes' job queue on the lean LDS layout (EnvLdsLean: three waves per SIMD) against the full layout -- same bits --,
the contact-overflow hand-over to kp_step_overflow_kernel, whole-episode parity against the CPU episode loop."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

pytestmark = pytest.mark.gpu
STD = np.load(os.path.join(ROOT, "tests", "golden", "standing_neutral.npz"))


def dev(a):
    return torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device="cuda")


def _run(kp, n, qpos, qvel, act, steps, **opts):
    sim = kp.KpSim(kp.KpModel(**opts), n)
    sim.set_state(dev(qpos), dev(qvel)); sim.set_target(dev(np.tile(STD["qpos"], (n, 1))))
    a = dev(act)
    for _ in range(steps):
        sim.step_ctrl(a, 15)
    return [sim.get(k).cpu().numpy() for k in ("qpos", "qvel", "xpos", "xquat", "xipos", "qpos_d")], sim.diag(), sim


@pytest.fixture(scope="module")
def kp():
    from kinpoly_amd import sim as kpsim
    return kpsim


def _states(n, seed, lying_every=0):
    rng = np.random.default_rng(seed)
    qpos = np.tile(STD["qpos"], (n, 1)); qpos[:, 7:] += np.clip(rng.normal(size=(n, 69)) * 0.2, -np.pi, np.pi)
    qvel = rng.normal(size=(n, 75)) * 0.5
    if lying_every:
        # every lying_every-th env lies on the floor (rolled about a horizontal axis by a seeded angle, pelvis 12 - 20 cm up): its hulls touch the plane all along
        # the body -- more contacts than the lean layout's 32
        idx = np.arange(0, n, lying_every)
        ang = rng.uniform(-np.pi, np.pi, idx.size)
        for k, e in enumerate(idx):
            c, s_ = np.cos(np.pi / 4), np.sin(np.pi / 4)             # 90 degrees about x: on the back / front ...
            q1 = np.array([c, s_, 0.0, 0.0])
            cz, sz = np.cos(ang[k] / 2), np.sin(ang[k] / 2)          # ... then any heading
            q2 = np.array([cz, 0.0, 0.0, sz])
            w1, x1, y1, z1 = q2; w2, x2, y2, z2 = q1
            qpos[e, 3:7] = [w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2, w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2, w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2, w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2]
            qpos[e, 2] = rng.uniform(0.12, 0.2)
            qpos[e, 7:] = STD["qpos"][7:] + rng.normal(size=69) * 0.05
            qvel[e] *= 0.2
    return qpos, qvel


def test_lean_queue_layout_is_the_full_layout_bit_for_bit(kp):
    """4096 floor envs, 3 control steps: kp_step_queue_kernel<false, true> (EnvLdsLean, 168 VGPRs, 10+ envs per CU) against the same queue on the full layout
    (lean_queue = 0) and against one workgroup per env -- every state, read-out and diagnostic identical; also with warm_extrap = 0.75, whose a_{k-2} the lean
    layout keeps in an HBM row."""
    n = 4096
    qpos, qvel = _states(n, 61)
    act = np.random.default_rng(62).normal(size=(n, 75)) * 0.2
    for extra in ({}, {"warm_extrap": 0.75}):
        lean, dl, sim = _run(kp, n, qpos, qvel, act, 3, lean_queue=1, **extra)
        assert sim.model.get_option("lds_bytes_per_env_lean") <= 12 * 1280
        full, df, _ = _run(kp, n, qpos, qvel, act, 3, lean_queue=0, **extra)
        for a_, b_ in zip(lean, full):
            assert (a_ == b_).all(), extra
        assert (dl == df).all()
        assert (dl[:, 3] & 255).max() <= 24 and (dl[:, 2] & 255).max() == 0
        if not extra:
            one, d1, _ = _run(kp, n, qpos, qvel, act, 3, substeps_per_job=0)
            for a_, b_ in zip(lean[:5], one[:5]):
                assert (a_ == b_).all()


def test_contact_overflow_goes_to_the_full_layout(kp):
    """A lean job that finds more contacts than its layout holds (24) hands the env to kp_step_overflow_kernel.  Floor scenes do not get there on their own (a
    humanoid lying flat has 12: mjc_PlaneConvex keeps at most 3 per hull), so the limit is lowered to 8 (model option lean_max_contacts): every 8th of 4096 envs
    lies on the floor and crosses it in some substep.  Those jobs (and the env's later ones) must come out of the overflow kernel exactly as the full-layout queue
    computes them, the other envs untouched by the detour, the queue drained and the status word clean."""
    n = 4096
    qpos, qvel = _states(n, 71, lying_every=8)
    act = np.random.default_rng(72).normal(size=(n, 75)) * 0.1
    lean, dl, sim = _run(kp, n, qpos, qvel, act, 4, lean_queue=1, lean_max_contacts=8, lean_adaptive=0)      # every launch on the lean layout + the overflow kernel
    assert sim.queue_counters()["lean_overflow_jobs"] >= 64
    full, df, _ = _run(kp, n, qpos, qvel, act, 4, lean_queue=0)
    maxcon = df[:, 3] & 255
    assert (maxcon > 8).sum() >= 64 and (maxcon <= 8).sum() >= 1024, f"some envs must cross the lowered limit and most must not (max contacts {maxcon.max()}, envs above 8: {(maxcon > 8).sum()})"
    assert (dl == df).all()
    for a_, b_ in zip(lean, full):
        assert (a_ == b_).all()
    assert int(sim.status_tensor()[2]) == 0
    assert np.isfinite(lean[0]).all() and (dl[:, 2] & 255).max() == 0


@pytest.mark.timeout(900, method="thread")
def test_whole_episodes_match_the_cpu_episode_loop():
    """128 envs x 99 control steps of the configs[2] rollout, VectorSampler against oracle/episode.py with the same clips, weights and exploration
    noise.  Outcomes: the step of every env's first termination, failures, mean reward; and |dqpos| along the episodes stays inside north_star's 1e-3 rad for
    all but the rows that follow a contact knife-edge flip (bounded in number)."""
    import episode_parity
    r = episode_parity.run(n=128, T=99, seed=7, objects=False, workers=min(32, os.cpu_count() or 1))
    print(r)
    assert r["bad_envs"] == 0
    assert r["first_termination_step_equal_frac"] >= 0.95
    assert r["mean_reward"]["rel_diff"] < 0.01
    assert abs(r["failures"]["hip"] - r["failures"]["oracle"]) <= max(2, int(0.02 * r["failures"]["oracle"]))
    # measured (MI355X, round 6): p50 8.1e-7, p99 7.4e-4, 0.83 % of the 12 672 env-steps above 1e-3 rad (the rows after a contact knife-edge flip), max 2e-2
    assert r["dqpos_aligned_rows"]["p50"] < 1e-5 and r["dqpos_aligned_rows"]["p99"] < 3e-3 and r["dqpos_aligned_rows"]["frac_above_1e-3"] < 0.03


def test_update_slices_are_reference_iterations():
    """AgentAR(min_batch_size): a sample() call's batch is cut into whole-env slices of about min_batch_size samples and each slice is ONE reference iteration
    (agent_ar.py:274-289: per_epoch_update -> sample(min_batch_size) -> update_params).  (i) a batch that fits min_batch_size takes the path
    without the option, bit for bit; (ii) 32 envs x 6 steps at min_batch_size 48 = 4 slices of 8 envs: four LambdaLR steps, 4 x num_optim_epoch Adam steps of the
    policy, 4 x num_step_update of the supervised optimiser, epoch += 4, and the ratios of the later slices are taken against the sampling policy."""
    from kinpoly_amd.agent import AgentAR
    from kinpoly_amd.env import standing_context
    from kinpoly_amd import sim as kpsim
    n, T = 32, 6
    fk_sim = kpsim.KpSim(kpsim.KpModel(), n, 0)

    def context_fn(m):
        return standing_context(m, T + 2, STD["qpos"], STD["qvel"], fk_sim, torch.zeros(m))
    kw = dict(device=0, horizon=T, num_optim_epoch=2, num_step_update=3, use_init_context=False, pool_depth=T, seed=11)
    ends = []
    for mbs in (0, n * T, 10 ** 6):
        agent = AgentAR(n, context_fn, min_batch_size=mbs, **kw)
        assert len(agent.update_slices(agent.sampler.sample(T))) == 1
        agent.sampler.start()
        agent.optimize_policy(0)
        ends.append(torch.cat([p.detach().reshape(-1) for p in agent.policy_net.parameters()]).clone())
        assert agent.epoch == 1
        del agent
    assert torch.equal(ends[0], ends[1]) and torch.equal(ends[0], ends[2])
    agent = AgentAR(n, context_fn, min_batch_size=48, **kw)
    info = agent.optimize_policy(0)
    assert info["update_slices"] == 4 and agent.epoch == 4 and len(info["surr_loss_per_slice"]) == 4
    assert agent.trainer.sched_p.last_epoch == 4 and agent.sched_sup.last_epoch == 4
    step_p = {int(st["step"]) for st in agent.trainer.opt_p.state.values()}
    step_s = {int(st["step"]) for st in agent.opt_sup.state.values()}
    assert step_p == {4 * 2} and step_s <= {4 * 3} and 4 * 3 in step_s, (step_p, step_s)
    assert np.isfinite(info["surr_loss_per_slice"]).all() and np.isfinite(info["step_loss_per_slice"]).all()
    # two ranks' worth of slicing arithmetic: the job-wide sample count decides the number of slices
    assert len(agent.update_slices(agent.sampler.sample(T))) == 4
    # the ratio against the sampling policy instead of the slice's starting parameters: same schedule arithmetic, finite losses
    agent = AgentAR(n, context_fn, min_batch_size=48, slice_ratio="behaviour", **kw)
    info = agent.optimize_policy(0)
    assert info["update_slices"] == 4 and agent.epoch == 4 and np.isfinite(info["surr_loss_per_slice"]).all()


def test_many_overflows_send_the_next_launches_to_the_full_layout(kp):
    """lean_adaptive (default): when more than 1 / 64 of the envs need more contact slots than the lean layout has -- a policy at random init resets every env onto
    a garbage pose -- the handle runs its next 64 control steps on the full layout (the overflow kernel's second pass costs more than the lean layout saves), then
    tries the lean one again.  The switch is a launch policy: states are those of either layout, bit for bit."""
    n = 4096
    qpos, qvel = _states(n, 81, lying_every=8)
    act = np.random.default_rng(82).normal(size=(n, 75)) * 0.1
    sim = kp.KpSim(kp.KpModel(lean_queue=1, lean_max_contacts=8), n)
    sim.set_state(dev(qpos), dev(qvel)); sim.set_target(dev(np.tile(STD["qpos"], (n, 1))))
    a = dev(act)
    seen = []
    for _ in range(6):
        sim.step_ctrl(a, 15)
        seen.append(sim.queue_counters())               # a host read: the count of the launch before is there when the next one is set up
    assert seen[0]["lean_layout_next_launch"] or seen[0]["fallbacks_to_full_layout"] >= 1
    assert seen[-1]["fallbacks_to_full_layout"] >= 1 and not seen[-1]["lean_layout_next_launch"] and seen[-1]["control_step_launches"] == 6
    full, df, _ = _run(kp, n, qpos, qvel, act, 6, lean_queue=0)
    for k, f in zip(("qpos", "qvel", "xpos", "xquat", "xipos", "qpos_d"), full):
        assert (sim.get(k).cpu().numpy() == f).all(), k
    assert (sim.diag() == df).all()
    quiet = kp.KpSim(kp.KpModel(), 4096)                 # the metric's kind of scene never overflows: the lean layout stays
    q2, v2 = _states(4096, 83)
    quiet.set_state(dev(q2), dev(v2)); quiet.set_target(dev(np.tile(STD["qpos"], (4096, 1))))
    for _ in range(4):
        quiet.step_ctrl(a, 15)
    c = quiet.queue_counters()
    assert c["fallbacks_to_full_layout"] == 0 and c["lean_layout_next_launch"] and c["lean_overflow_jobs"] == 0
