#!/usr/bin/env python
"""bench.py -- env-steps/sec of the dynamics-regulated rollout hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run, one rank per GPU)

One "step" = one batched env-step of the kin_poly.yml rollout over ENVS_PER_GPU = 4096 environments
per GPU: kinematic policy (GRU+MLP) -> step_ar -> target FK -> UHC obs (784) + ZFilter
-> PolicyMCP -> 15 physics substeps (stable-PD + RFC + forward dynamics + hull-plane contact) ->
termination + reward -> AR obs (105) -> device-side auto-reset.  Inputs are synthetic and resident in HBM
before the timed region (standing clip contexts, seeded random-init networks).

Workloads (all BASELINE configs[2] env-steps at 4096 envs/GPU unless noted):
    tracked      (default, the metric) episodes that last: the kinematic policy's GEMMs run, its output is replaced by the clip's own pose +
                 N(0, e^-3.2) exploration noise -- what a pretrained kinematic policy emits (the reference starts RL from a supervised one)
    random_init  seeded random-init networks: the body-diff termination ends (almost) every episode after one step, so every timed step
                 contains the device-side reset (round 1's headline; secondary now)
    wild_eval    BASELINE configs[4]: the --wild evaluation path (humanoid_smpl_neutral_mesh_all.xml, mode "test" = mean actions of both
                 policies, no GT termination, fail-safe on), inference only
    objects      BASELINE configs[3] on one GPU: the four action classes of dataset.synthetic_takes (sit: chair / push: box on table /
                 avoid: Can / step: step box) uniformly over the 4096 envs, the active objects simulated as free
                 bodies (kp_step_queue_kernel<true>), the kinematic policy's output replaced by the clip's next pose + noise
    train_iter   one AgentAR.optimize_policy per "step": sample 4096 x 24 env-steps, GAE, all-gather of advantages / returns,
                 data-parallel PPO + supervised update (agent_ar.py:271-297); value = env-steps/s of the whole iteration

Prints ONE JSON line on rank 0 (see the task contract) with `roofline` (dominant kernel kp_step_queue_kernel = kp_step_kernel scheduled as jobs,
live HIP-event launch durations; `traffic` / `valu` only from a rocprofv3 --pmc pass of THIS command and workload committed under profiles/)
and, at N = 1, `mfma` (the policy GEMMs timed in isolation), secondary workloads and `cpu_baseline` (the fp64 oracle port in 35
single-threaded worker processes, with its counted FLOPs per env-step).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ENVS_PER_GPU = 4096
CLIP_LEN = 100                      # fr_num (config/statear/kin_poly.yml:11)
ALGO_BYTES_PER_ENV_STEP = 2772      # humanoid-only compulsory fp32 traffic of do_simulation
ALGO_BYTES_PER_ENV_STEP_OBJ = 3292  # with the object block (nq 111 / nv 105)
TRAIN_HORIZON = 24                  # env-steps per env and iteration of the train_iter workload (4096 x 24 = 98 304 samples; kin_poly.yml asks for >= 10 000)
HBM_PEAK_GBS = 8000.0               # MI355X_MICROARCH.md: 8.0 TB/s spec
VALU_FP32_PEAK_TFLOPS = 157.3       # spec-sheet vector fp32 = the PACKED rate (v_pk_fma_f32): 256 CUs x 4 SIMDs x 16 lanes per clock x 2 (packed pair) x 2 (FMA) x 2.4 GHz.
                                    # Un-packed VALU fp32 -- what kp_step_kernel issues -- tops out at half of it, 78.6 TF = one wave-instruction per 4 cycles per SIMD =
                                    # 614.4 G wave-instructions/s, the ceiling roofline.issue.valu_issue_frac is measured against
MFMA_FP32_PEAK_TFLOPS = 157.3       # dense fp32 matrix peak (spec sheet): the policy / value GEMMs run in fp32
PROFILE_DIR = os.path.join(ROOT, "profiles", "r06")


POLICY_CKPT = CC_CKPT = None       # --policy-ckpt / --cc-ckpt: the networks of the `trained` workload (reference checkpoint layout, kinpoly_amd/checkpoint.py)


def build_engine(device_index, seed, threads, workload="tracked", n_envs=None):
    from kinpoly_amd.env import BatchedHumanoidAREnv, standing_context
    from kinpoly_amd.nets import KinPolicy, enable_tuned_gemms
    from kinpoly_amd.rollout import VectorSampler
    build_engine.tuned = enable_tuned_gemms()
    torch.manual_seed(seed)
    std = np.load(os.path.join(ROOT, "tests", "golden", "standing_neutral.npz"))
    wild = workload == "wild_eval"
    n_envs = ENVS_PER_GPU if n_envs is None else int(n_envs)
    opts = {"threads_per_env": threads, **({"substeps_per_job": int(os.environ["KP_SUBSTEPS_PER_JOB"])} if "KP_SUBSTEPS_PER_JOB" in os.environ else {}),
            **({"lpt_order": int(os.environ["KP_LPT_ORDER"])} if "KP_LPT_ORDER" in os.environ else {}),
            **({"queue_heavy": int(os.environ["KP_QUEUE_HEAVY"])} if "KP_QUEUE_HEAVY" in os.environ else {}),
            **({"queue_prio": int(os.environ["KP_QUEUE_PRIO"])} if "KP_QUEUE_PRIO" in os.environ else {}),
            **({"queue_late": int(os.environ["KP_QUEUE_LATE"])} if "KP_QUEUE_LATE" in os.environ else {}),
            **({"queue_fence": int(os.environ["KP_QUEUE_FENCE"])} if "KP_QUEUE_FENCE" in os.environ else {}),
            **({"queue_slots": int(os.environ["KP_QUEUE_SLOTS"])} if "KP_QUEUE_SLOTS" in os.environ else {}),
            **({"lean_queue": int(os.environ["KP_LEAN_QUEUE"])} if "KP_LEAN_QUEUE" in os.environ else {}),
            **({"lds_pad": int(os.environ["KP_LDS_PAD"])} if "KP_LDS_PAD" in os.environ else {}),
            **({"warm_extrap": float(os.environ["KP_WARM_EXTRAP"])} if "KP_WARM_EXTRAP" in os.environ else {})}
    trained = None
    if workload == "trained":
        from kinpoly_amd.checkpoint import load_bench_policies
        if not POLICY_CKPT:
            raise SystemExit("--workload trained needs --policy-ckpt (and --cc-ckpt for the UHC's ZFilter)")
        trained = load_bench_policies(POLICY_CKPT, CC_CKPT, torch.device("cuda", device_index))
    env = BatchedHumanoidAREnv(n_envs, device_index, mode="test" if wild else "train", wild=wild, seed=seed, model_options=opts,
                               **({"cc_policy": trained["cc_policy"], "cc_running_state": trained["cc_running_state"]} if trained else {}))
    policy = trained["kin_policy"] if trained else KinPolicy().to(env.device).float()
    g = torch.Generator().manual_seed(seed)
    headings = (torch.rand(n_envs, generator=g) * 2 - 1) * np.pi
    if workload == "objects":
        ctx = object_contexts(env, std, seed)
    else:
        ctx = standing_context(n_envs, CLIP_LEN, std["qpos"], std["qvel"], env.sim, headings)
    if wild:            # the kinematic roll-out the fail-safe falls back to (ar_context['ar_qpos' / 'ar_qvel']): the clip itself
        ctx["ar_qpos"], ctx["ar_qvel"] = ctx["qpos"].clone(), torch.zeros((n_envs, CLIP_LEN, 75), device=env.device)
    env.load_context(ctx)
    sampler = VectorSampler(env, policy, mean_action=wild)
    sampler.start()
    return env, policy, sampler, std


def object_contexts(env, std, seed):
    """BASELINE configs[3] stand-in (the MoCap set is absent): SURVEY 8(d) config 4 = dataset.synthetic_takes -- standing, then seeded
    smooth joint-space sinusoids, the action's object(s) at constant poses around the humanoid, random yaw -- 8 takes per action class,
    CLIP_LEN-frame clips drawn uniformly over the takes (StateARDataset.sample_batch), so the four classes share the 4096 envs evenly."""
    from kinpoly_amd import dataset as D
    from kinpoly_amd.model_compiler import read_kpm
    from kinpoly_amd.sim import STEP_KPM
    takes = D.synthetic_takes(env.sim, std["qpos"], n_per_action=8, T_range=(CLIP_LEN + 10, CLIP_LEN + 60), body_mass=read_kpm(STEP_KPM)["body_mass"], seed=seed)
    ds = D.StateARDataset(takes, fr_num=CLIP_LEN, seed=seed, device=env.device)
    ctx = ds.sample_batch(env.n, use_freq=False)
    starts = ds.rng.randint(0, 10, env.n)
    ctx = ds.batch(ctx["take_ind"].numpy(), starts, CLIP_LEN)
    ctx["init_qpos"], ctx["init_qvel"] = ctx["qpos"][:, 0].contiguous(), ctx["qvel"][:, 0].contiguous()
    return ctx


def tracking_action(env):
    """The kinematic action a converged policy emits on the standing clip: next pose = the clip's pose (step_ar's encoding:
    root height, de-headed root quaternion, 69 joint angles, zero root velocities).  Used by --workload tracked only."""
    q0 = env.ctx["init_qpos"]
    obs0 = env.reset().clone()                                   # obs_ar[0:74] = qpos[2:] with the root quaternion de-headed
    a = torch.zeros((env.n, 80), device=env.device)
    a[:, :74] = torch.cat([q0[:, 2:3], obs0[:, 1:5], q0[:, 7:]], 1)
    return a


def rollout_steps(sampler, k, a_track=None, wild=False, follow_clip=False):
    """k batched env-steps without keeping the experience (identical work to VectorSampler.sample's loop body).
    wild: the eval_ar_policy.py --wild loop (:196-215): mean actions, an env that terminates early is put back on the kinematic
    roll-out (ar_fail_safe) and keeps going, an env that finishes its clip starts it again."""
    env, pol = sampler.env, sampler.policy
    n_early = torch.zeros((), dtype=torch.int64, device=env.device)
    env.done_count.zero_()
    with torch.no_grad():
        # exploration noise of the whole call in one launch: 80 kinematic + 75 UHC (+ 80 for the stand-in action's noise)
        noise = None if wild else torch.randn((k, env.n, 235), device=env.device, generator=env.gen)
        a_noisy = None
        if a_track is not None and not wild and not follow_clip:        # the stand-in actions of the whole call in one launch, like the noise
            a_noisy = torch.add(a_track.unsqueeze(0), noise[:, :, 155:], alpha=0.04)
        if follow_clip and getattr(env, "_clip_action", None) is None:
            # step_ar's encoding of every clip frame (root height, [root quaternion: the env's own], 69 joint angles, zero velocities), one table per context
            q = env.ctx["qpos"]
            env._clip_action = torch.cat([q[..., 2:3], torch.zeros_like(q[..., :4]), q[..., 7:], torch.zeros_like(q[..., :6])], -1).contiguous()
        for t in range(k):
            action, sampler.hx = pol.select_action(sampler.obs, sampler.hx, wild, env.gen, None if wild else noise[t, :, :80])
            if follow_clip:                                      # moving clips (objects workload): the action that reproduces the clip's NEXT pose
                a_track = env._clip_action[env.row, torch.clamp(env.cur_t + 1, max=env._clip_action.shape[1] - 1)]
                a_track[:, 1:5] = sampler.obs[:, 1:5]
            if a_track is not None:                              # the policy's GEMMs ran; a trained policy's output stands in for theirs
                action = a_track if wild else (a_noisy[t] if a_noisy is not None else torch.add(a_track, noise[t, :, 155:], alpha=0.04))
            obs, _, done, info = env.step(action.contiguous(), need_obs=False, cc_noise=None if wild else noise[t, :, 80:155])
            if wild:
                early = done & (info["percent"] != 1)
                env.ar_fail_safe(early)                          # masked, device side
                done = done & ~early
                n_early += early.sum()
            sampler.obs = env.reset(done, policy_state=sampler.hx)        # finished envs: state, target, observation and the GRU state in one pass
    return n_early if wild else env.done_count.to(torch.int64)[0]       # episodes ended: counted inside kp_sim_post_step


def parity_summary(workload):
    """One-substep parity of THIS workload's own states against the fp64 oracle, as committed under profiles/ by tools/substep_parity.py
    (both sides restarted from a common fp32-rounded state at every substep): not measured in this run, read from the log."""
    import re
    path = os.path.join(PROFILE_DIR, "substep_parity_bench.log")
    if not os.path.exists(path):
        return None
    lines = open(path).read().splitlines()
    from kinpoly_amd.build import kernel_source_sha256
    stamp = [ln.split()[1] for ln in lines if ln.startswith("kernel_source_sha256 ") and len(ln.split()) > 1]
    if not stamp or stamp[0] != kernel_source_sha256():          # a log of another kernel does not ride on this record (parity_live is this run's own)
        return {"stale": True, "source": os.path.relpath(path, ROOT), "kernel_source_sha256_of_the_log": stamp[0] if stamp else None}
    for i, ln in enumerate(lines):
        if ln.startswith(f"bench:{workload}:") and i + 1 < len(lines):
            m = re.search(r"differ between the two sides at the same state: (\d+) of (\d+)", lines[i + 1])
            v = re.search(r"another vertex of a hull at the same height \(to 1e-7\): (\d+)", lines[i + 1])
            q = re.search(r"with the same contact points: max \|dqpos\| ([0-9.e+-]+)", lines[i + 1])
            med = re.search(r"one-substep \|dqpos\| median ([0-9.e+-]+) p99 ([0-9.e+-]+)", ln)
            if m and q:
                return {"substeps": int(m.group(2)), "contact_set_diffs": int(m.group(1)), "same_entities_other_hull_vertex": int(v.group(1)) if v else None,
                        "max_same_set_dqpos": float(q.group(1)), "median_dqpos": float(med.group(1)) if med else None, "p99_dqpos": float(med.group(2)) if med else None,
                        "note": "HIP kernel vs fp64 oracle, ONE substep from a common state, on states of this workload; contact-set differences sit on the knife edges of "
                                "MuJoCo's own contact rules (dist == margin, level hull vertices), where per-step agreement to 1e-3 rad does not hold on either side",
                        "source": os.path.relpath(path, ROOT)}
    return None


def parity_live(env, sampler, workload, a_track, n=256):
    """One-substep parity measured IN THIS RUN on the engine that was just timed: one more env-step is taken (its UHC action and target recorded), `n`
    envs' states after it are handed to tools/substep_parity.run, which restarts the HIP kernel and the fp64 oracle from a common fp32-rounded state at
    each of the 15 substeps of a control step and, where their contact sets differ (a knife edge of MuJoCo's contact rules), follows both sides to the
    end of the control step.  The oracle is the checker here, as in cpu_baseline; nothing of it is on the timed path."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import substep_parity
    keep, orig = {}, env.step

    def recording_step(*a, **k):
        out = orig(*a, **k); keep["info"] = out[3]; return out
    env.step = recording_step
    try:
        rollout_steps(sampler, 1, a_track, False, workload == "objects")
    finally:
        env.step = orig
    t0 = time.perf_counter()
    S = substep_parity.states_from_engine(env, keep["info"]["cc_action"], n, workload == "objects")
    R = substep_parity.run("bench:" + workload, states=S, nsub=15)
    out = substep_parity.summary(R)
    out.update(envs=n, seconds=time.perf_counter() - t0,
               note="measured in this run: HIP kernel vs fp64 oracle, one substep at a time from common states of this engine after the timed region; flip_* = both sides "
                    "followed free-running for the rest of the control step from a substep whose contact sets differed")
    return out


def episode_parity_live(envs=64, steps=99, workers=None):
    """Whole-episode, outcome-level parity measured IN THIS RUN (tools/episode_parity.py): `envs` x `steps` control steps of the configs[2] rollout through
    VectorSampler and through oracle/episode.py (fp64 CPU restatement of sample_worker, spawned single-threaded workers) with the same clips, weights and
    exploration noise.  The oracle is the checker; this runs after the timed regions, on an engine of its own."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import episode_parity
    t0 = time.perf_counter()
    r = episode_parity.run(n=envs, T=steps, seed=7, objects=False, workers=workers or min(32, os.cpu_count() or 1))
    keep = ("envs", "steps", "policy", "first_termination_step_equal_frac", "done_flags_equal_frac_of_rows", "episodes_ended", "failures", "failures_per_env_equal_frac", "mean_reward", "dqpos_aligned_rows", "bad_envs")
    out = {k: r[k] for k in keep}
    out["dqpos_p50_vs_control_step"] = {k: v["p50"] for k, v in r["dqpos_vs_control_step"].items()}
    out["seconds"] = time.perf_counter() - t0
    out["note"] = ("measured in this run: HIP sampler vs the fp64 CPU episode loop on the same episodes; |dqpos| over the rows whose done-flag history is identical on both sides "
                   "(rows above 1e-3 rad follow a contact knife-edge flip in a falling humanoid)")
    return out


def mujoco_pin_report():
    """null until a MuJoCo binding is importable on the box; then the live pin of tests/mj_pin.py (model arrays, free fall, contact workload)"""
    try:
        sys.path.insert(1, os.path.join(ROOT, "tests"))
        import mj_pin as mujoco_pin
        if mujoco_pin.find_mujoco() is None:
            return None
        return mujoco_pin.pin_report(n_free_fall=300, n_contact=30)
    except Exception as ex:
        return {"error": f"{type(ex).__name__}: {ex}"[:200]}


def policy_gemm_probe(env, policy, iters=30):
    """The MFMA side of an env-step in isolation: kinematic policy (GRU cell + MLP 1129-1024-512-256-80) and PolicyMCP (8 primitives
    784-512-256-75 + composer 784-300-200-8) forward on the bench's batch, fp32, timed with events on torch's stream.  FLOPs = 2 x MACs."""
    n = env.n
    obs = torch.randn((n, 105), device=env.device); hx = torch.randn((n, 1024), device=env.device); cc = torch.randn((n, 784), device=env.device)
    macs = (105 * 3072 + 1024 * 3072) + (1129 * 1024 + 1024 * 512 + 512 * 256 + 256 * 80) + 8 * (784 * 512 + 512 * 256 + 256 * 75) + (784 * 300 + 300 * 200 + 200 * 8)
    with torch.no_grad():
        for _ in range(5):
            policy.select_action(obs, hx, True); env.cc_policy.select_action(cc, True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            policy.select_action(obs, hx, True); env.cc_policy.select_action(cc, True)
        e1.record(); e1.synchronize()
    ms = e0.elapsed_time(e1) / iters
    tf = 2.0 * macs * n / (ms * 1e-3) / 1e12
    return {"gemm_flops_per_env_step": 2 * macs, "ms_per_step_isolated": ms, "achieved_tflops": tf, "peak_tflops": MFMA_FP32_PEAK_TFLOPS, "frac": tf / MFMA_FP32_PEAK_TFLOPS,
            "note": "policy GEMMs + their elementwise epilogues through hipBLASLt / rocBLAS (MFMA, fp32 in / fp32 accumulate), timed outside the env-step; "
                    "inside the step they overlap nothing (one stream), so ms_per_step_isolated / ms_per_step is their share of the step"}


def cpu_baseline(std, seconds_budget=15.0, tracked=True):
    """The fp64 oracle port of the same env-step on ONE host core: C physics (oracle/kp_oracle.c) + numpy
    obs / FK / reward (oracle/np_oracle.py) + fp64 torch policies on 1 thread (the reference samples on CPU,
    OMP_NUM_THREADS=1, agent_ar.py:29,654).  tracked: the metric's workload -- the kinematic policy runs, its output is replaced by the clip
    pose + N(0, 0.04) noise (episodes last); False: the random-init rollout."""
    from kinpoly_amd.model_compiler import DEFAULT_KPM, read_kpm
    from kinpoly_amd.nets import KinPolicy, PolicyMCP
    from oracle import np_oracle as O
    from oracle import kpo
    from oracle.kpo import OracleSim
    torch.set_num_threads(1)
    kpo.flops_reset()
    kpm = read_kpm(DEFAULT_KPM)
    bp, bi, par = kpm["body_pos"].reshape(24, 3), kpm["body_ipos"].reshape(24, 3), kpm["body_parent"]
    torch.manual_seed(0)
    mcp, kin = PolicyMCP().double(), KinPolicy().double()
    sim = OracleSim()
    qpos0, qvel0 = std["qpos"], std["qvel"]
    fk0 = O.qpos_fk(qpos0, bp, bi, par)
    head = np.concatenate([fk0["wbpos"][13], fk0["wbquat"][13]])
    obj_rel = np.concatenate([O.transform_vec(-head[:3], head[3:], "heading"), O.quaternion_multiply(O.quaternion_inverse(O.get_heading_q(head[3:])), [1, 0, 0, 0])])
    one_hot, hv = np.zeros(4), np.zeros(6)
    gt_bquat = fk0["bquat"].reshape(-1)
    rng = np.random.default_rng(os.getpid())
    n_steps, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds_budget:
        sim.reset(qpos0, qvel0)
        hx = torch.zeros(1, 1024, dtype=torch.float64)
        x = {k: sim.get(k) for k in ("qpos", "qvel", "xpos", "xquat", "xipos")}
        obs = O.obs_ar(x["qpos"], x["xpos"].reshape(24, 3), x["xquat"].reshape(24, 4), head, hv, obj_rel, one_hot, None)
        obs0 = obs.copy()
        for t in range(CLIP_LEN - 1):
            with torch.no_grad():
                a, hx = kin.select_action(torch.from_numpy(obs)[None], hx)
            a = a[0].numpy()
            if tracked:                                                  # tracking_action(): root height, de-headed root quaternion, joint angles of the clip
                a = np.concatenate([qpos0[2:3], obs0[1:5], qpos0[7:], np.zeros(6)]) + 0.04 * rng.standard_normal(80)
            prev_bquat = O.get_body_quat(x["qpos"]); prev_hpos = np.concatenate([x["xpos"].reshape(24, 3)[13], x["xquat"].reshape(24, 4)[13]])
            tgt = O.qpos_fk(O.step_ar(x["qpos"], a), bp, bi, par)
            cc_obs = O.zfilter(O.obs_cc(x["qpos"], x["qvel"], x["xpos"].reshape(24, 3), x["xquat"].reshape(24, 4), x["xipos"].reshape(24, 3), tgt), 0.0, 1.0, 5.0)
            with torch.no_grad():
                cc_a = mcp.select_action(torch.from_numpy(cc_obs)[None])[0].numpy()
            sim.do_simulation(cc_a, tgt["qpos"], 15)
            x = {k: sim.get(k) for k in ("qpos", "qvel", "xpos", "xquat", "xipos")}
            xp, xq = x["xpos"].reshape(24, 3), x["xquat"].reshape(24, 4)
            O.dynamic_supervision_v1(np.concatenate([xp[13], xq[13]]), prev_hpos, O.get_body_quat(x["qpos"]), prev_bquat, xp, tgt, head, gt_bquat, gt_bquat, 1 / 30, O.REWARD_WEIGHTS)
            fail = O.calc_body_diff(xp, tgt["wbpos"], kpm["body_diffw"]) > 10 or O.calc_body_diff(xp, fk0["wbpos"], kpm["body_diffw"]) > 12
            obs = O.obs_ar(x["qpos"], xp, xq, head, hv, obj_rel, one_hot, None)
            n_steps += 1
            if fail or time.perf_counter() - t0 > seconds_budget:
                break
    dt = time.perf_counter() - t0
    return {"value": n_steps / dt, "unit": "env-steps/s", "cores": 1, "kind": "port", "physics_flops_per_env_step": kpo.flops() / max(n_steps, 1),
            "sample": f"{n_steps} env-steps of the same standing-clip rollout ({'tracked' if tracked else 'random-init'} workload; fp64 C physics oracle + numpy obs/reward + fp64 torch policies, 1 thread) in {dt:.1f} s; host has {os.cpu_count()} cores"}


def cpu_baseline_workers(std, workers, seconds_budget=15.0):
    """The same oracle roll-out in `workers` single-threaded processes at once -- the shape of the reference's sampler
    (`--num_threads 35`: one env per forked worker, OMP_NUM_THREADS=1, agent_ar.py:29,651-680).  Each worker is a fresh
    interpreter (a HIP context does not survive fork) that runs cpu_baseline() and prints its step count."""
    import subprocess
    env = dict(os.environ, OMP_NUM_THREADS="1", MKL_NUM_THREADS="1", HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", str(seconds_budget)]
    t0 = time.perf_counter()
    procs = [subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for _ in range(workers)]
    steps, wall, flops = 0, 0.0, 0.0
    for p in procs:
        try:
            out, _ = p.communicate(timeout=seconds_budget * 4 + 120)
            rec = json.loads(out.strip().splitlines()[-1])
            steps += rec["steps"]; wall = max(wall, rec["seconds"]); flops += rec.get("flops", 0.0)
        except Exception:
            p.kill()
            raise
    return {"value": steps / wall, "unit": "env-steps/s", "cores": workers, "kind": "port", "physics_flops_per_env_step": flops / max(steps, 1),
            "sample": f"{steps} env-steps of the same standing-clip rollout (the metric's tracked workload) in {workers} single-threaded worker processes (fp64 C physics oracle + numpy obs/reward "
                      f"+ fp64 torch policies each; the reference samples with 35 such workers) over {wall:.1f} s of rollout ({time.perf_counter() - t0:.1f} s with start-up); "
                      f"host has {os.cpu_count()} cores"}


WORKLOAD_DESC = {
    "tracked": "BASELINE configs[2] rollout: kin_poly.yml dynamics-regulated env-step (kin GRU policy, step_ar, target FK, UHC obs+ZFilter+PolicyMCP, 15 substeps "
               "SPD+RFC+contact, term/reward, AR obs, auto-reset), standing MoCap clip, seeded networks; episodes that last: the kinematic policy's GEMMs "
               "run, its output is replaced by the clip pose + N(0, e^-3.2) noise (what a pretrained kinematic policy emits)",
    "random_init": "BASELINE configs[2] rollout, same env-step, seeded random-init networks: every env terminates and is reset on (almost) every step",
    "wild_eval": "BASELINE configs[4]: --wild eval_ar_policy path (humanoid_smpl_neutral_mesh_all.xml, mode test: mean actions of both policies, no GT "
                 "termination, fail-safe on), batched inference, standing clip",
    "objects": "BASELINE configs[3] on one GPU: same env-step with the scene's free objects simulated (sit: chair, push: box on table, avoid: Can, step: step box; "
               "SURVEY 8(d) config 4 synthetic takes, four action classes evenly over the envs), kinematic policy's output replaced by the clip's next pose + noise",
    "trained": "BASELINE configs[2] rollout with NOTHING overridden: the kinematic policy and the UHC of --policy-ckpt / --cc-ckpt (trained by this engine: tools/learning_demo.sh), "
               "the policy's own sampled output drives env.step, episodes end when the policy fails or the clip ends",
    "train_iter": "one AgentAR.optimize_policy per step: sample 4096 x 24 env-steps of the configs[2] rollout (random-init TrajARNet; every episode on a clip drawn from a "
                  "StateARDataset of synthetic takes and run through init_context, as scripts/train_ar_policy.py does), GAE, all-gather of "
                  "advantages / returns, 10 PPO epochs + 20 supervised step updates, gradients all-reduced (kin_poly.yml:36-71)",
}
ROLLOUT_WORKLOADS = ("tracked", "random_init", "wild_eval", "objects")       # the default line's workloads (`trained` runs when its checkpoint is given)
TRAIN_KEYS = ("T_sample", "T_update", "samples_per_s_per_gpu", "samples_per_iter_per_gpu", "iters", "avg_reward", "fail_rate", "episodes_per_iter",
              "clips_drawn_per_iter", "clips_through_init_context_per_iter", "pool_exhausted", "update_tflops", "update_mfma_frac", "update_flops_per_iter", "episode_source")


def run_workload(workload, device_index, seed, threads, steps, warmup, barrier=None, repeats=1):
    """Build the engine for `workload`, run `warmup` untimed + `steps` timed env-steps.  Returns (record, env, policy, sampler, std)."""
    env, policy, sampler, std = build_engine(device_index, seed, threads, workload)
    a_track = None
    if workload in ("tracked", "wild_eval"):
        a_track = tracking_action(env)
        env._a_track = a_track if workload == "tracked" else None
        sampler.start()
    wild, follow = workload == "wild_eval", workload == "objects"
    if workload in ("tracked", "objects", "trained"):
        stagger_episodes(env, sampler, seed, follow)
    rollout_steps(sampler, warmup, a_track, wild, follow)
    # `repeats` timed blocks of exactly `steps` env-steps each, every block bracketed by barrier + synchronize on both sides (the contract's timed region,
    # repeated inside one command so that the line carries its own spread).  The caller reports the median block.
    el_blocks, kern_blocks, done_blocks, n_launch = [], [], [], 0
    for _ in range(max(1, repeats)):
        env.sim.timing_reset()
        if barrier is not None:
            barrier()
        else:
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        n_done = rollout_steps(sampler, steps, a_track, wild, follow)
        if barrier is not None:
            barrier()
        else:
            torch.cuda.synchronize()
        el_blocks.append(time.perf_counter() - t0)
        kern_s, n_launch = env.sim.timing_mean_seconds()
        kern_blocks.append(kern_s); done_blocks.append(float(n_done.item()))
    diag = env.sim.diag()
    mid = int(np.argsort(el_blocks)[len(el_blocks) // 2])
    rec = {"elapsed": el_blocks[mid], "kern_s": kern_blocks[mid], "n_launch": n_launch, "diag": diag, "n_done": done_blocks[mid],
           "elapsed_blocks": el_blocks, "kern_s_blocks": kern_blocks, "n_done_blocks": done_blocks}
    return rec, env, policy, sampler, std


def stagger_episodes(env, sampler, seed, follow):
    """Episodes at every stage of their clip, as in a running job: cur_t staggered uniformly, so that ~1 / (CLIP_LEN - 1) of the envs end their
    clip on every step and the device-side reset is part of every timed step.  The standing clip is the same at every
    t; an env of the objects workload is put on its clip's pose at that frame."""
    g = torch.Generator().manual_seed(seed + 1000)
    env.cur_t.copy_(torch.randint(0, CLIP_LEN - 1, (env.n,), generator=g).to(env.device, torch.int32))
    if follow:
        q = env.ctx["qpos"][env.row.long(), env.cur_t.long()].contiguous()
        env.sim.set_state(q, env.ctx["init_qvel"][env.row.long()].contiguous())
        env.sim.set_target(q)
    sampler.obs = env.sim.obs_ar(env._ctx_struct, env._obs).clone()


def object_scene_launches(device_index, threads):
    """Physics kernel alone on the two contact-heavy object scenes of tools/obj_bench.py (4096 envs, kp_step_queue_kernel<true>): standing on the
    free step box and the push scene with the table within reach -- the launch durations VERDICT r2 asks to see on the driver's record."""
    from kinpoly_amd.sim import STEP_KPM, KpModel, KpSim
    std = np.load(os.path.join(ROOT, "tests", "golden", "standing_neutral.npz"))
    n, rng = ENVS_PER_GPU, np.random.default_rng(3)
    x0, y0 = std["qpos"][0], std["qpos"][1]
    out = {}
    for name, active, lift in (("standing_on_step_box", {4: [x0, y0, 0.3705, 1, 0, 0, 0]}, 0.341),
                               ("push_table_within_reach", {1: [x0 + 0.75, y0, 0.921, 1, 0, 0, 0], 2: [x0 + 0.75, y0, 0.7905, 1, 0, 0, 0]}, 0.0),
                               ("step_box_untouched", {4: [x0 + 2.0, y0, 0.3705, 1, 0, 0, 0]}, 0.0)):
        blk = np.zeros((n, 35))
        for i in range(5):
            blk[:, 7 * i: 7 * i + 3] = [(i + 1) * 100, 100, 0]
        for oi, pose in active.items():
            blk[:, 7 * oi: 7 * oi + 7] = pose
        sim = KpSim(KpModel(STEP_KPM, threads_per_env=threads), n, device_index)
        qpos = np.tile(std["qpos"], (n, 1)); qpos[:, 2] += lift; qpos[:, 7:] += rng.normal(size=(n, 69)) * 0.05
        dev = lambda a: torch.tensor(a, dtype=torch.float32, device=sim.device)      # noqa: E731
        q = dev(qpos)
        sim.set_objects(dev(blk)); sim.set_state(q, dev(rng.normal(size=(n, 75)) * 0.2)); sim.set_target(q.clone())
        a = dev(rng.normal(size=(n, 75)) * 0.1)
        ts = []
        for _ in range(8):
            sim.step_ctrl(a, 15)
            ts.append(sim.last_step_seconds())
        dg = sim.diag()
        out[name] = {"launch_ms": float(np.mean(ts[2:]) * 1e3), "contacts_mean": float(dg[:, 0].mean()), "newton_iters_per_substep": float(dg[:, 1].mean() / 15.0),
                     "bad_envs": int(((dg[:, 2] & 255) != 0).sum())}
        del sim
        torch.cuda.empty_cache()
    return out


def sampler_regime(device_index, seed, horizon=TRAIN_HORIZON, calls=4, warm=2, threads=64, amp=0.05, profile=False, pool_depth=4, lagged=True):
    """`VectorSampler.sample(horizon)` in the regime of a policy that has learnt to track: every episode on a clip drawn from a
    StateARDataset through init_context (ring of pool_depth + 1 rows per env, on-demand top-up), all twelve memory fields' worth of per-step records,
    and a kinematic policy whose GEMMs run but whose output is replaced by the clip's next pose + N(0, e^-3.2) noise -- so episodes last, a few per cent
    of the envs end per step (clip ends + failures), and what is timed is the sampler around the env-step, not init_context at fail rate 1.
    Returns T_sample per call, the bare env-steps' time for the same number of steps (rollout_steps on the same engine), and the done / fail rates."""
    from kinpoly_amd import dataset as D
    from kinpoly_amd import sim as kpsim
    from kinpoly_amd.context import PolicyARContext, TrajARNet
    from kinpoly_amd.env import BatchedHumanoidAREnv
    from kinpoly_amd.model_compiler import read_kpm
    from kinpoly_amd.nets import enable_tuned_gemms
    from kinpoly_amd.rollout import EpisodeSource, VectorSampler
    enable_tuned_gemms()
    torch.manual_seed(seed)
    std = np.load(os.path.join(ROOT, "tests", "golden", "standing_neutral.npz"))
    env = BatchedHumanoidAREnv(ENVS_PER_GPU, device_index, mode="train", seed=seed, model_options={"threads_per_env": threads})
    kin_sim = kpsim.KpSim(env.model, ENVS_PER_GPU, device_index)
    takes = D.synthetic_takes(kin_sim, std["qpos"], n_per_action=8, T_range=(CLIP_LEN + 10, CLIP_LEN + 60), body_mass=read_kpm(kpsim.STEP_KPM)["body_mass"],
                              seed=seed, with_objects=False, amp_max=amp)
    ds = D.StateARDataset(takes, fr_num=CLIP_LEN, seed=seed, device=env.device)

    class Tracking(TrajARNet):
        """the policy's own GEMMs, then the action a converged policy would emit: the clip's next frame in step_ar's encoding + exploration noise"""

        def select_action(self, state, hx, mean_action=False, generator=None, noise=None):
            _, hx = self.get_action(state, hx)
            row = env.row.long()
            q = env.ctx["qpos"][row, torch.minimum(env.cur_t.long() + 1, env.row_len[row].long())]
            a = torch.cat([q[:, 2:3], state[:, 1:5], q[:, 7:], torch.zeros((q.shape[0], 6), device=q.device)], 1)
            if noise is None:
                noise = torch.randn(a.shape, device=a.device, generator=generator)
            return torch.addcmul(a, self.std(), noise), hx
    pol = Tracking().to(env.device)
    builder = PolicyARContext(pol, kin_sim, smooth=True, need_rollout=False, keep_context_feat=False)
    # init_qpos / init_qvel of an episode = the clip's own first frame (what a trained context network predicts), through the real init_context call
    src = EpisodeSource(dataset=ds, ctx_builder=builder, sampling_temp=0.3, sampling_freq=0.5)
    orig_draw = src.draw

    def draw(n, device):
        data = orig_draw(n, device)
        data["init_qpos"], data["init_qvel"] = data["qpos"][:, 0].contiguous(), data["qvel"][:, 0].contiguous()
        return data
    src.draw = draw
    sampler = VectorSampler(env, pol, record_qpos=True, source=src, pool_depth=pool_depth, lagged=lagged)
    sampler.start()
    for _ in range(warm):
        sampler.sample(horizon)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    done = fail = 0.0
    prof = None
    drawn0, tops0 = src.n_drawn, sampler.top_ups
    if profile:
        prof = torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA])
        prof.__enter__()
    for _ in range(calls):
        b = sampler.sample(horizon)
        done += float((1 - b.masks).mean()); fail += float(b.fails.float().mean())
    torch.cuda.synchronize()
    ts = (time.perf_counter() - t0) / calls
    if prof is not None:
        prof.__exit__(None, None, None)
    # the same number of bare env-steps on the same engine and states (no records, no ring, no top-up)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(calls):
        rollout_steps(sampler, horizon, None, False, False)
    torch.cuda.synchronize()
    tb = (time.perf_counter() - t0) / calls
    out = {"T_sample": ts, "T_bare_env_steps": tb, "overhead": ts / tb, "horizon": horizon, "envs": ENVS_PER_GPU, "calls": calls, "done_per_step_frac": done / calls, "fail_per_step_frac": fail / calls,
           "top_ups_per_call": (sampler.top_ups - tops0) / calls, "pool_depth": pool_depth, "lagged_ring_read": lagged, "ring_rows_per_env": sampler.n_slots, "clips_drawn_per_call": (src.n_drawn - drawn0) / calls, "pool_exhausted": sampler.pool_exhausted,
           "env_steps_per_s": ENVS_PER_GPU * horizon / ts,
           "note": "VectorSampler.sample with a tracking stand-in policy (its GEMMs run; output = clip's next pose + noise): dataset-driven episodes through init_context, ring top-ups, "
                   "per-step records of all memory fields; T_bare_env_steps = rollout_steps on the same engine (random-init policy output, episodes end more often: an upper bound of the bare cost)"}
    if prof is not None:
        out["_profile_table"] = prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=45, max_name_column_width=70)
    del sampler, env, kin_sim
    torch.cuda.empty_cache()
    return out


def train_iteration(device_index, seed, horizon, iters, warm, barrier=None, threads=64, objects=False, rank=0, cache_init_context=False, model_opts=None):
    """`iters` timed AgentAR.optimize_policy calls (after `warm` untimed) at ENVS_PER_GPU x horizon env-steps per rank, the way
    scripts/train_ar_policy.py runs them: every episode draws its clip from a StateARDataset (adaptive take sampling, freq_dict feedback) and goes
    through init_context (context GRU over the 100-frame clip -> init_qpos / init_qvel) -- inside the timed region, at whatever failure rate the
    random-init networks produce.  objects=False: BASELINE configs[2] (object-free takes); True: configs[3]'s four action classes with free objects."""
    from kinpoly_amd.agent import AgentAR
    from kinpoly_amd import dataset as D
    from kinpoly_amd import sim as kpsim
    from kinpoly_amd.model_compiler import read_kpm
    std = np.load(os.path.join(ROOT, "tests", "golden", "standing_neutral.npz"))
    fk_sim = kpsim.KpSim(kpsim.KpModel(kpsim.STEP_KPM), ENVS_PER_GPU, device_index)
    # one data set for the whole job (take seed independent of the rank), one draw stream per rank
    takes = D.synthetic_takes(fk_sim, std["qpos"], n_per_action=8, T_range=(CLIP_LEN + 10, CLIP_LEN + 60), body_mass=read_kpm(kpsim.STEP_KPM)["body_mass"],
                              seed=seed, with_objects=objects)
    ds = D.StateARDataset(takes, fr_num=CLIP_LEN, seed=seed + rank, device=fk_sim.device)
    agent = AgentAR(ENVS_PER_GPU, dataset=ds, device=device_index, horizon=horizon, seed=seed, use_init_context=True, pool_depth=4,
                    model_options={"threads_per_env": threads, **(model_opts or {})}, sampling_temp=0.3, sampling_freq=0.5, cache_init_context=cache_init_context)
    for i in range(warm):
        agent.optimize_policy(i)
    (barrier or torch.cuda.synchronize)()
    t0 = time.perf_counter()
    ts = tu = 0.0
    info = {}
    drawn0, hits0, eps, fails = agent.source.n_drawn, agent.source.n_memo_hits, 0, 0.0
    for i in range(iters):
        info = agent.optimize_policy(warm + i)
        ts += info["T_sample"]; tu += info["T_update"]; eps += info["episodes"]; fails += info["fail_rate"]
    (barrier or torch.cuda.synchronize)()
    el = time.perf_counter() - t0
    n_s = ENVS_PER_GPU * horizon
    # FLOPs of the update, counted from its GEMM shapes (2 x MACs; backward = 2 x forward): the policy = GRU re-unroll (105 -> 3 x 1024 input gates,
    # 1024 -> 3 x 1024 recurrent gates) + MLP 1129-1024-512-256-80 per sample, the value net 105-512-256-1
    pol_macs = 105 * 3072 + 1024 * 3072 + 1129 * 1024 + 1024 * 512 + 512 * 256 + 256 * 80
    val_macs = 105 * 512 + 512 * 256 + 256
    ne, ns = agent.trainer.num_optim_epoch, agent.num_step_update
    # policy: (forward + backward = 3 forward-equivalents) x (PPO epochs + supervised steps); fixed_log_probs is epoch 0's own forward
    # (PPOTrainer.update), so no extra pass.  value: one forward for GAE + 3 x the regression steps
    upd_flops = 2.0 * n_s * (pol_macs * 3 * (ne + ns) + val_macs * (1 + 3 * ne * agent.trainer.value_opt_niter))
    rec = {"elapsed": el, "T_sample": ts / iters, "T_update": tu / iters, "samples_per_iter_per_gpu": n_s, "horizon": horizon,
           "iters": iters, "warmup": warm, "samples_per_s_per_gpu": n_s * iters / el, "avg_reward": info.get("avg_reward"),
           "pool_exhausted": info.get("pool_exhausted"), "episodes_per_iter": eps / iters,
           "clips_through_init_context_per_iter": ((agent.source.n_drawn - drawn0) - (agent.source.n_memo_hits - hits0)) / iters, "clips_drawn_per_iter": (agent.source.n_drawn - drawn0) / iters,
           "fail_rate": fails / iters, "episode_source": f"StateARDataset of {ds.get_len()} synthetic takes ({'four action classes with free objects' if objects else 'no objects'}), "
                                                         "adaptive take sampling + init_context per episode, pool_depth 4",
           "update_tflops": upd_flops / (tu / iters) / 1e12, "update_mfma_frac": upd_flops / (tu / iters) / 1e12 / MFMA_FP32_PEAK_TFLOPS,
           "update_flops_per_iter": upd_flops}
    del agent, fk_sim
    torch.cuda.empty_cache()
    return rec


def relaunch_under_torchrun(n_gpus):
    """`python bench.py --gpus N` without a launcher (the form the driver's scaling run may use): start N ranks of this same command under
    torch.distributed.run on this node (the reference forks its own workers too, agent_ar.py:651-663) and hand its exit code back."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__), *sys.argv[1:]]
    env = dict(os.environ, KP_BENCH_CHILD="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main():
    global ENVS_PER_GPU                 # the envs_per_gpu_8192 block of the default run changes it for two extra rollouts
    if len(sys.argv) >= 3 and sys.argv[1] == "--cpu-baseline-worker":
        std = np.load(os.path.join(ROOT, "tests", "golden", "standing_neutral.npz"))
        r = cpu_baseline(std, float(sys.argv[2]))
        n = int(r["sample"].split()[0])
        print(json.dumps({"steps": n, "seconds": n / r["value"], "flops": r["physics_flops_per_env_step"] * n}), flush=True)
        return
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default 60; 3 for --workload train_iter)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed steps (default 20; 1 for --workload train_iter)")
    ap.add_argument("--threads-per-env", type=int, default=int(os.environ.get("KP_THREADS_PER_ENV", "64")))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary workloads and the GEMM probe (profiling runs)")
    ap.add_argument("--workload", choices=tuple(WORKLOAD_DESC), default="tracked")
    ap.add_argument("--repeats", type=int, default=5, help="timed blocks of --steps env-steps inside this command; value / ms_per_step = the median block (rollout workloads)")
    ap.add_argument("--policy-ckpt", default=None, help="kinematic policy checkpoint (reference layout) of --workload trained")
    ap.add_argument("--cc-ckpt", default=None, help="UHC checkpoint (reference layout: weights + ZFilter) of --workload trained")
    ap.add_argument("--no-parity-live", action="store_true", help="skip the live one-substep parity sample against the fp64 oracle (profiling runs)")
    args = ap.parse_args()
    global POLICY_CKPT, CC_CKPT
    POLICY_CKPT, CC_CKPT = args.policy_ckpt, args.cc_ckpt
    train = args.workload == "train_iter"
    if args.steps is None:
        args.steps = 3 if train else 60
    if args.warmup is None:
        args.warmup = 1 if train else 20

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    if args.gpus > torch.cuda.device_count() and os.environ.get("KP_BENCH_SHARED_DEVICE") != "1":
        # fail fast, before any rendezvous: ranks without a device would hang the others in init_process_group
        raise SystemExit(f"bench.py --gpus {args.gpus}: this node shows {torch.cuda.device_count()} HIP device(s); one rank per GPU is the only layout "
                         "(KP_BENCH_SHARED_DEVICE=1 puts all ranks on device 0 for plumbing tests)")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:        # no launcher around us: become one
        raise SystemExit(relaunch_under_torchrun(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    shared = os.environ.get("KP_BENCH_SHARED_DEVICE") == "1"     # plumbing test on a 1-GPU box: all ranks on device 0, gloo barrier
    if shared:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    force_pg = os.environ.get("KP_BENCH_FORCE_PG") == "1"        # a 1-rank run that still creates the nccl group (RCCL smoke on a 1-GPU box)
    if world > 1 or force_pg:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if shared:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    in_group = dist.is_initialized()
    cdev = "cpu" if shared else torch.device("cuda", local_rank)

    def barrier():
        if in_group:
            dist.barrier()
        torch.cuda.synchronize()

    # every rank that takes part adds 1 through the job's collective backend (RCCL all-reduce of a device tensor under nccl)
    ranks_seen = 1
    if in_group:
        one = torch.ones(1, device=cdev, dtype=torch.float32)
        dist.all_reduce(one)
        ranks_seen = int(one.item())

    # which physical device every rank runs on: a scaling record must show N DISTINCT GPUs (uuid / PCI bus id from the runtime's device properties)
    pr = torch.cuda.get_device_properties(local_rank)
    me = {"rank": rank, "local_rank": local_rank, "name": pr.name, "uuid": str(getattr(pr, "uuid", "")), "pci_bus_id": getattr(pr, "pci_bus_id", None), "pci_device_id": getattr(pr, "pci_device_id", None)}
    devices = [me]
    if in_group:
        devices = [None] * world
        dist.all_gather_object(devices, me)
    distinct = len({(d["uuid"], d["pci_bus_id"], d["pci_device_id"]) for d in devices})

    if train:
        rec = train_iteration(local_rank, 4, TRAIN_HORIZON, args.steps, args.warmup, barrier, args.threads_per_env, rank=rank)
        env = policy = sampler = None
        std = np.load(os.path.join(ROOT, "tests", "golden", "standing_neutral.npz"))
    else:
        rec, env, policy, sampler, std = run_workload(args.workload, local_rank, 4 + rank, args.threads_per_env, args.steps, args.warmup, barrier, args.repeats)
    # per timed block: max over ranks; the line reports the MEDIAN block (its ms_per_step x steps is that block's bracketed wall time) and the spread
    blocks = list(rec.get("elapsed_blocks", [rec["elapsed"]]))
    per_rank_ms = [float(np.median(blocks)) / args.steps * 1e3]
    if in_group:
        t = torch.tensor(blocks, device=cdev, dtype=torch.float64)
        every = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(every, t)
        per_rank_ms = [float(x.median().item()) / args.steps * 1e3 for x in every]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        blocks = [float(x) for x in t.tolist()]
    mid = int(np.argsort(blocks)[len(blocks) // 2])
    elapsed = blocks[mid]
    if "kern_s_blocks" in rec:
        rec["kern_s"], rec["n_done"] = rec["kern_s_blocks"][mid], rec["n_done_blocks"][mid]

    train_n = None
    live = None
    if not train:       # read what the line needs from the engine before it is torn down
        cost = env.sim.launch_cost().astype(np.float64)
        spj = int(env.model.get_option("substeps_per_job"))
        per_cu = min(12 if env.model.get_option("lean_queue") and args.workload != "objects" else 8, 128 // -(-int(env.model.get_option("lds_bytes_per_env_objects" if args.workload == "objects" else ("lds_bytes_per_env_lean" if env.model.get_option("lean_queue") else "lds_bytes_per_env"))) // 1280))
        slots = int(os.environ.get("KP_QUEUE_SLOTS", 0)) or torch.cuda.get_device_properties(local_rank).multi_processor_count * per_cu
        qctr = {**env.sim.queue_counters(), "envs_per_cu": per_cu, "lean_layout": bool(env.model.get_option("lean_queue")), "lds_bytes_per_env": int(env.model.get_option("lds_bytes_per_env_lean" if env.model.get_option("lean_queue") else "lds_bytes_per_env")),
                "lean_max_contacts": int(env.model.get_option("lean_max_contacts")),
                "schedule": ("5+5+5 substeps, issue priority by remaining substeps (queue_prio 3), late envs kept by their wave (queue_late)" if (env.model.get_option("lean_queue") and env.model.get_option("job_auto") and args.workload != "objects" and "KP_JOB_SCHEDULE" not in os.environ)
                             else f"substeps_per_job {spj}, job_taper {int(env.model.get_option('job_taper'))}, queue_prio {int(env.model.get_option('queue_prio'))}, queue_late {int(env.model.get_option('queue_late'))}" + (f", KP_JOB_SCHEDULE={os.environ['KP_JOB_SCHEDULE']}" if "KP_JOB_SCHEDULE" in os.environ else ""))}
        if rank == 0 and world == 1 and not args.no_parity_live and args.workload in ("tracked", "random_init", "objects"):
            try:
                live = parity_live(env, sampler, args.workload, getattr(env, "_a_track", None))
            except Exception as ex:
                live = {"error": f"{type(ex).__name__}: {ex}"[:300]}
    if world > 1 and not train and not args.no_secondary:
        # the Amdahl term of a training job on N GPUs: one whole optimize_policy (sampling + all-gather of advantages / returns + data-parallel update with
        # gradient all-reduces) timed on every rank, max over ranks
        del sampler, env, policy
        sampler = env = policy = None
        torch.cuda.empty_cache()
        r3 = train_iteration(local_rank, 4, TRAIN_HORIZON, 1, 1, barrier, args.threads_per_env, rank=rank)
        t3 = torch.tensor([r3["elapsed"], r3["T_sample"], r3["T_update"]], device=cdev, dtype=torch.float64)
        dist.all_reduce(t3, op=dist.ReduceOp.MAX)
        train_n = {**{k: r3[k] for k in TRAIN_KEYS}, "T_iteration_max_over_ranks": float(t3[0]), "T_sample_max_over_ranks": float(t3[1]), "T_update_max_over_ranks": float(t3[2]),
                   "samples_per_s_whole_job": ENVS_PER_GPU * TRAIN_HORIZON * world / float(t3[0]), "n_gpus": world}
    if rank == 0 and train:
        per_step = ENVS_PER_GPU * TRAIN_HORIZON
        out = {"metric": "env-steps/sec (whole node) at 4096 envs/GPU, 69-DoF SMPL", "value": per_step * world * args.steps / elapsed, "unit": "env-steps/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": WORKLOAD_DESC["train_iter"], "workload_id": "train_iter", "envs_per_gpu": ENVS_PER_GPU, "horizon": TRAIN_HORIZON,
                          "samples_per_step": per_step * world, "parallelism": f"env-sharded x{world}, data-parallel update"},
               "ranks_seen": ranks_seen, "devices": devices, "distinct_devices": distinct, "ms_per_step_per_rank": per_rank_ms, "collective_backend": (dist.get_backend() if in_group else None),
               "train_iteration": {k: rec[k] for k in TRAIN_KEYS}}
        print(json.dumps(out), flush=True)
    elif rank == 0:
        kern_s, n_launch, diag = rec["kern_s"], rec["n_launch"], rec["diag"]
        objects = args.workload == "objects"
        value = ENVS_PER_GPU * world * args.steps / elapsed
        algo_bytes = (ALGO_BYTES_PER_ENV_STEP_OBJ if objects else ALGO_BYTES_PER_ENV_STEP) * ENVS_PER_GPU
        achieved = algo_bytes / kern_s / 1e9
        kernel_name = ("kp_step_queue_kernel" if spj > 0 else "kp_step_kernel") + ("<true>" if objects else "<false>")
        # HBM bytes / instruction counts: ONLY from a rocprofv3 --pmc pass of this same command and workload (tools/profile_bench.sh writes
        # profiles/r03/pmc_bench_<workload>.json); otherwise null -- nothing canned from another workload enters the line
        traffic, traffic_src, valu = None, None, None
        from kinpoly_amd.build import kernel_source_sha256
        sha_now = kernel_source_sha256()
        pmc_path = os.path.join(PROFILE_DIR, f"pmc_bench_{args.workload}.json")
        if os.path.exists(pmc_path):
            pj = json.load(open(pmc_path))
            fresh = pj.get("kernel_source_sha256") == sha_now          # profiled on THIS device code (csrc/* + flags)?  otherwise the figures are nulled
            traffic = pj.get("hbm_bytes_per_launch") if fresh else None
            traffic_src = {"file": os.path.relpath(pmc_path, ROOT), "command": pj.get("command"), "launch_ms_in_those_passes": pj.get("launch_ms"),
                           "kernel_source_sha256_of_the_profile": pj.get("kernel_source_sha256"), "kernel_source_sha256_now": sha_now, "stale": not fresh,
                           "note": "separate rocprofv3 --pmc passes of this command (FETCH_SIZE x2 per the guide's gfx950 correction, WRITE_SIZE as reported), "
                                   "median per launch of the same kernel; not measured in this run; nulled when the device code has changed since (stale)"}
            valu = pj.get("issue") if fresh else None
        # instruction issue: instructions per env-step from that PMC pass, wave cycles per env-step live from this run's last
        # launch (kp_sim_launch_cost), against what two waves per SIMD can issue (tools/micro/valu_probe.hip: one instruction per 2.9 SIMD cycles)
        issue, valu_active = None, None
        if valu and valu.get("valu_insts_per_launch"):
            insts = (valu["valu_insts_per_launch"] + valu.get("salu_insts_per_launch", 0.0) + valu.get("lds_insts_per_launch", 0.0)) / ENVS_PER_GPU
            cyc = float(cost.mean())
            issue = {"wave_insts_per_env_step": insts, "wave_cycles_per_env_step": cyc, "wave_cycles_per_inst": cyc / insts, "waves_per_simd": per_cu / 4.0,
                     "simd_cycles_per_inst": cyc / insts / (per_cu / 4.0), "attainable_simd_cycles_per_inst_at_2_waves": 2.9, "frac_of_attainable_issue": 2.9 / (cyc / insts / (per_cu / 4.0)),
                     # the kernel's own arithmetic throughput: VALU wave-instructions it executes per launch / this run's launch time, against one VALU
                     # wave-instruction per 4 cycles per SIMD (1024 SIMDs x 2.4 GHz / 4 = 614.4 G/s)
                     "valu_wave_insts_per_s_G": valu["valu_insts_per_launch"] / kern_s / 1e9, "valu_wave_inst_peak_G": 614.4,
                     "valu_issue_frac": valu["valu_insts_per_launch"] / kern_s / 1e9 / 614.4,
                     "note": "instructions: VALU + SALU + LDS wave-instructions of the PMC pass named in traffic_source; cycles: shader clock inside the jobs of this run's last launch "
                             "(hand-overs included, tail of the launch excluded)"}
            valu_active = valu.get("valu_active_frac_of_launch")
        out = {
            "metric": "env-steps/sec (whole node) at 4096 envs/GPU, 69-DoF SMPL", "value": value, "unit": "env-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "repeats": len(blocks), "ms_per_step_min": min(blocks) / args.steps * 1e3, "ms_per_step_max": max(blocks) / args.steps * 1e3,
            "ms_per_step_blocks": [b / args.steps * 1e3 for b in blocks],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": WORKLOAD_DESC[args.workload], "workload_id": args.workload, "envs_per_gpu": ENVS_PER_GPU, "substeps": 15, "clip_len": CLIP_LEN,
                       "threads_per_env": args.threads_per_env, "parallelism": f"env-sharded x{world}",
                       "gemm_selection": "kinpoly_amd/assets/tunableop_gfx950.csv (rocBLAS / hipBLASLt solution per shape, fp32)" if getattr(build_engine, "tuned", False) else "library default"},
            "ranks_seen": ranks_seen, "devices": devices, "distinct_devices": distinct, "ms_per_step_per_rank": per_rank_ms, "collective_backend": (dist.get_backend() if in_group else None),
            # the kernel is bound by wave-level instruction issue, not by HBM or MFMA (its state lives in LDS); the HBM figures the
            # contract asks for are kept: achieved = algorithmic bytes / launch, traffic = PMC bytes / launch, traffic_over_algorithmic = their ratio
            "roofline": {"bound": "valu-issue", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_over_algorithmic": (traffic / algo_bytes) if traffic else None, "traffic_source": traffic_src,
                         "kernel": kernel_name, "launch_ms": kern_s * 1e3, "launch_ms_min": min(rec["kern_s_blocks"]) * 1e3, "launch_ms_max": max(rec["kern_s_blocks"]) * 1e3,
                         "launches_timed": n_launch, "algorithmic_bytes_per_launch": algo_bytes,
                         "valu_active_frac_of_launch": valu_active, "frac_of_attainable_issue": issue["frac_of_attainable_issue"] if issue else None,
                         "limiter": "wave-level instruction issue of latency-bound dependent chains: residency (LDS bytes per env) sets how many waves hide them -- three per SIMD on the lean layout of floor scenes, two on the full layout (state lives in LDS, so the compulsory HBM traffic is tiny by construction; DESIGN.md section 6)",
                         "valu": valu, "issue": issue},
            "kernel_share_of_step": kern_s / (elapsed / args.steps),
            "contacts_mean": float(diag[:, 0].mean()), "contacts_max_in_a_substep": int((diag[:, 3] & 255).max()), "newton_iters_per_substep": float(diag[:, 1].mean() / 15.0),
            "queue": qctr,
            "hessian_factorisations_per_substep": float((diag[:, 3] >> 8).mean() / 15.0),
            "bad_envs": int(((diag[:, 2] & 255) != 0).sum()), "newton_cap_hits": int((diag[:, 2] >> 8).sum()),
            "episodes_ended_per_step_frac": rec["n_done"] / (ENVS_PER_GPU * args.steps),
            "parity_live": live,
            "parity": parity_summary(args.workload),
            "mujoco_pin": mujoco_pin_report(),
            # per-env shader-clock cycles of the last launch (kp_sim_launch_cost): what the launch would take if its waves were perfectly
            # packed on the resident slots vs its longest env
            "launch_balance": {"substeps_per_job": spj, "slots": slots, "sum_env_cycles_over_slots_ms": float(cost.sum() / slots / 2.38e6),
                               "longest_env_ms": float(cost.max() / 2.38e6), "median_env_ms": float(np.median(cost) / 2.38e6)},
        }
        if world == 1 and not args.no_secondary:
            try:
                out["mfma"] = policy_gemm_probe(env, policy)
            except Exception as ex:
                out["mfma"] = {"error": type(ex).__name__}
            del sampler, env, policy
            torch.cuda.empty_cache()
            out["secondary_workloads"] = {}
            for wl in [w for w in ROLLOUT_WORKLOADS if w != args.workload]:
                try:
                    r2, e2, p2, s2, _ = run_workload(wl, local_rank, 4 + rank, args.threads_per_env, 40, 15)
                    ab = (ALGO_BYTES_PER_ENV_STEP_OBJ if wl == "objects" else ALGO_BYTES_PER_ENV_STEP) * ENVS_PER_GPU
                    out["secondary_workloads"][wl] = {"value": ENVS_PER_GPU * 40 / r2["elapsed"], "unit": "env-steps/s", "ms_per_step": r2["elapsed"] / 40 * 1e3, "steps": 40, "warmup": 15,
                                                      "launch_ms": r2["kern_s"] * 1e3, "contacts_mean": float(r2["diag"][:, 0].mean()),
                                                      "newton_iters_per_substep": float(r2["diag"][:, 1].mean() / 15.0),
                                                      "bad_envs": int(((r2["diag"][:, 2] & 255) != 0).sum()),
                                                      "algorithmic_bytes_per_launch": ab, "achieved_gbs": ab / r2["kern_s"] / 1e9,
                                                      ("fail_safe_per_step_frac" if wl == "wild_eval" else "episodes_ended_per_step_frac"): r2["n_done"] / (ENVS_PER_GPU * 40),
                                                      "workload": WORKLOAD_DESC[wl]}
                    del r2, e2, p2, s2
                    torch.cuda.empty_cache()
                except Exception as ex:
                    out["secondary_workloads"][wl] = {"error": f"{type(ex).__name__}: {ex}"[:200]}
            # the same two rollouts with twice the envs on the GPU (NOT the metric: BASELINE's figure is at 4096 envs per GPU).  The control-step launch ends
            # on the serial chain of its costliest env, which does not grow with the batch: what more envs buy is the idle tail
            out["envs_per_gpu_8192"] = {}
            for wl in ("tracked", "objects"):
                try:
                    ENVS_PER_GPU = 8192
                    r2, e2, p2, s2, _ = run_workload(wl, local_rank, 4 + rank, args.threads_per_env, 30, 10)
                    out["envs_per_gpu_8192"][wl] = {"value": 8192 * 30 / r2["elapsed"], "unit": "env-steps/s", "ms_per_step": r2["elapsed"] / 30 * 1e3, "launch_ms": r2["kern_s"] * 1e3,
                                                    "steps": 30, "warmup": 10}
                    del r2, e2, p2, s2
                except Exception as ex:
                    out["envs_per_gpu_8192"][wl] = {"error": f"{type(ex).__name__}: {ex}"[:200]}
                finally:
                    ENVS_PER_GPU = 4096
                    torch.cuda.empty_cache()
            # the headline's workload is named at top level, next to the other workloads' rates (round 1's headline was random_init)
            out["workload_id"] = args.workload
            out["values_by_workload"] = {args.workload: value, **{k: v.get("value") for k, v in out["secondary_workloads"].items()}}
            try:        # physics kernel alone on the contact-heavy object scenes (kp_step_queue_kernel<true>)
                out["object_scenes"] = object_scene_launches(local_rank, args.threads_per_env)
            except Exception as ex:
                out["object_scenes"] = {"error": f"{type(ex).__name__}: {ex}"[:200]}
            try:        # a whole training iteration: the update is its larger part
                out["train_iteration"] = {}
                for name, hz, it, obj, memo in ((f"4096x{TRAIN_HORIZON}", TRAIN_HORIZON, 2, False, False), (f"4096x{TRAIN_HORIZON}_objects", TRAIN_HORIZON, 1, True, False),
                                                (f"4096x{TRAIN_HORIZON}_init_context_memo", TRAIN_HORIZON, 2, False, True)):
                    r3 = train_iteration(local_rank, 4, hz, it, 1, objects=obj, cache_init_context=memo)
                    out["train_iteration"][name] = {k: r3[k] for k in TRAIN_KEYS}
                out["train_iteration"]["sampler_tracking_regime"] = sampler_regime(local_rank, 4)
                out["train_iteration"]["note"] = ("random-init networks: (almost) every episode fails after one step, so every env-step draws a clip and runs it through init_context "
                                                  "(context GRU over its 100 frames) -- the worst case of the episode source; `_init_context_memo` is the opt-in lookup of windows "
                                                  "already computed under the same context-network parameters (EpisodeSource.cache_init_context; the synthetic set has few windows, a MoCap set has ~1e5)")
            except Exception as ex:
                out["train_iteration"] = {"error": f"{type(ex).__name__}: {ex}"[:200]}
        else:
            del sampler, env, policy
            if train_n is not None:
                out["train_iteration"] = {f"4096x{TRAIN_HORIZON}_x{world}gpus": train_n}
        if world == 1 and not args.no_parity_live and not args.no_secondary:
            try:
                out["episode_parity"] = episode_parity_live()
            except Exception as ex:
                out["episode_parity"] = {"error": f"{type(ex).__name__}: {ex}"[:300]}
        if world == 1 and not args.no_cpu_baseline:
            workers = min(35, os.cpu_count() or 1)
            try:
                out["cpu_baseline"] = cpu_baseline_workers(std, workers) if workers > 1 else cpu_baseline(std)
            except Exception as ex:        # e.g. no room for 35 interpreters: fall back to the one-core figure
                out["cpu_baseline"] = cpu_baseline(std)
                out["cpu_baseline"]["sample"] += f" (multi-worker run failed: {type(ex).__name__})"
            # algorithmic FLOPs of the reference formulation (counted in the fp64 oracle on its rollout) x the GPU's env-step rate
            fl = out["cpu_baseline"].get("physics_flops_per_env_step")
            if fl:
                out["roofline"]["reference_algorithm_flops"] = {"physics_flops_per_env_step_oracle": fl, "reference_algorithm_tflops_at_this_rate": fl * value / world / 1e12,
                                                                "peak_tflops": VALU_FP32_PEAK_TFLOPS, "frac_if_the_kernel_executed_them": fl * value / world / 1e12 / VALU_FP32_PEAK_TFLOPS,
                                                                "note": "NOT the kernel's FLOPs: counted at the loop bodies of the fp64 oracle (dense stable-PD Cholesky, sparse L'DL, dense Newton Hessian: the "
                                                                        "arithmetic the reference + MuJoCo execute) on the CPU baseline's run of the same workload; the HIP kernel's matrix-free passes execute "
                                                                        "fewer, so this figure flatters -- the kernel's own throughput is roofline.issue.valu_issue_frac (instruction-derived)"}
        print(json.dumps(out), flush=True)
    if in_group:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
