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
HBM_PEAK_GBS = 8000.0               # MI355X_MICROARCH.md: 8.0 TB/s spec
VALU_FP32_PEAK_TFLOPS = 157.3       # 256 CUs x 4 SIMDs x 64 lanes x 2 (FMA) x 2.4 GHz (vector fp32; MI355X spec sheet)
MFMA_FP32_PEAK_TFLOPS = 157.3       # dense fp32 matrix peak (spec sheet): the policy / value GEMMs run in fp32
PROFILE_DIR = os.path.join(ROOT, "profiles", "r02")


def build_engine(device_index, seed, threads, workload="tracked"):
    from kinpoly_amd.env import BatchedHumanoidAREnv, standing_context
    from kinpoly_amd.nets import KinPolicy, enable_tuned_gemms
    from kinpoly_amd.rollout import VectorSampler
    build_engine.tuned = enable_tuned_gemms()
    torch.manual_seed(seed)
    std = np.load(os.path.join(ROOT, "tests", "golden", "standing_neutral.npz"))
    wild = workload == "wild_eval"
    opts = {"threads_per_env": threads, **({"substeps_per_job": int(os.environ["KP_SUBSTEPS_PER_JOB"])} if "KP_SUBSTEPS_PER_JOB" in os.environ else {})}
    env = BatchedHumanoidAREnv(ENVS_PER_GPU, device_index, mode="test" if wild else "train", wild=wild, seed=seed, model_options=opts)
    policy = KinPolicy().to(env.device).float()
    g = torch.Generator().manual_seed(seed)
    headings = (torch.rand(ENVS_PER_GPU, generator=g) * 2 - 1) * np.pi
    ctx = standing_context(ENVS_PER_GPU, CLIP_LEN, std["qpos"], std["qvel"], env.sim, headings)
    if wild:            # the kinematic roll-out the fail-safe falls back to (ar_context['ar_qpos' / 'ar_qvel']): the clip itself
        ctx["ar_qpos"], ctx["ar_qvel"] = ctx["qpos"].clone(), torch.zeros((ENVS_PER_GPU, CLIP_LEN, 75), device=env.device)
    env.load_context(ctx)
    sampler = VectorSampler(env, policy, mean_action=wild)
    sampler.start()
    return env, policy, sampler, std


def tracking_action(env):
    """The kinematic action a converged policy emits on the standing clip: next pose = the clip's pose (step_ar's encoding:
    root height, de-headed root quaternion, 69 joint angles, zero root velocities).  Used by --workload tracked only."""
    q0 = env.ctx["init_qpos"]
    obs0 = env.reset().clone()                                   # obs_ar[0:74] = qpos[2:] with the root quaternion de-headed
    a = torch.zeros((env.n, 80), device=env.device)
    a[:, :74] = torch.cat([q0[:, 2:3], obs0[:, 1:5], q0[:, 7:]], 1)
    return a


def rollout_steps(sampler, k, a_track=None, wild=False):
    """k batched env-steps without keeping the experience (identical work to VectorSampler.sample's loop body).
    wild: the eval_ar_policy.py --wild loop (:196-215): mean actions, an env that terminates early is put back on the kinematic
    roll-out (ar_fail_safe) and keeps going, an env that finishes its clip starts it again."""
    env, pol = sampler.env, sampler.policy
    n_done = torch.zeros((), dtype=torch.int64, device=env.device)
    with torch.no_grad():
        for _ in range(k):
            action, sampler.hx = pol.select_action(sampler.obs, sampler.hx, wild, env.gen)
            if a_track is not None:                              # the policy's GEMMs ran; a trained policy's output stands in for theirs
                action = a_track if wild else torch.add(a_track, torch.randn(action.shape, device=action.device, generator=env.gen), alpha=0.04)
            obs, _, done, info = env.step(action.contiguous())
            if wild:
                early = done & (info["percent"] != 1)
                env.ar_fail_safe(early)                          # masked, device side
                done = done & ~early
                n_done += early.sum()
            else:
                n_done += done.sum()
            sampler.obs = env.reset(done).clone()
            sampler.hx = sampler.hx.masked_fill(done.unsqueeze(1), 0.0)
    return n_done


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
}


def run_workload(workload, device_index, seed, threads, steps, warmup, barrier=None):
    """Build the engine for `workload`, run `warmup` untimed + `steps` timed env-steps.  Returns (record, env, policy, sampler, std)."""
    env, policy, sampler, std = build_engine(device_index, seed, threads, workload)
    a_track = None
    if workload in ("tracked", "wild_eval"):
        a_track = tracking_action(env)
        sampler.start()
    wild = workload == "wild_eval"
    rollout_steps(sampler, warmup, a_track, wild)
    env.sim.timing_reset()
    if barrier is not None:
        barrier()
    else:
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    n_done = rollout_steps(sampler, steps, a_track, wild)
    if barrier is not None:
        barrier()
    else:
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kern_s, n_launch = env.sim.timing_mean_seconds()
    diag = env.sim.diag()
    rec = {"elapsed": elapsed, "kern_s": kern_s, "n_launch": n_launch, "diag": diag, "n_done": float(n_done.item())}
    return rec, env, policy, sampler, std


def main():
    if len(sys.argv) >= 3 and sys.argv[1] == "--cpu-baseline-worker":
        std = np.load(os.path.join(ROOT, "tests", "golden", "standing_neutral.npz"))
        r = cpu_baseline(std, float(sys.argv[2]))
        n = int(r["sample"].split()[0])
        print(json.dumps({"steps": n, "seconds": n / r["value"], "flops": r["physics_flops_per_env_step"] * n}), flush=True)
        return
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--threads-per-env", type=int, default=int(os.environ.get("KP_THREADS_PER_ENV", "64")))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary workloads and the GEMM probe (profiling runs)")
    ap.add_argument("--workload", choices=tuple(WORKLOAD_DESC), default="tracked")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    shared = os.environ.get("KP_BENCH_SHARED_DEVICE") == "1"     # plumbing test on a 1-GPU box: all ranks on device 0, gloo barrier
    if shared:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if shared:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    rec, env, policy, sampler, std = run_workload(args.workload, local_rank, 4 + rank, args.threads_per_env, args.steps, args.warmup, barrier)
    elapsed = rec["elapsed"]
    if world > 1:
        t = torch.tensor([elapsed], device="cpu" if shared else "cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kern_s, n_launch, diag = rec["kern_s"], rec["n_launch"], rec["diag"]
    cost = env.sim.launch_cost().astype(np.float64)

    if rank == 0:
        value = ENVS_PER_GPU * world * args.steps / elapsed
        algo_bytes = ALGO_BYTES_PER_ENV_STEP * ENVS_PER_GPU
        achieved = algo_bytes / kern_s / 1e9
        kernel_name = "kp_step_queue_kernel" if int(env.model.get_option("substeps_per_job")) > 0 else "kp_step_kernel"
        # HBM bytes / instruction counts: ONLY from a rocprofv3 --pmc pass of this same command and workload (tools/profile_bench.sh writes
        # profiles/r02/pmc_bench_<workload>.json); otherwise null -- nothing canned from another workload enters the line
        traffic, traffic_src, valu = None, None, None
        pmc_path = os.path.join(PROFILE_DIR, f"pmc_bench_{args.workload}.json")
        if os.path.exists(pmc_path):
            pj = json.load(open(pmc_path))
            traffic = pj.get("hbm_bytes_per_launch")
            traffic_src = {"file": os.path.relpath(pmc_path, ROOT), "command": pj.get("command"), "launch_ms_in_those_passes": pj.get("launch_ms"),
                           "note": "separate rocprofv3 --pmc passes of this command (FETCH_SIZE x2 per the guide's gfx950 correction, WRITE_SIZE as reported), "
                                   "median per launch of the same kernel; not measured in this run"}
            valu = pj.get("issue")
        # instruction issue: instructions per env-step from that PMC pass, wave cycles per env-step live from this run's last
        # launch (kp_sim_launch_cost), against what two waves per SIMD can issue (tools/micro/valu_probe.hip: one instruction per 2.9 SIMD cycles)
        issue = None
        if valu and valu.get("valu_insts_per_launch"):
            insts = (valu["valu_insts_per_launch"] + valu.get("salu_insts_per_launch", 0.0) + valu.get("lds_insts_per_launch", 0.0)) / ENVS_PER_GPU
            cyc = float(cost.mean())
            issue = {"wave_insts_per_env_step": insts, "wave_cycles_per_env_step": cyc, "wave_cycles_per_inst": cyc / insts, "waves_per_simd": 2,
                     "simd_cycles_per_inst": cyc / insts / 2.0, "attainable_simd_cycles_per_inst_at_2_waves": 2.9, "frac_of_attainable_issue": 2.9 / (cyc / insts / 2.0),
                     "note": "instructions: VALU + SALU + LDS wave-instructions of the PMC pass named in traffic_source; cycles: shader clock inside the jobs of this run's last launch "
                             "(hand-overs included, tail of the launch excluded); a third wave per SIMD buys ~1 % (profiles/r02/occupancy_premise.log)"}
        out = {
            "metric": "env-steps/sec (whole node) at 4096 envs/GPU, 69-DoF SMPL", "value": value, "unit": "env-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": WORKLOAD_DESC[args.workload], "workload_id": args.workload, "envs_per_gpu": ENVS_PER_GPU, "substeps": 15, "clip_len": CLIP_LEN,
                       "threads_per_env": args.threads_per_env, "parallelism": f"env-sharded x{world}",
                       "gemm_selection": "kinpoly_amd/assets/tunableop_gfx950.csv (rocBLAS / hipBLASLt solution per shape, fp32)" if getattr(build_engine, "tuned", False) else "library default"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_over_algorithmic": (traffic / algo_bytes) if traffic else None, "traffic_source": traffic_src,
                         "kernel": kernel_name, "launch_ms": kern_s * 1e3, "launches_timed": n_launch, "algorithmic_bytes_per_launch": algo_bytes,
                         "limiter": "wave-level instruction issue: the SIMDs are saturated by their two resident waves (state lives in LDS, so the compulsory HBM traffic is tiny by construction; DESIGN.md section 6)",
                         "valu": valu, "issue": issue},
            "kernel_share_of_step": kern_s / (elapsed / args.steps),
            "contacts_mean": float(diag[:, 0].mean()), "newton_iters_per_substep": float(diag[:, 1].mean() / 15.0),
            "hessian_factorisations_per_substep": float((diag[:, 3] >> 8).mean() / 15.0),
            "bad_envs": int(((diag[:, 2] & 255) != 0).sum()), "newton_cap_hits": int((diag[:, 2] >> 8).sum()),
            "episodes_ended_per_step_frac": rec["n_done"] / (ENVS_PER_GPU * args.steps),
            # per-env shader-clock cycles of the last launch (kp_sim_launch_cost): what the launch would take if its waves were perfectly
            # packed on the 2048 resident slots vs its longest env
            "launch_balance": {"substeps_per_job": int(env.model.get_option("substeps_per_job")), "sum_env_cycles_over_2048_slots_ms": float(cost.sum() / 2048 / 2.38e6),
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
            for wl in [w for w in WORKLOAD_DESC if w != args.workload]:
                try:
                    r2, e2, p2, s2, _ = run_workload(wl, local_rank, 4 + rank, args.threads_per_env, 40, 15)
                    out["secondary_workloads"][wl] = {"value": ENVS_PER_GPU * 40 / r2["elapsed"], "unit": "env-steps/s", "ms_per_step": r2["elapsed"] / 40 * 1e3, "steps": 40, "warmup": 15,
                                                      "launch_ms": r2["kern_s"] * 1e3, "contacts_mean": float(r2["diag"][:, 0].mean()),
                                                      "newton_iters_per_substep": float(r2["diag"][:, 1].mean() / 15.0),
                                                      ("fail_safe_per_step_frac" if wl == "wild_eval" else "episodes_ended_per_step_frac"): r2["n_done"] / (ENVS_PER_GPU * 40),
                                                      "workload": WORKLOAD_DESC[wl]}
                    del r2, e2, p2, s2
                    torch.cuda.empty_cache()
                except Exception as ex:
                    out["secondary_workloads"][wl] = {"error": f"{type(ex).__name__}: {ex}"[:200]}
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
                out["roofline"]["valu_flops"] = {"physics_flops_per_env_step_oracle": fl, "achieved_tflops": fl * value / world / 1e12, "peak_tflops": VALU_FP32_PEAK_TFLOPS,
                                                 "frac": fl * value / world / 1e12 / VALU_FP32_PEAK_TFLOPS,
                                                 "note": "FLOPs counted at the loop bodies of the fp64 oracle (dense stable-PD Cholesky, sparse L'DL, dense Newton Hessian: the arithmetic "
                                                         "the reference + MuJoCo execute) on the CPU baseline's run of the same workload; the HIP kernel's matrix-free passes do fewer"}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
