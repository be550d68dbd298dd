#!/usr/bin/env python
"""bench.py -- env-steps/sec of the dynamics-regulated rollout hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run, one rank per GPU)

One "step" = one batched env-step of the kin_poly.yml rollout over ENVS_PER_GPU = 4096 environments
per GPU: kinematic policy (GRU+MLP) -> step_ar -> target FK -> UHC obs (784) + ZFilter
-> PolicyMCP -> 15 physics substeps (stable-PD + RFC + forward dynamics + hull-plane contact) ->
termination + reward -> AR obs (105) -> device-side auto-reset.  Inputs are synthetic and resident in HBM
before the timed region (standing clip contexts, seeded random-init networks).

Prints ONE JSON line on rank 0 (see the task contract) with `roofline` (dominant kernel kp_step_queue_kernel = kp_step_kernel scheduled as jobs,
live HIP-event launch durations) and, at N = 1, `cpu_baseline` (the fp64 oracle port in 35 single-threaded worker processes).
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


def build_engine(device_index, seed, threads):
    from kinpoly_amd.env import BatchedHumanoidAREnv, standing_context
    from kinpoly_amd.nets import KinPolicy, enable_tuned_gemms
    from kinpoly_amd.rollout import VectorSampler
    build_engine.tuned = enable_tuned_gemms()
    torch.manual_seed(seed)
    std = np.load(os.path.join(ROOT, "tests", "golden", "standing_neutral.npz"))
    env = BatchedHumanoidAREnv(ENVS_PER_GPU, device_index, mode="train", seed=seed, model_options={"threads_per_env": threads, **({"substeps_per_job": int(os.environ["KP_SUBSTEPS_PER_JOB"])} if "KP_SUBSTEPS_PER_JOB" in os.environ else {})})
    policy = KinPolicy().to(env.device).float()
    g = torch.Generator().manual_seed(seed)
    headings = (torch.rand(ENVS_PER_GPU, generator=g) * 2 - 1) * np.pi
    env.load_context(standing_context(ENVS_PER_GPU, CLIP_LEN, std["qpos"], std["qvel"], env.sim, headings))
    sampler = VectorSampler(env, policy)
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


def rollout_steps(sampler, k, a_track=None):
    """k batched env-steps without keeping the experience (identical work to VectorSampler.sample's loop body)."""
    env, pol = sampler.env, sampler.policy
    n_done = torch.zeros((), dtype=torch.int64, device=env.device)
    with torch.no_grad():
        for _ in range(k):
            action, sampler.hx = pol.select_action(sampler.obs, sampler.hx, False, env.gen)
            if a_track is not None:                              # the policy's GEMMs ran; a trained policy's output stands in for theirs
                action = action * 0.0 + a_track + 0.04 * torch.randn(action.shape, device=action.device, generator=env.gen)
            _, _, done, info = env.step(action.contiguous())
            n_done += done.sum()
            sampler.obs = env.reset(done).clone()
            sampler.hx = sampler.hx * (~done).float().unsqueeze(1)
    return n_done


def cpu_baseline(std, seconds_budget=15.0):
    """The fp64 oracle port of the same env-step on ONE host core: C physics (oracle/kp_oracle.c) + numpy
    obs / FK / reward (oracle/np_oracle.py) + fp64 torch policies on 1 thread (the reference samples on CPU,
    OMP_NUM_THREADS=1, agent_ar.py:29,654)."""
    from kinpoly_amd.model_compiler import DEFAULT_KPM, read_kpm
    from kinpoly_amd.nets import KinPolicy, PolicyMCP
    from oracle import np_oracle as O
    from oracle.kpo import OracleSim
    torch.set_num_threads(1)
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
    n_steps, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds_budget:
        sim.reset(qpos0, qvel0)
        hx = torch.zeros(1, 1024, dtype=torch.float64)
        x = {k: sim.get(k) for k in ("qpos", "qvel", "xpos", "xquat", "xipos")}
        obs = O.obs_ar(x["qpos"], x["xpos"].reshape(24, 3), x["xquat"].reshape(24, 4), head, hv, obj_rel, one_hot, None)
        for t in range(CLIP_LEN - 1):
            with torch.no_grad():
                a, hx = kin.select_action(torch.from_numpy(obs)[None], hx)
            a = a[0].numpy()
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
    return {"value": n_steps / dt, "unit": "env-steps/s", "cores": 1, "kind": "port",
            "sample": f"{n_steps} env-steps of the same standing-clip rollout (fp64 C physics oracle + numpy obs/reward + fp64 torch policies, 1 thread) in {dt:.1f} s; host has {os.cpu_count()} cores"}


def cpu_baseline_workers(std, workers, seconds_budget=15.0):
    """The same oracle roll-out in `workers` single-threaded processes at once -- the shape of the reference's sampler
    (`--num_threads 35`: one env per forked worker, OMP_NUM_THREADS=1, agent_ar.py:29,651-680).  Each worker is a fresh
    interpreter (a HIP context does not survive fork) that runs cpu_baseline() and prints its step count."""
    import subprocess
    env = dict(os.environ, OMP_NUM_THREADS="1", MKL_NUM_THREADS="1", HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", str(seconds_budget)]
    t0 = time.perf_counter()
    procs = [subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for _ in range(workers)]
    steps, wall = 0, 0.0
    for p in procs:
        try:
            out, _ = p.communicate(timeout=seconds_budget * 4 + 120)
            rec = json.loads(out.strip().splitlines()[-1])
            steps += rec["steps"]; wall = max(wall, rec["seconds"])
        except Exception:
            p.kill()
            raise
    return {"value": steps / wall, "unit": "env-steps/s", "cores": workers, "kind": "port",
            "sample": f"{steps} env-steps of the same standing-clip rollout in {workers} single-threaded worker processes (fp64 C physics oracle + numpy obs/reward "
                      f"+ fp64 torch policies each; the reference samples with 35 such workers) over {wall:.1f} s of rollout ({time.perf_counter() - t0:.1f} s with start-up); "
                      f"host has {os.cpu_count()} cores"}


def main():
    if len(sys.argv) >= 3 and sys.argv[1] == "--cpu-baseline-worker":
        std = np.load(os.path.join(ROOT, "tests", "golden", "standing_neutral.npz"))
        r = cpu_baseline(std, float(sys.argv[2]))
        n = int(r["sample"].split()[0])
        print(json.dumps({"steps": n, "seconds": n / r["value"]}), flush=True)
        return
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--threads-per-env", type=int, default=int(os.environ.get("KP_THREADS_PER_ENV", "64")))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", choices=("random_init", "tracked"), default="random_init",
                    help="random_init (default, BASELINE configs[2]): seeded random-init networks; tracked: same step, but the kinematic "
                         "policy's output is replaced by the clip's own pose + exploration noise, i.e. episodes that last (secondary figure)")
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

    env, policy, sampler, std = build_engine(local_rank, 4 + rank, args.threads_per_env)
    a_track = None
    if args.workload == "tracked":
        a_track = tracking_action(env)
        sampler.start()
    rollout_steps(sampler, args.warmup, a_track)
    env.sim.timing_reset()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    n_done = rollout_steps(sampler, args.steps, a_track)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kern_s, n_launch = env.sim.timing_mean_seconds()
    diag = env.sim.diag()
    cost = env.sim.launch_cost().astype(np.float64)

    if rank == 0:
        value = ENVS_PER_GPU * world * args.steps / elapsed
        algo_bytes = ALGO_BYTES_PER_ENV_STEP * ENVS_PER_GPU
        achieved = algo_bytes / kern_s / 1e9
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
        # what actually bounds the kernel: VALU issue.  Counts and kernel duration both from the committed PMC pass (tools/pmc_step.py workload)
        valu = None
        if os.path.exists(pmc):
            valu = json.load(open(pmc)).get("issue")
        out = {
            "metric": "env-steps/sec (whole node) at 4096 envs/GPU, 69-DoF SMPL", "value": value, "unit": "env-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[2] rollout: kin_poly.yml dynamics-regulated env-step (kin GRU policy, step_ar, target FK, "
                                   "UHC obs+ZFilter+PolicyMCP, 15 substeps SPD+RFC+contact, term/reward, AR obs, auto-reset), standing MoCap clip, "
                                   "random-init seeded networks" + ("; SECONDARY workload 'tracked': kinematic policy output replaced by the clip pose + N(0, 0.04) noise" if args.workload == "tracked" else ""), "envs_per_gpu": ENVS_PER_GPU, "substeps": 15, "clip_len": CLIP_LEN,
                       "threads_per_env": args.threads_per_env, "parallelism": f"env-sharded x{world}",
                       "gemm_selection": "kinpoly_amd/assets/tunableop_gfx950.csv (rocBLAS / hipBLASLt solution per shape, fp32)" if getattr(build_engine, "tuned", False) else "library default"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "kernel": "kp_step_queue_kernel" if int(env.model.get_option("substeps_per_job")) > 0 else "kp_step_kernel", "launch_ms": kern_s * 1e3, "launches_timed": n_launch,
                         "algorithmic_bytes_per_launch": algo_bytes,
                         "note": "latency/VALU-bound tree recursion with state resident in LDS: compulsory HBM traffic is tiny by construction (DESIGN.md)",
                         "valu": valu},
            "kernel_share_of_step": kern_s / (elapsed / args.steps),
            "contacts_mean": float(diag[:, 0].mean()), "newton_iters_per_substep": float(diag[:, 1].mean() / 15.0), "hessian_factorisations_per_substep": float((diag[:, 3] >> 8).mean() / 15.0), "bad_envs": int((diag[:, 2] != 0).sum()),
            # random-init networks put the kinematic target far from the humanoid, so (as in the reference with untrained weights) the
            # body-diff termination (humanoid_ar_v1.py:303-309) fires on almost every step: each timed step includes the device-side reset
            "episodes_ended_per_step_frac": float(n_done.item()) / (ENVS_PER_GPU * args.steps),
            # per-env shader-clock cycles of the last launch (kp_sim_launch_cost): what the launch would take if its waves were perfectly
            # packed on the 2048 resident slots vs its longest env
            "launch_balance": {"substeps_per_job": int(env.model.get_option("substeps_per_job")), "sum_env_cycles_over_2048_slots_ms": float(cost.sum() / 2048 / 2.38e6),
                               "longest_env_ms": float(cost.max() / 2.38e6), "median_env_ms": float(np.median(cost) / 2.38e6)},
        }
        if world == 1 and args.workload == "random_init":
            # secondary figure (not the metric): the same step with episodes that last -- see --workload tracked
            try:
                a_tr = tracking_action(env)
                sampler.start()
                rollout_steps(sampler, 20, a_tr)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                nd = rollout_steps(sampler, 60, a_tr)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t1
                out["secondary_tracked_workload"] = {"value": ENVS_PER_GPU * 60 / dt, "unit": "env-steps/s", "ms_per_step": dt / 60 * 1e3, "steps": 60, "warmup": 20,
                                                     "episodes_ended_per_step_frac": float(nd.item()) / (ENVS_PER_GPU * 60),
                                                     "note": "kinematic policy output replaced by the clip pose + N(0, e^-3.2) noise (its GEMMs still run): what a pretrained policy emits"}
            except Exception as ex:
                out["secondary_tracked_workload"] = {"error": type(ex).__name__}
        if world == 1 and not args.no_cpu_baseline:
            workers = min(35, os.cpu_count() or 1)
            try:
                out["cpu_baseline"] = cpu_baseline_workers(std, workers) if workers > 1 else cpu_baseline(std)
            except Exception as ex:        # e.g. no room for 35 interpreters: fall back to the one-core figure
                out["cpu_baseline"] = cpu_baseline(std)
                out["cpu_baseline"]["sample"] += f" (multi-worker run failed: {type(ex).__name__})"
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
