"""Whole-episode, distribution-level parity of the product rollout against the fp64 CPU episode loop.

`VectorSampler.sample` (HIP engine, fp32) and `oracle/episode.py::EpisodeOracle.rollout` (the CPU restatement of AgentAR.sample_worker,
kin_poly/core/agent_ar.py:510-611, around HumanoidAREnv.step, kin_poly/envs/humanoid_ar_v1.py:295-323) run the SAME episodes: same clips, same
network weights, same exploration noise for both policies (drawn once, fed to both sides), `n` envs x `T` control steps with episodes ending (clip
end or body-diff failure) and restarting inside the window.  One-substep and one-control-step parity are measured elsewhere (tools/substep_parity.py);
this is what the knife-edge contact flips and fp32 rounding add up to over a whole 100-frame episode:

  * |dqpos| percentiles as a function of the control step, over the envs whose episode history (done flags) is still identical on both sides
    (the growth curve: do differences stay damped?);
  * OUTCOMES: per-env step of the first termination, per-env number of failures, total failures, mean reward, episodes started.

configs[2]: standing MoCap clip (SURVEY 8(d) config 3 stand-in), no objects.  configs[3] (--objects): the four action classes of
dataset.synthetic_takes with their free objects simulated.  The kinematic policy is one whose mean tracks the standing pose, modulated by its GRU / MLP
path (scaled random weights) -- or the networks of --policy-ckpt (checkpoint.py layout); the UHC is the seeded random-init PolicyMCP or the checkpoint's.

The oracle is the CHECKER: nothing here is on a timed or product path.  Oracle episodes run in `--workers` spawned single-threaded processes.

    python tools/episode_parity.py [--envs 128] [--steps 99] [--objects] [--workers 32] [--json out.json]
"""
import argparse
import copy
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CTX_KEYS = ("qpos", "head_pose", "head_vels", "obj_head_relative_poses", "action_one_hot", "init_qpos", "init_qvel", "obj_pose")
_W = {}


def _worker_init(kin_sd, mcp_sd, kpm_path, zf=None):
    import torch
    torch.set_num_threads(1)
    from kinpoly_amd.model_compiler import read_kpm
    from kinpoly_amd.nets import KinPolicy, PolicyMCP
    from oracle.episode import EpisodeOracle
    kin, mcp = KinPolicy().double(), PolicyMCP().double()
    kin.load_state_dict({k: torch.from_numpy(v).double() for k, v in kin_sd.items() if not k.startswith("context_")}); mcp.load_state_dict({k: torch.from_numpy(v).double() for k, v in mcp_sd.items()})
    _W["ep"] = EpisodeOracle(read_kpm(kpm_path), kin, mcp, kpm_path=kpm_path, zfilter=zf)


def _worker_run(job):
    e, ctx, T, noise = job
    want = _W["ep"].rollout(ctx, T, noise=noise)
    out = {k: want[k] for k in ("res_qpos", "reward", "done", "fail", "mask", "percent", "episode_start")}
    out["first_row"] = {k: want[k][0] for k in ("state", "action", "cc_state", "cc_action")}       # the very first control step: no dynamics behind it yet
    return e, out


def tracking_policy(env, ctx, seed):
    """a kinematic policy whose mean tracks the standing pose, modulated a little by its GRU / MLP path (as tests/test_gpu_sampler.py builds it)"""
    import torch
    from kinpoly_amd.nets import KinPolicy
    torch.manual_seed(seed)
    pol = KinPolicy().to(env.device)
    with torch.no_grad():
        pol.action_fc.weight.mul_(0.02); pol.action_fc.bias.zero_()
        q0 = ctx["init_qpos"]
        pol.action_fc.bias[:74] = torch.cat([q0[0, 2:3], torch.tensor([1.0, 0, 0, 0], device=env.device), q0[0, 7:]])
    return pol


def run(n=128, T=99, seed=7, objects=False, workers=32, policy_ckpt=None, device=0, clip_len=100, cc_ckpt=None):
    import torch
    from kinpoly_amd.env import BatchedHumanoidAREnv, standing_context
    from kinpoly_amd.rollout import VectorSampler
    from kinpoly_amd.sim import STEP_KPM
    std = np.load(os.path.join(ROOT, "tests", "golden", "standing_neutral.npz"))
    torch.manual_seed(seed)
    cc_policy = cc_rs = None
    if policy_ckpt:
        from kinpoly_amd.checkpoint import load_bench_policies
        ck = load_bench_policies(policy_ckpt, cc_ckpt, torch.device("cuda", device))
        cc_policy, cc_rs = ck["cc_policy"], ck["cc_running_state"]
    env = BatchedHumanoidAREnv(n, device, mode="train", seed=seed, cc_policy=cc_policy, cc_running_state=cc_rs)
    g = torch.Generator().manual_seed(seed)
    if objects:
        from kinpoly_amd import dataset as D
        from kinpoly_amd.model_compiler import read_kpm
        takes = D.synthetic_takes(env.sim, std["qpos"], n_per_action=8, T_range=(clip_len + 10, clip_len + 60), body_mass=read_kpm(STEP_KPM)["body_mass"], seed=seed)
        ds = D.StateARDataset(takes, fr_num=clip_len, seed=seed, device=env.device)
        ctx = ds.sample_batch(n, use_freq=False)
        ctx = ds.batch(ctx["take_ind"].numpy(), ds.rng.randint(0, 10, n), clip_len)
        ctx["init_qpos"], ctx["init_qvel"] = ctx["qpos"][:, 0].contiguous(), ctx["qvel"][:, 0].contiguous()
    else:
        ctx = standing_context(n, clip_len, std["qpos"], std["qvel"], env.sim, (torch.rand(n, generator=g) * 2 - 1) * np.pi)
    env.load_context(ctx)
    pol = ck["kin_policy"] if policy_ckpt else tracking_policy(env, ctx, seed)
    env.reset()
    noise = torch.randn((T, n, 155), generator=g).to(env.device)
    sampler = VectorSampler(env, pol, record_full=True)
    t0 = time.perf_counter()
    b = sampler.sample(T, noise=noise)
    torch.cuda.synchronize()
    t_hip = time.perf_counter() - t0
    bad = int(((env.sim.diag()[:, 2] & 255) != 0).sum())
    first_hip = {"state": b.states[:, 0], "action": b.actions[:, 0], "cc_state": b.cc_state[:, 0], "cc_action": b.cc_action[:, 0]}
    first_hip = {k: v.double().cpu().numpy() for k, v in first_hip.items()}
    hip = {"res_qpos": b.res_qpos.double().cpu().numpy(), "reward": b.rewards.double().cpu().numpy(), "done": (b.masks == 0).cpu().numpy(), "fail": b.fails.cpu().numpy().astype(bool)}
    # ---- the same episodes on the CPU: fp64 copies of the networks (ZFilter identity unless the checkpoint carries one: EpisodeOracle applies zfilter(0, 1, 5))
    zf = None if cc_rs is None else (cc_rs.mean.double().cpu().numpy(), cc_rs.std.double().cpu().numpy(), float(cc_rs.clip))
    sd = lambda m: {k: v.detach().double().cpu().numpy() for k, v in copy.deepcopy(m).state_dict().items()}       # noqa: E731
    c = {k: v.double().cpu().numpy() for k, v in ctx.items() if k in CTX_KEYS}
    nz = noise.double().cpu().numpy()
    jobs = []
    for e in range(n):
        one = {k: (c[k][e] if c[k].ndim > 1 else c[k]) for k in c}
        if one["action_one_hot"].ndim == 2:                      # a data set's batch carries the one-hot per frame; it is constant over a take
            one["action_one_hot"] = one["action_one_hot"][0]
        if not objects:
            one.pop("obj_pose", None)
        jobs.append((e, one, T, nz[:, e]))
    t0 = time.perf_counter()
    want = [None] * n
    with mp.get_context("spawn").Pool(min(workers, n), initializer=_worker_init, initargs=(sd(pol), sd(env.cc_policy), STEP_KPM, zf)) as pool:
        for e, r in pool.imap_unordered(_worker_run, jobs):
            want[e] = r
    t_cpu = time.perf_counter() - t0
    ora = {k: np.stack([w[k] for w in want]) for k in ("res_qpos", "reward", "done", "fail")}
    # where a first-step difference comes from: the 784-d UHC observation AFTER the ZFilter ((x - mean) / (std + 1e-8): a feature that barely varied in training has
    # a tiny std, which multiplies the fp32 rounding of x), the UHC action, the kinematic action
    first = {k: float(np.abs(first_hip[k] - np.stack([w["first_row"][k] for w in want])).max()) for k in first_hip}
    first["zfilter_min_std"] = None if zf is None else float(np.min(zf[1]))
    first["zfilter_stds_below_1e-3"] = None if zf is None else int((zf[1] < 1e-3).sum())
    return summarise(hip, ora, dict(first_control_step_max_abs_diff=first, envs=n, steps=T, seed=seed, objects=bool(objects), policy=("checkpoint " + os.path.basename(policy_ckpt)) if policy_ckpt else "standing-pose tracker (scaled random GRU / MLP)",
                                    seconds_hip=t_hip, seconds_oracle=t_cpu, oracle_workers=min(workers, n), bad_envs=bad))


def summarise(hip, ora, meta):
    n, T = hip["done"].shape
    same_hist = np.cumprod(hip["done"] == ora["done"], axis=1).astype(bool)            # [n, T]: done flags identical up to and including step t
    aligned = np.concatenate([np.ones((n, 1), bool), same_hist[:, :-1]], 1)             # the state at step t comes from identical episode histories
    dq = np.abs(hip["res_qpos"] - ora["res_qpos"]).max(axis=2)                          # [n, T]
    curve = {}
    for t in sorted({0, 1, 2, 4, 9, 19, 29, 49, 69, 89, T - 1}):
        if t < T and aligned[:, t].any():
            d = dq[aligned[:, t], t]
            curve[str(t + 1)] = {"aligned_envs": int(aligned[:, t].sum()), "p50": float(np.percentile(d, 50)), "p90": float(np.percentile(d, 90)), "p99": float(np.percentile(d, 99)), "max": float(d.max())}
    # within an episode: |dqpos| by the episode's own step count (steps since the last reset), aligned rows only
    age = np.zeros((n, T), int)
    for t in range(1, T):
        age[:, t] = np.where(ora["done"][:, t - 1], 0, age[:, t - 1] + 1)
    by_age = {}
    for a in (0, 1, 2, 4, 9, 19, 39, 59, 79, 98):
        m = aligned & (age == a)
        if m.any():
            by_age[str(a + 1)] = {"rows": int(m.sum()), "p50": float(np.percentile(dq[m], 50)), "p99": float(np.percentile(dq[m], 99)), "max": float(dq[m].max())}
    first = lambda d: np.where(d.any(1), d.argmax(1), T)                                # noqa: E731   step of the first termination (T: none in the window)
    fh, fo = first(hip["done"]), first(ora["done"])
    out = dict(meta)
    out.update({
        "first_termination_step_equal_frac": float((fh == fo).mean()),
        "first_termination_step_abs_diff_max": int(np.abs(fh - fo).max()),
        "done_flags_equal_frac_of_rows": float((hip["done"] == ora["done"]).mean()),
        "envs_with_identical_done_history": int(same_hist[:, -1].sum()),
        "episodes_ended": {"hip": int(hip["done"].sum()), "oracle": int(ora["done"].sum())},
        "failures": {"hip": int(hip["fail"].sum()), "oracle": int(ora["fail"].sum())},
        "failures_per_env_equal_frac": float((hip["fail"].sum(1) == ora["fail"].sum(1)).mean()),
        "mean_reward": {"hip": float(hip["reward"].mean()), "oracle": float(ora["reward"].mean()), "rel_diff": float(abs(hip["reward"].mean() - ora["reward"].mean()) / abs(ora["reward"].mean()))},
        "mean_episode_return_first_episode": {"hip": float(np.mean([hip["reward"][e, :min(fh[e] + 1, T)].sum() for e in range(n)])),
                                              "oracle": float(np.mean([ora["reward"][e, :min(fo[e] + 1, T)].sum() for e in range(n)]))},
        "dqpos_vs_control_step": curve,
        "dqpos_vs_episode_step": by_age,
        "dqpos_aligned_rows": {"rows": int(aligned.sum()), "p50": float(np.percentile(dq[aligned], 50)), "p99": float(np.percentile(dq[aligned], 99)), "max": float(dq[aligned].max()),
                               "frac_above_1e-3": float((dq[aligned] > 1e-3).mean())},
        "note": "VectorSampler (HIP, fp32) vs oracle/episode.py (fp64 CPU restatement of sample_worker) on the same clips, weights and exploration noise; |dqpos| = max over the 76 "
                "coordinates of res_qpos after a control step, over rows whose done-flag history is identical on both sides",
    })
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=128); ap.add_argument("--steps", type=int, default=99); ap.add_argument("--seed", type=int, default=7)
    ap.add_argument("--objects", action="store_true"); ap.add_argument("--workers", type=int, default=min(32, os.cpu_count() or 1))
    ap.add_argument("--policy-ckpt", default=None); ap.add_argument("--cc-ckpt", default=None); ap.add_argument("--json", default=None)
    a = ap.parse_args()
    r = run(a.envs, a.steps, a.seed, a.objects, a.workers, a.policy_ckpt, cc_ckpt=a.cc_ckpt)
    s = json.dumps(r, indent=1)
    print(s)
    if a.json:
        open(a.json, "w").write(s + "\n")
