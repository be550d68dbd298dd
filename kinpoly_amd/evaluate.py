"""Batched evaluation roll-outs: `run_seq` / `test_coverage` (here `run_sequences` / `write_coverage`) of scripts/eval_ar_policy.py:178-262 for N sequences at once.

Every environment plays one sequence with the mean action of the kinematic policy (test mode => mean UHC action too); the
per-sequence record has the reference's keys (`target`, `pred`, `obj_pose`, `percent`, `fail_safe`) so that the reference's
metric script (eval_pose_all.py) reads the `*_coverage_full.pkl` this writes.  With `fail_safe` a sequence that terminates
early is put back on the kinematic roll-out (`env.ar_fail_safe`) and continues, exactly as the reference does.
"""
from __future__ import annotations

import os

import numpy as np
import torch


@torch.no_grad()
def run_sequences(env, policy_net, keys, fail_safe=False, max_steps=100000):
    """env: BatchedHumanoidAREnv with its context loaded (one sequence per env, mode 'test').  Returns {key: seq_result}."""
    n = env.n
    if fail_safe and "ar_qpos" not in env.ctx:            # checked once, up front: the fail-safe runs (masked) every step, whether or not an env ended early
        raise ValueError("run_sequences(fail_safe=True) needs the kinematic roll-out in the context (ctx['ar_qpos'] / ['ar_qvel']: PolicyARContext(need_rollout=True))")
    env.set_mode("test")
    obs = env.reset()
    hx = policy_net.init_hidden(n, env.device)
    active = torch.ones(n, dtype=torch.bool, device=env.device)
    used_fs = torch.zeros(n, dtype=torch.bool, device=env.device)
    percent = torch.zeros(n, device=env.device)
    rec = {"target": [], "pred": [], "obj_pose": [], "active": []}
    for step in range(max_steps):
        rec["target"].append(env.sim.get("target_qpos")); rec["pred"].append(env.get_humanoid_qpos())
        rec["obj_pose"].append(env.obj_qpos); rec["active"].append(active.clone())
        action, hx = policy_net.select_action(obs, hx, True, env.gen)
        obs, _, done, info = env.step(action.contiguous())
        newly = done & active
        percent = torch.where(newly, info["percent"], percent)
        early = newly & (info["percent"] != 1)
        if fail_safe:                               # masked on the device (no host read per step): envs that ended early go back onto the kinematic roll-out
            used_fs |= early
            env.ar_fail_safe(early)
            obs = env.sim.obs_ar(env._ctx_struct, env._obs)
            newly = newly & ~early
        active = active & ~newly
        if step % 8 == 7 and not bool(active.any()):      # one host read every 8 steps; the records of finished envs are dropped by `active` below
            break
    act = torch.stack(rec["active"], 0).cpu().numpy()                       # [steps, n]
    tgt = torch.stack(rec["target"], 0).double().cpu().numpy(); pred = torch.stack(rec["pred"], 0).double().cpu().numpy()
    parked = np.zeros(35); parked[0::7] = [100.0 * (i + 1) for i in range(5)]; parked[1::7] = 100.0
    objs = None if rec["obj_pose"][0] is None else torch.stack(rec["obj_pose"], 0).double().cpu().numpy()
    out = {}
    for e, key in enumerate(keys):
        steps = np.nonzero(act[:, e])[0]
        out[key] = {"target": [tgt[t, e] for t in steps], "pred": [pred[t, e] for t in steps],
                    "obj_pose": [(objs[t, e] if objs is not None else parked.copy()) for t in steps],
                    "percent": float(percent[e]), "fail_safe": bool(used_fs[e])}
    return out


@torch.no_grad()
def eval_dataset(env, policy_net, ctx_builder, dataset, fail_safe=False, inds=None):
    """`eval_seq` for every take of a data set (agent_ar.py:463-503; eval_ar_policy.py's run_seq loop :178-262), env.n takes at a time: each
    env plays one WHOLE take (`get_seq_by_ind(ind, full_sample=True)`), takes of different length share a batch padded to its longest
    (`ctx['len']` ends every env on its own last frame; init_context averages every row's context over its own frames).  ctx_builder:
    PolicyARContext (its kinematic twin may hold any number of rows: other batch sizes go through it in chunks); the kinematic roll-out is
    computed (need_rollout=True: the fail-safe and ar_mode read it).  Returns {take name: seq_result} in the data set's order."""
    inds = list(range(dataset.get_len())) if inds is None else [int(i) for i in inds]
    out = {}
    for i in range(0, len(inds), env.n):
        chunk = inds[i:i + env.n]
        rows = chunk + [chunk[-1]] * (env.n - len(chunk))                  # a short last chunk is filled with copies of its last take
        data = dataset.batch(np.asarray(rows, np.int64), None, None)
        data = {k: (v.to(env.device) if torch.is_tensor(v) else v) for k, v in data.items()}
        ctx = ctx_builder.init_context(data, need_rollout=True)
        env.load_context(ctx)
        keys = [dataset.takes[j] for j in chunk] + [f"__fill_{j}" for j in range(env.n - len(chunk))]
        res = run_sequences(env, policy_net, keys, fail_safe=fail_safe)
        out.update({k: res[k] for k in keys[:len(chunk)]})
    return out


def write_coverage(results: dict, result_dir: str, iter_num: int, data_file: str):
    """Write `<iter>_<data_file>_coverage.pkl` / `_coverage_full.pkl` (eval_ar_policy.py:225-262); returns the coverage count."""
    import joblib
    cov = {k: {"percent": r["percent"], "values": r.get("values", []), "fail_safe": r["fail_safe"]} for k, r in results.items()}
    coverage = sum(1 for r in results.values() if r["percent"] == 1 and not r["fail_safe"])
    os.makedirs(result_dir, exist_ok=True)
    joblib.dump(cov, os.path.join(result_dir, f"{iter_num:04d}_{data_file}_coverage.pkl"))
    joblib.dump(results, os.path.join(result_dir, f"{iter_num:04d}_{data_file}_coverage_full.pkl"))
    return coverage
