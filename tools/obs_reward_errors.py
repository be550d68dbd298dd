"""Where do the observation / reward kernels lose digits?  HIP vs the fp64 oracle on the simulator's own states, per feature block of the 784-d UHC
observation, the 105-d kinematic-policy observation and the six reward terms -- on the golden fixture states (large target distances) and on
tracking states (target = current pose + 1e-3: the differences of near-equal numbers).    python tools/obs_reward_errors.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from kinpoly_amd import sim as kp  # noqa: E402
from kinpoly_amd.model_compiler import DEFAULT_KPM, read_kpm  # noqa: E402
from oracle import np_oracle as O  # noqa: E402

KPM = read_kpm(DEFAULT_KPM)
BODY_POS, BODY_IPOS, PARENT = KPM["body_pos"].reshape(24, 3), KPM["body_ipos"].reshape(24, 3), KPM["body_parent"]
dev = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device="cuda")  # noqa: E731
CC_BLOCKS = [("hq", 0, 4), ("target qpos[2:]", 4, 78), ("cur qpos[2:]", 78, 152), ("diff", 152, 226), ("qvel", 226, 301), ("rel_heading", 301, 302), ("rel_pos", 302, 304),
             ("body pos", 304, 376), ("d joint pos", 376, 448), ("body com", 448, 520), ("d com", 520, 592), ("body quat", 592, 688), ("rel quat", 688, 784)]
AR_BLOCKS = [("cur qpos[2:]", 0, 74), ("d head pos", 74, 77), ("d head rot", 77, 81), ("obj rel head", 81, 88), ("head vels", 88, 94), ("target obj rel", 94, 101), ("one hot", 101, 105)]


def report(tag, g_qpos, g_qvel, g_target, steps):
    n = len(g_qpos)
    rng = np.random.default_rng(6)
    sim = kp.KpSim(kp.KpModel(), n)
    sim.set_state(dev(g_qpos), dev(g_qvel)); sim.set_target(dev(g_target))
    sim.step_begin()
    sim.step_ctrl(dev(rng.normal(size=(n, 75)) * 0.05), steps)
    T = 6
    t = np.full(n, 3, np.int32)
    head_pose = rng.normal(size=(n, T, 7)); head_vels = rng.normal(size=(n, T, 6)); obj_rel = rng.normal(size=(n, T, 7))
    rd = {k: sim.get(k).cpu().numpy().astype(np.float64) for k in ("qpos", "qvel", "xpos", "xquat", "xipos", "target_qpos", "prev_bquat", "prev_hpos")}
    # ground truth of the clip = the current pose + a small perturbation (what a tracking policy sees), previous frame = the pre-step pose
    gt_bquat = np.tile(np.array([1.0, 0, 0, 0]), (n, T, 24)); gt_wbpos = np.zeros((n, T, 72))
    for i in range(n):
        head_pose[i, 3, :3] = rd["xpos"][i, 39:42] + rng.normal(size=3) * 1e-3
        head_pose[i, 3, 3:] = rd["xquat"][i, 52:56]
        qn = rd["qpos"][i].copy(); qn[7:] += rng.normal(size=69) * 1e-3
        gt_bquat[i, 3] = O.get_body_quat(qn); gt_bquat[i, 2] = rd["prev_bquat"][i]
        gt_wbpos[i, 3] = rd["xpos"][i] + rng.normal(size=72) * 1e-3
    onehot = np.zeros((n, 4)); onehot[:, 0] = 1
    obj7 = np.tile(np.array([0.3, 0.2, 0.4, 1, 0, 0, 0.0]), (n, 1))
    ctx = sim.make_ctx(T, dev(head_pose), dev(head_vels), dev(obj_rel), dev(onehot), dev(gt_bquat), dev(gt_wbpos), torch.tensor(t, dtype=torch.int32, device="cuda"), obj_qpos=dev(obj7))
    cc = sim.obs_cc().cpu().numpy().astype(np.float64)
    ar = sim.obs_ar(ctx).cpu().numpy().astype(np.float64)
    rew, info, fail, diffs = sim.term_reward(ctx, kp.KpRewardCfg.default())
    rew, info, diffs = rew.cpu().numpy().astype(np.float64), info.cpu().numpy().astype(np.float64), diffs.cpu().numpy().astype(np.float64)
    r32 = lambda x: np.asarray(x, np.float32).astype(np.float64)  # noqa: E731
    e_cc = np.zeros((n, 784)); e_ar = np.zeros((n, 105)); e_info = np.zeros((n, info.shape[1])); e_rew = np.zeros(n); e_diff = np.zeros((n, 2))
    for i in range(n):
        xpos, xquat = rd["xpos"][i].reshape(24, 3), rd["xquat"][i].reshape(24, 4)
        tg = O.qpos_fk(rd["target_qpos"][i], BODY_POS, BODY_IPOS, PARENT)
        e_cc[i] = np.abs(cc[i] - O.obs_cc(rd["qpos"][i], rd["qvel"][i], xpos, xquat, rd["xipos"][i].reshape(24, 3), tg))
        e_ar[i] = np.abs(ar[i] - O.obs_ar(rd["qpos"][i], xpos, xquat, r32(head_pose[i, 3]), r32(head_vels[i, 3]), r32(obj_rel[i, 3]), onehot[i], r32(obj7[i])))
        head = np.concatenate([xpos[13], xquat[13]])
        r, inf = O.dynamic_supervision_v1(head, rd["prev_hpos"][i], O.get_body_quat(rd["qpos"][i]), rd["prev_bquat"][i], xpos, tg, r32(head_pose[i, 3]),
                                          r32(gt_bquat[i, 3]), r32(gt_bquat[i, 2]), 1.0 / 30.0, O.REWARD_WEIGHTS)
        e_info[i] = np.abs(info[i] - inf); e_rew[i] = abs(rew[i] - r)
        DIFFW = KPM["body_diffw"] if "body_diffw" in KPM else np.ones(24)
        e_diff[i] = np.abs(diffs[i] - [O.calc_body_diff(xpos, tg["wbpos"], DIFFW), O.calc_body_diff(xpos, r32(gt_wbpos[i, 3]).reshape(24, 3), DIFFW)])
    print(f"== {tag}: {n} states")
    print("   obs_cc |error| max per block: " + ", ".join(f"{nm} {e_cc[:, a:b].max():.1e}" for nm, a, b in CC_BLOCKS))
    print("   obs_ar |error| max per block: " + ", ".join(f"{nm} {e_ar[:, a:b].max():.1e}" for nm, a, b in AR_BLOCKS))
    print("   reward terms |error| max: " + ", ".join(f"{x:.1e}" for x in e_info.max(0)) + f"; reward {e_rew.max():.1e}; body diffs {e_diff.max(0)[0]:.1e} / {e_diff.max(0)[1]:.1e}")
    print("   reward terms (oracle, env 0): " + ", ".join(f"{x:.4f}" for x in inf))


g = np.load(os.path.join(ROOT, "tests", "golden", "env_funcs.npz"))
report("fixture states, far targets, one control step", g["qpos"], g["qvel"], g["target_qpos"], 15)
STD = np.load(os.path.join(ROOT, "tests", "golden", "standing_neutral.npz"))
rng = np.random.default_rng(3)
n = 64
q = np.tile(STD["qpos"], (n, 1)); q[:, 7:] += rng.normal(size=(n, 69)) * 0.05
tq = q.copy(); tq[:, 7:] += rng.normal(size=(n, 69)) * 1e-3; tq[:, :2] += rng.normal(size=(n, 2)) * 1e-3
report("tracking states: target = pose + 1e-3, one control step", q, np.zeros((n, 75)), tq, 15)
report("tracking states, one SUBSTEP (bodies turn 1e-4 rad between the two reward frames)", q, np.zeros((n, 75)), tq, 1)
