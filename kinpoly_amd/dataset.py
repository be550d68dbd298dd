"""The data formats either side of the rollout path: the per-take feature file the reference trains from
(`<data_dir>/features/<data_file>.p`, a joblib dict  take -> {qpos, qvel, head_pose, head_vels, obj_pose,
obj_head_relative_poses, action_one_hot, wbpos, wbquat, bquat, of_files, ...}, written by
kin_poly/data_process/process_smpl.py:140-235) and the sampler that serves clips from it
(kin_poly/data_loaders/statear_smpl_dataset.py), here batched: N episodes per call instead of one.

    build_take_features   process_smpl.post_process_expert on top of the batched get_expert (one FK launch per take)
    StateARDataset        preprocess_data (derived `target` trajectory), sample_seq x N -> [N, fr_num, .] tensors,
                          adaptive take sampling (freq_dict / ewma), get_seq_by_ind / iter_seq, padded ragged batches
    synthetic_takes       SURVEY.md section 8(d) config 4 stand-in for the absent MoCap set, in the same schema
"""
from __future__ import annotations

import math
import os

import numpy as np
import torch

from .context import heading_q, quat_acos_w, quat_inv, quat_mul, quat_rotate_t, quat_sin_half, quat_small

ACTIONS = ("sit", "push", "avoid", "step")             # cfg.all_actions order = action_one_hot columns
TRAIN_KEYS = ("wbpos", "wbquat", "bquat")


# ------------------------------------------------------------------ feature construction (torch, any device, any float dtype)
def _fd_vel(cur7, nxt7, dt):
    """get_head_vel / get_root_vel body (process_smpl.py:30-55, statear_smpl_dataset.py:186-214): linear velocity in the
    heading frame, angular velocity (axis * angle / dt, angle wrapped once) in the root frame; [T-1, 6]."""
    v = quat_rotate_t(heading_q(cur7[:, 3:7]), (nxt7[:, :3] - cur7[:, :3]) / dt)
    qrel = quat_mul(nxt7[:, 3:7], quat_inv(cur7[:, 3:7]))
    w = qrel[:, 0]
    small = quat_small(qrel)                          # 1 - |w| < 1e-8, sqrt(1 - w^2), acos(w): in the forms that keep a slow turn's digits in fp32
    s = quat_sin_half(qrel).clamp_min(1e-30)
    angle = torch.where(small, torch.zeros_like(w), 2 * quat_acos_w(qrel))
    axis = torch.where(small[:, None], torch.tensor([1.0, 0.0, 0.0], device=w.device, dtype=w.dtype).expand_as(qrel[:, 1:]), qrel[:, 1:] / s[:, None])
    angle = torch.where(angle > math.pi, angle - 2 * math.pi, angle)
    rv = quat_rotate_t(cur7[:, 3:7], axis * angle[:, None] / dt)
    return torch.cat([v, rv], 1)


def get_head_vel(pose7, dt=1.0 / 30.0):
    v = _fd_vel(pose7[:-1], pose7[1:], dt)
    return torch.cat([v, v[-1:]], 0)


def get_obj_relative_pose(obj_poses, ref_poses, num_objs=1):
    """process_smpl.py:110-135: object position in the reference's heading frame + heading^-1 (x) object quaternion."""
    qh = heading_q(ref_poses[:, 3:7])
    out = []
    for o in range(num_objs):
        out.append(quat_rotate_t(qh, obj_poses[:, 7 * o:7 * o + 3] - ref_poses[:, :3]))
        out.append(quat_mul(quat_inv(qh), obj_poses[:, 7 * o + 3:7 * o + 7]))
    return torch.cat(out, 1)


def get_traj_de_heading(qpos):
    """statear_smpl_dataset.py:153-181 with cfg.has_z: qpos[2:] with the root quaternion de-headed."""
    t = qpos[:, 2:].clone()
    t[:, 1:5] = quat_mul(quat_inv(heading_q(qpos[:, 3:7])), qpos[:, 3:7])
    return t


def build_take_features(sim, qpos, obj_pose=None, action: str | None = None, body_mass=None, dt=1.0 / 30.0) -> dict:
    """One take of the feature file from its qpos clip [T, 76] (+ object poses [T, 7k]): get_expert features, then
    post_process_expert (process_smpl.py:137-152, 217-226).  Returns numpy float64 arrays like the reference's file."""
    from .uhc_env import get_expert_batch
    q = torch.as_tensor(np.asarray(qpos), dtype=torch.float32)
    T = q.shape[0]
    mass = body_mass if body_mass is not None else torch.ones(24)
    ex = get_expert_batch(sim, q[None], torch.as_tensor(mass, dtype=torch.float32, device=sim.device), dt)
    out = {k: ex[k][0].double().cpu().numpy() for k in ("qpos", "qvel", "wbpos", "wbquat", "bquat", "head_pose", "body_com", "com", "ee_pos", "ee_wpos", "bangvel", "rq_rmh")}
    out["qpos"] = np.asarray(qpos, np.float64)
    if obj_pose is None:
        obj = np.tile(np.array([0.0, 0, 0, 1, 0, 0, 0]), (T, 1)); one_hot = np.zeros((T, 4))
    else:
        obj = np.asarray(obj_pose, np.float64); one_hot = np.zeros((T, 4)); one_hot[:, ACTIONS.index(action)] = 1.0
    nobj = obj.shape[1] // 7
    hp, ob = torch.as_tensor(out["head_pose"]), torch.as_tensor(obj)
    out.update(obj_pose=obj, action_one_hot=one_hot, action=action or "none", head_vels=get_head_vel(hp, dt).numpy(),
               obj_head_relative_poses=get_obj_relative_pose(ob, hp, nobj).numpy(),
               obj_root_relative_poses=get_obj_relative_pose(ob, torch.as_tensor(out["qpos"][:, :7]), nobj).numpy(),
               of_files=[f"{i:05d}.npy" for i in range(T)], len=T)
    return out


def write_features(path, takes: dict):
    import joblib
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    joblib.dump(takes, path)


# ------------------------------------------------------------------ the dataset
class StateARDataset:
    KEYS = ("qvel", "target", "qpos", "head_vels", "head_pose", "action_one_hot", "obj_head_relative_poses", "obj_pose")

    def __init__(self, features, takes=None, data_mode="train", fr_num=100, wild=False, dt=1.0 / 30.0, seed=0, device="cpu"):
        if isinstance(features, (str, os.PathLike)):
            import joblib
            features = joblib.load(features)
        self.rng = np.random.RandomState(seed)
        self.data_mode, self.fr_num, self.dt, self.wild, self.device = data_mode, int(fr_num), dt, wild, torch.device(device)
        self.takes, self.data = [], {k: [] for k in self.KEYS + (TRAIN_KEYS if data_mode == "train" and not wild else ())}
        for take in (takes if takes is not None else sorted(features)):
            e = features[take]
            q = torch.as_tensor(np.asarray(e["qpos"]), dtype=torch.float64)
            if data_mode == "train" and q.shape[0] < self.fr_num and not wild:
                continue                                        # :100-102
            assert len(e["of_files"]) == q.shape[0]
            target = torch.cat([get_traj_de_heading(q), torch.cat([_fd_vel(q[:-1, :7], q[1:, :7], dt)] + [_fd_vel(q[-2:-1, :7], q[-1:, :7], dt)], 0)], 1)
            row = dict(qvel=e["qvel"], target=target, qpos=q, head_vels=e["head_vels"], head_pose=e["head_pose"], action_one_hot=e["action_one_hot"],
                       obj_head_relative_poses=np.asarray(e["obj_head_relative_poses"])[:, :7], obj_pose=e["obj_pose"])
            for k in self.data:
                v = row[k] if k in row else e[k]
                v = torch.as_tensor(np.asarray(v) if not torch.is_tensor(v) else v, dtype=torch.float32)
                if k == "obj_pose" and v.shape[1] < 14:       # one width for every take (push carries two objects): zero-padded, the env
                    v = torch.cat([v, torch.zeros((v.shape[0], 14 - v.shape[1]))], 1)      # reads only the action's slice (convert_obj_qpos)
                self.data[k].append(v)
            self.takes.append(take)
        self.freq_indices = np.array([i for i, q in enumerate(self.data["qpos"]) for _ in range(int(np.ceil(q.shape[0] / self.fr_num)))])
        self.all_indices = list(range(len(self.takes)))
        self.traj_dim = self.data["target"][0].shape[1] if self.takes else 0
        self.counter = 0

    def get_len(self):
        return len(self.takes)

    def get_seq_len(self, ind):
        return self.data["qpos"][ind].shape[0]

    def get_seq_key(self, ind):
        return self.takes[ind]

    def _slice(self, ind, start, end):
        return {k: v[ind][start:end] for k, v in self.data.items()}

    def take_probs(self, freq_dict, sampling_temp=0.5):
        """:281-286: exp(-ewma(success history) / T), normalised; takes without history get ewma 0.  The reference's recursion
        avg <- alpha x_i + (1 - alpha) avg from avg = x_0 in closed form: (1 - alpha)^(n-1) x_0 + sum_i>0 alpha (1 - alpha)^(n-1-i) x_i."""
        alpha = 0.05
        e = np.zeros(len(freq_dict))
        for j, k in enumerate(freq_dict):
            h = freq_dict[k]
            if len(h) > 0:
                x = (np.asarray(h, np.float64)[:, 0] == 1).astype(np.float64)
                w = alpha * (1.0 - alpha) ** np.arange(len(x) - 1, -1, -1.0)
                w[0] = (1.0 - alpha) ** (len(x) - 1)
                e[j] = float(w @ x)
        p = np.exp(-e / sampling_temp)
        return p / p.sum()

    @property
    def has_objects(self):
        """does any take carry an action object (action_one_hot != 0)?  host-side, evaluated once"""
        if getattr(self, "_has_obj", None) is None:
            self._has_obj = any(bool((a.abs().sum() > 0).item()) for a in self.data["action_one_hot"])
        return self._has_obj

    def sample_batch(self, n, freq_dict=None, use_freq=True, full_sample=False, sampling_temp=0.5, sampling_freq=0.9, probs=None):
        """n independent `sample_seq` draws (:264-327) -> dict of [n, fr_num, .] tensors (+ 'take_ind', 'fr_start').  The draws are made as
        arrays (one rng call per quantity, not per row): the same distributions as n sequential sample_seq calls."""
        starts = np.zeros(n, np.int64)
        if use_freq and freq_dict is None:
            inds = self.rng.choice(self.freq_indices, size=n)
        elif use_freq:
            probs = self.take_probs(freq_dict, sampling_temp) if probs is None else probs      # probs: the caller's cached take_probs(freq_dict)
            coin = self.rng.binomial(1, sampling_freq, size=n).astype(bool)
            inds = np.where(coin, self.rng.choice(self.all_indices, size=n, p=probs), self.rng.choice(self.all_indices, size=n))
            if not full_sample:
                hi = np.maximum(self._seq_lens()[inds] - self.fr_num, 1)
                starts = np.minimum((self.rng.random_sample(n) * hi).astype(np.int64), hi - 1)
        else:
            inds = self.rng.choice(self.all_indices, size=n)
        return self.batch(np.asarray(inds, np.int64), starts, None if full_sample else self.fr_num)

    def _seq_lens(self):
        if getattr(self, "_lens_np", None) is None or len(self._lens_np) != len(self.takes):
            self._lens_np = np.array([q.shape[0] for q in self.data["qpos"]], np.int64)
        return self._lens_np

    def _flat_store(self):
        """every take's rows back to back on self.device, per key [sum T, dim], + the takes' offsets: a batch is one gather per key"""
        if getattr(self, "_flat", None) is None or self._flat_n != len(self.takes):
            lens = self._seq_lens()
            self._flat_off = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int64) if len(lens) else np.zeros(0, np.int64)
            self._flat = {k: torch.cat(v, 0).to(self.device) for k, v in self.data.items()} if len(lens) else {}
            self._flat_n = len(self.takes)
        return self._flat, self._flat_off

    def batch(self, inds, starts=None, length=None):
        """Rows (take, start) as one padded batch; `len` holds each row's valid frame count (ragged when length is None).  Rows shorter than
        the batch's longest are padded with their last frame."""
        inds = np.asarray(inds, np.int64); starts = np.zeros(len(inds), np.int64) if starts is None else np.asarray(starts, np.int64)
        avail = self._seq_lens()[inds] - starts
        lens = np.minimum(avail, length) if length else avail
        T = int(lens.max())
        flat, off = self._flat_store()
        idx = (off[inds] + starts)[:, None] + np.minimum(np.arange(T)[None, :], (lens - 1)[:, None])             # [n, T] rows of the flat store
        idx_t = torch.as_tensor(idx, device=self.device)
        out = {k: v[idx_t] for k, v in flat.items()}
        out["len"] = torch.as_tensor(lens.astype(np.int32), device=self.device)
        out["ragged"] = bool(lens.min() < T)           # host-side flag: rows are padded (init_context then averages every row over its own frames)
        out["take_ind"], out["fr_start"] = torch.as_tensor(inds), torch.as_tensor(starts)
        return out

    def get_seq_by_ind(self, ind, full_sample=False):
        return self.batch([ind], [0], None if full_sample else self.fr_num)

    def iter_seq(self):
        ind = self.counter % len(self.takes)
        self.counter += 1
        self.curr_key = self.takes[ind]
        return self.batch([ind], [0], None)

    def set_seq_counter(self, idx):
        self.counter = idx


# ------------------------------------------------------------------ synthetic stand-in for the absent MoCap set
def synthetic_takes(sim, std_qpos, n_per_action=2, T_range=(110, 160), body_mass=None, seed=0, with_objects=True, amp_max=0.3):
    """standing -> seeded smooth joint-space sinusoids (amplitude <= 0.3 rad, <= 1 Hz), four action
    classes with their object(s) at constant poses in front of / behind the humanoid, yaw U(-pi, pi).  with_objects=False: the same
    motions as takes without an action (obj_pose = [0,0,0,1,0,0,0], action_one_hot = 0; process_smpl.py:223-225) -- config 3's
    object-free MoCap clips.  amp_max: the sinusoids' amplitude bound (0.3 rad = SURVEY's figure; the pelvis does not move, so at that amplitude the
    legs swing the feet through the floor and no controller can follow the clip for long -- fine as a workload, not as a learning target)."""
    rng = np.random.default_rng(seed)
    std_qpos = np.asarray(std_qpos, np.float64)
    obj_local = {"sit": [[0.0, -0.6, 0.3805]], "push": [[0.0, 0.8, 0.921], [0.0, 0.8, 0.7905]], "avoid": [[0.0, 1.0, 0.69]], "step": [[0.0, 0.8, 0.3705]]}
    takes = {}
    for a in ACTIONS:
        for j in range(n_per_action):
            T = int(rng.integers(*T_range))
            yaw = rng.uniform(-np.pi, np.pi)
            qz = np.array([np.cos(yaw / 2), 0, 0, np.sin(yaw / 2)])
            q = np.tile(std_qpos, (T, 1))
            w, x, y, z = std_qpos[3:7]
            q[:, 3:7] = [qz[0] * w - qz[3] * z, qz[0] * x - qz[3] * y, qz[0] * y + qz[3] * x, qz[0] * z + qz[3] * w]     # qz (x) q_root
            amp, fr, ph = rng.uniform(0, amp_max, 69), rng.uniform(0.1, 1.0, 69), rng.uniform(0, 2 * np.pi, 69)
            tt = np.arange(T)[:, None] / 30.0
            q[:, 7:] += amp * (np.sin(2 * np.pi * fr * tt + ph) - np.sin(ph)) * np.minimum(tt / 0.5, 1.0)
            c, s_ = np.cos(yaw), np.sin(yaw)
            obj = []
            for lx, ly, lz in obj_local[a]:
                obj += [std_qpos[0] + c * lx - s_ * ly, std_qpos[1] + s_ * lx + c * ly, lz, *qz]
            if with_objects:
                takes[f"{a}-synthetic-{j:02d}"] = build_take_features(sim, q, np.tile(np.array(obj), (T, 1)), a, body_mass)
            else:
                takes[f"none-synthetic-{a}-{j:02d}"] = build_take_features(sim, q, None, None, body_mass)
    return takes
