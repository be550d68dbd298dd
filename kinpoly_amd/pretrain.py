"""The supervised warm start of the kinematic policy: AgentAR.train_init (kin_poly/core/agent_ar.py:366-385) =
PolicyAR.update_init_supervised x 500 epochs + PolicyAR.train_full_supervised(scheduled_sampling=0.3) x 50 epochs
(kin_poly/models/policy_ar.py:243-275), before the first PPO iteration.

Off the rollout hot path (it runs once, at epoch 0) but differentiable end to end: the kinematic roll-out of a clip
(TrajARNet.forward in train form, traj_ar_smpl_net.py:346-383) -- observation (:203-290), GRU + MLP, kinematic step (:292-330), scheduled sampling
-- is written in torch ops on the device so that autograd reaches the context network and the action network through all T frames; the HIP
roll-out of kinpoly_amd/context.py is the same computation without a tape (both are held to the reference's fixtures: traj_ar_net.npz,
pretrain.npz).  Losses: compute_loss (:390-457) and compute_loss_init (:499-527) with model_specs' weights.
"""
from __future__ import annotations

import numpy as np
import torch

from .context import get_qvel_fd_batch, heading_q, quat_inv, quat_mul, quat_rotate_t
from .supervised import TorchFK, kinematic_step

HEAD = 13                 # body "Head" of the SMPL tree (fk_model.get_head_idx(), torch_smpl_humanoid.py:43-44)
LOSS_WEIGHTS = dict(w_rp=50.0, w_rr=50.0, w_p=1.0, w_v=1.0, w_ee=10.0, w_op=1.0, w_or=10.0)        # config/statear/kin_poly.yml:28-34


def _one_hot_at(data, t):
    oh = data["action_one_hot"]
    return oh if oh.dim() == 2 else oh[:, t]


def observe(fk: TorchFK, qpos, data, t, noise_std=0.0, generator=None):
    """TrajARNet.get_obs (:203-290) for kin_poly.yml (use_head, use_action; no use_vel / use_of / use_context), differentiable in qpos.
    Returns (obs [B, 105], features: pred_wbpos [B,72], obj_2_head [B,7])."""
    wbpos = fk.wbpos(qpos)                     # float32 device rows: k_target_fk forward / k_fk_wbpos_grad backward (supervised.TorchFK)
    hpos, hrot = wbpos[:, HEAD], fk.body_quat(qpos, HEAD)
    local = torch.cat([qpos[:, 2:3], quat_mul(quat_inv(heading_q(qpos[:, 3:7])), qpos[:, 3:7]), qpos[:, 7:]], 1)        # height, de-headed root, pose: 74
    t_hpos, t_hrot = data["head_pose"][:, t, :3], data["head_pose"][:, t, 3:]
    t_hlvel, t_havel = data["head_vels"][:, t, :3], data["head_vels"][:, t, 3:]
    t_obj = data["obj_head_relative_poses"][:, t]
    if noise_std > 0.0:                        # cfg.add_noise in train mode (:235-240): N(0, noise_std) on every target-head quantity
        n = lambda x: x + torch.randn(x.shape, device=x.device, dtype=x.dtype, generator=generator) * noise_std      # noqa: E731
        t_hrot, t_hpos, t_havel, t_hlvel, t_obj = n(t_hrot), n(t_hpos), n(t_havel), n(t_hlvel), n(t_obj)
    diff_hpos = quat_rotate_t(heading_q(hrot), t_hpos - hpos)
    diff_hrot = quat_mul(quat_inv(t_hrot), hrot)
    obj = data["obj_pose"][:, t]
    obj_rel = torch.cat([quat_rotate_t(heading_q(hrot), obj[:, :3] - hpos), quat_mul(quat_inv(heading_q(hrot)), obj[:, 3:7])], 1)
    obs = torch.cat([local, diff_hpos, diff_hrot, obj_rel, t_havel, t_hlvel, t_obj, _one_hot_at(data, t)], 1)
    return obs, wbpos.reshape(qpos.shape[0], 72), obj_rel


def forward_supervised(net, fk: TorchFK, data, gt_rate=0.0, rng=None, noise_std=0.0, generator=None, dt=1.0 / 30.0):
    """TrajARNet.forward (:346-383): init_states -> T observations / actions / kinematic steps with autograd; with gt_rate > 0 the simulated
    pose is put back on the clip's GT pose with that probability at the initial state and after every step (scheduled sampling; `rng` =
    np.random.RandomState-like with .binomial, default numpy's global one as in the reference).  Returns the reference's feature_pred:
    qpos [B,T,76], qvel [B,T,75] (after fix_qvel), action [B,T,80], pred_wbpos [B,T,72], obj_2_head [B,T,7]."""
    rng = np.random if rng is None else rng
    B, T = data["qpos"].shape[:2]
    qpos, qvel, _ = net.init_states(data, keep_feat=False)
    if gt_rate > 0.0 and rng.binomial(1, gt_rate):
        qpos, qvel = data["qpos"][:, 0], data["qvel"][:, 0]
    hx = torch.zeros((B, net.rnn_hdim), device=qpos.device, dtype=qpos.dtype)
    Q, V, A, W, O = [], [], [], [], []
    for t in range(T):
        obs, wb, orel = observe(fk, qpos, data, t, noise_std, generator)
        Q.append(qpos); V.append(qvel); W.append(wb); O.append(orel)
        action, hx = net.get_action(obs, hx)
        A.append(action)
        if t == T - 1:
            break
        nxt = kinematic_step(qpos, action, dt)
        qvel = get_qvel_fd_batch(qpos, nxt, dt)
        qpos = nxt
        if gt_rate > 0.0 and rng.binomial(1, gt_rate):
            qpos, qvel = data["qpos"][:, t + 1], data["qvel"][:, t + 1]
    V = torch.stack(V, 1)
    return {"qpos": torch.stack(Q, 1), "qvel": torch.cat([V[:, 1:], V[:, -2:-1]], 1), "action": torch.stack(A, 1),
            "pred_wbpos": torch.stack(W, 1), "obj_2_head": torch.stack(O, 1)}


def _orientation_loss(gt_q, pred_q):
    """compute_loss.py:38-43 / 54-61: (|gt (x) pred^-1| - identity)^2 summed"""
    d = quat_mul(gt_q, quat_inv(pred_q)).abs()
    iden = torch.tensor([1.0, 0.0, 0.0, 0.0], device=d.device, dtype=d.dtype)
    return (d - iden).pow(2).sum(1)


def compute_loss(pred: dict, data: dict, weights=None):
    """TrajARNet.compute_loss (:390-457).  data['wbpos'] [B,T,72] is the feature file's GT joint positions.  -> (loss, 8 components)."""
    w = {**LOSS_WEIGHTS, **(weights or {})}
    B, T = pred["qpos"].shape[:2]
    pq, gq = pred["qpos"].reshape(B * T, -1), data["qpos"].reshape(B * T, -1)
    pv, gv = pred["qvel"][:, :-1].reshape(B * (T - 1), -1), data["qvel"][:, 1:].reshape(B * (T - 1), -1)        # GT qvel is one step ahead
    po, go = pred["obj_2_head"].reshape(B * T, -1), data["obj_head_relative_poses"].reshape(B * T, -1)
    r_pos = (gq[:, :3] - pq[:, :3]).pow(2).sum(1).mean()
    r_rot = _orientation_loss(gq[:, 3:7], pq[:, 3:7]).mean()
    p_rot = (gq[:, 7:] - pq[:, 7:]).pow(2).sum(1).mean()
    vl = (gv[:, :3] - pv[:, :3]).pow(2).sum(1).mean()
    va = (gv[:, 3:6] - pv[:, 3:6]).pow(2).sum(1).mean()
    ee = (data["wbpos"].reshape(B * T, -1) - pred["pred_wbpos"].reshape(B * T, -1)).pow(2).sum(1).mean()
    o_pos = (go[:, :3] - po[:, :3]).pow(2).sum(1).mean()
    o_rot = _orientation_loss(go[:, 3:7], po[:, 3:7]).mean()
    loss = w["w_rp"] * r_pos + w["w_rr"] * r_rot + w["w_p"] * p_rot + w["w_v"] * (vl + va) + w["w_ee"] * ee + w["w_op"] * o_pos + w["w_or"] * o_rot
    return loss, [r_pos, r_rot, p_rot, vl, va, ee, o_pos, o_rot]


def compute_loss_init(fk: TorchFK, pred_qpos, gt_qpos, weights=None):
    """TrajARNet.compute_loss_init (:499-527): root position / orientation, pose and joint positions of the predicted first frame (the
    predicted qvel enters the reference's signature but not its loss).  -> (loss, 4 components)."""
    w = {**LOSS_WEIGHTS, **(weights or {})}
    r_pos = (gt_qpos[:, :3] - pred_qpos[:, :3]).pow(2).sum(1).mean()
    r_rot = _orientation_loss(gt_qpos[:, 3:7], pred_qpos[:, 3:7]).mean()
    p_rot = (gt_qpos[:, 7:] - pred_qpos[:, 7:]).pow(2).sum(1).mean()
    with torch.no_grad():
        gt_wb = fk.wbpos(gt_qpos)
    ee = (gt_wb - fk.wbpos(pred_qpos)).reshape(pred_qpos.shape[0], -1).pow(2).sum(1).mean()
    return w["w_rp"] * r_pos + w["w_rr"] * r_rot + w["w_p"] * p_rot + w["w_ee"] * ee, [r_pos, r_rot, p_rot, ee]


def job_wide_rng(epoch, seed=4):
    """The scheduled-sampling coins of one call of train_full_supervised, identical on every rank: a RandomState seeded by (seed, epoch) with NO rank
    offset.  The reference draws them from numpy's global generator in its single process; with one process per GPU every rank must throw the
    same frames back onto the GT clip, or ranks disagree on which parameters got a gradient.  epoch -1 = the warm start."""
    return np.random.RandomState((int(seed) * 1000003 + int(epoch) + 2) % (2 ** 31 - 1))


def sampling_batches(dataset, num_samples, batch_size, device, dtype=None):
    """DatasetBatch.sampling_generator (statear_smpl_dataset.py:378-399): `num_samples` takes drawn with replacement from freq_indices (a take is
    listed once per fr_num frames of its length), a uniform window of fr_num frames each, served in shuffled batches of `batch_size`."""
    inds = dataset.rng.choice(dataset.freq_indices, size=num_samples)
    hi = np.maximum(dataset._seq_lens()[inds] - dataset.fr_num, 1)
    starts = np.minimum((dataset.rng.random_sample(num_samples) * hi).astype(np.int64), hi - 1)
    order = dataset.rng.permutation(num_samples)
    for i in range(0, num_samples, batch_size):
        sel = order[i:i + batch_size]
        b = dataset.batch(inds[sel], starts[sel], dataset.fr_num)
        # dtype: the network's (fp64 master copies of --update_dtype fp64 read fp32 data sets: floating tensors are cast, index tensors are not)
        yield {k: ((v.to(device, dtype) if (dtype is not None and v.is_floating_point()) else v.to(device)) if torch.is_tensor(v) else v) for k, v in b.items()}


def update_init_supervised(net, optimizer, fk: TorchFK, dataset, num_epoch=500, num_sample=2000, batch_size=256, weights=None, grad_allreduce=None):
    """PolicyAR.update_init_supervised (:261-275): the context network learns to predict the clip's first pose."""
    dev = next(net.parameters()).device
    last = None
    for _ in range(num_epoch):
        for data in sampling_batches(dataset, num_sample, batch_size, dev, next(net.parameters()).dtype):
            pred_qpos, _, _ = net.init_states(data, keep_feat=False)
            loss, _ = compute_loss_init(fk, pred_qpos, data["qpos"][:, 0], weights)
            optimizer.zero_grad(); loss.backward()
            if grad_allreduce is not None:
                grad_allreduce([p for p in net.parameters() if p.requires_grad])
            optimizer.step()
            last = loss.detach()
    return None if last is None else float(last)


def train_full_supervised(net, optimizer, fk: TorchFK, dataset, num_epoch=50, scheduled_sampling=0.3, num_sample=2000, batch_size=256, weights=None,
                          noise_std=0.0, scheduler=None, rng=None, grad_allreduce=None):
    """PolicyAR.train_full_supervised (:243-258): whole-clip roll-outs against the GT clip with scheduled sampling; `scheduler.step()` per epoch."""
    dev = next(net.parameters()).device
    last = None
    for _ in range(num_epoch):
        for data in sampling_batches(dataset, num_sample, batch_size, dev, next(net.parameters()).dtype):
            pred = forward_supervised(net, fk, data, scheduled_sampling, rng, noise_std)
            loss, _ = compute_loss(pred, data, weights)
            optimizer.zero_grad(); loss.backward()
            if grad_allreduce is not None:
                grad_allreduce([p for p in net.parameters() if p.requires_grad])
            optimizer.step()
            last = loss.detach()
        if scheduler is not None:
            scheduler.step()
    return None if last is None else float(last)
