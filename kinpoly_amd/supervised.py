"""Supervised one-step update of the kinematic policy (`step_update: true`, config/statear/kin_poly.yml:65,70):
PolicyAR.update_supervised_step (kin_poly/models/policy_ar.py:277-287) = policy forward -> kinematic step from the
recorded sim pose -> TrajARNet.compute_loss_lite against the GT next pose (traj_ar_smpl_net.py:459-497).

Everything here is differentiable torch on the device (autograd through the kinematic step and the forward
kinematics); the roll-out itself uses the HIP kernels, this runs only in the optimiser step.
"""
from __future__ import annotations

import torch

from .context import heading_q, quat_inv, quat_mul


def _qmat(q):
    """quaternion_matrix_batch: rotation matrix of q / |q| -> [..., 3, 3]."""
    q = q / q.norm(dim=-1, keepdim=True)
    w, x, y, z = q.unbind(-1)
    return torch.stack([torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], -1),
                        torch.stack([2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)], -1),
                        torch.stack([2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], -1)], -2)


def _qrot(q, v):
    """quat_mul_vec_batch (kin_poly/utils/torch_utils.py): v + 2 (w (u x v) + u x (u x v)) with q AS IT IS -- the reference does not normalise
    here, so a root quaternion that is unit to 1e-7 only (the data set's are) rotates the angular velocity with that error; reproduced
    (3.5e-7 on the fixture's poses, visible in fp64).  No matrix: a [B,3,3] x [B,3,1] batched GEMM of 98 k tiny matrices costs over a
    millisecond each way."""
    u = q[..., 1:]
    uv = torch.cross(u, v, dim=-1)
    return v + 2.0 * (q[..., :1] * uv + torch.cross(u, uv, dim=-1))


def quat_from_expmap(e):
    """quat_from_expmap_batch (kin_poly/utils/torch_utils.py:239-248)."""
    angle = e.norm(dim=1)
    safe = angle.clamp_min(1e-8)
    axis = torch.where((angle < 1e-8)[:, None], torch.tensor([1.0, 0.0, 0.0], device=e.device, dtype=e.dtype).expand_as(e), e / safe[:, None])
    return torch.cat([torch.cos(angle / 2)[:, None], axis * torch.sin(angle / 2)[:, None]], 1)


def kinematic_step(curr_qpos, action, dt=1.0 / 30.0):
    """TrajARNet.step (traj_ar_smpl_net.py:292-330), has_z, no pose_delta: -> next_qpos [B,76] (root quat normalised)."""
    rot = curr_qpos[:, 3:7]
    linv = _qrot(heading_q(rot), action[:, 74:77])
    angv = _qrot(rot, action[:, 77:80])
    new_rot = quat_mul(quat_from_expmap(angv * dt), rot)
    new_rot = new_rot / new_rot.norm(dim=1, keepdim=True)
    return torch.cat([curr_qpos[:, :2] + linv[:, :2] * dt, action[:, :1], new_rot, action[:, 5:74]], 1)


class _FKWbpos(torch.autograd.Function):
    """qpos [B,76] -> wbpos [B,24,3] through the library's kernels: k_target_fk forward, k_fk_wbpos_grad backward."""

    @staticmethod
    def forward(ctx, qpos, sim):
        q = qpos.contiguous()
        out = sim.fk(q)
        ctx.sim = sim
        ctx.save_for_backward(q, out["wbpos"], out["wbquat"])
        return out["wbpos"].view(-1, 24, 3)

    @staticmethod
    def backward(ctx, g):
        q, wbpos, wbquat = ctx.saved_tensors
        return ctx.sim.fk_backward(q, wbpos, wbquat, g.reshape(-1, 72).contiguous()), None


class TorchFK:
    """Differentiable Humanoid.qpos_fk (kin_poly/utils/torch_smpl_humanoid.py:125-202): world joint positions [B,24,3].
    The chain is walked per tree LEVEL (9 batched steps over [B, bodies of the level]) instead of per body (23 steps)."""

    def __init__(self, body_pos, body_parent, device, dtype=torch.float32, sim=None):
        self.sim = sim                         # a KpSim: float32 device rows then go through the HIP forward / backward kernels
        self.offsets = torch.as_tensor(body_pos, dtype=dtype, device=device).view(24, 3)
        self.parents = [int(p) for p in body_parent]
        depth = [0] * 24
        for i in range(1, 24):
            depth[i] = depth[self.parents[i]] + 1
        self.levels = []                       # per level: (body ids, parent ids) as index tensors
        for d in range(1, max(depth) + 1):
            ids = [i for i in range(24) if depth[i] == d]
            self.levels.append((torch.tensor(ids, device=device), torch.tensor([self.parents[i] for i in ids], device=device)))

    def wbpos(self, qpos):
        if self.sim is not None and qpos.is_cuda and qpos.dtype == torch.float32:
            return _FKWbpos.apply(qpos, self.sim)
        return self.wbpos_torch(qpos)

    def wbpos_torch(self, qpos):
        return self.chain_torch(qpos)[0]

    def body_quat(self, qpos, body):
        """world quaternion of ONE body [B, 4], differentiable: the product of the local 'rzyx' quaternions along its root path only (the
        supervised roll-out's observation reads the head's orientation every frame; the full tree walk is 23 bodies, the head's path 5)"""
        path = []
        b = int(body)
        while b > 0:
            path.append(b); b = self.parents[b]
        q = qpos[:, 3:7] / qpos[:, 3:7].norm(dim=1, keepdim=True)
        if not path:
            return q
        ids = torch.tensor(path[::-1], device=qpos.device)
        ang = qpos[:, 7:].view(qpos.shape[0], 23, 3)[:, ids - 1] * 0.5                      # [B, n, 3]
        s, c = torch.sin(ang), torch.cos(ang)
        z = torch.zeros_like(c[..., 0])
        qz = torch.stack([c[..., 0], z, z, s[..., 0]], -1); qy = torch.stack([c[..., 1], z, s[..., 1], z], -1); qx = torch.stack([c[..., 2], s[..., 2], z, z], -1)
        local = quat_mul(quat_mul(qz, qy), qx)
        for k in range(len(path)):
            q = quat_mul(q, local[:, k])
        return q

    def chain_torch(self, qpos):
        """(wbpos [B,24,3], wbquat [B,24,4]) by differentiable torch ops (the observation of the supervised roll-out reads the head's quaternion too)"""
        B = qpos.shape[0]
        root_q = qpos[:, 3:7] / qpos[:, 3:7].norm(dim=1, keepdim=True)
        ang = qpos[:, 7:].view(B, 23, 3) * 0.5
        s, c = torch.sin(ang), torch.cos(ang)
        z = torch.zeros_like(c[..., 0])
        qz = torch.stack([c[..., 0], z, z, s[..., 0]], -1); qy = torch.stack([c[..., 1], z, s[..., 1], z], -1); qx = torch.stack([c[..., 2], s[..., 2], z, z], -1)
        local = quat_mul(quat_mul(qz, qy), qx)                       # 'rzyx', [B, 23, 4]
        pos = [None] * 24; quat = [None] * 24
        pos[0], quat[0] = qpos[:, :3], root_q
        for ids, par in self.levels:
            pq = torch.stack([quat[p] for p in par.tolist()], 1)                               # [B, n, 4]
            pp = torch.stack([pos[p] for p in par.tolist()], 1)                                # [B, n, 3]
            npos = (_qmat(pq) @ self.offsets[ids][None, :, :, None])[..., 0] + pp
            nq = quat_mul(pq, local[:, ids - 1])
            for k, i in enumerate(ids.tolist()):
                pos[i], quat[i] = npos[:, k], nq[:, k]
        return torch.stack(pos, 1), torch.stack(quat, 1)


def compute_loss_lite(fk: TorchFK, pred_qpos, gt_qpos, w_rp=50.0, w_rr=50.0, w_p=1.0, w_ee=10.0, gt_wbpos=None):
    """TrajARNet.compute_loss_lite with kin_poly.yml weights (model_specs: w_rp 50, w_rr 50, w_p 1, w_ee 10)."""
    r_pos = (gt_qpos[:, :3] - pred_qpos[:, :3]).pow(2).sum(1)
    dist = quat_mul(gt_qpos[:, 3:7], quat_inv(pred_qpos[:, 3:7]))
    iden = torch.tensor([1.0, 0.0, 0.0, 0.0], device=dist.device, dtype=dist.dtype)
    r_rot = (dist.abs() - iden).pow(2).sum(1)
    p_rot = (gt_qpos[:, 7:] - pred_qpos[:, 7:]).pow(2).sum(1)
    ee = ((fk.wbpos(gt_qpos) if gt_wbpos is None else gt_wbpos) - fk.wbpos(pred_qpos)).reshape(pred_qpos.shape[0], -1).pow(2).sum(1)
    loss = w_rp * r_pos.mean() + w_rr * r_rot.mean() + w_p * p_rot.mean() + w_ee * ee.mean()
    return loss, [r_pos.mean(), r_rot.mean(), p_rot.mean(), ee.mean()]


def update_supervised_step(policy, optimizer, fk: TorchFK, batch, num_epoch=20, grad_allreduce=None, target=None, history=None):
    """batch: RolloutBatch with curr_qpos / gt_target_qpos recorded by VectorSampler(record_qpos=True).  target: another [N, T, 76] pose to regress
    the kinematic step onto -- `batch.res_qpos`, the pose the simulation reached, makes this PolicyAR.update_supervised_dyna (policy_ar.py:289-301).
    The update runs in the policy's dtype (fp64 master copies: the batch is read as fp64, `fk` must be an fp64 TorchFK).  history: a list that
    receives every epoch's loss (device scalars)."""
    N, T, _ = batch.states.shape
    dt = next(policy.parameters()).dtype
    states, hx0 = batch.states.to(dt), (None if batch.hx0 is None else batch.hx0.to(dt))
    curr, tgt = batch.curr_qpos.to(dt).reshape(N * T, 76), (batch.gt_target_qpos if target is None else target).to(dt).reshape(N * T, 76)
    loss_val = None
    with torch.no_grad():
        tgt_wbpos = fk.wbpos(tgt)              # the GT side of the end-effector term does not change between epochs
    for _ in range(num_epoch):
        means = policy.unroll(states, batch.episode_start, hx0).reshape(N * T, -1)
        loss, _ = compute_loss_lite(fk, kinematic_step(curr, means), tgt, gt_wbpos=tgt_wbpos)
        if history is not None:
            history.append(loss.detach())
        optimizer.zero_grad()
        loss.backward()
        if grad_allreduce is not None:
            grad_allreduce([p for p in policy.parameters() if p.requires_grad])
        optimizer.step()
        loss_val = loss.detach()
    return None if loss_val is None else float(loss_val)      # one host read per call, not one per epoch
