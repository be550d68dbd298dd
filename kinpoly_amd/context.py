"""Per-episode context pipeline, batched over environments: the vectorised `PolicyAR.init_context`
(kin_poly/models/policy_ar.py:124-182) on top of `TrajARNet.init_states / forward`
(kin_poly/models/traj_ar_smpl_net.py:157-201, 346-383).

Per episode the reference (i) runs a context GRU over the whole clip and predicts the initial pose / velocity,
(ii) rolls the kinematic policy over the whole clip (T GRU steps + T forward-kinematics calls), (iii) "smooths" the
roll-out and re-runs FK.  NOTE on (iii): with `cfg.smooth` the reference executes
`ar_qpos[:, 7:] = gaussian_filter1d(ar_qpos[:, 7:], 1, axis=0)` on a `[1, T, 76]` tensor (policy_ar.py:150-152): the slice takes
FRAMES 7.. (not joint columns) and the filter runs along the length-1 batch axis, which leaves every value unchanged
(fixture `smooth_effective.npz`, generated with the reference's statement).  The default here reproduces that effective
behaviour (ar_qpos untouched); `smooth_time_axis=True` is the documented deviation that really smooths the joint angles in time.  Here all N environments' clips go
through these three stages together: the GRUs are [N, .] GEMMs, and the kinematic roll-out reuses the HIP
kernels of the rollout path (`step_kin`, forward kinematics, `obs_ar`) on a second, physics-free `KpSim`.

Parameter names mirror the reference module (`action_rnn.rnn_f`, `action_mlp`, `action_fc`, `context_rnn.rnn_f`,
`context_mlp`, `context_fc`) so `policy_dict['traj_ar_net.*']` of a reference checkpoint loads directly.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from . import sim as kpsim
from .nets import MLP, KinPolicy, _StepRNN


def quat_mul(a, b):
    """Gohlke quaternion_multiply(a, b), batched (w, x, y, z)."""
    w1, x1, y1, z1 = a.unbind(-1); w0, x0, y0, z0 = b.unbind(-1)
    return torch.stack([-x1 * x0 - y1 * y0 - z1 * z0 + w1 * w0, x1 * w0 + y1 * z0 - z1 * y0 + w1 * x0,
                        -x1 * z0 + y1 * w0 + z1 * x0 + w1 * y0, x1 * y0 - y1 * x0 + z1 * w0 + w1 * z0], -1)


def quat_inv(q):
    return torch.cat([q[..., :1], -q[..., 1:]], -1) / (q * q).sum(-1, keepdim=True)


def heading_q(q):
    h = torch.zeros_like(q); h[..., 0] = q[..., 0]; h[..., 3] = q[..., 3]
    return h / h.norm(dim=-1, keepdim=True)


def quat_rotate_t(q, v):
    """transform_vec_batch(v, q, 'root'): R(q)^T v with R normalised by |q|^2."""
    qn = q / q.norm(dim=-1, keepdim=True)
    qc = torch.cat([qn[..., :1], -qn[..., 1:]], -1)
    u = qc[..., 1:]
    t = 2.0 * torch.cross(u, v, dim=-1)
    return v + qc[..., :1] * t + torch.cross(u, t, dim=-1)


# The reference evaluates sqrt(1 - w^2), acos(w) and 1 - |w| in fp64, where they are harmless.  In fp32 they lose a small rotation's digits (1 - w^2 of a
# 1e-3 rad turn is 2.5e-7, known to 6e-8), so below fp64 the same quantities are taken from |xyz| of the unit quaternion; in fp64 the reference's own
# expressions are kept, so that the fixtures generated from it (whose quaternions are unit to 1e-7 only) are met to the last digit.
def quat_sin_half(q):
    """sin(angle / 2) >= 0 of a unit quaternion [..., 4]: sqrt(1 - w^2) (fp64) / |xyz| (fp32)."""
    if q.dtype == torch.float64:
        return torch.sqrt((1 - q[..., 0] * q[..., 0]).clamp_min(0.0))
    return q[..., 1:].norm(dim=-1)


def quat_acos_w(q):
    """acos(w) of a unit quaternion [..., 4]: as written (fp64) / atan2(|xyz|, w) (fp32)."""
    if q.dtype == torch.float64:
        return torch.acos(q[..., 0].clamp(-1.0, 1.0))
    return torch.atan2(q[..., 1:].norm(dim=-1), q[..., 0])


def quat_small(q, eps=1e-8):
    """the reference's `1 - |w| < eps` (uhc/khrylib/utils/transformation.py:349): as written (fp64) / |xyz|^2 < eps (1 + |w|) (fp32)."""
    if q.dtype == torch.float64:
        return (1 - q[..., 0].abs()) < eps
    return (q[..., 1:] ** 2).sum(-1) < eps * (1 + q[..., 0].abs())


def get_qvel_fd_batch(cur_qpos, next_qpos, dt):
    """kin_poly/utils/torch_utils.py:315-331 (transform=None)."""
    v = (next_qpos[:, :3] - cur_qpos[:, :3]) / dt
    qrel = quat_mul(next_qpos[:, 3:7], quat_inv(cur_qpos[:, 3:7]))
    w = qrel[:, 0].clamp(-1.0, 1.0)
    # sin(acos(w)) and 2 acos(w) of rotation_from_quaternion_batch (:109-130).  Its safe_acos (:32-36) clamps w to +-(1 - 1e-7), so a rotation too
    # small for 1 - w to be seen comes out as xyz / sin(acos(1 - 1e-7)) * 2 acos(1 - 1e-7) = 2 xyz, the right small-angle limit (and the `< 1e-5`
    # branch can never be taken); fp64 keeps that form to the letter, fp32 gets the same limit from |xyz| and atan2
    if qrel.dtype == torch.float64:
        half = torch.acos(qrel[:, 0].clamp(-1.0 + 1e-7, 1.0 - 1e-7))
        s = torch.sin(half)
        small = s.abs() < 1e-5
        two_half = 2 * half
    else:
        s = quat_sin_half(qrel)
        small = ~(s > 0)
        two_half = 2 * quat_acos_w(qrel)
    s = s.clamp_min(1e-30)
    axis = torch.where(small[:, None], torch.tensor([1.0, 0.0, 0.0], device=w.device, dtype=w.dtype).expand_as(qrel[:, 1:]), qrel[:, 1:] / s[:, None])
    angle = torch.where(small, torch.zeros_like(w), two_half)
    angle = torch.where(angle > math.pi, angle - 2 * math.pi, angle)
    angle = torch.where(angle < -math.pi, angle + 2 * math.pi, angle)
    rv = quat_rotate_t(cur_qpos[:, 3:7], axis * angle[:, None] / dt)
    return torch.cat([v, rv, (next_qpos[:, 7:] - cur_qpos[:, 7:]) / dt], 1)


def gaussian_filter1d_time(x, sigma=1.0, truncate=4.0):
    """scipy.ndimage.gaussian_filter1d(x, sigma, axis=time) with its default 'reflect' boundary, for x [N, T, C]."""
    r = int(truncate * sigma + 0.5)
    k = torch.arange(-r, r + 1, device=x.device, dtype=x.dtype)
    w = torch.exp(-0.5 * (k / sigma) ** 2); w = w / w.sum()
    T = x.shape[1]
    idx = torch.arange(-r, T + r, device=x.device)
    period = 2 * T
    idx = idx % period
    idx = torch.where(idx >= T, period - 1 - idx, idx)          # half-sample symmetric extension (d c b a | a b c d | d c b a)
    xp = x[:, idx]                                              # [N, T + 2r, C]
    return sum(w[j] * xp[:, j:j + T] for j in range(2 * r + 1))


class TrajARNet(KinPolicy):
    """KinPolicy (the per-step part) + the context network of TrajARNet."""

    def __init__(self, state_dim=105, action_dim=80, context_dim=17, rnn_hdim=1024, mlp_hsize=(1024, 512, 256), htype="relu", log_std=-3.2):
        super().__init__(state_dim, action_dim, rnn_hdim, mlp_hsize, htype, log_std)
        self.context_dim, self.init_dim = context_dim, action_dim + 75
        self.context_rnn = _StepRNN(context_dim, rnn_hdim)
        self.context_mlp = MLP(rnn_hdim, mlp_hsize, htype)
        self.context_fc = nn.Linear(mlp_hsize[-1], self.init_dim)

    def _context_input(self, data):
        one_hot = data["action_one_hot"]
        T = data["head_vels"].shape[1]
        if one_hot.dim() == 2:
            one_hot = one_hot[:, None].expand(-1, T, -1)
        return torch.cat([data["obj_head_relative_poses"], data["head_vels"], one_hot], 2)

    def get_context_feat(self, data):
        """get_context_feat (:138-167): GRU over [obj_head_relative_poses, head_vels, action_one_hot] -> [N, T, rnn_hdim]."""
        feat = self._context_input(data)
        hx = torch.zeros((feat.shape[0], self.rnn_hdim), device=feat.device, dtype=feat.dtype)
        outs = []
        for t in range(feat.shape[1]):
            hx = self.context_rnn.rnn_f(feat[:, t], hx)
            outs.append(hx)
        return torch.stack(outs, 1)

    def get_context_mean(self, data):
        """context_feat_rnn.mean(dim=1) of init_states (:185-186) without keeping the [N, T, rnn_hdim] sequence: the only consumer of the
        sequence itself is the `use_context` / `use_of` observation block (humanoid_ar_v1.py:151-155), off in kin_poly.yml, and 4096 clips of 100
        frames would be 1.7 GB of it per draw.  On the device every step is the two gate GEMMs + kp_gru_cell_step (as the rollout's GRU step)."""
        feat = self._context_input(data)
        N0, T = feat.shape[:2]
        cell = self.context_rnn.rnn_f
        w = self._frame_weights(data, T, feat)                     # None unless the batch is ragged
        fast = feat.is_cuda and feat.dtype == torch.float32 and not torch.is_grad_enabled()
        if fast and N0 % 64:
            # a sampler's top-up draws however many clips were used up (497, 530, ...): the T recurrent GEMMs [N, H] x [H, 3H] then run on whatever kernel the
            # library's heuristics pick for that odd N (48 TFLOP/s picks were seen).  Rows are independent, so the batch is padded with zero rows to the next
            # multiple of 64 (regular tile counts) and the padding is dropped at the end (at most 63 wasted rows)
            pad = 64 - N0 % 64
            feat = torch.cat([feat, feat.new_zeros((pad, T, feat.shape[2]))], 0)
            if w is not None:
                w = torch.cat([w, w.new_zeros((pad, T))], 0)
        N = feat.shape[0]
        hx = torch.zeros((N, self.rnn_hdim), device=feat.device, dtype=feat.dtype)
        acc = torch.zeros_like(hx)
        ft = feat.transpose(0, 1).contiguous()                     # time-major: every step's input rows are contiguous
        CH = max(1, min(T, (1 << 27) // max(1, N * 3 * self.rnn_hdim)))      # input-gate GEMM for CH frames at a time (<= 512 MB of gates): one launch instead of CH
        gi_chunk = None
        for t in range(T):
            if fast:
                if t % CH == 0:
                    gi_chunk = torch.nn.functional.linear(ft[t:t + CH].reshape(-1, ft.shape[2]), cell.weight_ih).view(-1, N, 3 * self.rnn_hdim)
                gh = torch.nn.functional.linear(hx, cell.weight_hh)
                hx = kpsim.gru_cell_step(gi_chunk[t % CH], gh, cell.bias_ih, cell.bias_hh, hx)
            else:
                hx = cell(ft[t], hx)
            if w is None:
                acc.add_(hx)
            else:
                acc.add_(hx * w[:, t, None])
        acc = acc[:N0]
        return acc / T if w is None else acc

    @staticmethod
    def _frame_weights(data, T, like):
        """Ragged batches (whole takes of different length padded to the longest, `data['ragged']` set by StateARDataset.batch): the reference
        runs init_context on one unpadded sequence at a time, so a row's context mean is over its own `len` frames -- weights 1 / len on those
        frames, 0 on the padding (the GRU runs forwards: the padding cannot reach the frames before it).  None for rectangular batches."""
        if not data.get("ragged", False):
            return None
        ln = torch.as_tensor(data["len"], device=like.device).to(like.dtype)
        return (torch.arange(T, device=like.device)[None, :] < ln[:, None]).to(like.dtype) / ln[:, None]

    def init_states(self, data, keep_feat: bool = True):
        """init_states (:180-201) + init_pred_qpos (:169-178): -> (init_qpos [N,76], init_qvel [N,75], context_feat_rnn or None)."""
        if keep_feat:
            ctx = self.get_context_feat(data)
            w = self._frame_weights(data, ctx.shape[1], ctx)
            mean = ctx.mean(1) if w is None else (ctx * w[:, :, None]).sum(1)
        else:
            ctx, mean = None, self.get_context_mean(data)
        init = self.context_fc(self.context_mlp(mean))
        pred, vel = init[:, :self.action_dim], init[:, self.action_dim:]
        q0 = data["qpos"][:, 0]
        pred_qpos = torch.cat([q0[:, :2], pred[:, :74]], 1)
        root = quat_mul(heading_q(q0[:, 3:7]), pred_qpos[:, 3:7])
        pred_qpos = torch.cat([pred_qpos[:, :3], root / root.norm(dim=1, keepdim=True), pred_qpos[:, 7:]], 1)
        return pred_qpos, vel, ctx

    @torch.no_grad()
    def rollout(self, data, kin_sim: kpsim.KpSim, init_qpos, init_qvel, dt=1.0 / 30.0):
        """TrajARNet.forward (:346-383) in test mode: kinematic roll-out of the whole clip.
        Returns ar_qpos [N,T,76], ar_qvel [N,T,75] (after fix_qvel), action [N,T,80]."""
        N, T = data["qpos"].shape[:2]
        dev = init_qpos.device
        one_hot = data["action_one_hot"] if data["action_one_hot"].dim() == 2 else data["action_one_hot"][:, 0]
        cur_t = torch.zeros(N, dtype=torch.int32, device=dev)
        z96 = torch.zeros((N, T, 96), device=dev); z72 = torch.zeros((N, T, 72), device=dev)
        obj = torch.empty((N, 7), device=dev)
        ctx = kin_sim.make_ctx(T, data["head_pose"].contiguous(), data["head_vels"].contiguous(), data["obj_head_relative_poses"].contiguous(),
                               one_hot.contiguous(), z96, z72, cur_t, obj_qpos=obj)
        # time-major records: frame t + 1 of Q / V is written in place by the fused step (kp_kin_advance), so the loop carries views, not copies
        Qb = torch.empty((T, N, 76), device=dev); Vb = torch.empty((T, N, 75), device=dev); Ab = torch.empty((T, N, 80), device=dev)
        Qb[0].copy_(init_qpos); Vb[0].copy_(init_qvel)
        hx = self.init_hidden(N, dev)
        for t in range(T):
            cur_t.fill_(t)
            obj.copy_(data["obj_pose"][:, t, :7])
            kin_sim.set_state(Qb[t], Vb[t])                    # forward kinematics of the kinematic state
            state = kin_sim.obs_ar(ctx)
            action, hx = self.get_action(state, hx)
            Ab[t].copy_(action)
            if t == T - 1:
                break
            # next pose (root quaternion normalised, TrajARNet.step :323-327) and finite-difference velocity in one launch
            kpsim.kin_advance(Qb[t], Ab[t], dt, Qb[t + 1], Vb[t + 1])
        Q, V, A = Qb.transpose(0, 1), Vb.transpose(0, 1), Ab.transpose(0, 1)
        V = torch.cat([V[:, 1:], V[:, -2:-1]], 1)              # fix_qvel (:385-388)
        return Q, V, A


class PolicyARContext:
    """`PolicyAR.init_context` for N episodes at once -> the `ar_context` tensors the env consumes.

    need_rollout (constructor default, per-call override): the kinematic roll-out of the whole clip (`ar_qpos`, `ar_qvel`, `ar_wbpos`, ...) is
    read by `ar_mode` (humanoid_ar_v1.py:263, 339-341), `ar_fail_safe` (:327-331), the legacy `policy_v == 2` observation (:209-210) and
    the evaluation scripts; a TRAINING episode consumes `init_qpos` / `init_qvel` only (reset_model :342-343 -- load_context's
    `target = qpos_fk(ar_qpos[0])`, :88, is overwritten by the reset that always follows, :384).  With need_rollout=False the T GRU + MLP +
    FK steps of the roll-out are skipped and the context keys they would fill are absent; everything a training episode reads is unchanged."""

    def __init__(self, net: TrajARNet, kin_sim: kpsim.KpSim, smooth: bool = True, smooth_time_axis: bool = False, need_rollout: bool = True,
                 keep_context_feat: bool = True):
        # smooth = cfg.smooth (kin_poly.yml:19): selects the branch of init_context (fix_height + FK of ar_qpos);
        # smooth_time_axis: deviation from the reference, see the module docstring
        self.net, self.kin_sim, self.smooth, self.smooth_time_axis = net, kin_sim, smooth, smooth_time_axis
        self.need_rollout, self.keep_context_feat = need_rollout, keep_context_feat

    def _rollout_any(self, data, init_qpos, init_qvel):
        """TrajARNet.rollout for any number of clips: the kinematic twin simulator holds kin_sim.n rows, so other batch sizes go through it in
        chunks of that many (the last one padded with copies of its last clip)."""
        m, n = init_qpos.shape[0], self.kin_sim.n
        if m == n:
            q, v, _ = self.net.rollout(data, self.kin_sim, init_qpos, init_qvel)
            return q, v
        keys = ("qpos", "head_pose", "head_vels", "obj_head_relative_poses", "action_one_hot", "obj_pose")
        qs, vs = [], []
        for i in range(0, m, n):
            k = min(n, m - i)
            idx = torch.arange(i, i + n, device=init_qpos.device).clamp_(max=m - 1)
            part = {key: data[key][idx].contiguous() for key in keys}
            q, v, _ = self.net.rollout(part, self.kin_sim, init_qpos[idx].contiguous(), init_qvel[idx].contiguous())
            qs.append(q[:k]); vs.append(v[:k])
        return torch.cat(qs, 0), torch.cat(vs, 0)

    @torch.no_grad()
    def init_context(self, data: dict, fix_height: bool = False, need_rollout: bool | None = None) -> dict:
        out = dict(data)
        need_rollout = self.need_rollout if need_rollout is None else need_rollout
        init_qpos, init_qvel, ctx_feat = self.net.init_states(data, keep_feat=self.keep_context_feat)
        out["init_qpos"], out["init_qvel"] = init_qpos.contiguous(), init_qvel.contiguous()
        if ctx_feat is not None:
            out["context_feat_rnn"] = ctx_feat
        begin_feet_offset = 0.01
        if self.smooth and fix_height:
            fk = self.kin_sim.fk(out["init_qpos"])
            N0 = init_qpos.shape[0]
            feet = torch.minimum(fk["wbpos"].view(N0, 24, 3)[:, 4, 2], fk["wbpos"].view(N0, 24, 3)[:, 8, 2]) - begin_feet_offset
            out["init_qpos"] = torch.cat([out["init_qpos"][:, :2], (out["init_qpos"][:, 2] - feet)[:, None], out["init_qpos"][:, 3:]], 1).contiguous()
        if not need_rollout:
            return out
        ar_qpos, ar_qvel = self._rollout_any(data, init_qpos, init_qvel)
        N, T = ar_qpos.shape[:2]
        if self.smooth:
            if self.smooth_time_axis:      # NOT what the reference computes (its filter call is a no-op, module docstring)
                ar_qpos = torch.cat([ar_qpos[:, :, :7], gaussian_filter1d_time(ar_qpos[:, :, 7:], 1.0)], 2)
            if fix_height:
                fk = self.kin_sim.fk(ar_qpos.reshape(-1, 76).contiguous())
                wb = fk["wbpos"].view(N, T, 24, 3)
                feet = torch.minimum(wb[:, 0, 4, 2], wb[:, 0, 8, 2]) - begin_feet_offset   # first frame of each clip (:157-158)
                ar_qpos = torch.cat([ar_qpos[:, :, :2], ar_qpos[:, :, 2:3] - feet[:, None, None], ar_qpos[:, :, 3:]], 2)
        fk = self.kin_sim.fk(ar_qpos.reshape(-1, 76).contiguous())
        out["ar_qpos"], out["ar_qvel"] = ar_qpos, ar_qvel
        out["ar_wbpos"], out["ar_wbquat"], out["ar_bquat"] = fk["wbpos"].view(N, T, 72), fk["wbquat"].view(N, T, 96), fk["bquat"].view(N, T, 96)
        return out
