"""The UHC training environment (`HumanoidEnv`, uhc/envs/humanoid_im.py) and its expert-feature precompute
(`get_expert`, uhc/utils/tools.py:20-85), batched over N environments on the device.

It runs on the same C-ABI entry points as the kinematic-policy env: `kp_sim_step_ctrl` is `do_simulation`, `kp_sim_obs_cc` is
`get_full_obs_v1` against the expert frame t + 1 (installed with `kp_sim_set_target`), `kp_sim_fk` does the clip's forward
kinematics.  Reward (`world_rfc_implicit_reward`, uhc/core/reward_function.py:4-53) and termination (`calc_body_diff`, the MEAN
form of humanoid_im.py:719-726, threshold 0.5) are a handful of [N, .] tensor ops.  SURVEY.md section 8(f) row 3.
"""
from __future__ import annotations

import math

import numpy as np
import torch

from . import sim as kpsim
from .context import quat_acos_w, quat_inv, quat_mul, quat_rotate_t, quat_sin_half, quat_small
from .env import RunningState
from .model_compiler import read_kpm
from .nets import PolicyMCP

EE_BODIES = (4, 8, 17, 22, 13)          # L_Toe, R_Toe, L_Wrist, R_Wrist, Head (humanoid_im.py:329)
UHC_REWARD_WEIGHTS = dict(w_p=0.3, w_v=0.1, w_e=0.45, w_c=0.1, w_vf=0.05, k_p=2.0, k_v=0.005, k_e=5.0, k_c=100.0, k_vf=1.0)   # uhc.yml:37-48


def rotation_from_quaternion_t(q):
    """uhc/khrylib/utils/transformation.py:348-356 on [..., 4]: axis * angle, no wrap; exactly 0 when 1 - |w| < 1e-8."""
    small = quat_small(q)
    s = quat_sin_half(q).clamp_min(1e-30)
    out = q[..., 1:] / s[..., None] * (2 * quat_acos_w(q))[..., None]
    return torch.where(small[..., None], torch.zeros_like(out), out)


def get_angvel_fd_t(prev_bquat, cur_bquat, dt):
    """math.py:68-74 on [..., 96] -> [..., 72]."""
    p, c = prev_bquat.reshape(*prev_bquat.shape[:-1], 24, 4), cur_bquat.reshape(*cur_bquat.shape[:-1], 24, 4)
    return (rotation_from_quaternion_t(quat_mul(c, quat_inv(p))) / dt).reshape(*prev_bquat.shape[:-1], 72)


def get_qvel_fd_new_t(cur_qpos, next_qpos, dt):
    """math.py:45-65 (transform=None) on [..., 76] -> [..., 75]: root angular velocity in the root frame, joint differences unwrapped."""
    v = (next_qpos[..., :3] - cur_qpos[..., :3]) / dt
    qrel = quat_mul(next_qpos[..., 3:7], quat_inv(cur_qpos[..., 3:7]))
    w = qrel[..., 0]
    small = quat_small(qrel)
    s = quat_sin_half(qrel).clamp_min(1e-30)
    angle = torch.where(small, torch.zeros_like(w), 2 * quat_acos_w(qrel))
    axis = torch.where(small[..., None], torch.tensor([1.0, 0.0, 0.0], device=w.device, dtype=w.dtype).expand_as(qrel[..., 1:]), qrel[..., 1:] / s[..., None])
    angle = torch.where(angle > math.pi, angle - 2 * math.pi, angle)
    rv = quat_rotate_t(cur_qpos[..., 3:7], axis * angle[..., None] / dt)
    diff = next_qpos[..., 7:] - cur_qpos[..., 7:]
    diff = diff - 2 * math.pi * torch.ceil((diff - math.pi) / (2 * math.pi))          # the two while-loops: into (-pi, pi]
    return torch.cat([v, rv, diff / dt], -1)


def get_expert_batch(sim: kpsim.KpSim, expert_qpos: torch.Tensor, body_mass: torch.Tensor, dt=1.0 / 30.0) -> dict:
    """get_expert for N clips at once: expert_qpos [N, T, 76] -> dict of [N, T, .] device tensors (same keys as the reference)."""
    N, T, _ = expert_qpos.shape
    q = expert_qpos.to(sim.device, torch.float32).contiguous()
    fk = sim.fk(q.reshape(-1, 76))
    wbpos, wbquat = fk["wbpos"].view(N, T, 24, 3), fk["wbquat"].view(N, T, 24, 4)
    body_com = fk["body_com"].view(N, T, 24, 3)
    bquat = fk["bquat"].view(N, T, 24, 4).clone()
    bquat[:, :, 0] = q[:, :, 3:7]                                   # env.get_body_quat() starts from the raw root quaternion
    ex = {"qpos": q, "wbpos": wbpos.reshape(N, T, 72), "wbquat": wbquat.reshape(N, T, 96), "bquat": bquat.reshape(N, T, 96),
          "body_com": body_com.reshape(N, T, 72), "com": (body_com * body_mass[None, None, :, None]).sum(2) / body_mass.sum(),
          "head_pose": torch.cat([wbpos[:, :, 13], wbquat[:, :, 13]], -1)}
    ee = wbpos[:, :, list(EE_BODIES)]
    ex["ee_wpos"] = ee.reshape(N, T, 15)
    rootq = q[:, :, None, 3:7].expand(N, T, 5, 4)
    ex["ee_pos"] = quat_rotate_t(rootq, ee - q[:, :, None, :3]).reshape(N, T, 15)
    h = torch.zeros_like(q[:, :, 3:7]); h[..., 0] = q[..., 3]; h[..., 3] = q[..., 6]
    h = h / h.norm(dim=-1, keepdim=True)
    ex["rq_rmh"] = quat_mul(quat_inv(h), q[:, :, 3:7])
    qvel = get_qvel_fd_new_t(q[:, :-1], q[:, 1:], dt).clamp(-10.0, 10.0)
    qvel = torch.cat([qvel[:, :1], qvel], 1)                        # frame 0 repeats frame 1's finite difference
    ex["qvel"], ex["rlinv"], ex["rangv"] = qvel, qvel[..., :3], qvel[..., 3:6]
    ex["rlinv_local"] = torch.cat([quat_rotate_t(q[:, 1:2, 3:7], qvel[:, :1, :3]), quat_rotate_t(q[:, 1:, 3:7], qvel[:, 1:, :3])], 1)
    bav = get_angvel_fd_t(ex["bquat"][:, :-1], ex["bquat"][:, 1:], dt)
    ex["bangvel"] = torch.cat([bav[:, :1], bav], 1)
    ex["len"] = T
    ex["height_lb"], ex["head_height_lb"] = q[:, :, 2].min(1).values, ex["head_pose"][:, :, 2].min(1).values
    return ex


def world_rfc_implicit_reward_t(xpos, bquat, prev_bquat, com, action, e_bquat, e_bangvel, e_ee_wpos, e_com, b_diffw, dt=1.0 / 30.0, ws=UHC_REWARD_WEIGHTS):
    """uhc/core/reward_function.py:4-53 on [N, .] tensors -> (reward [N], info [N, 5])."""
    N = xpos.shape[0]
    cur_ee = xpos.view(N, 24, 3)[:, list(EE_BODIES)].reshape(N, 15)
    cur_bangvel = get_angvel_fd_t(prev_bquat, bquat, dt)
    qd = quat_mul(bquat.view(N, 24, 4), quat_inv(e_bquat.view(N, 24, 4)))
    pose_diff = (torch.acos(qd[..., 0].abs().clamp(-1.0, 1.0)) if qd.dtype == torch.float64 else torch.atan2(quat_sin_half(qd), qd[..., 0].abs())) * b_diffw[None]   # acos(|w|)
    pose_r = torch.exp(-ws["k_p"] * (pose_diff ** 2).sum(1))
    vel_r = torch.exp(-ws["k_v"] * ((cur_bangvel - e_bangvel) ** 2).sum(1))
    ee_r = torch.exp(-ws["k_e"] * ((cur_ee - e_ee_wpos) ** 2).sum(1))
    com_r = torch.exp(-ws["k_c"] * ((com - e_com) ** 2).sum(1))
    vf_r = torch.exp(-ws["k_vf"] * (action[:, -6:] ** 2).sum(1)) if ws["w_vf"] > 0 else torch.zeros_like(pose_r)
    r = ws["w_p"] * pose_r + ws["w_v"] * vel_r + ws["w_e"] * ee_r + ws["w_c"] * com_r + ws["w_vf"] * vf_r
    return r / (ws["w_p"] + ws["w_v"] + ws["w_e"] + ws["w_c"] + ws["w_vf"]), torch.stack([pose_r, vel_r, ee_r, com_r, vf_r], 1)


class BatchedHumanoidEnv:
    """HumanoidEnv (UHC imitation env) x N.  `load_expert(qpos [N, T, 76])`, `reset(mask)`, `step(a [N, 75])`."""

    def __init__(self, n_envs, device=0, kpm_path=None, env_episode_len=100000, env_init_noise=0.0, env_expert_trail_steps=0,
                 body_diff_thresh=0.5, reward_weights=None, model_options=None, seed=0):
        self.n = int(n_envs)
        kpm_path = kpm_path or kpsim.DEFAULT_KPM
        self.model = kpsim.KpModel(kpm_path, **(model_options or {}))
        self.sim = kpsim.KpSim(self.model, self.n, device)
        self.device = self.sim.device
        kpm = read_kpm(kpm_path)
        self.body_mass = torch.tensor(kpm["body_mass"], dtype=torch.float32, device=self.device)
        self.b_diffw = torch.tensor(kpm["uhc_b_diffw"], dtype=torch.float32, device=self.device)        # pose_diff[1:] *= cfg.b_diffw (root weight 1)
        self.jpos_diffw = torch.tensor(kpm["body_diffw"], dtype=torch.float32, device=self.device)
        self.env_episode_len, self.env_init_noise, self.trail = env_episode_len, env_init_noise, env_expert_trail_steps
        self.body_diff_thresh, self.ws = body_diff_thresh, dict(reward_weights or UHC_REWARD_WEIGHTS)
        self.frame_skip, self.dt = 15, self.model.get_option("timestep") * 15
        self.gen = torch.Generator(device=self.device); self.gen.manual_seed(seed)
        self.cur_t = torch.zeros(self.n, dtype=torch.long, device=self.device)
        self.expert = None
        self._obs = torch.empty((self.n, kpsim.CC_OBS_DIM), dtype=torch.float32, device=self.device)
        self.obs_dim, self.action_dim = kpsim.CC_OBS_DIM, kpsim.CC_ACTION_DIM
        self._ar = torch.arange(self.n, device=self.device)

    def load_expert(self, expert_qpos: torch.Tensor):
        self.expert = get_expert_batch(self.sim, expert_qpos, self.body_mass, self.dt)

    def _e(self, key, t):
        return self.expert[key][self._ar, t.clamp(max=self.expert["len"] - 1)].contiguous()

    def _obs_now(self):
        self.sim.set_target(self._e("qpos", self.cur_t + 1))            # get_full_obs_v1 looks at expert frame t + 1 (humanoid_im.py:158)
        return self.sim.obs_cc(self._obs)

    def reset(self, env_mask: torch.Tensor | None = None):
        m8 = None if env_mask is None else env_mask.to(self.device, torch.uint8).contiguous()
        if env_mask is None:
            self.cur_t.zero_()
        else:
            self.cur_t.masked_fill_(env_mask.to(self.device, torch.bool), 0)
        q0 = self.expert["qpos"][:, 0].clone()
        if self.env_init_noise > 0:
            q0[:, 7:] += torch.randn((self.n, 69), device=self.device, generator=self.gen) * self.env_init_noise
        self.sim.set_state(q0.contiguous(), self.expert["qvel"][:, 0].contiguous(), m8)
        return self._obs_now()

    def step(self, a: torch.Tensor):
        sim = self.sim
        sim.step_begin()                                               # prev_bquat
        sim.set_target(self._e("qpos", self.cur_t))                    # compute_torque's base pose is get_expert_kin_pose(delta_t=0) (humanoid_im.py:441, 678)
        sim.step_ctrl(a.contiguous(), self.frame_skip)
        self.cur_t += 1
        xpos, bquat = sim.get("xpos"), sim.get("bquat")
        com = (sim.get("xipos").view(self.n, 24, 3) * self.body_mass[None, :, None]).sum(1) / self.body_mass.sum()
        e_wbpos = self._e("wbpos", self.cur_t)
        body_diff = (((xpos - e_wbpos).view(self.n, 24, 3) * self.jpos_diffw[None, :, None]).norm(dim=2)).mean(1)
        fail = body_diff > self.body_diff_thresh
        end = (self.cur_t >= self.env_episode_len) | (self.cur_t >= self.expert["len"] + self.trail)
        reward, rinfo = world_rfc_implicit_reward_t(xpos, bquat, sim.get("prev_bquat"), com, a, self._e("bquat", self.cur_t), self._e("bangvel", self.cur_t),
                                                    self._e("ee_wpos", self.cur_t), self._e("com", self.cur_t), self.b_diffw, self.dt, self.ws)
        obs = self._obs_now()
        return obs, torch.ones(self.n, device=self.device), fail | end, {"fail": fail, "end": end, "percent": self.cur_t.float() / self.expert["len"],
                                                                        "custom_reward": reward, "custom_info": rinfo, "body_diff": body_diff}


class RunningStateOnline(RunningState):
    """ZFilter with update=True (uhc/khrylib/utils/zfilter.py): RunningStat.push over a whole [B, dim] batch at once
    (Chan et al. pairwise merge == pushing the rows one by one, in exact arithmetic)."""

    def __init__(self, dim=784, clip=5.0, device="cuda"):
        super().__init__(np.zeros(dim), np.ones(dim), clip, device)
        self.count = 0
        self._mean64 = torch.zeros(dim, dtype=torch.float64, device=self.mean.device)
        self._m2 = torch.zeros(dim, dtype=torch.float64, device=self.mean.device)

    def update(self, x: torch.Tensor):
        x = x.double()
        b = x.shape[0]
        bm = x.mean(0); bm2 = ((x - bm) ** 2).sum(0)
        tot = self.count + b
        delta = bm - self._mean64
        self._m2 += bm2 + delta ** 2 * (self.count * b / tot)
        self._mean64 += delta * (b / tot)
        self.count = tot
        self.mean.copy_(self._mean64.float())
        std = torch.sqrt(self._m2 / (self.count - 1)) if self.count > 1 else self._mean64.abs()
        self.std.copy_(std.float())

    def __call__(self, x, update=True):
        if update:
            self.update(x)
        y = (x - self.mean) / (self.std + 1e-8)
        return y.clamp(-self.clip, self.clip) if self.clip else y


class CopycatAgent:
    """The UHC training iteration (uhc/agents/agent_copycat.py: sample with the running state updating, GAE, PPO on PolicyMCP)
    on the batched env: one process per GPU, gradients all-reduced like kinpoly_amd.rollout.PPOTrainer."""

    def __init__(self, env: BatchedHumanoidEnv, policy: PolicyMCP | None = None, value=None, gamma=0.95, tau=0.95, clip_epsilon=0.2,
                 policy_lr=5e-5, value_lr=3e-4, num_optim_epoch=10, group=None):
        from .nets import MLP, Value
        from .rollout import _allreduce_grads, estimate_advantages, ppo_surrogate
        self.env, self.group = env, group
        self.policy = (policy or PolicyMCP()).to(env.device).float()
        self.value = (value or Value(MLP(kpsim.CC_OBS_DIM, (1024, 512), "relu"))).to(env.device).float()
        self.running_state = RunningStateOnline(kpsim.CC_OBS_DIM, 5.0, env.device)
        self.gamma, self.tau, self.clip_epsilon, self.num_optim_epoch = gamma, tau, clip_epsilon, num_optim_epoch
        self.opt_p = torch.optim.Adam([p for p in self.policy.parameters() if p.requires_grad], lr=policy_lr)
        self.opt_v = torch.optim.Adam(self.value.parameters(), lr=value_lr)
        self._ar, self._ea, self._surr = _allreduce_grads, estimate_advantages, ppo_surrogate

    @torch.no_grad()
    def sample(self, horizon):
        env = self.env
        S, A, R, M = [], [], [], []
        obs = self.running_state(env.reset(), update=True)
        for _ in range(horizon):
            a = self.policy.select_action(obs, False, env.gen).contiguous()
            nobs, _, done, info = env.step(a)
            S.append(obs); A.append(a); R.append(info["custom_reward"]); M.append((~done).float())
            nobs = env.reset(done)
            obs = self.running_state(nobs, update=True)
        return torch.stack(S, 1), torch.stack(A, 1), torch.stack(R, 1), torch.stack(M, 1)

    def log_prob(self, states, actions):
        mean, log_std = self.policy(states)
        return (-(actions - mean) ** 2 / (2 * torch.exp(2 * log_std)) - 0.5 * math.log(2 * math.pi) - log_std).sum(1, keepdim=True)

    def optimize_policy(self, horizon=32):
        S, A, R, M = self.sample(horizon)
        N, T, _ = S.shape
        fs, fa = S.reshape(N * T, -1), A.reshape(N * T, -1)
        with torch.no_grad():
            values = self.value(fs).view(N, T)
            fixed = self.log_prob(fs, fa)
        adv, ret = self._ea(R, M, values, self.gamma, self.tau, self.group)
        adv, ret = adv.reshape(-1, 1), ret.reshape(-1, 1)
        stats = {}
        for _ in range(self.num_optim_epoch):
            vloss = (self.value(fs) - ret).pow(2).mean()
            self.opt_v.zero_grad(); vloss.backward(); self._ar(list(self.value.parameters()), self.group); self.opt_v.step()
            surr = self._surr(self.log_prob(fs, fa), fixed, adv, self.clip_epsilon)
            self.opt_p.zero_grad(); surr.backward()
            params = [p for p in self.policy.parameters() if p.requires_grad]
            self._ar(params, self.group); torch.nn.utils.clip_grad_norm_(params, 40.0); self.opt_p.step()
            stats = {"value_loss": float(vloss.detach()), "surr_loss": float(surr.detach())}
        stats.update(avg_reward=float(R.mean()), fail_rate=float((1 - M).mean()), num_steps=N * T)
        return stats
