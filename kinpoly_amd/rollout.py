"""Vectorised rollout driver + GAE/PPO update: the replacement of AgentAR.sample / sample_worker /
update_params (kin_poly/core/agent_ar.py:510-611, 651-680, 682-772) for N environments per GPU.

Instead of forking `num_threads` workers that each own one MjSim and push Python rows through a
multiprocessing.Queue, one process per GPU steps all N environments in lock-step; experience lives in
device-resident SoA buffers laid out env-major [N, T, .] (each env's rows contiguous and in time order --
the ordering GAE and the GRU re-unroll of the reference rely on).  Across GPUs the
environments shard by rank; the only exchange is the all-gather of advantages / returns for the global
normalisation (uhc/khrylib/rl/core/common.py:22) and, for a data-parallel update, a gradient all-reduce.
"""
from __future__ import annotations

from dataclasses import dataclass

import torch
import torch.distributed as dist

from . import sim as kpsim
from .env import BatchedHumanoidAREnv
from .nets import KinPolicy, Value


@dataclass
class RolloutBatch:
    states: torch.Tensor         # [N, T, 105]
    actions: torch.Tensor        # [N, T, 80]
    rewards: torch.Tensor        # [N, T]
    masks: torch.Tensor          # [N, T]   0 where the episode ended at this row (or at the horizon)
    episode_start: torch.Tensor  # [N, T]   bool: hidden state is zero before this row
    fails: torch.Tensor          # [N, T]   bool
    curr_qpos: torch.Tensor | None = None   # [N, T, 76]  (TrajBatchEgo fields used by the supervised step update)
    gt_target_qpos: torch.Tensor | None = None


class VectorSampler:
    """Fixed-horizon lock-step sampler with device-side auto-reset (no host sync inside the loop)."""

    def __init__(self, env: BatchedHumanoidAREnv, policy: KinPolicy, record_qpos: bool = False, mean_action: bool = False):
        self.env, self.policy, self.record_qpos, self.mean_action = env, policy, record_qpos, mean_action
        self.obs = None
        self.hx = None
        self.fresh = None

    def start(self):
        self.obs = self.env.reset().clone()
        self.hx = self.policy.init_hidden(self.env.n, self.env.device)
        self.fresh = torch.ones(self.env.n, dtype=torch.bool, device=self.env.device)

    @torch.no_grad()
    def sample(self, T: int) -> RolloutBatch:
        env, pol, N, dev = self.env, self.policy, self.env.n, self.env.device
        if self.obs is None:
            self.start()
        S = torch.empty((N, T, 105), device=dev); A = torch.empty((N, T, 80), device=dev)
        R = torch.empty((N, T), device=dev); M = torch.empty((N, T), device=dev)
        E = torch.empty((N, T), dtype=torch.bool, device=dev); F = torch.empty((N, T), dtype=torch.bool, device=dev)
        Q = torch.empty((N, T, 76), device=dev) if self.record_qpos else None
        G = torch.empty((N, T, 76), device=dev) if self.record_qpos else None
        ar = torch.arange(N, device=dev)
        for t in range(T):
            S[:, t] = self.obs
            E[:, t] = self.fresh
            action, self.hx = pol.select_action(self.obs, self.hx, self.mean_action, env.gen)
            action = action.contiguous()
            if self.record_qpos:
                Q[:, t] = env.sim.get("qpos")
                G[:, t] = env.ctx["qpos"][ar, torch.minimum(env.cur_t.long() + 1, env.ctx_len.long())]
            _, _, done, info = env.step(action)
            A[:, t] = action
            R[:, t] = info["custom_reward"]
            F[:, t] = info["fail"]
            M[:, t] = (~done).float()
            # device-side auto reset of finished episodes (masked kernels; untouched envs keep their state)
            self.obs = env.reset(done).clone()
            self.hx = self.hx * (~done).float().unsqueeze(1)
            self.fresh = done
        M[:, T - 1] = 0.0  # horizon cut: the flat-batch GAE of the reference never carries value across workers
        return RolloutBatch(S, A, R, M, E, F, Q, G)


# ---------------------------------------------------------------------------------------------- update
def env_shard(rank: int, world_size: int, envs_per_gpu: int):
    """Rank r owns global environments [r * envs_per_gpu, (r + 1) * envs_per_gpu) and seed stream 4 + r."""
    return range(rank * envs_per_gpu, (rank + 1) * envs_per_gpu), 4 + rank


def normalize_advantages_global(adv: torch.Tensor, ret: torch.Tensor, group=None):
    """The reference normalises advantages over its whole concatenated batch (common.py:22, unbiased std).
    Sharded over ranks, the per-rank advantages / returns are all-gathered (one RCCL all-gather over xGMI per
    PPO iteration; gloo in the CPU tests) so every rank applies the whole-job mean / std.  Returns
    (normalised local adv, local ret, gathered returns [world * n])."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        ws = dist.get_world_size(group)
        packed = torch.stack([adv.reshape(-1), ret.reshape(-1)], 1).contiguous()
        gathered = [torch.empty_like(packed) for _ in range(ws)]
        dist.all_gather(gathered, packed, group=group)
        all_adv = torch.cat([g[:, 0] for g in gathered]); all_ret = torch.cat([g[:, 1] for g in gathered])
    else:
        all_adv, all_ret = adv.reshape(-1), ret.reshape(-1)
    return (adv - all_adv.mean()) / all_adv.std(), ret, all_ret


def estimate_advantages(rewards, masks, values, gamma, tau, group=None):
    """GAE on the device (k_gae, env-major reverse scan) + the global normalisation above."""
    adv, ret = kpsim.gae(rewards.contiguous(), masks.contiguous(), values.contiguous(), gamma, tau)
    adv, ret, _ = normalize_advantages_global(adv, ret, group)
    return adv, ret


def _allreduce_grads(params, group=None):
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return
    grads = [p.grad for p in params if p.grad is not None]
    if not grads:
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, group=group)
    flat /= dist.get_world_size(group)
    off = 0
    for g in grads:
        g.copy_(flat[off:off + g.numel()].view_as(g)); off += g.numel()


def ppo_surrogate(log_probs, fixed_log_probs, advantages, clip_epsilon=0.2, ind=None):
    """AgentPPO.ppo_loss (uhc/khrylib/rl/agents/agent_ppo.py:58-65): clipped surrogate over the rows `ind` (the `exps` mask)."""
    if ind is not None:
        log_probs, fixed_log_probs, advantages = log_probs[ind], fixed_log_probs[ind], advantages[ind]
    ratio = torch.exp(log_probs - fixed_log_probs)
    return -torch.min(ratio * advantages, torch.clamp(ratio, 1.0 - clip_epsilon, 1.0 + clip_epsilon) * advantages).mean()


class PPOTrainer:
    """AgentPPO.update_policy / ppo_loss / update_value (agent_ar.py:756-772, 852-870; agent_ppo.py:53-56)."""

    def __init__(self, policy: KinPolicy, value: Value, gamma=0.95, tau=0.95, clip_epsilon=0.2, policy_lr=1e-5, value_lr=3e-4,
                 num_optim_epoch=10, policy_grad_clip=40.0, group=None):
        self.policy, self.value, self.group = policy, value, group
        self.gamma, self.tau, self.clip_epsilon, self.num_optim_epoch, self.policy_grad_clip = gamma, tau, clip_epsilon, num_optim_epoch, policy_grad_clip
        self.opt_p = torch.optim.Adam([p for p in policy.parameters() if p.requires_grad], lr=policy_lr)
        self.opt_v = torch.optim.Adam(value.parameters(), lr=value_lr)

    def update(self, batch: RolloutBatch):
        N, T, _ = batch.states.shape
        flat_states = batch.states.reshape(N * T, -1)
        with torch.no_grad():
            values = self.value(flat_states).view(N, T)
            means = self.policy.unroll(batch.states, batch.episode_start)
            fixed_log_probs = self.policy.log_prob(means.reshape(N * T, -1), batch.actions.reshape(N * T, -1))
        adv, ret = estimate_advantages(batch.rewards, batch.masks, values, self.gamma, self.tau, self.group)
        adv, ret = adv.reshape(-1, 1), ret.reshape(-1, 1)
        stats = {}
        for _ in range(self.num_optim_epoch):
            vloss = (self.value(flat_states) - ret).pow(2).mean()
            self.opt_v.zero_grad(); vloss.backward(); _allreduce_grads(list(self.value.parameters()), self.group); self.opt_v.step()
            means = self.policy.unroll(batch.states, batch.episode_start)
            log_probs = self.policy.log_prob(means.reshape(N * T, -1), batch.actions.reshape(N * T, -1))
            surr = ppo_surrogate(log_probs, fixed_log_probs, adv, self.clip_epsilon)
            self.opt_p.zero_grad(); surr.backward()
            params = [p for p in self.policy.parameters() if p.requires_grad]
            _allreduce_grads(params, self.group)
            torch.nn.utils.clip_grad_norm_(params, self.policy_grad_clip)
            self.opt_p.step()
            stats = {"value_loss": float(vloss.detach()), "surr_loss": float(surr.detach())}
        return stats
