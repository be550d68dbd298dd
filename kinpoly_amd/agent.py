"""`AgentAR` on the batched engine: the per-iteration driver of kin_poly/core/agent_ar.py
(`optimize_policy` :271-297 = sample :651-680 + update_params :682-752), one process per GPU.

    sample         VectorSampler over N envs (device SoA, auto-reset through the batched init_context)
    rl_update      GAE (k_gae) + global advantage normalisation (RCCL all-gather) + PPO epochs (:756-772)
    step_update    supervised one-step update x num_step_update (:277-287 of policy_ar.py)
    checkpoints    reference pickle layout (kinpoly_amd/checkpoint.py)

Dataset files of the reference are not in its repository, so episodes come from a `context_fn(n) -> dict` callable
(kinpoly_amd.env.standing_context for the synthetic configs of SURVEY.md section 8(d)).
"""
from __future__ import annotations

import time

import torch
import torch.distributed as dist

from . import checkpoint as ck
from . import sim as kpsim
from .context import PolicyARContext, TrajARNet
from .env import BatchedHumanoidAREnv
from .model_compiler import read_kpm
from .nets import MLP, Value, enable_tuned_gemms
from .rollout import PPOTrainer, VectorSampler, _allreduce_grads
from .supervised import TorchFK, update_supervised_step


class AgentAR:
    def __init__(self, n_envs, context_fn, device=0, horizon=99, seed=4, wild=False, use_init_context=True,
                 policy_lr=1e-5, value_lr=3e-4, supervised_lr=5e-4, num_optim_epoch=10, num_step_update=20, gamma=0.95, tau=0.95,
                 clip_epsilon=0.2, rl_update=True, step_update=True, model_options=None):
        rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
        enable_tuned_gemms()          # library GEMM solution per shape (rollout and update shapes at 4096 envs); selection only
        torch.manual_seed(seed + rank)
        self.env = BatchedHumanoidAREnv(n_envs, device, mode="train", wild=wild, seed=seed + rank, model_options=model_options)
        self.device = self.env.device
        self.policy_net = TrajARNet().to(self.device)
        self.value_net = Value(MLP(105, (512, 256), "relu")).to(self.device)
        self._sync_params()
        self.kin_sim = kpsim.KpSim(self.env.model, n_envs, self.device.index)      # physics-free twin for the kinematic roll-out
        self.ctx_builder = PolicyARContext(self.policy_net, self.kin_sim, smooth=True)
        self.context_fn, self.use_init_context = context_fn, use_init_context
        self.horizon, self.rl_update, self.step_update, self.num_step_update = horizon, rl_update, step_update, num_step_update
        self.trainer = PPOTrainer(self.policy_net, self.value_net, gamma, tau, clip_epsilon, policy_lr, value_lr, num_optim_epoch)
        self.opt_sup = torch.optim.Adam([p for p in self.policy_net.parameters() if p.requires_grad], lr=supervised_lr)
        kpm = read_kpm(kpsim.DEFAULT_KPM)
        self.fk = TorchFK(kpm["body_pos"], kpm["body_parent"], self.device, sim=self.kin_sim)   # HIP forward / backward kernels for the loss FK
        self.sampler = VectorSampler(self.env, self.policy_net, record_qpos=True)
        self.epoch = 0
        self._new_episodes()

    def _sync_params(self):
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            for p in list(self.policy_net.parameters()) + list(self.value_net.parameters()):
                dist.broadcast(p.data, 0)

    def _new_episodes(self):
        """sample_seq + init_context + load_context + reset for every env (agent_ar.py:519-537)."""
        data = self.context_fn(self.env.n)
        if self.use_init_context:
            data = self.ctx_builder.init_context(data, fix_height=False)
        self.env.load_context(data)
        self.sampler.start()

    def optimize_policy(self, i_iter=None):
        t0 = time.time()
        batch = self.sampler.sample(self.horizon)
        torch.cuda.synchronize(self.device)
        t1 = time.time()
        info = {}
        if self.rl_update:
            info.update(self.trainer.update(batch))
        if self.step_update:
            info["step_loss"] = update_supervised_step(self.policy_net, self.opt_sup, self.fk, batch, self.num_step_update, _allreduce_grads)
        torch.cuda.synchronize(self.device)
        t2 = time.time()
        self._new_episodes()
        self.epoch += 1
        n = batch.rewards.numel()
        info.update(T_sample=t1 - t0, T_update=t2 - t1, T_total=time.time() - t0, num_steps=n, avg_reward=float(batch.rewards.mean()),
                    fail_rate=float(batch.fails.float().mean()), env_steps_per_s=n / (t1 - t0))
        return info

    def save_checkpoint(self, path):
        return ck.save_checkpoint(path, self.policy_net, self.value_net, None, self.env.cc_policy)

    def load_checkpoint(self, path):
        cp = ck.load_checkpoint(path)
        self.policy_net.load_state_dict(ck.split_policy_dict(cp["policy_dict"]), strict=False)
        self.value_net.load_state_dict(cp["value_dict"])
        if "cc_dict" in cp:
            self.env.cc_policy.load_state_dict(cp["cc_dict"])
        return cp
