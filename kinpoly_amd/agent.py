"""`AgentAR` on the batched engine: the per-iteration driver of kin_poly/core/agent_ar.py
(`optimize_policy` :271-297 = per_epoch_update + sample :651-680 + update_params :682-752), one process per GPU.

    per_epoch_update  LambdaLR schedules of the policy / value optimisers (:215-225, 268-269)
    sample            VectorSampler over N envs: device SoA, every episode on a freshly drawn clip at any failure rate (a ring of pool_depth + 1
                      context rows per env, topped up on demand by EpisodeSource: batched sample_seq + init_context, freq_dict feedback, :518-606)
    rl_update         GAE (k_gae) + global advantage normalisation (RCCL all-gather) + PPO epochs (:756-772); with
                      joint_controller also update_controller (:774-794), which -- as in the reference, whose optimiser holds
                      policy_net only -- leaves the UHC weights alone unless `train_uhc` is set (PPOTrainer)
    step_update       supervised one-step update x num_step_update (policy_ar.py:277-287) + its own LambdaLR (`step_lr`, :88-89)
    checkpoints       reference pickle layout (kinpoly_amd/checkpoint.py)

Episodes come from a `StateARDataset` (the reference's feature-file sampler, kinpoly_amd/dataset.py) or, for the synthetic
single-clip configs of SURVEY.md section 8(d), from a `context_fn(n) -> dict` callable.
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
from .rollout import EpisodeSource, PPOTrainer, VectorSampler, _allreduce_grads, lambda_lr
from .supervised import TorchFK, update_supervised_step


class AgentAR:
    def __init__(self, n_envs, context_fn=None, device=0, horizon=99, seed=4, wild=False, use_init_context=True,
                 policy_lr=1e-5, value_lr=3e-4, supervised_lr=5e-4, num_optim_epoch=10, num_step_update=20, gamma=0.95, tau=0.95,
                 clip_epsilon=0.2, rl_update=True, step_update=True, model_options=None, dataset=None, sampling_temp=0.3, sampling_freq=0.5,
                 pool_depth=4, num_epoch_fix=100, num_epoch=10000, joint_controller=False, grad_joint=False, grad_alternate=False, train_uhc=False,
                 cache_init_context=False):
        rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
        enable_tuned_gemms()          # library GEMM solution per shape (rollout and update shapes at 4096 envs); selection only
        torch.manual_seed(seed + rank)
        self.env = BatchedHumanoidAREnv(n_envs, device, mode="train", wild=wild, seed=seed + rank, model_options=model_options,
                                        joint_controller=joint_controller)
        self.device = self.env.device
        self.policy_net = TrajARNet().to(self.device)
        self.value_net = Value(MLP(105, (512, 256), "relu")).to(self.device)
        self._sync_params()
        self.kin_sim = kpsim.KpSim(self.env.model, n_envs, self.device.index)      # physics-free twin for the kinematic roll-out
        # a training episode reads init_qpos / init_qvel of its context only: the whole-clip kinematic roll-out and the [N, T, 1024] context
        # feature sequence of init_context are not computed (PolicyARContext; evaluate.py builds its own with need_rollout=True)
        self.ctx_builder = PolicyARContext(self.policy_net, self.kin_sim, smooth=True, need_rollout=False, keep_context_feat=False)
        # sampling_temp / sampling_freq: kin_poly.yml:67-68; freq_dict lives in the source (agent_ar.py:228-234)
        self.source = EpisodeSource(dataset=dataset, context_fn=context_fn if dataset is None else None,
                                    ctx_builder=self.ctx_builder if use_init_context else None,
                                    sampling_temp=sampling_temp, sampling_freq=sampling_freq, fix_height=False, cache_init_context=cache_init_context)
        self.horizon, self.rl_update, self.step_update, self.num_step_update = horizon, rl_update, step_update, num_step_update
        self.grad_joint, self.grad_alternate = grad_joint, grad_alternate       # policy_specs.grad_joint / grad_alternate (agent_ar.py:703, 746-747)
        self.trainer = PPOTrainer(self.policy_net, self.value_net, gamma, tau, clip_epsilon, policy_lr, value_lr, num_optim_epoch,
                                  num_epoch_fix=num_epoch_fix, num_epoch=num_epoch, cc_policy=self.env.cc_policy if joint_controller else None, train_uhc=train_uhc)
        self.opt_sup = torch.optim.Adam([p for p in self.policy_net.parameters() if p.requires_grad], lr=supervised_lr)
        self.sched_sup = lambda_lr(self.opt_sup, num_epoch_fix, num_epoch)          # PolicyAR.setup_optimizers / step_lr (policy_ar.py:45-62, 88-89)
        kpm = read_kpm(kpsim.DEFAULT_KPM)
        self.fk = TorchFK(kpm["body_pos"], kpm["body_parent"], self.device, sim=self.kin_sim)   # HIP forward / backward kernels for the loss FK
        self.sampler = VectorSampler(self.env, self.policy_net, record_qpos=True, source=self.source, pool_depth=pool_depth,
                                     record_full=joint_controller)
        self.epoch = 0
        self.sampler.start()

    @property
    def freq_dict(self):
        return self.source.freq_dict

    def _sync_params(self):
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            for p in list(self.policy_net.parameters()) + list(self.value_net.parameters()):
                dist.broadcast(p.data, 0)

    def optimize_policy(self, i_iter=None):
        t0 = time.time()
        self.trainer.per_epoch_update()
        batch = self.sampler.sample(self.horizon)
        torch.cuda.synchronize(self.device)
        t1 = time.time()
        info = {}
        if self.grad_joint:               # update_params' other branch (agent_ar.py:746-747): surrogate + supervised loss in one step
            info.update(self.trainer.update_joint(batch, self.fk, self.grad_alternate, self.epoch, self.opt_sup))
        else:
            if self.rl_update:
                info.update(self.trainer.update(batch))
            if self.step_update:
                info["step_loss"] = update_supervised_step(self.policy_net, self.opt_sup, self.fk, batch, self.num_step_update, _allreduce_grads)
        self.sched_sup.step()
        torch.cuda.synchronize(self.device)
        t2 = time.time()
        self.epoch += 1
        n = batch.rewards.numel()
        info.update(T_sample=t1 - t0, T_update=t2 - t1, T_total=time.time() - t0, num_steps=n, avg_reward=float(batch.rewards.mean()),
                    fail_rate=float(batch.fails.float().mean()), env_steps_per_s=n / (t1 - t0), episodes=len(batch.episodes.get("percent", ())),
                    pool_exhausted=self.sampler.pool_exhausted, clips_drawn=self.source.n_drawn, init_context_memo_hits=self.source.n_memo_hits, top_ups=self.sampler.top_ups, policy_lr=self.trainer.opt_p.param_groups[0]["lr"])
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
