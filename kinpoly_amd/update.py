"""`AgentAR.update_params` and everything it steps, without the environment: the two Adam optimisers and LambdaLR schedules of the agent
(kin_poly/core/agent_ar.py:184-225), PolicyAR's own supervised optimiser + schedule (kin_poly/models/policy_ar.py:72-89), `per_epoch_update`
(agent_ar.py:264-269) and the update itself (agent_ar.py:682-752: rl_update -> init_update -> step_update -> step_update_dyna -> full_update,
or the grad_joint branch; update_controller; `policy_net.step_lr()`).

Kept apart from `AgentAR` so that the whole loop can be run -- and pinned -- on a recorded batch with no simulator: tests/golden/update_params.npz
holds two consecutive iterations of the REFERENCE's update_params in fp64 (tools/make_golden.py::gen_update_params), and
tests/test_update_cpu.py replays them here.

update_dtype: the dtype the optimisers' parameters live in.  None = the roll-out modules themselves (fp32 on the device: fused HIP GRU re-unroll,
k_gae, HIP FK kernels).  torch.float64 = the reference's training precision (scripts/train_ar_policy.py:76-77): fp64 master copies of the policy
and the value net are updated (GRUCell loop, `gae_scan`, torch FK) and written back into the fp32 roll-out modules after every update.
"""
from __future__ import annotations

import copy

import torch

from .rollout import PPOTrainer, _allreduce_grads, lambda_lr
from .supervised import TorchFK, update_supervised_step


class ParamUpdate:
    def __init__(self, policy_net, value_net, body_pos, body_parent, kin_sim=None, update_dtype=None, reference_bugs=True,
                 policy_lr=1e-5, value_lr=3e-4, supervised_lr=5e-4, num_optim_epoch=10, num_step_update=20, gamma=0.95, tau=0.95, clip_epsilon=0.2,
                 rl_update=True, step_update=True, num_epoch_fix=100, num_epoch=10000, grad_joint=False, grad_alternate=False, cc_policy=None, train_uhc=False,
                 policy_weightdecay=0.0, value_weightdecay=0.0, init_update=False, num_init_update=5, step_update_dyna=False, num_step_dyna_update=10,
                 full_update=False, num_sample=20000, batch_size=128, noise_std=0.0, group=None):
        self.rollout_policy, self.rollout_value = policy_net, value_net
        p0 = next(policy_net.parameters())
        self.dtype = p0.dtype if update_dtype is None else update_dtype
        if self.dtype == p0.dtype:
            self.policy, self.value = policy_net, value_net
        else:                                                     # master copies in the update's precision
            self.policy, self.value = copy.deepcopy(policy_net).to(self.dtype), copy.deepcopy(value_net).to(self.dtype)
            if hasattr(self.policy, "refresh_log_std"):
                self.policy.refresh_log_std()                     # exp(-3.2) in fp64, not the fp32 module's rounded constant
        hip_fk = kin_sim is not None and self.dtype == torch.float32 and p0.is_cuda
        self.fk = TorchFK(body_pos, body_parent, p0.device, dtype=self.dtype, sim=kin_sim if hip_fk else None)
        self.rl_update, self.step_update, self.num_step_update = rl_update, step_update, num_step_update
        self.grad_joint, self.grad_alternate = grad_joint, grad_alternate       # policy_specs.grad_joint / grad_alternate (agent_ar.py:703, 746-747)
        # the optional supervised branches of update_params (agent_ar.py:711-745; all off in kin_poly.yml)
        self.init_update, self.num_init_update, self.step_update_dyna, self.num_step_dyna_update, self.full_update = init_update, num_init_update, step_update_dyna, num_step_dyna_update, full_update
        self.num_sample, self.batch_size, self.noise_std = num_sample, batch_size, noise_std
        self.trainer = PPOTrainer(self.policy, self.value, gamma, tau, clip_epsilon, policy_lr, value_lr, num_optim_epoch, group=group,
                                  num_epoch_fix=num_epoch_fix, num_epoch=num_epoch, cc_policy=cc_policy, train_uhc=train_uhc,
                                  policy_weightdecay=policy_weightdecay, value_weightdecay=value_weightdecay, reference_bugs=reference_bugs)
        self._sup_cfg = (supervised_lr, num_epoch_fix, num_epoch)
        self.setup_supervised_optimizer()
        self.step_history = []

    @property
    def has_master(self):
        return self.policy is not self.rollout_policy

    def setup_supervised_optimizer(self):
        """PolicyAR.setup_optimizers / step_lr (policy_ar.py:45-62, 72-89): Adam(lr) over the kinematic policy + its LambdaLR"""
        lr, fix, total = self._sup_cfg
        self.opt_sup = torch.optim.Adam([p for p in self.policy.parameters() if p.requires_grad], lr=lr)
        self.sched_sup = lambda_lr(self.opt_sup, fix, total)

    def per_epoch_update(self):
        self.trainer.per_epoch_update()

    @torch.no_grad()
    def sync_rollout(self):
        """master (update dtype) -> the fp32 modules the sampler steps; nothing to do when the update runs on those modules themselves"""
        if self.has_master:
            for dst, src in ((self.rollout_policy, self.policy), (self.rollout_value, self.value)):
                for d, s in zip(dst.parameters(), src.parameters()):
                    d.copy_(s)

    @torch.no_grad()
    def load_from_rollout(self):
        """the other way round, after the roll-out modules were replaced (checkpoint load)"""
        if self.has_master:
            for dst, src in ((self.policy, self.rollout_policy), (self.value, self.rollout_value)):
                for d, s in zip(dst.parameters(), src.parameters()):
                    d.copy_(s)
            if hasattr(self.policy, "refresh_log_std"):
                self.policy.refresh_log_std()          # the constructor's log_std in the master's precision again (an fp32 copy of -3.2 is -3.2000000477)

    def update_params(self, batch, epoch=0, dataset=None):
        """AgentAR.update_params (agent_ar.py:682-752) on a RolloutBatch; returns the losses it saw."""
        info = {}
        if self.grad_joint:               # update_params' other branch (agent_ar.py:746-747): surrogate + supervised loss in one step
            info.update(self.trainer.update_joint(batch, self.fk, self.grad_alternate, epoch, self.opt_sup))
        else:
            if self.rl_update:
                info.update(self.trainer.update(batch))
            if self.init_update:             # :711-718
                from . import pretrain as P
                info["init_loss"] = P.update_init_supervised(self.policy, self.opt_sup, self.fk, dataset, self.num_init_update, self.num_sample, self.batch_size,
                                                             grad_allreduce=_allreduce_grads)
            if self.step_update:
                self.step_history = []
                info["step_loss"] = update_supervised_step(self.policy, self.opt_sup, self.fk, batch, self.num_step_update, _allreduce_grads, history=self.step_history)
            if self.step_update_dyna:        # :728-734: the same step regressed onto the pose the simulation reached
                info["step_dyna_loss"] = update_supervised_step(self.policy, self.opt_sup, self.fk, batch, self.num_step_dyna_update, _allreduce_grads, target=batch.res_qpos)
            if self.full_update:             # :736-744
                from . import pretrain as P
                info["full_loss"] = P.train_full_supervised(self.policy, self.opt_sup, self.fk, dataset, 1, 0.3, self.num_sample, self.batch_size,
                                                            noise_std=self.noise_std, grad_allreduce=_allreduce_grads, rng=P.job_wide_rng(epoch))
        self.sched_sup.step()                 # self.policy_net.step_lr() (:751)
        self.sync_rollout()
        return info
