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
    log / eval        LoggerRL statistics of every sample() call (`info['log']`, merged over ranks) and the `log_train` line (:243-262); `eval_policy`
                      (:394-448): every take of a data set played whole with mean actions, coverage, freq_dict feedback, eval_dict_<mode>.pt;
                      freq_dict.pt written after every iteration when a result_dir is given (:297) and read back at start (:228-234)

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
from .rollout import EpisodeSource, LoggerRL, VectorSampler, _allreduce_grads, _collective_on
from .update import ParamUpdate


class AgentAR:
    def __init__(self, n_envs, context_fn=None, device=0, horizon=99, seed=4, wild=False, use_init_context=True,
                 policy_lr=1e-5, value_lr=3e-4, supervised_lr=5e-4, num_optim_epoch=10, num_step_update=20, gamma=0.95, tau=0.95,
                 clip_epsilon=0.2, rl_update=True, step_update=True, model_options=None, dataset=None, sampling_temp=0.3, sampling_freq=0.5,
                 pool_depth=4, num_epoch_fix=100, num_epoch=10000, joint_controller=False, grad_joint=False, grad_alternate=False, train_uhc=False,
                 cache_init_context=False, log_std=-3.2, policy_weightdecay=0.0, value_weightdecay=0.0, smooth=True, result_dir=None, eval_envs=None,
                 init_update=False, num_init_update=5, step_update_dyna=False, num_step_dyna_update=10, full_update=False, num_sample=20000, batch_size=128,
                 noise_std=0.0, cc_checkpoint=None, update_dtype=None, reference_bugs=True, min_batch_size=0, slice_ratio="slice_start"):
        """update_dtype: None = the update runs on the fp32 roll-out modules (fused HIP re-unroll); torch.float64 = the reference's training
        precision on fp64 master copies (kinpoly_amd/update.py).  reference_bugs: reproduce the reference's generator-consumed gradient clip
        (PPOTrainer) and LoggerRL.merge's max-of-mins; False = the corrected forms.
        min_batch_size (kin_poly.yml:56, 10 000 in the reference; 0 = off): the reference updates once per >= min_batch_size samples.  4096 envs x H steps are ten
        times that, so one update per sample() call would spend a tenth of the reference's optimiser steps (and LambdaLR / Adam steps) per sample.  With
        min_batch_size > 0 a call's batch is cut into ceil(N H / min_batch_size) whole-env slices and every slice is ONE reference iteration: per_epoch_update,
        update_params (PPO epochs, value steps, supervised steps), epoch += 1.  N H <= min_batch_size: one slice, exactly the path without the option.
        slice_ratio: what the PPO ratio of a later slice is taken against.  "slice_start" (default): the parameters that slice's update starts from -- plain
        consecutive update_params calls, each the reference's own (fixed_log_probs at the top of update_policy, agent_ar.py:758-759); the rows were sampled up to
        k - 1 updates earlier, which the ratio then ignores.  "behaviour": the log-probabilities under the policy that sampled the rows, recorded before the first
        slice moves it -- the importance ratio PPO defines; while the supervised step updates move the policy far between slices (random init: step loss 77 -> 5
        within one call) it leaves the clip range at once and the surrogate of the later slices has no gradient (profiles/r06/scripts_run.log)."""
        rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
        enable_tuned_gemms()          # library GEMM solution per shape (rollout and update shapes at 4096 envs); selection only
        torch.manual_seed(seed + rank)
        self.env = BatchedHumanoidAREnv(n_envs, device, mode="train", wild=wild, seed=seed + rank, model_options=model_options,
                                        joint_controller=joint_controller)
        self.device = self.env.device
        if cc_checkpoint is not None:      # the pre-trained UHC the env is built around (humanoid_ar_v1.py:60-81)
            self.env.load_uhc_checkpoint(cc_checkpoint, load_policy=not joint_controller)
        self.policy_net = TrajARNet(log_std=log_std).to(self.device)
        self.value_net = Value(MLP(105, (512, 256), "relu")).to(self.device)
        self._sync_params()
        self.kin_sim = kpsim.KpSim(self.env.model, n_envs, self.device.index)      # physics-free twin for the kinematic roll-out
        # a training episode reads init_qpos / init_qvel of its context only: the whole-clip kinematic roll-out and the [N, T, 1024] context
        # feature sequence of init_context are not computed (PolicyARContext; evaluate.py builds its own with need_rollout=True)
        self.ctx_builder = PolicyARContext(self.policy_net, self.kin_sim, smooth=smooth, need_rollout=False, keep_context_feat=False)
        # sampling_temp / sampling_freq: kin_poly.yml:67-68; freq_dict lives in the source (agent_ar.py:228-234)
        self.source = EpisodeSource(dataset=dataset, context_fn=context_fn if dataset is None else None,
                                    ctx_builder=self.ctx_builder if use_init_context else None,
                                    sampling_temp=sampling_temp, sampling_freq=sampling_freq, fix_height=False, cache_init_context=cache_init_context)
        self.horizon = horizon
        kpm = read_kpm(kpsim.DEFAULT_KPM)
        self.upd = ParamUpdate(self.policy_net, self.value_net, kpm["body_pos"], kpm["body_parent"], kin_sim=self.kin_sim, update_dtype=update_dtype,
                               reference_bugs=reference_bugs, policy_lr=policy_lr, value_lr=value_lr, supervised_lr=supervised_lr, num_optim_epoch=num_optim_epoch,
                               num_step_update=num_step_update, gamma=gamma, tau=tau, clip_epsilon=clip_epsilon, rl_update=rl_update, step_update=step_update,
                               num_epoch_fix=num_epoch_fix, num_epoch=num_epoch, grad_joint=grad_joint, grad_alternate=grad_alternate,
                               cc_policy=self.env.cc_policy if joint_controller else None, train_uhc=train_uhc, policy_weightdecay=policy_weightdecay,
                               value_weightdecay=value_weightdecay, init_update=init_update, num_init_update=num_init_update, step_update_dyna=step_update_dyna,
                               num_step_dyna_update=num_step_dyna_update, full_update=full_update, num_sample=num_sample, batch_size=batch_size, noise_std=noise_std)
        self.reference_bugs, self.min_batch_size = bool(reference_bugs), int(min_batch_size)
        if slice_ratio not in ("slice_start", "behaviour"):
            raise ValueError("slice_ratio must be 'slice_start' or 'behaviour'")
        self.slice_ratio = slice_ratio
        self.sampler = VectorSampler(self.env, self.policy_net, record_qpos=True, source=self.source, pool_depth=pool_depth,
                                     record_full=joint_controller or step_update_dyna)
        self.epoch = 0
        self.result_dir, self.eval_envs, self.test_datasets, self._eval = result_dir, eval_envs, [], {}
        if result_dir is not None and dataset is not None:          # setup_logging (:228-234): resume the sampling history of an earlier run
            import os
            fp = os.path.join(result_dir, "freq_dict.pt")
            if os.path.exists(fp):
                try:
                    import joblib
                    got = joblib.load(fp)
                    if set(got) == set(self.source.freq_dict):
                        self.source.freq_dict = {k: [list(x) for x in v] for k, v in got.items()}
                except Exception:                                    # "error parsing freq_dict, using empty one" (:232-234)
                    pass
        self.sampler.start()

    # the update's state lives in self.upd (kinpoly_amd/update.py); the names the callers and tools know are kept
    trainer = property(lambda self: self.upd.trainer)
    opt_sup = property(lambda self: self.upd.opt_sup)
    sched_sup = property(lambda self: self.upd.sched_sup)
    fk = property(lambda self: self.upd.fk)
    rl_update = property(lambda self: self.upd.rl_update)
    step_update = property(lambda self: self.upd.step_update)
    num_step_update = property(lambda self: self.upd.num_step_update)

    def _setup_supervised_optimizer(self):
        self.upd.setup_supervised_optimizer()

    def train_init(self, warm_update_init=500, warm_update_full=50, num_sample=2000, batch_size=256, scheduled_sampling=0.3, noise_std=0.0):
        """AgentAR.train_init (agent_ar.py:366-385), run once before the first iteration of a fresh run: `update_init_supervised` x
        warm_update_init epochs (the context network learns the clip's first pose), `train_full_supervised(scheduled_sampling=0.3)` x
        warm_update_full epochs (whole-clip kinematic roll-outs against the GT clip), then fresh supervised optimiser / schedule
        (`setup_optimizers`).  cfg.num_sample / cfg.batch_size clips per epoch; kinpoly_amd/pretrain.py.  Gradients are all-reduced over ranks
        (over a fixed parameter list); the scheduled-sampling coins come from ONE job-wide stream, so every rank throws the same frames back
        onto the GT clip and the same parameters get a gradient everywhere."""
        from . import pretrain as P
        ds = self.source.dataset
        assert ds is not None and "wbpos" in ds.data, "the warm start needs a training data set (GT joint positions)"
        u = self.upd
        li = P.update_init_supervised(u.policy, u.opt_sup, u.fk, ds, warm_update_init, num_sample, batch_size, grad_allreduce=_allreduce_grads)
        lf = P.train_full_supervised(u.policy, u.opt_sup, u.fk, ds, warm_update_full, scheduled_sampling, num_sample, batch_size,
                                     noise_std=noise_std, scheduler=u.sched_sup, grad_allreduce=_allreduce_grads, rng=P.job_wide_rng(-1))
        u.setup_supervised_optimizer()
        u.sync_rollout()
        self.restart_sampler()
        return {"init_loss": li, "full_loss": lf}

    def restart_sampler(self):
        """Throw the queued clips away and start every env on a fresh episode: the ring's rows carry init_qpos / init_qvel computed by the context
        network AS IT WAS when they were drawn (INTEGRATION deviation 2), which is fine across PPO iterations (nothing there trains the context
        network) but not across a warm start or a checkpoint load that replaces it."""
        self.sampler.start()

    @property
    def freq_dict(self):
        return self.source.freq_dict

    def _sync_params(self):
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            for p in list(self.policy_net.parameters()) + list(self.value_net.parameters()):
                dist.broadcast(p.data, 0)

    def update_params(self, batch):
        """AgentAR.update_params (agent_ar.py:682-752): kinpoly_amd/update.py::ParamUpdate.update_params on this iteration's batch."""
        return self.upd.update_params(batch, self.epoch, self.source.dataset)

    def update_slices(self, batch):
        """The whole-env slices one sample() call's batch is updated on (min_batch_size; see the constructor): [(lo, hi), ...] over this rank's envs."""
        N, T = batch.rewards.shape
        world = dist.get_world_size(self.trainer.group) if _collective_on(self.trainer.group) else 1
        k = 1 if self.min_batch_size <= 0 else min(N, max(1, -(-N * T * world // self.min_batch_size)))
        per = -(-N // k)
        return [(lo, min(lo + per, N)) for lo in range(0, N, per)]

    def optimize_policy(self, i_iter=None):
        t0 = time.time()
        self.trainer.per_epoch_update()
        batch = self.sampler.sample(self.horizon)
        torch.cuda.synchronize(self.device)
        t1 = time.time()
        slices = self.update_slices(batch)
        if len(slices) == 1:
            info = self.update_params(batch)
            self.epoch += 1
        else:
            N, T, _ = batch.states.shape
            if self.slice_ratio == "behaviour":
                with torch.no_grad():                           # the behaviour policy's log-probabilities of the whole batch, before any slice moves the parameters
                    pol, tr = self.upd.policy, self.trainer
                    means = pol.unroll(tr._cast(batch.states), batch.episode_start, tr._cast(batch.hx0))
                    batch.behaviour_log_probs = pol.log_prob(means.reshape(N * T, -1), tr._cast(batch.actions).reshape(N * T, -1)).detach()
            infos = []
            for k, (lo, hi) in enumerate(slices):
                if k > 0:
                    self.trainer.per_epoch_update()
                infos.append(self.update_params(batch.env_slice(lo, hi)))
                self.epoch += 1
            info = dict(infos[-1])
            info["update_slices"] = len(slices)
            for key in ("surr_loss", "value_loss", "step_loss", "ppo_log_ratio_std"):
                if all(key in x for x in infos):
                    info[key + "_per_slice"] = [x[key] for x in infos]
        torch.cuda.synchronize(self.device)
        t2 = time.time()
        n = batch.rewards.numel()
        log = self.sampler.log
        log.sample_time = t1 - t0
        if _collective_on(self.trainer.group):                   # LoggerRL.merge over the workers (agent_ar.py:677) = over the ranks here
            every = [None] * dist.get_world_size(self.trainer.group)
            dist.all_gather_object(every, log, group=self.trainer.group)
            log = LoggerRL.merge(every, self.reference_bugs)
        info["log"] = log
        if self.result_dir is not None and self.source.dataset is not None and (not dist.is_initialized() or dist.get_rank() == 0):
            import os
            os.makedirs(self.result_dir, exist_ok=True)
            self.source.save_freq_dict(os.path.join(self.result_dir, "freq_dict.pt"))       # joblib.dump(self.freq_dict, ...) after every iteration (:297)
        with torch.no_grad():       # size of the action network's weights (the supervised step updates at lr 5e-4 grow it; a fixed per-weight PPO step then moves the mean further)
            info["policy_param_norm"] = float(torch.sqrt(sum((p.float() ** 2).sum() for n_, p in self.upd.policy.named_parameters() if n_.startswith(("action_mlp", "action_fc", "action_rnn")))))
        info.update(reference_bugs=self.reference_bugs, T_sample=t1 - t0, T_update=t2 - t1, T_total=time.time() - t0, num_steps=n, avg_reward=float(batch.rewards.mean()),
                    fail_rate=float(batch.fails.float().mean()), env_steps_per_s=n / (t1 - t0), episodes=len(batch.episodes.get("percent", ())),
                    pool_exhausted=self.sampler.pool_exhausted, clips_drawn=self.source.n_drawn, init_context_memo_hits=self.source.n_memo_hits, top_ups=self.sampler.top_ups, policy_lr=self.trainer.opt_p.param_groups[0]["lr"])
        return info

    def log_train(self, info, cfg_id="kin_poly", max_iter_num=20000) -> str:
        """The reference's per-iteration log line (agent_ar.py:243-255), from info['log']."""
        log = info["log"]
        done, left = self.epoch, max(max_iter_num - self.epoch, 0)
        eta = left * info["T_total"]
        eta_str = "%02d:%02d:%02d" % (eta // 3600, (eta % 3600) // 60, eta % 60)
        c_info = ",".join("%.4f" % x for x in log.avg_c_info)
        return (f"Ep: {done - 1}\t {cfg_id} \tT_s {info['T_sample']:.2f}\t T_u {info['T_update']:.2f}\tETA {eta_str} \texpert_R_avg {log.avg_c_reward:.4f} [{c_info}]"
                f"\texpert_R_range ({log.min_c_reward:.4f}, {log.max_c_reward:.4f})\teps_len {log.avg_episode_len:.2f}")

    def _eval_engine(self, wild=None):
        """A second, smaller engine for evaluation roll-outs (the training envs are in the middle of their episodes and of their clip rings):
        test mode, its own kinematic twin for the whole-take roll-out of init_context.  The reference switches its one env to test mode
        instead (eval_seq, :463-470)."""
        wild = bool(self.env.wild if wild is None else wild)          # `curr_env = self.env if not loader.cfg.wild else self.env_wild` (:464)
        if wild not in self._eval:
            n = int(self.eval_envs or min(self.env.n, 256))
            env = BatchedHumanoidAREnv(n, self.device.index, mode="test", wild=wild, seed=0, cc_policy=self.env.cc_policy, cc_running_state=self.env.cc_running_state,
                                       model_options=getattr(self.env, "model_options", None))
            import copy
            gt_term = env.reward_cfg.use_gt_term                         # a property of the mode (test: no GT-diff termination), not of the configuration
            env.reward_cfg = copy.deepcopy(self.env.reward_cfg)          # thresholds AND the reward weights cfg.apply_reward_weights put on the training env
            env.reward_cfg.use_gt_term = gt_term
            if hasattr(self.env, "env_episode_len"):
                env.env_episode_len = self.env.env_episode_len
            builder = PolicyARContext(self.policy_net, kpsim.KpSim(env.model, n, self.device.index), smooth=self.ctx_builder.smooth, need_rollout=True, keep_context_feat=False)
            self._eval[wild] = (env, builder)
        return self._eval[wild]

    def eval_policy(self, data_mode="train"):
        """AgentAR.eval_policy (:394-448): every take of the training set (`train`) or of every set in `self.test_datasets` (`test`) is played whole
        with mean actions; a take counts as covered when it runs to its end (`percent == 1`).  `train` feeds the result back into the sampling
        history (one [percent, 0] entry for a covered take, three for a failed one, :427-433).  Writes / extends `eval_dict_<mode>.pt` under
        result_dir.  Returns the reference's list of {"coverage_<name>": {"mean_coverage", "num_coverage", "all_coverage"}}."""
        from .evaluate import eval_dataset
        sets = [self.source.dataset] if data_mode == "train" else list(self.test_datasets)
        out = []
        for ds in sets:
            env, builder = self._eval_engine(getattr(ds, "wild", None))
            res = eval_dataset(env, self.policy_net, builder, ds)
            if _collective_on(self.trainer.group):
                # every rank plays the takes (replicas are identical), but run-to-run differences of the device arithmetic could flip a `percent == 1` on one
                # rank only and the job-wide sampling history would part: rank 0's percents are everybody's
                box = [{k: r["percent"] for k, r in res.items()}]
                dist.broadcast_object_list(box, src=0, group=self.trainer.group)
                for k, pc in box[0].items():
                    res[k]["percent"] = pc
            ok = {k: r["percent"] == 1 for k, r in res.items()}
            if data_mode == "train":
                for k, r in res.items():
                    self.source.freq_dict[k].extend([[r["percent"], 0]] * (1 if ok[k] else 3))
                self.source._probs = None
            if self.result_dir is not None and (not dist.is_initialized() or dist.get_rank() == 0):
                import os
                import joblib
                os.makedirs(self.result_dir, exist_ok=True)
                path = os.path.join(self.result_dir, f"eval_dict_{data_mode}.pt")
                hist = joblib.load(path) if os.path.exists(path) else {}
                hist[self.epoch] = {k: r["percent"] for k, r in res.items()}
                joblib.dump(hist, path)
            name = getattr(ds, "name", data_mode)
            out.append({f"coverage_{name}": {"mean_coverage": float(sum(ok.values())) / max(len(ok), 1), "num_coverage": int(sum(ok.values())), "all_coverage": len(ok)}})
        return out

    def save_checkpoint(self, path):
        return ck.save_checkpoint(path, self.policy_net, self.value_net, None, self.env.cc_policy)

    def load_checkpoint(self, path):
        cp = ck.load_checkpoint(path)
        ck.load_state_strict(self.policy_net, ck.split_policy_dict(cp["policy_dict"]), what=str(path))      # missing / unknown keys raise (allow-list: checkpoint.py)
        self.value_net.load_state_dict(cp["value_dict"])
        self.upd.load_from_rollout()
        if "cc_dict" in cp:
            self.env.cc_policy.load_state_dict(cp["cc_dict"])
        self.restart_sampler()
        return cp
