"""Vectorised rollout driver + GAE/PPO update: the replacement of AgentAR.sample / sample_worker /
update_params (kin_poly/core/agent_ar.py:510-611, 651-680, 682-772) for N environments per GPU.

Instead of forking `num_threads` workers that each own one MjSim and push Python rows through a
multiprocessing.Queue, one process per GPU steps all N environments in lock-step; experience lives in
device-resident SoA buffers laid out env-major [N, T, .] (each env's rows contiguous and in time order --
the ordering GAE and the GRU re-unroll of the reference rely on).  Across GPUs the
environments shard by rank; the only exchange is the all-gather of advantages / returns for the global
normalisation (uhc/khrylib/rl/core/common.py:22) and, for a data-parallel update, a gradient all-reduce.

Episode semantics follow sample_worker (agent_ar.py:518-606): every episode draws its own clip
(`sample_seq` -> `init_context` -> `load_context` -> `reset`), runs until `done`, and leaves `[percent, fr_start]` in the
take's `freq_dict` entry, which steers the next draws.  The draws of the NEXT episodes are made ahead of the rollout, batched
(`EpisodeSource`: N clips + one batched init_context per pool level), and kept resident as extra context rows; a finished env
moves to its next row on the device (no host sync inside the loop).  What differs, by construction of a lock-step sampler: the
horizon T is fixed, so an episode can straddle two sample() calls -- its hidden state is carried over (`RolloutBatch.hx0`) and
the cut is bootstrapped with V (kp_gae_bootstrap) instead of being forced terminal.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np
import torch
import torch.distributed as dist

from . import sim as kpsim
from .env import BatchedHumanoidAREnv
from .nets import KinPolicy, Value


@dataclass
class RolloutBatch:
    """TrajBatchEgo (kin_poly/core/trajbatch_ego.py:5-14 over uhc/khrylib/rl/core/trajbatch.py:4-16), env-major [N, T, .]."""
    states: torch.Tensor         # [N, T, 105]
    actions: torch.Tensor        # [N, T, 80]
    rewards: torch.Tensor        # [N, T]
    masks: torch.Tensor          # [N, T]   0 where the episode ended at this row
    episode_start: torch.Tensor  # [N, T]   bool: hidden state is zero before this row (initialize_rnn's episode boundaries)
    fails: torch.Tensor          # [N, T]   bool
    curr_qpos: torch.Tensor | None = None       # [N, T, 76]  env.get_humanoid_qpos() before the step
    gt_target_qpos: torch.Tensor | None = None  # [N, T, 76]  ar_context['qpos'][cur_t + 1]
    next_states: torch.Tensor | None = None     # [N, T, 105]
    exps: torch.Tensor | None = None            # [N, T]      1 (agent_ar.py:582)
    v_metas: torch.Tensor | None = None         # [N, T, 3]   curr_take_ind, fr_start, fr_num (agent_ar.py:627-631)
    res_qpos: torch.Tensor | None = None        # [N, T, 76]  env.get_humanoid_qpos() after the step
    cc_action: torch.Tensor | None = None       # [N, T, 75]  info['cc_action']
    cc_state: torch.Tensor | None = None        # [N, T, 784] info['cc_state']
    hx0: torch.Tensor | None = None             # [N, H]      GRU state before row 0 (episodes continuing from the previous call)
    last_states: torch.Tensor | None = None     # [N, 105]    observation after the last row (bootstrap of cut episodes)
    episodes: dict = field(default_factory=dict)   # finished episodes: take_ind, fr_start, percent (numpy), for freq_dict


class EpisodeSource:
    """`sample_seq` + `init_context` for n episodes at once (agent_ar.py:519-533).

    dataset: kinpoly_amd.dataset.StateARDataset (or None with `context_fn(n) -> dict`); ctx_builder: PolicyARContext or None
    (contexts already carry init_qpos / init_qvel).  Keeps the reference's `freq_dict` (take -> list of [percent, fr_start],
    last 5000 kept, agent_ar.py:668-676) and passes it with sampling_temp / sampling_freq to every draw."""

    def __init__(self, dataset=None, ctx_builder=None, context_fn=None, sampling_temp=0.5, sampling_freq=0.9, fix_height=False):
        assert (dataset is None) != (context_fn is None), "give a dataset or a context_fn"
        self.dataset, self.ctx_builder, self.context_fn = dataset, ctx_builder, context_fn
        self.sampling_temp, self.sampling_freq, self.fix_height = sampling_temp, sampling_freq, fix_height
        self.freq_dict = {k: [] for k in dataset.takes} if dataset is not None else {}

    def draw(self, n: int, device) -> dict:
        if self.dataset is not None:
            data = self.dataset.sample_batch(n, freq_dict=self.freq_dict, sampling_temp=self.sampling_temp, sampling_freq=self.sampling_freq)
        else:
            data = self.context_fn(n)
        data = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in data.items()}
        if self.ctx_builder is not None:
            data = self.ctx_builder.init_context(data, fix_height=self.fix_height)
        elif "init_qpos" not in data:         # no context network: the episode starts on the clip's first frame
            data["init_qpos"] = data["qpos"][:, 0].contiguous()
            data["init_qvel"] = data["qvel"][:, 0].contiguous() if "qvel" in data else torch.zeros((n, 75), device=device)
        return data

    def record(self, take_ind, fr_start, percent, group=None):
        """freq_dict[curr_key].append([info['percent'], fr_start]) for every finished episode (agent_ar.py:601-603).

        ONE job-wide freq_dict, as in the reference: AgentAR.sample merges every worker's list into the agent's dict
        (agent_ar.py:664-673) before the next draws.  With several ranks the finished episodes of all ranks are exchanged once per
        sample() call (`all_gather_object`: a few KB of Python lists, host side) and appended in rank order, so every rank holds the
        same dict and draws its next clips from the same take probabilities."""
        if self.dataset is None:
            return
        mine = [[int(ti), int(fs), float(pc)] for ti, fs, pc in zip(take_ind, fr_start, percent)]
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            every = [None] * dist.get_world_size(group)
            dist.all_gather_object(every, mine, group=group)
        else:
            every = [mine]
        for part in every:
            for ti, fs, pc in part:
                self.freq_dict[self.dataset.takes[ti]].append([pc, fs])
        self.freq_dict = {k: (v if len(v) < 5000 else v[-5000:]) for k, v in self.freq_dict.items()}

    def save_freq_dict(self, path):
        """joblib.dump(self.freq_dict, 'freq_dict.pt') of the reference (agent_ar.py:297): a pickle of {take: [[percent, fr_start], ...]}."""
        import pickle
        with open(path, "wb") as f:
            pickle.dump(self.freq_dict, f)


_ROW_KEYS = ("qpos", "head_pose", "head_vels", "obj_head_relative_poses", "action_one_hot", "init_qpos", "init_qvel", "obj_pose", "ar_qpos", "ar_qvel")


class VectorSampler:
    """Fixed-horizon lock-step sampler with device-side auto-reset (no host sync inside the loop).

    Without a `source` a finished env restarts on its own clip (the synthetic single-clip configs of SURVEY 8d).  With one, every
    episode gets a freshly drawn clip: `pool_depth` levels of N pre-drawn contexts sit behind the current ones as extra context
    rows, `done` moves an env one level down.  An env that exhausts the pool inside one call restarts on its last clip and is
    counted in `pool_exhausted` (size pool_depth for the shortest episodes you expect: T / mean episode length)."""

    def __init__(self, env: BatchedHumanoidAREnv, policy: KinPolicy, record_qpos: bool = False, mean_action: bool = False,
                 source: EpisodeSource | None = None, pool_depth: int = 2, record_full: bool = False):
        self.env, self.policy, self.record_qpos, self.mean_action = env, policy, record_qpos or record_full, mean_action
        self.source, self.pool_depth, self.record_full = source, int(pool_depth), record_full
        self.obs = self.hx = self.fresh = None
        self.level = None
        self.pool_exhausted = 0
        self._replay = None       # bool [N]: the env is re-running a clip it already finished (pool exhausted): not evidence for freq_dict
        self.group = None         # process group of the job-wide freq_dict exchange (None = the default group)

    # ------------------------------------------------------------------ episode pool
    def _refill(self):
        """Row table of this call: level 0 = the clips the envs are on now, levels 1..D = freshly drawn next episodes."""
        env, N, dev = self.env, self.env.n, self.env.device
        levels = []
        if env.ctx is not None and self.obs is not None:
            r = env.row.long()
            cur = {k: env.ctx[k][r] for k in _ROW_KEYS if k in env.ctx}
            cur["len"] = env.row_len[r] + 1
            cur["take_ind"], cur["fr_start"] = env.row_meta[r, 0], env.row_meta[r, 1]
            levels.append(cur)
        n_new = self.pool_depth + (0 if levels else 1)
        for _ in range(n_new):
            d = self.source.draw(N, dev)
            T = d["qpos"].shape[1]
            lv = {k: d[k] for k in _ROW_KEYS if k in d}
            if lv["action_one_hot"].dim() == 3:
                lv["action_one_hot"] = lv["action_one_hot"][:, 0]
            lv["len"] = torch.as_tensor(d["len"], device=dev).to(torch.int32) if "len" in d else torch.full((N,), T, dtype=torch.int32, device=dev)
            for k in ("take_ind", "fr_start"):
                lv[k] = torch.as_tensor(d[k]).to(dev, torch.float32) if k in d else torch.zeros(N, device=dev)
            levels.append(lv)
        keys = [k for k in levels[0] if all(k in lv for lv in levels)]
        # draws can come back with different clip lengths (batch() pads to the longest take of THAT draw): bring every level to the
        # common T by repeating its last frame, as StateARDataset.batch pads, so the row table is one [R, T, .] block
        T_all = max(lv["qpos"].shape[1] for lv in levels)
        for lv in levels:
            T_lv = lv["qpos"].shape[1]
            if T_lv < T_all:
                for k in keys:
                    v = lv[k]
                    if torch.is_tensor(v) and v.dim() == 3 and v.shape[1] == T_lv:
                        lv[k] = torch.cat([v, v[:, -1:].expand(-1, T_all - T_lv, -1)], 1)
        table = {k: torch.cat([lv[k].to(dev) for lv in levels], 0) for k in keys}
        first = self.obs is None
        env.load_context(table, row=torch.arange(N, device=dev, dtype=torch.int32), keep_state=not first)
        self.level = torch.zeros(N, dtype=torch.int64, device=dev)
        self._n_levels = len(levels)

    def start(self):
        if self.source is not None:
            self.obs = None
            self._refill()
        self.obs = self.env.reset().clone()
        self.hx = self.policy.init_hidden(self.env.n, self.env.device)
        self.fresh = torch.ones(self.env.n, dtype=torch.bool, device=self.env.device)

    @torch.no_grad()
    def sample(self, T: int) -> RolloutBatch:
        env, pol, N, dev = self.env, self.policy, self.env.n, self.env.device
        if self.obs is None:
            self.start()
        elif self.source is not None:
            self._refill()
        f = lambda *s: torch.empty((N, T, *s), device=dev)  # noqa: E731
        S, A, R = f(105), f(80), f()
        E = torch.empty((N, T), dtype=torch.bool, device=dev); F = torch.empty((N, T), dtype=torch.bool, device=dev)
        Q = f(76) if self.record_qpos else None
        G = f(76) if self.record_qpos else None
        full = self.record_full
        NS, VM, RQ, CA, CS = (f(105), f(3), f(76), f(75), f(784)) if full else (None,) * 5
        D = torch.empty((N, T), dtype=torch.bool, device=dev); PC = f(); MT = f(2)
        REC = torch.empty((N, T), dtype=torch.bool, device=dev)      # finished episodes that count for freq_dict (not the replays of an exhausted pool)
        if self._replay is None:
            self._replay = torch.zeros(N, dtype=torch.bool, device=dev)
        ar = torch.arange(N, device=dev)
        hx0 = self.hx.clone()
        fr_num = float(env.ctx["qpos"].shape[1])
        exhausted = torch.zeros((), dtype=torch.int64, device=dev)
        # the exploration noise of both policies for the whole horizon in one launch: [T, N, 80 kinematic + 75 UHC]
        cc_mean = env.mode == "test" or (env.mode == "train" and env.joint_controller)
        noise = None if (self.mean_action and cc_mean) else torch.randn((T, N, 155), device=dev, generator=env.gen)
        for t in range(T):
            S[:, t] = self.obs
            E[:, t] = self.fresh
            action, self.hx = pol.select_action(self.obs, self.hx, self.mean_action, env.gen, None if noise is None else noise[t, :, :80])
            action = action.contiguous()
            row = env.row.long()
            if self.record_qpos:
                Q[:, t] = env.sim.get("qpos")
                G[:, t] = env.ctx["qpos"][row, torch.minimum(env.cur_t.long() + 1, env.ctx_len.long())]
            meta = env.row_meta[row]
            obs, _, done, info = env.step(action, need_obs=full, cc_noise=None if noise is None else noise[t, :, 80:])
            A[:, t] = action
            R[:, t] = info["custom_reward"]
            F[:, t] = info["fail"]
            D[:, t] = done; PC[:, t] = info["percent"]; MT[:, t] = meta
            REC[:, t] = done & ~self._replay
            if full:
                NS[:, t] = obs; RQ[:, t] = env.sim.get("qpos"); CA[:, t] = info["cc_action"]; CS[:, t] = info["cc_state"]
                VM[:, t, :2] = meta; VM[:, t, 2] = fr_num
            # device-side episode turnover: finished envs move to their next pre-drawn clip (masked row switch), then the masked reset
            if self.source is not None:
                nxt = self.level + done.long()
                over = nxt >= self._n_levels
                exhausted += (over & done).sum()
                self._replay = torch.where(done, over, self._replay)     # a finished env that found no fresh clip replays its last one
                self.level = torch.where(over, self.level, nxt)
                env.set_rows((self.level * N + ar).to(torch.int32), done)
            # env._obs: step() writes its own observation elsewhere, so this stays valid through the next step; the same launch zeroes the GRU state
            # of the finished envs in place (self.hx is this step's fresh output of select_action; hx0 above is a copy)
            self.obs = env.reset(done, policy_state=self.hx)
            self.fresh = D[:, t]                # `done` itself lives in a buffer the step after next reuses
        M = (~D).float()
        # one host transfer per call: finished episodes -> freq_dict, launch status
        status = int(env.sim.status_tensor()[2])
        if status:
            raise kpsim.KinPolyNativeError("kp_step_queue_kernel reported a stalled job queue during the rollout (states are incomplete)")
        self.pool_exhausted += int(exhausted)
        dm = REC.cpu().numpy()
        eps = {"take_ind": MT[..., 0].cpu().numpy()[dm].astype(np.int64), "fr_start": MT[..., 1].cpu().numpy()[dm].astype(np.int64),
               "percent": PC.cpu().numpy()[dm].astype(np.float64)}
        if self.source is not None:
            self.source.record(eps["take_ind"], eps["fr_start"], eps["percent"], self.group)
        return RolloutBatch(S, A, R, M, E, F, Q, G, NS, torch.ones((N, T), device=dev) if full else None, VM, RQ, CA, CS, hx0, self.obs.clone(), eps)


# ---------------------------------------------------------------------------------------------- update
def env_shard(rank: int, world_size: int, envs_per_gpu: int):
    """Rank r owns global environments [r * envs_per_gpu, (r + 1) * envs_per_gpu) and seed stream 4 + r."""
    return range(rank * envs_per_gpu, (rank + 1) * envs_per_gpu), 4 + rank


FORCE_COLLECTIVES = False      # tests: run the all-gather / all-reduce even in a 1-rank group (so a 1-GPU box pushes device tensors through RCCL)


def _collective_on(group=None):
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or FORCE_COLLECTIVES)


def normalize_advantages_global(adv: torch.Tensor, ret: torch.Tensor, group=None):
    """The reference normalises advantages over its whole concatenated batch (common.py:22, unbiased std).
    Sharded over ranks, the per-rank advantages / returns are all-gathered (one RCCL all-gather over xGMI per
    PPO iteration; gloo in the CPU tests) so every rank applies the whole-job mean / std.  Returns
    (normalised local adv, local ret, gathered returns [world * n])."""
    if _collective_on(group):
        ws = dist.get_world_size(group)
        packed = torch.stack([adv.reshape(-1), ret.reshape(-1)], 1).contiguous()
        gathered = [torch.empty_like(packed) for _ in range(ws)]
        dist.all_gather(gathered, packed, group=group)
        all_adv = torch.cat([g[:, 0] for g in gathered]); all_ret = torch.cat([g[:, 1] for g in gathered])
    else:
        all_adv, all_ret = adv.reshape(-1), ret.reshape(-1)
    return (adv - all_adv.mean()) / all_adv.std(), ret, all_ret


def estimate_advantages(rewards, masks, values, gamma, tau, group=None, last_values=None):
    """GAE on the device (k_gae, env-major reverse scan) + the global normalisation above."""
    adv, ret = kpsim.gae(rewards.contiguous(), masks.contiguous(), values.contiguous(), gamma, tau,
                         None if last_values is None else last_values.contiguous())
    adv, ret, _ = normalize_advantages_global(adv, ret, group)
    return adv, ret


def _allreduce_grads(params, group=None):
    if not _collective_on(group):
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


def lambda_lr(optimizer, nepoch_fix, nepoch):
    """get_scheduler(policy='lambda') (uhc/khrylib/utils/torch.py:166-171): lr factor 1 for nepoch_fix epochs, then linear decay."""
    return torch.optim.lr_scheduler.LambdaLR(optimizer, lr_lambda=lambda epoch: 1.0 - max(0, epoch - nepoch_fix) / float(nepoch - nepoch_fix + 1))


class PPOTrainer:
    """AgentPPO.update_policy / ppo_loss / update_value (agent_ar.py:756-772, 852-870; agent_ppo.py:53-56), the LambdaLR schedules of
    agent_ar.py:215-225 stepped once per iteration (`per_epoch_update`, :268-269), and `update_controller` (:774-794) for
    joint_controller runs.

    The optimiser owns the kinematic policy's parameters ONLY, as the reference's does (`Adam(self.policy_net.parameters())`,
    agent_ar.py:184-199; `policy_grad_clip=[(self.policy_net.parameters(), 40)]`, :93): `joint_controller` appends env.cc_policy to
    update_modules / sample_modules (:97-99, train / eval mode and device moves), not to the optimiser.  The reference's
    update_controller therefore back-propagates the surrogate into the UHC and then steps an optimiser that does not hold the UHC
    weights: the UHC is never trained there, and neither is it here by default.  `train_uhc=True` is this engine's opt-in extension:
    the UHC gets its own Adam (same lr / weight decay) and its own 40-norm clip, so that the kinematic policy's clip is untouched."""

    def __init__(self, policy: KinPolicy, value: Value, gamma=0.95, tau=0.95, clip_epsilon=0.2, policy_lr=1e-5, value_lr=3e-4,
                 num_optim_epoch=10, policy_grad_clip=40.0, group=None, num_epoch_fix=100, num_epoch=10000, value_opt_niter=1,
                 cc_policy=None, policy_weightdecay=0.0, value_weightdecay=0.0, train_uhc=False):
        self.policy, self.value, self.group, self.cc_policy = policy, value, group, cc_policy
        self.gamma, self.tau, self.clip_epsilon, self.num_optim_epoch, self.policy_grad_clip = gamma, tau, clip_epsilon, num_optim_epoch, policy_grad_clip
        self.value_opt_niter = value_opt_niter
        self.opt_p = torch.optim.Adam([p for p in policy.parameters() if p.requires_grad], lr=policy_lr, weight_decay=policy_weightdecay)
        self.opt_v = torch.optim.Adam(value.parameters(), lr=value_lr, weight_decay=value_weightdecay)
        self.sched_p = lambda_lr(self.opt_p, num_epoch_fix, num_epoch)
        self.sched_v = lambda_lr(self.opt_v, num_epoch_fix, num_epoch)
        self.opt_cc = self.sched_cc = None
        if cc_policy is not None and train_uhc:
            cc_params = [p for p in cc_policy.parameters() if p.dtype.is_floating_point and p is not cc_policy.action_log_std]
            for p in cc_params:
                p.requires_grad_(True)
            self.opt_cc = torch.optim.Adam(cc_params, lr=policy_lr, weight_decay=policy_weightdecay)
            self.sched_cc = lambda_lr(self.opt_cc, num_epoch_fix, num_epoch)

    def per_epoch_update(self):
        """scheduler_policy.step(); scheduler_value.step()   (agent_ar.py:268-269, called at the top of optimize_policy)."""
        self.sched_p.step(); self.sched_v.step()
        if self.sched_cc is not None:
            self.sched_cc.step()

    def _clip(self, opt=None):
        params = [p for g in (opt or self.opt_p).param_groups for p in g["params"]]
        _allreduce_grads(params, self.group)
        torch.nn.utils.clip_grad_norm_(params, self.policy_grad_clip)

    def update(self, batch: RolloutBatch, bootstrap: bool = True):
        N, T, _ = batch.states.shape
        flat_states = batch.states.reshape(N * T, -1)
        ind = None
        if batch.exps is not None:                # `ind = exps.nonzero()` (agent_ar.py:763): the rows the surrogate is taken over
            ind = batch.exps.reshape(-1).nonzero(as_tuple=False).squeeze(1)
            if ind.numel() == N * T:
                ind = None
        with torch.no_grad():
            values = self.value(flat_states).view(N, T)
            last_v = self.value(batch.last_states).view(N) if (bootstrap and batch.last_states is not None) else None
            means = self.policy.unroll(batch.states, batch.episode_start, batch.hx0)
            fixed_log_probs = self.policy.log_prob(means.reshape(N * T, -1), batch.actions.reshape(N * T, -1))
        adv, ret = estimate_advantages(batch.rewards, batch.masks, values, self.gamma, self.tau, self.group, last_v)
        adv, ret = adv.reshape(-1, 1), ret.reshape(-1, 1)
        stats = {}
        for _ in range(self.num_optim_epoch):
            for _ in range(self.value_opt_niter):
                vloss = (self.value(flat_states) - ret).pow(2).mean()
                self.opt_v.zero_grad(); vloss.backward(); _allreduce_grads(list(self.value.parameters()), self.group); self.opt_v.step()
            means = self.policy.unroll(batch.states, batch.episode_start, batch.hx0)
            log_probs = self.policy.log_prob(means.reshape(N * T, -1), batch.actions.reshape(N * T, -1))
            surr = ppo_surrogate(log_probs, fixed_log_probs, adv, self.clip_epsilon, ind)
            self.opt_p.zero_grad(); surr.backward()
            self._clip()
            self.opt_p.step()
            stats = {"value_loss": float(vloss.detach()), "surr_loss": float(surr.detach())}
        if self.cc_policy is not None and batch.cc_state is not None:
            stats["cc_surr_loss"] = self.update_controller(batch, adv, ind)
        return stats

    def update_joint(self, batch: RolloutBatch, fk, grad_alternate: bool = False, epoch: int = 0, sup_optimizer=None, bootstrap: bool = True):
        """AgentAR.update_policy_joint (agent_ar.py:796-850; `grad_joint` runs): per epoch one value step, then the PPO surrogate and the
        supervised one-step loss (TrajARNet.step on the mean action + compute_loss_lite against the GT next pose) in ONE policy step,
        loss = 10 * loss_step + surr; with `grad_alternate` odd epochs take the surrogate step and even epochs the supervised step (on
        `sup_optimizer`, the reference's policy_net.optimizer).  fk: kinpoly_amd.supervised.TorchFK."""
        from .supervised import compute_loss_lite, kinematic_step
        assert batch.curr_qpos is not None and batch.gt_target_qpos is not None, "sample with record_qpos=True"
        N, T, _ = batch.states.shape
        flat_states = batch.states.reshape(N * T, -1)
        curr, tgt = batch.curr_qpos.reshape(N * T, 76), batch.gt_target_qpos.reshape(N * T, 76)
        ind = None
        if batch.exps is not None:                # `ind = exps.nonzero()` (agent_ar.py:813): the rows the surrogate is taken over
            ind = batch.exps.reshape(-1).nonzero(as_tuple=False).squeeze(1)
            if ind.numel() == N * T:
                ind = None
        with torch.no_grad():
            values = self.value(flat_states).view(N, T)
            last_v = self.value(batch.last_states).view(N) if (bootstrap and batch.last_states is not None) else None
            means = self.policy.unroll(batch.states, batch.episode_start, batch.hx0)
            fixed_log_probs = self.policy.log_prob(means.reshape(N * T, -1), batch.actions.reshape(N * T, -1))
            tgt_wbpos = fk.wbpos(tgt)
        adv, ret = estimate_advantages(batch.rewards, batch.masks, values, self.gamma, self.tau, self.group, last_v)
        adv, ret = adv.reshape(-1, 1), ret.reshape(-1, 1)
        stats = {}
        for _ in range(self.num_optim_epoch):
            vloss = (self.value(flat_states) - ret).pow(2).mean()
            self.opt_v.zero_grad(); vloss.backward(); _allreduce_grads(list(self.value.parameters()), self.group); self.opt_v.step()
            means = self.policy.unroll(batch.states, batch.episode_start, batch.hx0).reshape(N * T, -1)
            surr = ppo_surrogate(self.policy.log_prob(means, batch.actions.reshape(N * T, -1)), fixed_log_probs, adv, self.clip_epsilon, ind)
            loss_step, _ = compute_loss_lite(fk, kinematic_step(curr, means), tgt, gt_wbpos=tgt_wbpos)
            if grad_alternate:
                if epoch % 2 == 1:
                    self.opt_p.zero_grad(); surr.backward(); self._clip(); self.opt_p.step()
                else:
                    opt = sup_optimizer if sup_optimizer is not None else self.opt_p
                    opt.zero_grad(); loss_step.backward()
                    _allreduce_grads([p for g in opt.param_groups for p in g["params"]], self.group)
                    opt.step()
            else:
                loss = loss_step * 10 + surr
                self.opt_p.zero_grad(); loss.backward(); self._clip(); self.opt_p.step()
            stats = {"value_loss": float(vloss.detach()), "surr_loss": float(surr.detach()), "step_loss": float(loss_step.detach())}
        if self.cc_policy is not None and batch.cc_state is not None:      # update_params runs update_controller after either branch (agent_ar.py:748-749)
            stats["cc_surr_loss"] = self.update_controller(batch, adv, ind)
        return stats

    def update_controller(self, batch: RolloutBatch, adv, ind=None):
        """AgentAR.update_controller (agent_ar.py:774-794): the clipped surrogate of env.cc_policy over the recorded (cc_state, cc_action)
        with the kinematic policy's advantages; no value step.  The reference steps `optimizer_policy`, which holds policy_net's
        parameters only, after a backward that reaches the UHC only: with torch >= 2.0 (`zero_grad(set_to_none=True)`) policy_net's
        gradients are None and the step changes nothing, so its num_optim_epoch passes all evaluate the same loss -- computed once here
        and reported.  (Under torch < 2.0 the same code takes ten zero-gradient Adam steps on policy_net, i.e. momentum drift; not
        reproduced.)  With `train_uhc` the epochs run on the UHC's own optimiser."""
        cs, ca = batch.cc_state.reshape(-1, batch.cc_state.shape[-1]), batch.cc_action.reshape(-1, batch.cc_action.shape[-1])
        pol = self.cc_policy

        def logp(x, a):
            mean, log_std = pol.forward(x)
            var = torch.exp(2 * log_std)
            return (-(a - mean) ** 2 / (2 * var) - 0.5 * np.log(2 * np.pi) - log_std).sum(1, keepdim=True)
        with torch.no_grad():
            fixed = logp(cs, ca)
            if self.opt_cc is None:
                return float(ppo_surrogate(fixed, fixed, adv, self.clip_epsilon, ind))
        loss = None
        for _ in range(self.num_optim_epoch):
            surr = ppo_surrogate(logp(cs, ca), fixed, adv, self.clip_epsilon, ind)
            self.opt_cc.zero_grad(); surr.backward()
            self._clip(self.opt_cc)
            self.opt_cc.step()
            loss = float(surr.detach())
        return loss
