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
moves to its next row on the device; one host read every `pool_depth` steps tells the sampler how many queued clips were used up, and
exactly that many are drawn and written in place, so the pool never runs dry and no clip is played twice.  What differs, by construction of a lock-step sampler: the
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
    behaviour_log_probs: torch.Tensor | None = None   # [N * T, 1] log-probabilities of `actions` under the policy that SAMPLED them (set when the batch is updated on in
                                                      # slices, AgentAR.min_batch_size: later slices start from parameters the earlier ones have moved)

    def env_slice(self, lo: int, hi: int) -> "RolloutBatch":
        """rows of the envs [lo, hi): a whole-env slice is a valid batch of its own (its own hx0 and bootstrap observations)"""
        N, T = self.rewards.shape
        cut = lambda x: None if x is None else x[lo:hi]      # noqa: E731
        blp = None if self.behaviour_log_probs is None else self.behaviour_log_probs.view(N, T, -1)[lo:hi].reshape((hi - lo) * T, -1)
        return RolloutBatch(cut(self.states), cut(self.actions), cut(self.rewards), cut(self.masks), cut(self.episode_start), cut(self.fails), cut(self.curr_qpos),
                            cut(self.gt_target_qpos), cut(self.next_states), cut(self.exps), cut(self.v_metas), cut(self.res_qpos), cut(self.cc_action), cut(self.cc_state),
                            cut(self.hx0), cut(self.last_states), self.episodes, blp)


def _agree_status(status: int, device, group=None) -> int:
    """max of a per-rank status word over the job (one tiny all-reduce), so that every rank raises -- or none does -- and no rank is left
    waiting in the next collective for one that threw."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        t = torch.tensor([int(status)], dtype=torch.int32, device=device if dist.get_backend(group) == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        return int(t.item())
    return int(status)


class LoggerRL:
    """The sampling statistics of uhc/khrylib/rl/core/logger_rl.py (the object AgentAR.sample returns next to the batch and log_train prints,
    agent_ar.py:243-262), for a lock-step sampler: built once per sample() call from the device buffers (`episode_log`), merged over ranks
    with `merge` (LoggerRL.merge :51-70).  An episode that straddles two sample() calls is counted in the call it ends in, with its whole return.

    `merge` reproduces the reference as it is: `min_episode_reward = max(...)` over the workers (logger_rl.py:60 -- a typo there, but "results identical
    to the reference's" is the bar); `LoggerRL.REFERENCE_BUGS = False` (or `merge(..., reference_bugs=False)`) takes the min."""

    REFERENCE_BUGS = True

    FIELDS = ("num_steps", "num_episodes", "total_reward", "min_episode_reward", "max_episode_reward", "total_c_reward", "min_c_reward", "max_c_reward")

    def __init__(self, **kw):
        self.num_steps = self.num_episodes = 0
        self.total_reward = self.total_c_reward = 0.0
        self.min_episode_reward = self.min_c_reward = float("inf")
        self.max_episode_reward = self.max_c_reward = float("-inf")
        self.total_c_info = np.zeros(6)
        self.sample_time = 0.0
        for k, v in kw.items():
            setattr(self, k, v)
        self.end_sampling()

    def end_sampling(self):
        ne, ns = max(self.num_episodes, 1), max(self.num_steps, 1)        # a call in which no episode ended reports per-episode figures over 1
        self.avg_episode_len = self.num_steps / ne
        self.avg_episode_reward = self.total_reward / ne
        self.avg_c_reward = self.total_c_reward / ns
        self.avg_c_info = self.total_c_info / ns
        self.avg_episode_c_reward = self.total_c_reward / ne
        self.avg_episode_c_info = self.total_c_info / ne

    @classmethod
    def merge(cls, loggers, reference_bugs=None):
        loggers = list(loggers)
        min_of_mins = max if (cls.REFERENCE_BUGS if reference_bugs is None else reference_bugs) else min
        out = cls(num_steps=sum(x.num_steps for x in loggers), num_episodes=sum(x.num_episodes for x in loggers),
                  total_reward=sum(x.total_reward for x in loggers), total_c_reward=sum(x.total_c_reward for x in loggers),
                  total_c_info=sum(x.total_c_info for x in loggers),
                  min_episode_reward=min_of_mins(x.min_episode_reward for x in loggers), max_episode_reward=max(x.max_episode_reward for x in loggers),
                  min_c_reward=min(x.min_c_reward for x in loggers), max_c_reward=max(x.max_c_reward for x in loggers))
        out.sample_time = max(x.sample_time for x in loggers)
        return out

    def as_dict(self):
        return {**{k: getattr(self, k) for k in self.FIELDS}, "total_c_info": self.total_c_info.tolist(), "avg_episode_len": self.avg_episode_len,
                "avg_episode_reward": self.avg_episode_reward, "avg_c_reward": self.avg_c_reward, "avg_c_info": self.avg_c_info.tolist()}


def episode_log(rewards: torch.Tensor, done: torch.Tensor, c_info: torch.Tensor | None, carry: torch.Tensor):
    """LoggerRL's counters from one sample() call's buffers, without a per-step launch: rewards [N, T] (>= 0: sums of exponentials), done bool
    [N, T], c_info [N, T, 6] or None, carry [N] = return collected so far by the episode every env was in when the call began.  Returns
    (stats float64 [8 + 6] in LoggerRL.FIELDS order followed by total_c_info, new carry [N]).  Pure torch (CPU-testable).

    The return of the episode that ends at (e, t) is cs[e, t] - cs[e, t'] with t' its predecessor's last row (cs = running sum over the call,
    + carry when there is no predecessor in this call); rewards are non-negative, so cs at the latest done row is a running max."""
    N, T = rewards.shape
    if carry is None:                          # a caller that drives the sampler's pieces by hand and never went through start(): every env begins an episode
        carry = torch.zeros(N, dtype=torch.float64, device=rewards.device)
    r = rewards.double()
    cs = torch.cumsum(r, 1)
    at_done = torch.where(done, cs, torch.zeros_like(cs))
    last = torch.cummax(at_done, 1).values                               # cs at the latest done row <= t
    prev = torch.cat([torch.zeros((N, 1), dtype=cs.dtype, device=cs.device), last[:, :-1]], 1)     # ... < t
    done_so_far = torch.cummax(done.to(torch.uint8), 1).values.bool()
    seen = torch.cat([torch.zeros((N, 1), dtype=torch.bool, device=cs.device), done_so_far[:, :-1]], 1)    # a done row before t
    ep_ret = cs - prev + torch.where(seen, torch.zeros_like(cs), carry.double()[:, None].expand(N, T))
    inf = torch.full_like(cs, float("inf"))
    n_ep = done.sum()
    stats = torch.stack([torch.tensor(float(N * T), dtype=cs.dtype, device=cs.device), n_ep.double(),
                         torch.where(done, ep_ret, torch.zeros_like(cs)).sum(), torch.where(done, ep_ret, inf).min(), torch.where(done, ep_ret, -inf).max(),
                         r.sum(), r.min(), r.max()])
    ci = c_info.double().sum((0, 1)) if c_info is not None else torch.zeros(6, dtype=cs.dtype, device=cs.device)
    any_done = done_so_far[:, -1]
    new_carry = (cs[:, -1] - last[:, -1] + torch.where(any_done, torch.zeros_like(carry.double()), carry.double())).to(carry.dtype)
    return torch.cat([stats, ci]), new_carry


class EpisodeSource:
    """`sample_seq` + `init_context` for n episodes at once (agent_ar.py:519-533).

    dataset: kinpoly_amd.dataset.StateARDataset (or None with `context_fn(n) -> dict`); ctx_builder: PolicyARContext or None
    (contexts already carry init_qpos / init_qvel).  Keeps the reference's `freq_dict` (take -> list of [percent, fr_start],
    last 5000 kept, agent_ar.py:668-676) and passes it with sampling_temp / sampling_freq to every draw.  As in the reference, the
    dict changes only BETWEEN sample() calls (the forked workers draw from the copy they were started with and the agent merges their
    lists afterwards, agent_ar.py:664-673): the take probabilities are evaluated once per version of the dict."""

    def __init__(self, dataset=None, ctx_builder=None, context_fn=None, sampling_temp=0.5, sampling_freq=0.9, fix_height=False, cache_init_context=False):
        assert (dataset is None) != (context_fn is None), "give a dataset or a context_fn"
        self.dataset, self.ctx_builder, self.context_fn = dataset, ctx_builder, context_fn
        self.sampling_temp, self.sampling_freq, self.fix_height = sampling_temp, sampling_freq, fix_height
        self.freq_dict = {k: [] for k in dataset.takes} if dataset is not None else {}
        self._probs = None
        self.n_drawn = 0          # clips drawn so far (every one of them went through init_context when there is a ctx_builder)
        # Opt-in memo of init_context (off by default: then every drawn clip runs through the context network, as in the reference).  What a training
        # episode takes from init_context -- init_qpos / init_qvel -- is a pure function of the window (take, fr_start) and of the context
        # network's parameters, which no update of the RL phase touches (neither the PPO surrogate nor the supervised step loss reaches
        # context_rnn / context_mlp / context_fc; they get no gradient and Adam skips them): a window that was seen under the same parameter
        # version is looked up instead of recomputed.  Needs a ctx_builder with need_rollout=False; any change of the parameters' versions
        # (load_state_dict, an optimiser that does train them) empties the memo.
        self.cache_init_context = bool(cache_init_context) and dataset is not None and ctx_builder is not None and not ctx_builder.need_rollout
        self._memo = None
        self.n_memo_hits = 0

    def draw(self, n: int, device) -> dict:
        if self.dataset is not None:
            if self._probs is None:
                self._probs = self.dataset.take_probs(self.freq_dict, self.sampling_temp)
            data = self.dataset.sample_batch(n, freq_dict=self.freq_dict, sampling_temp=self.sampling_temp, sampling_freq=self.sampling_freq, probs=self._probs)
        else:
            data = self.context_fn(n)
        self.n_drawn += n
        keys = (data["take_ind"].numpy().astype(np.int64), data["fr_start"].numpy().astype(np.int64)) if self.cache_init_context else None
        data = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in data.items()}
        if self.cache_init_context:
            data = self._init_context_memo(data, keys, device)
        elif self.ctx_builder is not None:
            data = self.ctx_builder.init_context(data, fix_height=self.fix_height)
        elif "init_qpos" not in data:         # no context network: the episode starts on the clip's first frame
            data["init_qpos"] = data["qpos"][:, 0].contiguous()
            data["init_qvel"] = data["qvel"][:, 0].contiguous() if "qvel" in data else torch.zeros((n, 75), device=device)
        return data

    def _init_context_memo(self, data, keys, device):
        net = self.ctx_builder.net
        params = list(net.context_rnn.parameters()) + list(net.context_mlp.parameters()) + list(net.context_fc.parameters())
        ver = tuple((p._version, p.data_ptr()) for p in params)
        lens = self.dataset._seq_lens()
        if self._memo is None or self._memo["ver"] != ver or self._memo["n_takes"] != len(lens):
            per_take = np.maximum(lens - self.dataset.fr_num, 1)                  # window starts a take offers (sample_batch)
            off = np.concatenate([[0], np.cumsum(per_take)]).astype(np.int64)
            self._memo = {"ver": ver, "n_takes": len(lens), "off": off, "have": np.zeros(int(off[-1]), bool),
                          "q": torch.zeros((int(off[-1]), 76), device=device), "v": torch.zeros((int(off[-1]), 75), device=device)}
        m = self._memo
        wid = m["off"][keys[0]] + keys[1]
        miss_w, first = np.unique(wid[~m["have"][wid]], return_index=True)
        if len(miss_w):
            rows = np.nonzero(~m["have"][wid])[0][first]                          # one drawn row per missing window
            idx = torch.as_tensor(rows, device=device)
            n = len(wid)
            part = {k: (v[idx] if torch.is_tensor(v) and v.dim() > 0 and v.shape[0] == n else v) for k, v in data.items()}
            out = self.ctx_builder.init_context(part, fix_height=self.fix_height)
            w_t = torch.as_tensor(miss_w, device=device)
            m["q"].index_copy_(0, w_t, out["init_qpos"]); m["v"].index_copy_(0, w_t, out["init_qvel"])
            m["have"][miss_w] = True
        self.n_memo_hits += len(wid) - len(miss_w)
        w_all = torch.as_tensor(wid, device=device)
        data["init_qpos"], data["init_qvel"] = m["q"][w_all].contiguous(), m["v"][w_all].contiguous()
        return data

    def record(self, take_ind, fr_start, percent, group=None):
        """freq_dict[curr_key].append([info['percent'], fr_start]) for every finished episode (agent_ar.py:601-603).

        ONE job-wide freq_dict, as in the reference: AgentAR.sample merges every worker's list into the agent's dict
        (agent_ar.py:664-673) before the next draws.  With several ranks the finished episodes of all ranks are exchanged once per
        sample() call (`all_gather_object`: a few KB of Python lists, host side) and appended in rank order, so every rank holds the
        same dict and draws its next clips from the same take probabilities."""
        if self.dataset is None:
            return
        # arrays, not Python rows: with random-init networks every env-step ends an episode (98 304 of them per 4096 x 24 sample() call)
        mine = (np.asarray(take_ind, np.int64), np.asarray(fr_start, np.int64), np.asarray(percent, np.float64))
        if _collective_on(group):
            every = [None] * dist.get_world_size(group)
            dist.all_gather_object(every, mine, group=group)
        else:
            every = [mine]
        for ti, fs, pc in every:
            for t in np.unique(ti):
                m = ti == t
                pcs, fss = pc[m][-5000:].tolist(), fs[m][-5000:].tolist()          # only the last 5000 entries of a take are kept (agent_ar.py:674-676)
                self.freq_dict[self.dataset.takes[int(t)]].extend(map(list, zip(pcs, fss)))
        self.freq_dict = {k: (v if len(v) < 5000 else v[-5000:]) for k, v in self.freq_dict.items()}
        self._probs = None

    def save_freq_dict(self, path):
        """joblib.dump(self.freq_dict, 'freq_dict.pt') of the reference (agent_ar.py:297): a pickle of {take: [[percent, fr_start], ...]}."""
        import joblib
        joblib.dump(self.freq_dict, path)


def ring_refill_plan(head: torch.Tensor, ahead: torch.Tensor, n_slots: int, total: int):
    """Which ring rows a top-up writes.  head / ahead: int [N] (slot every env is playing, clips queued behind it); every env is brought back to
    n_slots - 1 queued clips.  total = sum of the deficits (the caller's one host read).  Returns (env [total], row [total]) int64: env e's
    j-th new clip goes to slot (head + ahead + 1 + j) mod n_slots, row = slot * N + e -- never the slot the env is playing.  Pure torch (CPU-testable)."""
    N = head.shape[0]
    dev = head.device
    deficit = (n_slots - 1) - ahead.long()
    ar = torch.arange(N, device=dev)
    env_idx = torch.repeat_interleave(ar, deficit, output_size=total)
    first = torch.cumsum(deficit, 0) - deficit                       # index of every env's first new clip in the flat list
    j = torch.arange(total, device=dev) - first[env_idx]
    slot = (head.long()[env_idx] + ahead.long()[env_idx] + 1 + j) % n_slots
    return env_idx, slot * N + env_idx


class VectorSampler:
    """Fixed-horizon lock-step sampler with device-side auto-reset (no host sync inside the loop).

    Without a `source` a finished env restarts on its own clip (the synthetic single-clip configs of SURVEY 8d).  With one, EVERY episode runs
    on a freshly drawn clip, whatever the failure rate (agent_ar.py:518-535: sample_seq -> init_context -> load_context -> reset per episode):
    every env owns a ring of `pool_depth + 1` context rows -- the clip it is on and up to pool_depth clips drawn ahead.  `done` moves the env one
    slot on (kp_pool_advance, on the device); every pool_depth steps the sampler reads ONE number from the device -- how many queued clips
    were used up -- draws exactly that many (one batched sample_seq + init_context), and writes them into the used-up slots in place.  An env
    ends at most one episode per step, so between two top-ups it cannot use more than the pool_depth clips it had: the pool cannot run dry,
    and clips that were not used stay queued across sample() calls instead of being drawn again."""

    NOISE_CHUNK = 16          # exploration noise is drawn for this many steps at a time (memory does not grow with the horizon)

    def __init__(self, env: BatchedHumanoidAREnv, policy: KinPolicy, record_qpos: bool = False, mean_action: bool = False,
                 source: EpisodeSource | None = None, pool_depth: int = 4, record_full: bool = False, lagged: bool = True):
        """lagged (default): the ring's one host read per pool_depth steps is taken ONE PERIOD LATE -- the counts are copied to pinned memory when a
        period ends and read when the next one does, by which time the copy has long completed, so sample() never waits for the device and the
        top-up's host work (the draw, init_context's launches, the row writes) overlaps the queued env-steps.  A refill then restores what was
        missing a period ago, so an env can be 2 x pool_depth episodes ahead of its last refill: the ring holds 2 x pool_depth + 1 rows per env
        instead of pool_depth + 1 (HBM: 0.11 MB per row).  lagged=False: the round-4 form, one blocking read per period."""
        self.env, self.policy, self.record_qpos, self.mean_action = env, policy, record_qpos or record_full, mean_action
        self.source, self.pool_depth, self.record_full = source, max(1, int(pool_depth)), record_full
        self.lagged, self._pending = bool(lagged), None
        self.obs = self.hx = self.fresh = None
        self.head = self.ahead = None      # int32 [N]: ring slot every env is playing / fresh clips queued behind it
        self._since = 0                    # steps since the last top-up
        self.pool_exhausted = 0            # 0 by construction; kept as the counter callers report
        self.top_ups = 0                   # host reads of the ring state so far (one per pool_depth steps)
        self.group = None                  # process group of the job-wide freq_dict exchange (None = the default group)
        self.ep_return = None              # [N] return collected so far by the episode every env is in (LoggerRL's episode_reward)
        self.log = None                    # LoggerRL of the last sample() call (this rank's envs)

    # ------------------------------------------------------------------ episode pool
    @property
    def n_slots(self):
        return (2 * self.pool_depth if self.lagged else self.pool_depth) + 1

    def _pool_init(self):
        """The ring: slot 0 = the clips the first episodes run on, slots 1 .. pool_depth = their successors, all freshly drawn."""
        env, N, dev, D = self.env, self.env.n, self.env.device, self.n_slots
        first = self.source.draw(N, dev)
        T = first["qpos"].shape[1]
        if self.source.dataset is not None:
            T = max(T, int(self.source.dataset.fr_num))
        # do the clips carry action objects?  A data set knows (host side); a context_fn source is asked once, here
        if "obj_pose" not in first:
            objects = False
        elif self.source.dataset is not None:
            objects = bool(self.source.dataset.has_objects)
        else:
            objects = bool((first["action_one_hot"].reshape(-1, 4).sum(1) > 0).any())
        env.alloc_context(D * N, T, objects=objects, with_ar="ar_qpos" in first, obj_width=first["obj_pose"].shape[2] if "obj_pose" in first else 14)
        ar = torch.arange(N, device=dev)
        env.write_context_rows(ar, first)
        for s in range(1, D):
            env.write_context_rows(ar + s * N, self.source.draw(N, dev))
        self.head = torch.zeros(N, dtype=torch.int32, device=dev)
        self.ahead = torch.full((N,), D - 1, dtype=torch.int32, device=dev)
        self._since, self._pending = 0, None

    def _top_up(self):
        """Replace the queued clips the envs used up: one host read (how many), one batched draw, in-place row writes.  Lagged form: the read is of
        the counts copied out ONE period ago (no wait), the refill restores the ring to the level it had then; see __init__."""
        env, D = self.env, self.n_slots
        self._since = 0
        self.top_ups += 1
        if not self.lagged:
            deficit = (D - 1) - self.ahead
            total, low = torch.stack([deficit.sum(), self.ahead.min()]).tolist()
            self._refill(self.head, self.ahead, deficit, int(total), int(low))
            return
        if self._pending is not None:
            head_s, ahead_s, deficit_s, host, ev = self._pending
            ev.synchronize()                       # recorded a whole period of env-steps ago: complete unless the host has run that far ahead
            total, low = host.tolist()
            self._refill(head_s, ahead_s, deficit_s, int(total), int(low))
        # this period's snapshot (after the refill above): the slots behind every env's current one that are free NOW stay free until they are rewritten
        head_s, ahead_s = self.head.clone(), self.ahead.clone()
        deficit_s = (D - 1) - ahead_s
        host = torch.empty(2, dtype=torch.int64, pin_memory=True)
        host.copy_(torch.stack([deficit_s.sum(), self.ahead.min().to(torch.int64)]), non_blocking=True)
        ev = torch.cuda.Event(); ev.record()
        self._pending = (head_s, ahead_s, deficit_s, host, ev)

    def _refill(self, head, ahead, deficit, total, low):
        if low < 0:            # cannot happen: an env ends at most one episode per step and the ring is sized for the periods between its refills
            self.pool_exhausted += 1
            raise kpsim.KinPolyNativeError("episode pool underflow: an env finished more episodes than steps since the last top-up")
        if total == 0:
            return
        env_idx, rows = ring_refill_plan(head, ahead, self.n_slots, total)
        self.env.write_context_rows(rows, self.source.draw(total, self.env.device))
        self.ahead.add_(deficit)

    def start(self):
        if self.source is not None:
            self._pool_init()
        self.obs = self.env.reset().clone()
        self.hx = self.policy.init_hidden(self.env.n, self.env.device)
        self.fresh = torch.ones(self.env.n, dtype=torch.bool, device=self.env.device)
        self.ep_return = torch.zeros(self.env.n, dtype=torch.float64, device=self.env.device)

    @torch.no_grad()
    def sample(self, T: int, noise: torch.Tensor | None = None) -> RolloutBatch:
        """noise (optional) [T, N, 80 + 75]: the standard-normal exploration draws of both policies, made by the caller instead of drawn here from env.gen
        (a parity run feeds the same draws to the CPU episode loop; columns a policy that acts with its mean would not read are ignored)."""
        env, pol, N, dev = self.env, self.policy, self.env.n, self.env.device
        if self.obs is None:
            self.start()
        f = lambda *s: torch.empty((N, T, *s), device=dev)  # noqa: E731
        S, A, R = f(105), f(80), f()
        E = torch.empty((N, T), dtype=torch.bool, device=dev); F = torch.empty((N, T), dtype=torch.bool, device=dev)
        Q = f(76) if self.record_qpos else None
        G = f(76) if self.record_qpos else None
        full = self.record_full
        NS, VM, RQ, CA, CS = (f(105), f(3), f(76), f(75), f(784)) if full else (None,) * 5
        D = torch.empty((N, T), dtype=torch.bool, device=dev); PC = f(); MT = f(2); CI = f(6)
        hx0 = self.hx.clone()
        fr_num = float(env.ctx["qpos"].shape[1])
        # exploration noise of both policies, NOISE_CHUNK steps per launch, only the columns a sampling policy reads: [chunk, N, 80 kinematic | 75 UHC]
        cc_mean = env.mode == "test" or (env.mode == "train" and env.joint_controller)
        n_kin, n_cc = (0 if self.mean_action else 80), (0 if cc_mean else 75)
        given = noise
        if given is not None and tuple(given.shape) != (T, N, 155):
            raise ValueError(f"sample(noise=...): expected shape ({T}, {N}, 155), got {tuple(given.shape)}")
        noise = None
        qview = env.sim.view("qpos") if (self.record_qpos or full) else None          # the simulator's own rows: the record kernels read them in place
        for t in range(T):
            if given is None and (n_kin + n_cc) and t % self.NOISE_CHUNK == 0:
                noise = torch.randn((min(self.NOISE_CHUNK, T - t), N, n_kin + n_cc), device=dev, generator=env.gen)
            nz = None if noise is None else noise[t % self.NOISE_CHUNK]
            if given is not None and (n_kin + n_cc):
                nz = torch.cat([given[t, :, :n_kin], given[t, :, 80:80 + n_cc]], 1)
            # Memory.push, first half (one launch): state, episode start, the pose before the step, the GT pose of the clip's next frame, (take, fr_start)
            kpsim.record_pre(t, T, obs=self.obs, fresh=self.fresh, qpos=qview if self.record_qpos else None, ctx_qpos=env.ctx["qpos"] if self.record_qpos else None,
                             row=env.row, cur_t=env.cur_t, row_len=env.row_len, row_meta=env.row_meta,
                             states=S, episode_start=E, curr_qpos=Q, gt_target_qpos=G, meta=MT)
            action, self.hx = pol.select_action(self.obs, self.hx, self.mean_action, env.gen, nz[:, :n_kin] if n_kin else None)
            action = action.contiguous()
            obs, _, done, info = env.step(action, need_obs=full, cc_noise=nz[:, n_kin:] if n_cc else None)
            # second half (one launch): action, reward, flags, custom_info and -- full record -- next state, pose after the step, UHC action / state, v_meta
            kpsim.record_post(t, T, fr_num, action=action, reward=info["custom_reward"], fail=info["fail"], done=done, percent=info["percent"], c_info=info["custom_info"],
                              obs=obs if full else None, qpos=qview if full else None, cc_action=info["cc_action"] if full else None, cc_state=info["cc_state"] if full else None,
                              meta=MT, actions=A, rewards=R, fails=F, dones=D, percents=PC, c_infos=CI, next_states=NS, res_qpos=RQ, cc_actions=CA, cc_states=CS, v_metas=VM)
            # device-side episode turnover: a finished env moves to the next clip of its ring (env.row in place), then the masked reset
            if self.source is not None:
                kpsim.pool_advance(done, self.head, self.ahead, env.row, self.n_slots)
            # env._obs: step() writes its own observation elsewhere, so this stays valid through the next step; the same launch zeroes the GRU state
            # of the finished envs in place (self.hx is this step's fresh output of select_action; hx0 above is a copy)
            self.obs = env.reset(done, policy_state=self.hx)
            self.fresh = done                   # read by the next step's record_pre, before that step reuses the buffer (step() keeps two alternating sets)
            if self.source is not None:
                self._since += 1
                if self._since >= self.pool_depth:
                    self._top_up()
        self.fresh = self.fresh.clone()         # across calls the env's buffers may be reused by others (evaluation, bare env-steps)
        M = (~D).float()
        # one host transfer per call: finished episodes -> freq_dict, launch status.  Every rank takes part in the job-wide exchanges BEFORE
        # any rank raises, so a stalled queue on one rank ends the job on all of them instead of leaving the others in a collective
        status = _agree_status(int(env.sim.status_tensor()[2]), dev, self.group)
        stats, self.ep_return = episode_log(R, D, CI, self.ep_return)
        st = stats.tolist()
        self.log = LoggerRL(num_steps=int(st[0]), num_episodes=int(st[1]), total_reward=st[2], min_episode_reward=st[3], max_episode_reward=st[4],
                            total_c_reward=st[5], min_c_reward=st[6], max_c_reward=st[7], total_c_info=np.asarray(st[8:14]))
        dm = D.cpu().numpy()
        eps = {"take_ind": MT[..., 0].cpu().numpy()[dm].astype(np.int64), "fr_start": MT[..., 1].cpu().numpy()[dm].astype(np.int64),
               "percent": PC.cpu().numpy()[dm].astype(np.float64)}
        if self.source is not None:
            self.source.record(eps["take_ind"], eps["fr_start"], eps["percent"], self.group)
        if status:
            raise kpsim.KinPolyNativeError("kp_step_queue_kernel reported a stalled job queue during the rollout on some rank (states are incomplete)")
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


def gae_scan(rewards, masks, values, gamma, tau, last_values=None):
    """estimate_advantages' recurrence (common.py:11-19) as a reverse scan over the time axis of env-major [N, T] tensors, in the tensors' own
    dtype: the fp64 update (`update_dtype=torch.float64`) and CPU tensors go through this; fp32 device tensors through k_gae.  The reference
    scans one flat batch whose workers' rows end on `masks == 0`; an env's row here may be cut by the horizon, then `last_values` [N] = V of
    the state after the last row enters as the value behind it (kp_gae_bootstrap's rule)."""
    N, T = rewards.shape
    adv = torch.empty_like(values)
    prev_v = torch.zeros(N, dtype=values.dtype, device=values.device) if last_values is None else last_values.to(values.dtype)
    prev_a = torch.zeros(N, dtype=values.dtype, device=values.device)
    for t in range(T - 1, -1, -1):
        delta = rewards[:, t] + gamma * prev_v * masks[:, t] - values[:, t]
        prev_a = delta + gamma * tau * prev_a * masks[:, t]
        adv[:, t] = prev_a
        prev_v = values[:, t]
    return adv, values + adv


def estimate_advantages(rewards, masks, values, gamma, tau, group=None, last_values=None):
    """GAE on the device (k_gae, env-major reverse scan; `gae_scan` for fp64 / CPU tensors) + the global normalisation above."""
    if values.is_cuda and values.dtype == torch.float32:
        adv, ret = kpsim.gae(rewards.contiguous(), masks.contiguous(), values.contiguous(), gamma, tau,
                             None if last_values is None else last_values.contiguous())
    else:
        adv, ret = gae_scan(rewards.to(values.dtype), masks.to(values.dtype), values, gamma, tau, last_values)
    adv, ret, _ = normalize_advantages_global(adv, ret, group)
    return adv, ret


def _allreduce_grads(params, group=None):
    """Mean of the ranks' gradients over a FIXED parameter list: every rank sends one flat buffer of the same length whatever its own backward
    reached (a parameter without a gradient on this rank -- scheduled sampling threw the context network's output away here but not there --
    contributes zeros), plus one flag per parameter, so that a parameter NO rank has a gradient for keeps `grad = None` and its optimiser state
    untouched, exactly as in a single process (ranks with different None sets used to call all_reduce with different lengths)."""
    if not _collective_on(group):
        return
    params = [p for p in params if p.requires_grad]
    if not params:
        return
    ref = params[0]
    flat = torch.cat([(p.grad.reshape(-1) if p.grad is not None else torch.zeros(p.numel(), dtype=p.dtype, device=p.device)).to(ref.dtype) for p in params]
                     + [torch.tensor([0.0 if p.grad is None else 1.0 for p in params], dtype=ref.dtype, device=ref.device)])
    dist.all_reduce(flat, group=group)
    n_flags = len(params)
    have = (flat[-n_flags:] > 0).tolist()
    flat = flat[:-n_flags] / dist.get_world_size(group)
    off = 0
    for p, h in zip(params, have):
        n = p.numel()
        if h:
            g = flat[off:off + n].view_as(p).to(p.dtype)
            if p.grad is None:
                p.grad = g.clone()
            else:
                p.grad.copy_(g)
        off += n


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
                 cc_policy=None, policy_weightdecay=0.0, value_weightdecay=0.0, train_uhc=False, reference_bugs=True):
        """policy / value: the modules the optimisers own.  Their dtype is the update's dtype: fp32 modules on the device run the fused HIP re-unroll
        and k_gae; fp64 modules (the reference trains in fp64, scripts/train_ar_policy.py:76-77) run the same update through the GRUCell loop and
        `gae_scan` -- `AgentAR(update_dtype=torch.float64)` keeps such fp64 master copies and writes them back into the fp32 roll-out modules.

        reference_bugs (default True: results identical to the reference's): the gradient clip acts on the FIRST optimiser step of a run only.
        The reference hands `policy_grad_clip=[(self.policy_net.parameters(), 40)]` (agent_ar.py:92-93) -- a generator -- to
        clip_policy_grad (agent_ppo.py:53-56); the first clip_grad_norm_ consumes it, every later call sees no parameters and returns 0
        (tests/golden/update_params.npz holds the norms the reference's calls reported: 45.99, 0, 0, ...).  False clips every step."""
        self.policy, self.value, self.group, self.cc_policy = policy, value, group, cc_policy
        self.reference_bugs, self._clip_calls, self.clip_norms = bool(reference_bugs), 0, []
        self.debug_reset_momentum = __import__("os").environ.get("KP_DEBUG_RESET_PPO_MOMENTUM") == "1"      # tools/update_ablation.sh only; read once, not per update
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
        self._clip_calls += 1
        if self.reference_bugs and self._clip_calls > 1:       # the reference's generator of parameters was consumed by the run's first call
            return
        norm = torch.nn.utils.clip_grad_norm_(params, self.policy_grad_clip)
        if self._clip_calls == 1:
            self.clip_norms.append(norm.detach())              # kept on the device (read by tests / the update fixture)

    def _cast(self, t):
        """batch tensors in the update's dtype (the roll-out records fp32; an fp64 update reads them as fp64 like the reference's `.to(self.dtype)`, agent_ar.py:685-695)"""
        dt = next(self.policy.parameters()).dtype
        return None if t is None else (t if t.dtype == dt or not t.dtype.is_floating_point else t.to(dt))

    def _value_epochs(self, flat_states, ret, n_steps):
        """the value net's regression steps of all epochs: `update_value` (agent_ppo.py:53-56) x n_steps.  They share nothing with the policy
        passes (fixed targets `ret`, own optimiser)."""
        vloss = None
        self.vloss_history = []
        for _ in range(n_steps):
            vloss = (self.value(flat_states) - ret).pow(2).mean()
            self.vloss_history.append(vloss.detach())
            self.opt_v.zero_grad(); vloss.backward(); _allreduce_grads(list(self.value.parameters()), self.group); self.opt_v.step()
        return vloss

    def update(self, batch: RolloutBatch, bootstrap: bool = True):
        N, T, _ = batch.states.shape
        states, hx0 = self._cast(batch.states), self._cast(batch.hx0)
        flat_states = states.reshape(N * T, -1)
        flat_actions = self._cast(batch.actions).reshape(N * T, -1)
        ind = None
        if batch.exps is not None:                # `ind = exps.nonzero()` (agent_ar.py:763): the rows the surrogate is taken over
            ind = batch.exps.reshape(-1).nonzero(as_tuple=False).squeeze(1)
            if ind.numel() == N * T:
                ind = None
        with torch.no_grad():
            values = self.value(flat_states).view(N, T)
            last_v = self.value(self._cast(batch.last_states)).view(N) if (bootstrap and batch.last_states is not None) else None
        adv, ret = estimate_advantages(self._cast(batch.rewards), self._cast(batch.masks), values, self.gamma, self.tau, self.group, last_v)
        adv, ret = adv.reshape(-1, 1), ret.reshape(-1, 1)
        self.last_adv, self.last_ret = adv, ret
        # the value net's steps of all epochs first: they share nothing with the policy passes (fixed targets, own optimiser).  Running them on a
        # side stream underneath the policy epochs was tried in round 4 (at most 15 of 850 ms to gain) and DEADLOCKED in the second or third
        # iteration on ROCm 7.2 / torch 2.10 (two autograd backward passes in flight on two streams; tools/micro/dbg_train_hang.py,
        # profiles/r04/side_stream_hang.log) -- one stream.
        vloss = self._value_epochs(flat_states, ret, self.num_optim_epoch * self.value_opt_niter)
        # fixed_log_probs (agent_ar.py:758-759) is the policy's forward at the parameters the update starts from: epoch 0's own forward, reused
        # (the reference evaluates it twice; one of its 11 policy forwards is redundant), so epoch 0's ratio is exactly 1 as it is there
        # a batch updated on in slices carries the log-probabilities under the policy that sampled it (behaviour_log_probs): the ratio is then taken against
        # those, not against parameters the earlier slices have already moved
        fixed_log_probs, surr = self._cast(batch.behaviour_log_probs), None
        self.surr_history = []                     # every epoch's surrogate, on the device (one host read at the end)
        if self.debug_reset_momentum:
            # diagnosis only (tools/update_ablation.sh, KP_DEBUG_RESET_PPO_MOMENTUM=1 read once by the constructor): Adam's first moment of the policy optimiser is
            # zeroed at the start of every iteration, to tell a stale momentum (built at parameters the supervised step updates have since moved) from a wrong gradient
            for st in self.opt_p.state.values():
                if "exp_avg" in st:
                    st["exp_avg"].zero_()
        for _ in range(self.num_optim_epoch):
            means = self.policy.unroll(states, batch.episode_start, hx0)
            log_probs = self.policy.log_prob(means.reshape(N * T, -1), flat_actions)
            if fixed_log_probs is None:
                fixed_log_probs = log_probs.detach()
            surr = ppo_surrogate(log_probs, fixed_log_probs, adv, self.clip_epsilon, ind)
            self.surr_history.append(surr.detach())
            self.opt_p.zero_grad(); surr.backward()
            self._clip()
            self.opt_p.step()
        stats = {"value_loss": float(vloss.detach()), "surr_loss": float(surr.detach())} if surr is not None else {}
        if surr is not None:
            # how far the epochs moved the policy on its own batch: the log-ratio of the LAST epoch's forward against the behaviour policy (one host read for
            # the three numbers).  A surrogate that ends above 0 is the signature of a log-ratio spread far beyond the clip range: min(r A, clip(r) A) caps the
            # gain of a sample the step moved the right way at 0.2 |A| and leaves the loss of one it moved the wrong way unbounded
            lr_ = (log_probs.detach() - fixed_log_probs).reshape(-1)
            d = torch.stack([lr_.std(), (lr_.abs() > 0.2).float().mean()]).tolist()
            stats.update(ppo_log_ratio_std=d[0], ppo_frac_outside_clip=d[1])
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
        states, hx0, actions = self._cast(batch.states), self._cast(batch.hx0), self._cast(batch.actions)
        flat_states = states.reshape(N * T, -1)
        curr, tgt = self._cast(batch.curr_qpos).reshape(N * T, 76), self._cast(batch.gt_target_qpos).reshape(N * T, 76)
        ind = None
        if batch.exps is not None:                # `ind = exps.nonzero()` (agent_ar.py:813): the rows the surrogate is taken over
            ind = batch.exps.reshape(-1).nonzero(as_tuple=False).squeeze(1)
            if ind.numel() == N * T:
                ind = None
        with torch.no_grad():
            values = self.value(flat_states).view(N, T)
            last_v = self.value(self._cast(batch.last_states)).view(N) if (bootstrap and batch.last_states is not None) else None
            tgt_wbpos = fk.wbpos(tgt)
        adv, ret = estimate_advantages(self._cast(batch.rewards), self._cast(batch.masks), values, self.gamma, self.tau, self.group, last_v)
        adv, ret = adv.reshape(-1, 1), ret.reshape(-1, 1)
        stats, fixed_log_probs = {}, None
        for _ in range(self.num_optim_epoch):
            vloss = (self.value(flat_states) - ret).pow(2).mean()
            self.opt_v.zero_grad(); vloss.backward(); _allreduce_grads(list(self.value.parameters()), self.group); self.opt_v.step()
            means = self.policy.unroll(states, batch.episode_start, hx0).reshape(N * T, -1)
            log_probs = self.policy.log_prob(means, actions.reshape(N * T, -1))
            if fixed_log_probs is None:            # the forward at the starting parameters is epoch 0's own (see update())
                fixed_log_probs = log_probs.detach()
            surr = ppo_surrogate(log_probs, fixed_log_probs, adv, self.clip_epsilon, ind)
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
        pol = self.cc_policy
        dt = next(pol.parameters()).dtype
        cs, ca = batch.cc_state.reshape(-1, batch.cc_state.shape[-1]).to(dt), batch.cc_action.reshape(-1, batch.cc_action.shape[-1]).to(dt)
        adv = adv.to(dt)

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
