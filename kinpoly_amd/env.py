"""Batched `HumanoidAREnv` on the HIP simulator + a single-env facade with the reference surface.

`BatchedHumanoidAREnv` runs N environments in lock-step on one GPU; its `step()` is the batched
restatement of HumanoidAREnv.step (kin_poly/envs/humanoid_ar_v1.py:243-323):

    step_begin        prev_bquat / prev_hpos snapshots                     :246-249
    step_kin          next_qpos = step_ar(a)                               :252  (216-241)
    set_target        target = smpl_humanoid.qpos_fk(next_qpos)            :256
    obs_cc (+ZFilter) cc_obs = cc_running_state(get_cc_obs(), update=False):265-266
    cc_policy         cc_action = cc_policy.select_action(cc_obs, mean)    :267-268
    step_ctrl         do_simulation(cc_action, frame_skip=15)              :286
    cur_t += 1; term_reward: fail / end / done                             :288-316
    obs_ar            obs = get_ar_obs_v1()                                :322

All tensors stay on the device; nothing here touches the CPU oracle.
`HumanoidAREnv` wraps a 1-env batch and speaks numpy float64, so the callers in
kin_poly/core/agent_ar.py:463-611 and scripts/eval_ar_policy.py:178-222 can use it unchanged.
"""
from __future__ import annotations

import numpy as np
import torch

from . import sim as kpsim
from .nets import PolicyMCP

CTX_KEYS = ("qpos", "head_pose", "head_vels", "obj_head_relative_poses", "action_one_hot", "init_qpos", "init_qvel")


class RunningState:
    """Device view of the reference's ZFilter/RunningStat (uhc/khrylib/utils/zfilter.py) used with update=False."""

    def __init__(self, mean, std, clip=5.0, device="cuda"):
        self.mean = torch.as_tensor(mean, dtype=torch.float32, device=device).contiguous()
        self.std = torch.as_tensor(std, dtype=torch.float32, device=device).contiguous()
        self.clip = float(clip)

    @classmethod
    def identity(cls, dim=784, clip=5.0, device="cuda"):
        return cls(np.zeros(dim), np.ones(dim), clip, device)


class BatchedHumanoidAREnv:
    def __init__(self, n_envs, device=0, kpm_path=None, cc_policy: PolicyMCP | None = None,
                 cc_running_state: RunningState | None = None, mode="train", wild=False, joint_controller=False,
                 env_episode_len=100000, body_diff_thresh=10.0, body_diff_gt_thresh=12.0, model_options=None, seed=0, ar_mode=False):
        self.n = int(n_envs)
        self.ar_mode = bool(ar_mode)
        if kpm_path is None:  # agent_ar.py:165-169: mocap training uses ..._all_step.xml, --wild uses ..._all.xml
            kpm_path = kpsim.DEFAULT_KPM if wild else kpsim.STEP_KPM
        self.model_options = dict(model_options or {})
        self.model = kpsim.KpModel(kpm_path, **self.model_options)
        self.sim = kpsim.KpSim(self.model, self.n, device)
        self.device = self.sim.device
        self.mode, self.wild, self.joint_controller = mode, wild, joint_controller
        self.env_episode_len = env_episode_len
        self.frame_skip = 15
        self.dt = self.model.get_option("timestep") * self.frame_skip
        self.cc_policy = (cc_policy if cc_policy is not None else PolicyMCP()).to(self.device).float()
        self.cc_running_state = cc_running_state if cc_running_state is not None else RunningState.identity(device=self.device)
        self.reward_cfg = kpsim.KpRewardCfg.default(use_gt_term=(mode == "train" and not wild))
        self.reward_cfg.body_diff_thresh = body_diff_thresh
        self.reward_cfg.body_diff_gt_thresh = body_diff_gt_thresh
        self.gen = torch.Generator(device=self.device)
        self.gen.manual_seed(seed)
        self.cur_t = torch.zeros(self.n, dtype=torch.int32, device=self.device)
        self.ctx = None
        self.row = None           # int32 [N]: context row every env reads (load_context / set_rows)
        self.row_len = None       # int32 [R]: ar_context['len'] of every row
        self.row_meta = None      # [R, 2]: take index, first frame of every row (v_meta of the reference's memory rows)
        self.obj7 = None          # [N,7]  = get_obj_qpos(action_one_hot), kept by kp_sim_reset_rows / kp_sim_post_step
        self._row_obj_qpos = None  # [R,35] = convert_obj_qpos(action_one_hot, obj_pose[0]) of every context row
        self._ctx_struct = None
        self.end_reward = 0.0
        self.action_dim, self.obs_dim, self.cc_action_dim = 80, kpsim.AR_OBS_DIM, kpsim.CC_ACTION_DIM
        # persistent I/O buffers (no per-step allocation)
        f = lambda d: torch.empty((self.n, d), dtype=torch.float32, device=self.device)  # noqa: E731
        self._next_qpos, self._cc_obs, self._obs, self._obs_next = f(76), f(784), f(105), f(105)
        # what step() hands back lives in two alternating sets of buffers: a returned tensor stays valid until the next-but-one step()
        u8 = lambda: torch.empty(self.n, dtype=torch.uint8, device=self.device)  # noqa: E731
        self._outs = [dict(reward=torch.empty(self.n, device=self.device), info=f(6), diffs=f(2), fail=u8(), done=u8(), end=u8(),
                           percent=torch.empty(self.n, device=self.device)) for _ in range(2)]
        self._flip = 0
        self.done_count = torch.zeros(1, dtype=torch.int32, device=self.device)      # episodes ended since the caller last zeroed it (kp_sim_post_step)
        self._unit_reward = torch.ones(self.n, device=self.device)

    # ------------------------------------------------------------------ reference surface
    def seed(self, seed):
        self.gen.manual_seed(int(seed))
        return [seed]

    def set_mode(self, mode):
        self.mode = mode
        self.reward_cfg.use_gt_term = int(mode == "train" and not self.wild)

    def alloc_context(self, R: int, T: int, objects: bool = False, with_ar: bool = False, obj_width: int = 14):
        """Empty context tables [R, T, .] (R a multiple of n_envs) that `write_context_rows` fills in place: what a sampler that keeps the next
        episodes' clips resident allocates once (VectorSampler's ring of pool_depth + 1 rows per env).  objects: the clips carry action objects
        (the per-row object block of reset_model is kept next to them); with_ar: room for the kinematic roll-out (ar_qpos / ar_qvel: ar_mode,
        ar_fail_safe, evaluation)."""
        if R % self.n != 0:
            raise ValueError(f"context rows ({R}) must be a multiple of n_envs ({self.n})")
        dev = self.device
        z = lambda *s: torch.zeros(s, dtype=torch.float32, device=dev)  # noqa: E731
        self.ctx = {"qpos": z(R, T, 76), "head_pose": z(R, T, 7), "head_vels": z(R, T, 6), "obj_head_relative_poses": z(R, T, 7), "action_one_hot": z(R, 4),
                    "init_qpos": z(R, 76), "init_qvel": z(R, 75), "gt_bquat": z(R, T, 96), "gt_wbpos": z(R, T, 72)}
        if objects:
            self.ctx["obj_pose"] = z(R, T, obj_width)
        if with_ar or self.ar_mode:
            self.ctx["ar_qpos"], self.ctx["ar_qvel"] = z(R, T, 76), z(R, T, 75)
        self.row_len = torch.full((R,), T - 1, dtype=torch.int32, device=dev)
        self.row_meta = z(R, 2)
        self.row = torch.arange(self.n, device=dev, dtype=torch.int32)
        self._row_obj_qpos = self.obj7 = None
        if objects:
            self._alloc_objects(R)
        self._bind_context()

    def _alloc_objects(self, R):
        # data.qpos[76:111] of every row as reset_model builds it (convert_obj_qpos); obj7 [N,7] = get_obj_qpos(action_one_hot) per env, kept by
        # kp_sim_reset_rows / kp_sim_post_step (the simulated poses themselves live in the simulator: the obj_qpos property)
        self._row_obj_qpos = convert_obj_qpos(torch.zeros((R, 4), device=self.device), torch.zeros((R, 7), device=self.device))[0]
        self.obj7 = torch.tensor([0.0, 0, 0, 1, 0, 0, 0], device=self.device).repeat(self.n, 1).contiguous()

    def _bind_context(self):
        c = self.ctx
        T = c["qpos"].shape[1]
        self._ctx_struct = self.sim.make_ctx(T, c["head_pose"], c["head_vels"], c["obj_head_relative_poses"], c["action_one_hot"],
                                             c["gt_bquat"], c["gt_wbpos"], self.cur_t, obj_qpos=self.obj7, row=self.row)
        # rows a reset starts from (reset_model, humanoid_ar_v1.py:339-341): the kinematic roll-out's first frame in ar_mode, else init_qpos / init_qvel
        if self.ar_mode:
            self._init_q, self._init_v = c["ar_qpos"][:, 0].contiguous(), c["ar_qvel"][:, 0].contiguous()
        else:
            self._init_q, self._init_v = c["init_qpos"], c["init_qvel"]

    def write_context_rows(self, rows: torch.Tensor, data: dict):
        """Overwrite the context rows `rows` (int64 [m]) in place with m freshly drawn clips: data[k] is [m, T', .] with T' <= the tables' T (shorter
        clips are padded with their last frame, as StateARDataset.batch pads), action_one_hot [m, T', 4] or [m, 4], init_qpos / init_qvel [m, .],
        optional len [m], take_ind / fr_start [m], obj_pose, ar_qpos / ar_qvel.  The GT clip's FK (load_context's gt_targets, humanoid_ar_v1.py:87)
        is computed for those rows only.  Envs that are playing one of these rows must be reset afterwards."""
        c, dev = self.ctx, self.device
        rows = rows.to(dev, torch.int64)
        m, T = rows.numel(), c["qpos"].shape[1]
        if m == 0:
            return

        def fit(v):                                  # [m, T', .] -> [m, T, .]
            v = v.to(dev, torch.float32)
            if v.dim() == 3 and v.shape[1] < T:
                v = torch.cat([v, v[:, -1:].expand(-1, T - v.shape[1], -1)], 1)
            return v
        if data["qpos"].shape[1] > T:
            raise ValueError(f"clips of {data['qpos'].shape[1]} frames do not fit context tables of {T}")
        one_hot = data["action_one_hot"].to(dev, torch.float32)
        if one_hot.dim() == 3:
            one_hot = one_hot[:, 0]
        for k in ("qpos", "head_pose", "head_vels", "obj_head_relative_poses", "init_qpos", "init_qvel"):
            c[k].index_copy_(0, rows, fit(data[k]))
        c["action_one_hot"].index_copy_(0, rows, one_hot)
        for k in ("ar_qpos", "ar_qvel"):
            if k in c:
                if k not in data:
                    raise ValueError(f"this env needs ctx['{k}'] (ar_mode / tables allocated with_ar)")
                c[k].index_copy_(0, rows, fit(data[k]))
        gt = self.sim.fk(fit(data["qpos"]).reshape(-1, 76).contiguous())
        c["gt_bquat"].index_copy_(0, rows, gt["bquat"].view(m, T, 96))
        c["gt_wbpos"].index_copy_(0, rows, gt["wbpos"].view(m, T, 72))
        lens = data.get("len")
        # no `len`: the clip runs over its OWN frames (data['qpos'].shape[1]), not over the padding up to the tables' T
        new_len = torch.full((m,), data["qpos"].shape[1] - 1, dtype=torch.int32, device=dev) if lens is None else torch.as_tensor(lens, device=dev).to(torch.int32) - 1
        self.row_len.index_copy_(0, rows, new_len)
        meta = torch.stack([torch.as_tensor(data[k]).to(dev, torch.float32) if k in data else torch.zeros(m, device=dev) for k in ("take_ind", "fr_start")], 1)
        self.row_meta.index_copy_(0, rows, meta)
        if self._row_obj_qpos is not None:
            if "obj_pose" in data:
                op = fit(data["obj_pose"])
                if "obj_pose" in c and op.shape[2] == c["obj_pose"].shape[2]:
                    c["obj_pose"].index_copy_(0, rows, op)
                self._row_obj_qpos.index_copy_(0, rows, convert_obj_qpos(one_hot, op[:, 0])[0])
            else:
                self._row_obj_qpos.index_copy_(0, rows, convert_obj_qpos(torch.zeros_like(one_hot), torch.zeros((m, 7), device=dev))[0])
        if self.ar_mode:
            self._init_q.index_copy_(0, rows, c["ar_qpos"][rows, 0]); self._init_v.index_copy_(0, rows, c["ar_qvel"][rows, 0])

    def load_context(self, ctx: dict, env_mask: torch.Tensor | None = None, row: torch.Tensor | None = None, keep_state: bool = False):
        """ctx tensors are [R, T, .] (action_one_hot [R, T, 4] or [R, 4]; init_qpos/init_qvel [R, .]) with R = n_envs context rows,
        or -- for a sampler that keeps the next episodes' clips resident -- R = k * n_envs rows with `row` (int32 [N]) naming the
        row every env starts on (default: env e on row e); `set_rows` later switches finished envs to other rows without copying.
        With env_mask (bool [N], R = N only) just those envs' rows are replaced (T must match).
        Ragged episodes: pad every clip to the longest T (repeat the last frame) and pass ctx["len"] = frames per row [R];
        an env's episode then ends at its own len - 1 (`ar_context['len']`, humanoid_ar_v1.py:312).
        keep_state: the envs are in the middle of episodes whose clips are among the new rows (a sampler re-laying its row table):
        per-env episode state (cur_t, the simulated object pose) is left alone."""
        R, T = ctx["qpos"].shape[:2]
        if R % self.n != 0:
            raise ValueError(f"context rows ({R}) must be a multiple of n_envs ({self.n})")
        if env_mask is not None and (R != self.n or row is not None):
            raise ValueError("masked load_context works on one row per env")
        lens = ctx.get("len")
        if lens is not None:
            lt = torch.as_tensor(lens)
            if int(lt.max()) > T or int(lt.min()) < 2:
                raise ValueError("ctx['len'] must lie in [2, T]")
        one_hot = ctx["action_one_hot"] if ctx["action_one_hot"].dim() == 2 else ctx["action_one_hot"][:, 0]
        has_obj = "obj_pose" in ctx and bool((one_hot.sum(1) > 0).any())
        fresh = self.ctx is None or env_mask is None or self.ctx["qpos"].shape[:2] != (R, T)
        if fresh:
            if env_mask is not None and self.ctx is not None:
                raise ValueError("masked load_context needs the same clip length T")
            obj7_before = self.obj7 if keep_state else None
            self.alloc_context(R, T, objects=has_obj, with_ar="ar_qpos" in ctx, obj_width=ctx["obj_pose"].shape[2] if "obj_pose" in ctx else 14)
            if row is not None:
                self.row.copy_(row.to(self.device, torch.int32))
            self.write_context_rows(torch.arange(R, device=self.device), ctx)
            if self.obj7 is not None:
                if obj7_before is not None:
                    self.obj7.copy_(obj7_before)       # the simulated objects stay where they are
                else:                                  # before the first reset: the clip's own object pose
                    r = self.row.long()
                    start = torch.tensor(ACTION_INDEX_MAP, device=self.device)[self.ctx["action_one_hot"][r].argmax(1)]
                    got = torch.gather(self._row_obj_qpos[r], 1, start[:, None] + torch.arange(7, device=self.device)[None])
                    self.obj7.copy_(torch.where(self.ctx["action_one_hot"][r].sum(1, keepdim=True) > 0, got, self.obj7))
        else:
            idx = env_mask.to(self.device, torch.bool).nonzero(as_tuple=True)[0]
            if has_obj and self._row_obj_qpos is None:             # the first clips with objects arrive through a masked load
                self._alloc_objects(R)
                self._bind_context()
            sub = {}
            for k, v in ctx.items():
                if k in ("len", "take_ind", "fr_start") and not torch.is_tensor(v):
                    v = torch.as_tensor(v)                      # per-row lists / arrays: sliced like the clips
                sub[k] = v.to(self.device)[idx] if torch.is_tensor(v) and v.dim() > 0 and v.shape[0] == R else v
            self.write_context_rows(idx, sub)

    def load_uhc_checkpoint(self, path, load_policy=True):
        """The trained UHC of the reference's constructor (humanoid_ar_v1.py:60-81): `running_state` (ZFilter of the 784-d observation) always,
        `policy_dict` into cc_policy unless the run trains the controller jointly (`if not kin_cfg.joint_controller`, :79-81).  `path`: a pickle in
        the reference's layout ({'policy_dict', 'value_dict', 'running_state'}; scripts/train_uhc.py --save writes one)."""
        from . import checkpoint as ck
        cp = ck.load_checkpoint(path)
        arr = ck.running_state_arrays(cp.get("running_state"))
        if arr is not None:
            self.cc_running_state = RunningState(arr[0], arr[1], arr[2], self.device)
        if load_policy:
            self.cc_policy.load_state_dict({k: (v if torch.is_tensor(v) else torch.as_tensor(v)) for k, v in cp["policy_dict"].items()})
            self.cc_policy.to(self.device).float()
        return cp

    @property
    def has_objects(self):
        """the loaded clips carry action objects (they are free bodies of the envs)"""
        return self._row_obj_qpos is not None

    @property
    def obj_qpos(self):
        """data.qpos[76:111] of every env [N, 35] (get_obj_qpos(), humanoid_ar_v1.py:466-477), or None when the clips carry no objects"""
        return self.sim.get("obj_qpos") if self.has_objects else None

    @property
    def ctx_len(self):
        """ar_context['len'] of every env's current clip (int32 [N])."""
        return self.row_len[self.row.long()]

    def set_rows(self, new_row: torch.Tensor, env_mask: torch.Tensor | None = None):
        """Put the masked envs on other context rows (device op, no copy): the new episode's clip of agent_ar.py:519-535.
        Follow with reset(env_mask)."""
        nr = new_row.to(self.device, torch.int32)
        if env_mask is None:
            self.row.copy_(nr)
        else:
            self.row.copy_(torch.where(env_mask.to(self.device, torch.bool), nr, self.row))

    def ctx_rows(self, key):
        """ctx[key] gathered to the envs' current rows: [N, ...]."""
        return self.ctx[key][self.row.long()]

    @staticmethod
    def _mask8(env_mask, device):
        """bool / uint8 mask -> uint8 view (a bool tensor's bytes are 0 / 1: no conversion launch)"""
        if env_mask is None:
            return None
        m = env_mask.to(device)
        return (m.view(torch.uint8) if m.dtype == torch.bool else m.to(torch.uint8)).contiguous()

    def reset(self, env_mask: torch.Tensor | None = None, policy_state: torch.Tensor | None = None):
        """sim.reset(); cur_t = 0; reset_model(): state <- ar_context init_qpos/init_qvel (+ the clip's object block, convert_obj_qpos), target =
        FK(init) (:334-387): one gather launch for the humanoid, one for the objects, sim.forward() and the target FK (kp_sim_reset_rows).
        policy_state [N, H] (optional): the caller's recurrent policy state, zeroed in place for the same envs by that launch (PolicyAR.reset at
        every episode start, policy_ar.py:124-131)."""
        m8 = self._mask8(env_mask, self.device)
        self.sim.reset_rows(self._init_q, self._init_v, self.row, m8, self.cur_t, set_target=True, aux_rows=policy_state,
                            row_obj_qpos=self._row_obj_qpos, row_one_hot=None if self._row_obj_qpos is None else self.ctx["action_one_hot"], obj7=self.obj7)
        return self.sim.obs_ar(self._ctx_struct, self._obs)

    def _ar_frame(self, key):
        """ar_context[key][cur_t + 1] per env."""
        t = (self.cur_t.long() + 1).clamp_(max=self.ctx[key].shape[1] - 1)
        return self.ctx[key][self.row.long(), t].contiguous()

    def ar_fail_safe(self, env_mask: torch.Tensor | None = None):
        """HumanoidAREnv.ar_fail_safe (:327-331): put the humanoid back on the kinematic roll-out (objects keep their state)."""
        self.sim.set_state(self._ar_frame("ar_qpos"), self._ar_frame("ar_qvel"), self._mask8(env_mask, self.device))

    def step(self, a: torch.Tensor, need_obs: bool = True, cc_noise: torch.Tensor | None = None):
        """One batched HumanoidAREnv.step.  need_obs=False skips get_ar_obs_v1 (a sampler that resets finished envs right after the step and
        does not record next_states gets its observation from reset()); cc_noise [N, 75]: standard-normal draws for the UHC's exploration
        made ahead by the caller.  Lifetime of what is returned (no per-step allocation): done / fail / end / percent / the rewards live in two
        alternating buffer sets and stay valid until the next-but-one step(); `obs` and info['cc_state'] live in single
        buffers and are overwritten by the NEXT step() (reset() likewise returns its one observation buffer, which the next reset() rewrites):
        a caller that keeps any of them longer clones them (VectorSampler copies each step's values into its [N, T, .] rollout rows; the
        numpy facade HumanoidAREnv copies on conversion)."""
        sim = self.sim
        if self.ar_mode:                 # the UHC tracks the kinematic roll-out (:263-264)
            sim.step_begin()
            sim.set_target(self._ar_frame("ar_qpos"))
        else:
            sim.step_head(a)             # prev_bquat / prev_hpos, next_qpos = step_ar(a), target = qpos_fk(next_qpos): one launch
        rs = self.cc_running_state
        cc_obs = sim.obs_cc(self._cc_obs, rs.mean, rs.std, rs.clip)
        mean_action = self.mode == "test" or (self.mode == "train" and self.joint_controller)
        with torch.no_grad():
            cc_action = self.cc_policy.select_action(cc_obs, mean_action, self.gen, cc_noise).contiguous()
        sim.step_ctrl(cc_action, self.frame_skip)
        # cur_t += 1; fail (body diffs), reward, end / done / percent, and the action object's simulated pose for the next observation, in one launch (:288-316)
        self._flip ^= 1
        o = self._outs[self._flip]
        sim.post_step(self._ctx_struct, self.reward_cfg, self.cur_t, self.row_len, int(min(self.env_episode_len, 2 ** 31 - 1)),
                      o["reward"], o["info"], o["fail"], o["diffs"], o["done"], o["end"], o["percent"], self.done_count, self.obj7)
        obs = sim.obs_ar(self._ctx_struct, self._obs_next) if need_obs else None
        info = {"fail": o["fail"].view(torch.bool), "end": o["end"].view(torch.bool), "percent": o["percent"], "cc_action": cc_action, "cc_state": cc_obs,
                "custom_reward": o["reward"], "custom_info": o["info"], "body_diff": o["diffs"]}
        return obs, self._unit_reward, o["done"].view(torch.bool), info        # the env's own reward is the constant 1.0 (humanoid_ar_v1.py:311); one shared read-only tensor

    # getters (device tensors; reference names)
    def get_humanoid_qpos(self):
        return self.sim.get("qpos")

    def get_humanoid_qvel(self):
        return self.sim.get("qvel")

    def get_head(self):
        return self.sim.get("head")

    def get_body_quat(self):
        return self.sim.get("bquat")

    def get_wbody_pos(self):
        return self.sim.get("xpos")

    def get_wbody_quat(self):
        return self.sim.get("xquat")

    def get_body_com(self):
        return self.sim.get("xipos")

    @property
    def target(self):
        g = self.sim.get
        return {"qpos": g("target_qpos"), "wbpos": g("target_wbpos"), "wbquat": g("target_wbquat"), "bquat": g("target_bquat"),
                "body_com": g("target_com")}


ACTION_INDEX_MAP, ACTION_LEN = (0, 7, 21, 28), (7, 14, 7, 7)   # sit / push / avoid / step  (humanoid_ar_v1.py:37-39)


def convert_obj_qpos(action_one_hot: torch.Tensor, obj_pose0: torch.Tensor):
    """HumanoidAREnv.convert_obj_qpos (humanoid_ar_v1.py:479-496) batched: every object parked at [(i+1)*100, 100, 0]
    with a zero quaternion, the active action's slice overwritten by obj_pose.  Returns ([N,35], [N,7] active slice).
    Branch-free (masked selects over the four action slots): no host read, so the sampler's ring top-up keeps its one read per pool_depth steps."""
    n, dev = action_one_hot.shape[0], action_one_hot.device
    blk = torch.zeros((n, 35), device=dev)
    for i in range(5):
        blk[:, 7 * i] = (i + 1) * 100.0; blk[:, 7 * i + 1] = 100.0
    obj7 = torch.tensor([0.0, 0, 0, 1, 0, 0, 0], device=dev).repeat(n, 1)
    for a in range(4):
        m = (action_one_hot[:, a] > 0).unsqueeze(1)
        st, ln = ACTION_INDEX_MAP[a], min(ACTION_LEN[a], obj_pose0.shape[1])
        blk[:, st:st + ln] = torch.where(m, obj_pose0[:, :ln], blk[:, st:st + ln])
        obj7 = torch.where(m, blk[:, st:st + 7], obj7)
    return blk.contiguous(), obj7.contiguous()


def standing_context(n, T, std_qpos, std_qvel, sim: kpsim.KpSim, headings: torch.Tensor | None = None):
    """Synthetic per-env context of SURVEY.md section 8(d) config 3: the standing clip repeated T frames, objects
    absent (obj_pose = [0,0,0,1,0,0,0], action_one_hot = 0; kin_poly/data_process/process_smpl.py:223-225),
    head_vels = 0, obj_head_relative_poses by the rule of process_smpl.py:110-135.  Optional per-env heading
    rotation (radians) about z."""
    dev = sim.device
    q = torch.tensor(std_qpos, dtype=torch.float32, device=dev).repeat(n, 1)
    if headings is not None:
        h = headings.to(dev, torch.float32)
        hq = torch.stack([torch.cos(h / 2), torch.zeros_like(h), torch.zeros_like(h), torch.sin(h / 2)], 1)
        w0, x0, y0, z0 = q[:, 3], q[:, 4], q[:, 5], q[:, 6]
        w1, x1, y1, z1 = hq.unbind(1)
        q[:, 3:7] = torch.stack([w1 * w0 - x1 * x0 - y1 * y0 - z1 * z0, w1 * x0 + x1 * w0 + y1 * z0 - z1 * y0,
                                 w1 * y0 - x1 * z0 + y1 * w0 + z1 * x0, w1 * z0 + x1 * y0 - y1 * x0 + z1 * w0], 1)
    fk = sim.fk(q.contiguous())
    head = torch.cat([fk["wbpos"].view(n, 24, 3)[:, 13], fk["wbquat"].view(n, 24, 4)[:, 13]], 1)  # [n,7]
    # object [0,0,0 | 1,0,0,0] relative to the head in the head's heading frame
    hw, hz = head[:, 3], head[:, 6]
    hn = torch.sqrt(hw * hw + hz * hz)
    c, s = hw / hn, hz / hn                       # heading quaternion (c,0,0,s)
    cos_t, sin_t = c * c - s * s, 2 * c * s       # rotation by the heading angle
    d = -head[:, :3]
    loc = torch.stack([cos_t * d[:, 0] + sin_t * d[:, 1], -sin_t * d[:, 0] + cos_t * d[:, 1], d[:, 2]], 1)  # R^T d
    objq = torch.stack([c, torch.zeros_like(c), torch.zeros_like(c), -s], 1)                               # inverse(heading) (x) identity
    obj_rel = torch.cat([loc, objq], 1)
    rep = lambda x: x.unsqueeze(1).repeat(1, T, 1).contiguous()  # noqa: E731
    return {"qpos": rep(q), "head_pose": rep(head), "head_vels": torch.zeros((n, T, 6), device=dev), "obj_head_relative_poses": rep(obj_rel),
            "action_one_hot": torch.zeros((n, 4), device=dev), "init_qpos": q.contiguous(),
            "init_qvel": torch.tensor(std_qvel, dtype=torch.float32, device=dev).repeat(n, 1).contiguous()}


SMPL_BODY_NAMES = ("Pelvis", "L_Hip", "L_Knee", "L_Ankle", "L_Toe", "R_Hip", "R_Knee", "R_Ankle", "R_Toe", "Torso", "Spine", "Chest", "Neck", "Head",
                   "L_Thorax", "L_Shoulder", "L_Elbow", "L_Wrist", "L_Hand", "R_Thorax", "R_Shoulder", "R_Elbow", "R_Wrist", "R_Hand")   # XML body order


class Box:
    """The two attributes-and-a-method of gym.spaces.Box the reference's callers touch (humanoid_ar_v1.py:106-112;
    agent_ar.py:146 reads action_space.shape[0]); gym itself is not a dependency of this engine."""

    def __init__(self, low, high, dtype=np.float32):
        self.low, self.high, self.dtype = np.asarray(low, dtype), np.asarray(high, dtype), dtype
        self.shape = self.low.shape

    def sample(self, rng=None):
        rng = rng or np.random
        lo, hi = np.where(np.isfinite(self.low), self.low, -1.0), np.where(np.isfinite(self.high), self.high, 1.0)
        return rng.uniform(lo, hi).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))


class _ModelView:
    """Read-only stand-in for the mujoco-py `env.model` attributes callers of the env read:
    nq / nv / nu, opt.timestep, body_names, actuator_names (one motor per hinge, XML order), actuator_ctrlrange."""

    def __init__(self, kp_model, has_objects=True):
        import types
        self.nu = 69
        self.nq, self.nv = (76 + 35, 75 + 30) if has_objects else (76, 75)
        self.opt = types.SimpleNamespace(timestep=kp_model.get_option("timestep"))
        self.body_names = ("world",) + SMPL_BODY_NAMES
        self._body_name2id = {n: i for i, n in enumerate(self.body_names)}
        self.actuator_names = tuple(f"{b}_{a}" for b in SMPL_BODY_NAMES[1:] for a in "zyx")
        self.actuator_ctrlrange = np.tile(np.array([[-1.0, 1.0]]), (self.nu, 1))


class HumanoidAREnv:
    """Single-environment facade with the reference constructor and numpy float64 I/O
    (kin_poly/envs/humanoid_ar_v1.py:28).  `cfg` / `cc_cfg` are duck-typed: only `policy_specs` thresholds,
    `joint_controller`, `env_episode_len` are read if present; a trained UHC checkpoint can be passed as
    `cc_state=(policy_dict, running_state mean, std)`."""

    def __init__(self, cfg=None, cc_cfg=None, init_context=None, cc_iter=-1, mode="train", wild=False, ar_mode=False, cc_state=None, device=0):
        ps = getattr(cfg, "policy_specs", {}) or {}
        pol = PolicyMCP()
        rs = None
        if cc_state is not None:
            pol.load_state_dict(cc_state[0])
            rs = RunningState(cc_state[1], cc_state[2], 5.0, torch.device("cuda", device))
        self.b = BatchedHumanoidAREnv(1, device, cc_policy=pol, cc_running_state=rs, mode=mode, wild=wild,
                                      joint_controller=bool(getattr(cfg, "joint_controller", False)),
                                      env_episode_len=int(getattr(cc_cfg, "env_episode_len", 100000)),
                                      body_diff_thresh=ps.get("body_diff_thresh", 10), body_diff_gt_thresh=ps.get("body_diff_gt_thresh", 12), ar_mode=ar_mode)
        self.kin_cfg, self.cc_cfg, self.ar_mode, self.wild, self.mode = cfg, cc_cfg, ar_mode, wild, mode
        self.cc_policy, self.cc_running_state = self.b.cc_policy, self.b.cc_running_state
        self.dt, self.end_reward = self.b.dt, 0.0
        self.prev_bquat = self.prev_hpos = self.prev_qpos = self.prev_qvel = None
        # constants / spaces of the reference constructor (humanoid_ar_v1.py:36-39, 50-59, 90-112; humanoid_im.py:30-49)
        self.model = _ModelView(self.b.model)
        self.frame_skip, self.sim_iter, self.start_ind = 15, 15, 0
        self.qpos_lim, self.qvel_lim, self.body_lim = 76, 75, 25
        self.num_obj, self.action_index_map, self.action_len = 5, list(ACTION_INDEX_MAP), list(ACTION_LEN)
        self.action_names = ["sit", "push", "avoid", "step"]
        self.ndof, self.vf_dim, self.cc_action_dim = 69, 6, 75
        self.action_dim, self.obs_dim = 75, kpsim.AR_OBS_DIM      # set_spaces() (:106-112): the Box is 75-d although step() takes the 80-d kinematic action
        self.action_space = Box(-np.ones(self.action_dim), np.ones(self.action_dim))
        self.observation_space = Box(-np.inf * np.ones(self.obs_dim), np.inf * np.ones(self.obs_dim))
        self.body_diff_thresh, self.body_diff_gt_thresh = ps.get("body_diff_thresh", 10), ps.get("body_diff_gt_thresh", 12)
        self.jpos_diffw = np.ones((24, 1))
        self.gt_targets = self.ar_context = None
        self.np_random = np.random.RandomState(0)                 # MujocoEnv.__init__ seeds at construction (mujoco_env.py:57)
        if init_context is not None:
            self.load_context(init_context)

    def seed(self, seed=None):
        self.np_random = np.random.RandomState(seed)
        return self.b.seed(0 if seed is None else seed)

    def render(self, mode="human"):
        """No viewer in this engine (the reference's mujoco-py viewer is out of scope): a no-op so that
        callers which render unconditionally keep working."""
        return None

    @property
    def bquat(self):
        return self.get_body_quat()

    def set_mode(self, mode):
        self.b.set_mode(mode)

    @property
    def cur_t(self):
        return int(self.b.cur_t[0])

    def load_context(self, data_dict):
        """data_dict: tensors [1, T, .] as produced by PolicyAR.init_context (policy_ar.py:124-182)."""
        self.ar_context = {k: (v[0].detach().cpu().numpy() if torch.is_tensor(v) else (np.asarray(v)[0] if np.ndim(v) > 0 else v)) for k, v in data_dict.items()}
        self.ar_context["len"] = self.ar_context["qpos"].shape[0] - 1
        ctx = {k: torch.as_tensor(np.asarray(self.ar_context[k]), dtype=torch.float32)[None] for k in
               ("qpos", "head_pose", "head_vels", "obj_head_relative_poses", "action_one_hot", "obj_pose", "ar_qpos", "ar_qvel") if k in self.ar_context}
        key_q, key_v = ("ar_qpos", "ar_qvel") if (self.ar_mode and "init_qpos" not in self.ar_context) else ("init_qpos", "init_qvel")
        iq, iv = self.ar_context[key_q], self.ar_context[key_v]
        ctx["init_qpos"] = torch.as_tensor(iq[0] if iq.ndim == 2 else iq, dtype=torch.float32)[None]
        ctx["init_qvel"] = torch.as_tensor(iv[0] if iv.ndim == 2 else iv, dtype=torch.float32)[None]
        self.b.load_context(ctx)
        # gt_targets = smpl_humanoid.qpos_fk_batch(ar_context['qpos'])   (:87)
        T = self.ar_context["qpos"].shape[0]
        fk = self.b.sim.fk(self.b.ctx["qpos"][0].contiguous())
        self.gt_targets = {"qpos": fk["qpos"].double().cpu().numpy(), "wbpos": fk["wbpos"].view(T, 24, 3).double().cpu().numpy(),
                           "wbquat": fk["wbquat"].view(T, 24, 4).double().cpu().numpy(), "bquat": fk["bquat"].view(T, 96).double().cpu().numpy(),
                           "body_com": fk["body_com"].view(T, 24, 3).double().cpu().numpy()}

    def reset(self):
        return self.b.reset()[0].double().cpu().numpy()

    def step(self, a):
        self.prev_qpos, self.prev_qvel = self.get_humanoid_qpos(), self.get_humanoid_qvel()
        obs, r, done, info = self.b.step(torch.as_tensor(np.asarray(a), dtype=torch.float32, device=self.b.device)[None].contiguous())
        self.prev_bquat = self.b.sim.get("prev_bquat")[0].double().cpu().numpy()
        self.prev_hpos = self.b.sim.get("prev_hpos")[0].double().cpu().numpy()
        out_info = {"fail": bool(info["fail"][0]), "end": bool(info["end"][0]), "percent": float(info["percent"][0]),
                    "cc_action": info["cc_action"][0].double().cpu().numpy(), "cc_state": info["cc_state"][0].double().cpu().numpy()}
        self._last_reward = (float(info["custom_reward"][0]), info["custom_info"][0].double().cpu().numpy())
        return obs[0].double().cpu().numpy(), 1.0, bool(done[0]), out_info

    def _g(self, name):
        return self.b.sim.get(name)[0].double().cpu().numpy()

    def get_humanoid_qpos(self): return self._g("qpos")
    def get_humanoid_qvel(self): return self._g("qvel")
    def get_head(self): return self._g("head")
    def get_body_quat(self): return self._g("bquat")
    def get_wbody_pos(self): return self._g("xpos")
    def get_wbody_quat(self): return self._g("xquat")
    def get_body_com(self): return self._g("xipos")

    def get_obj_qpos(self, action_one_hot=None):
        """humanoid_ar_v1.py:466-477: the whole object block, or the pose of the action's (first) object."""
        full = self._g("obj_qpos") if self.b.has_objects else convert_obj_qpos(torch.zeros((1, 4)), torch.zeros((1, 7)))[0][0].double().numpy()
        if action_one_hot is None:
            return full
        if np.sum(action_one_hot) == 0:
            return np.array([0, 0, 0, 1, 0, 0, 0.0])
        a = int(np.nonzero(action_one_hot)[0][0])
        return full[ACTION_INDEX_MAP[a]:ACTION_INDEX_MAP[a] + ACTION_LEN[a]][:7]

    def get_obj_qvel(self):
        return self._g("obj_qvel") if self.b.has_objects else np.zeros(30)

    def ar_fail_safe(self):
        self.b.ar_fail_safe()

    @property
    def target(self):
        return {k: v[0].double().cpu().numpy() for k, v in self.b.target.items()}
