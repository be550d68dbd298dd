"""ctypes binding of libkinpoly_sim.so (include/kinpoly_sim.h) for torch device tensors.

This is the thin host layer: tensors in, tensors out.  A `KpSim` enqueues on the torch stream that was current when it
was created; `KpSim.use_current_stream()` rebinds it (kp_sim_set_stream) when the caller moves to another stream -- the library
never guesses.  There is NO CPU fallback: if the extension or a HIP device is missing, construction
fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch

from . import build as _build

NQ, NV, NU, NBODY = 76, 75, 69, 24
CC_OBS_DIM, AR_OBS_DIM, KIN_ACTION_DIM, CC_ACTION_DIM = 784, 105, 80, 75
DEFAULT_KPM = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets", "smpl_humanoid.kpm")
STEP_KPM = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets", "smpl_humanoid_step.kpm")

FIELDS = dict(qpos=0, qvel=1, xpos=2, xquat=3, xipos=4, bquat=5, head=6, target_qpos=7, target_wbpos=8,
              target_wbquat=9, target_bquat=10, target_com=11, qpos_d=12, qvel_d=13, prev_bquat=14, prev_hpos=15, obj_qpos=16, obj_qvel=17,
              M=18, bias=19)

_lib = None

# every symbol include/kinpoly_sim.h declares (tests check the .so exports all of them)
ABI_SYMBOLS = [
    "kp_model_load", "kp_model_free", "kp_model_set_option", "kp_model_get_option", "kp_sim_create", "kp_sim_destroy",
    "kp_sim_n_envs", "kp_sim_set_state", "kp_sim_set_target", "kp_sim_step_ctrl", "kp_sim_step_kin", "kp_sim_obs_cc",
    "kp_field_dim", "kp_sim_get", "kp_sim_diag", "kp_sim_last_step_seconds", "kp_last_error", "kp_version",
    "kp_sim_step_begin", "kp_sim_obs_ar", "kp_sim_term_reward", "kp_gae", "kp_sim_set_full_state", "kp_sim_fk",
    "kp_sim_timing_reset", "kp_sim_timing_mean_seconds", "kp_sim_phase_cycles", "kp_sim_set_objects", "kp_sim_set_obj_state",
    "kp_sim_launch_cost", "kp_job_schedule", "kp_sim_fk_backward", "kp_sim_set_stream", "kp_sim_status_device", "kp_sim_mass_matrix",
    "kp_sim_contacts", "kp_gae_bootstrap", "kp_gru_gates_forward", "kp_gru_gates_backward", "kp_sim_phase_cycles_env",
    "kp_sim_post_step", "kp_sim_reset_rows", "kp_mcp_compose", "kp_sim_step_head", "kp_model_compile", "kp_model_load_xml",
    "kp_mcp_tail", "kp_gru_cell_step", "kp_kin_advance", "kp_pool_advance", "kp_rollout_record_pre", "kp_rollout_record_post", "kp_sim_field_device",
    "kp_sim_lean_state",
]


class KpCtx(C.Structure):
    """mirror of kp_ctx (include/kinpoly_sim.h)"""
    _fields_ = [("T", C.c_int), ("head_pose", C.c_void_p), ("head_vels", C.c_void_p), ("obj_head_relative_poses", C.c_void_p),
                ("action_one_hot", C.c_void_p), ("gt_bquat", C.c_void_p), ("gt_wbpos", C.c_void_p), ("obj_qpos", C.c_void_p),
                ("cur_t", C.c_void_p), ("row", C.c_void_p)]


class KpRecordPre(C.Structure):
    """mirror of kp_record_pre (include/kinpoly_sim.h)"""
    _fields_ = [(k, C.c_int) for k in ("n", "T", "t", "ctx_T")] + [(k, C.c_void_p) for k in (
        "obs", "fresh", "qpos", "ctx_qpos", "row", "cur_t", "row_len", "row_meta", "states", "episode_start", "curr_qpos", "gt_target_qpos", "meta")]


class KpRecordPost(C.Structure):
    """mirror of kp_record_post"""
    _fields_ = [("n", C.c_int), ("T", C.c_int), ("t", C.c_int), ("fr_num", C.c_float)] + [(k, C.c_void_p) for k in (
        "action", "reward", "fail", "done", "percent", "c_info", "obs", "qpos", "cc_action", "cc_state", "meta",
        "actions", "rewards", "fails", "dones", "percents", "c_infos", "next_states", "res_qpos", "cc_actions", "cc_states", "v_metas")]


class KpRewardCfg(C.Structure):
    """mirror of kp_reward_cfg; defaults = config/statear/kin_poly.yml:72-86, humanoid_ar_v1.py:53-54"""
    _fields_ = [(k, C.c_float) for k in ("w_hp", "w_hq", "w_p", "w_jp", "w_act_p", "w_act_v", "k_hp", "k_hq", "k_p", "k_jp", "k_act_p",
                                          "k_act_v", "dt", "body_diff_thresh", "body_diff_gt_thresh")] + [("use_gt_term", C.c_int)]

    @classmethod
    def default(cls, use_gt_term=True):
        return cls(0.15, 0.15, 0.2, 0.2, 0.2, 0.1, 45.0, 45.0, 50.0, 50.0, 5.0, 0.005, 1.0 / 30.0, 10.0, 12.0, int(use_gt_term))


class KinPolyNativeError(RuntimeError):
    pass


def load_library(path: str | None = None):
    """dlopen the in-tree extension (never builds implicitly on a GPU box: the .so must be there)."""
    global _lib
    if _lib is not None:
        return _lib
    path = path or os.environ.get("KP_SIM_LIBRARY") or _build.LIB      # KP_SIM_LIBRARY: another build of the same ABI (A/B measurements)
    if not os.path.exists(path):
        raise KinPolyNativeError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                                 "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    L = C.CDLL(path)
    P, F, U8 = C.c_void_p, C.c_void_p, C.c_void_p
    L.kp_model_load.restype = P; L.kp_model_load.argtypes = [C.c_char_p]
    L.kp_model_compile.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p]; L.kp_model_compile.restype = C.c_int
    L.kp_model_load_xml.restype = P; L.kp_model_load_xml.argtypes = [C.c_char_p, C.c_char_p]
    L.kp_model_free.argtypes = [P]
    L.kp_model_set_option.argtypes = [P, C.c_char_p, C.c_double]; L.kp_model_set_option.restype = C.c_int
    L.kp_model_get_option.argtypes = [P, C.c_char_p]; L.kp_model_get_option.restype = C.c_double
    L.kp_sim_create.restype = P; L.kp_sim_create.argtypes = [P, C.c_int, C.c_int, C.c_void_p]
    L.kp_sim_destroy.argtypes = [P]
    L.kp_sim_n_envs.argtypes = [P]; L.kp_sim_n_envs.restype = C.c_int
    L.kp_sim_set_state.argtypes = [P, F, F, U8]; L.kp_sim_set_state.restype = C.c_int
    L.kp_sim_set_target.argtypes = [P, F, U8]; L.kp_sim_set_target.restype = C.c_int
    L.kp_sim_step_ctrl.argtypes = [P, F, C.c_int, U8]; L.kp_sim_step_ctrl.restype = C.c_int
    L.kp_sim_step_kin.argtypes = [P, F, F]; L.kp_sim_step_kin.restype = C.c_int
    L.kp_sim_step_head.argtypes = [P, F]; L.kp_sim_step_head.restype = C.c_int
    L.kp_sim_obs_cc.argtypes = [P, F, F, F, C.c_float]; L.kp_sim_obs_cc.restype = C.c_int
    L.kp_field_dim.argtypes = [C.c_int]; L.kp_field_dim.restype = C.c_int
    L.kp_sim_get.argtypes = [P, C.c_int, F]; L.kp_sim_get.restype = C.c_int
    L.kp_sim_diag.argtypes = [P, C.c_void_p]; L.kp_sim_diag.restype = C.c_int
    L.kp_sim_last_step_seconds.argtypes = [P]; L.kp_sim_last_step_seconds.restype = C.c_double
    L.kp_sim_step_begin.argtypes = [P]; L.kp_sim_step_begin.restype = C.c_int
    L.kp_sim_obs_ar.argtypes = [P, C.POINTER(KpCtx), F]; L.kp_sim_obs_ar.restype = C.c_int
    L.kp_sim_term_reward.argtypes = [P, C.POINTER(KpCtx), C.POINTER(KpRewardCfg), F, F, U8, F]; L.kp_sim_term_reward.restype = C.c_int
    L.kp_sim_post_step.argtypes = [P, C.POINTER(KpCtx), C.POINTER(KpRewardCfg), C.c_void_p, C.c_void_p, C.c_int, F, F, U8, F, U8, U8, F, C.c_void_p, F]; L.kp_sim_post_step.restype = C.c_int
    L.kp_sim_reset_rows.argtypes = [P, F, F, C.c_void_p, U8, C.c_void_p, C.c_int, F, C.c_int, F, F, F]; L.kp_sim_reset_rows.restype = C.c_int
    L.kp_pool_advance.argtypes = [C.c_int, C.c_int, U8, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]; L.kp_pool_advance.restype = C.c_int
    L.kp_sim_field_device.argtypes = [P, C.c_int]; L.kp_sim_field_device.restype = C.c_void_p
    L.kp_rollout_record_pre.argtypes = [C.POINTER(KpRecordPre), C.c_void_p]; L.kp_rollout_record_pre.restype = C.c_int
    L.kp_rollout_record_post.argtypes = [C.POINTER(KpRecordPost), C.c_void_p]; L.kp_rollout_record_post.restype = C.c_int
    L.kp_mcp_tail.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, F, F, F, C.c_int, F, F, F, C.c_int, F, F, C.c_void_p]; L.kp_mcp_tail.restype = C.c_int
    L.kp_kin_advance.argtypes = [C.c_int, F, F, C.c_float, F, F, C.c_void_p]; L.kp_kin_advance.restype = C.c_int
    L.kp_gru_cell_step.argtypes = [C.c_int, C.c_int, C.c_int, F, F, F, F, F, F, F, F, C.c_void_p]; L.kp_gru_cell_step.restype = C.c_int
    L.kp_mcp_compose.argtypes = [C.c_int, C.c_int, C.c_int, F, F, F, C.c_int, F, F, C.c_void_p]; L.kp_mcp_compose.restype = C.c_int
    L.kp_gae.argtypes = [C.c_int, C.c_int, F, F, F, C.c_float, C.c_float, F, F, C.c_void_p]; L.kp_gae.restype = C.c_int
    L.kp_gae_bootstrap.argtypes = [C.c_int, C.c_int, F, F, F, F, C.c_float, C.c_float, F, F, C.c_void_p]; L.kp_gae_bootstrap.restype = C.c_int
    L.kp_gru_gates_forward.argtypes = [C.c_int, C.c_int, F, F, F, F, F, F, C.c_void_p]; L.kp_gru_gates_forward.restype = C.c_int
    L.kp_gru_gates_backward.argtypes = [C.c_int, C.c_int, F, F, F, F, F, F, F, F, F, C.c_void_p]; L.kp_gru_gates_backward.restype = C.c_int
    L.kp_sim_set_full_state.argtypes = [P, F, F, F, F, U8]; L.kp_sim_set_full_state.restype = C.c_int
    L.kp_sim_timing_reset.argtypes = [P]; L.kp_sim_timing_reset.restype = C.c_int
    L.kp_sim_timing_mean_seconds.argtypes = [P, C.POINTER(C.c_int)]; L.kp_sim_timing_mean_seconds.restype = C.c_double
    L.kp_sim_phase_cycles.argtypes = [P, C.POINTER(C.c_double)]; L.kp_sim_phase_cycles.restype = C.c_int
    L.kp_sim_phase_cycles_env.argtypes = [P, C.c_void_p]; L.kp_sim_phase_cycles_env.restype = C.c_int
    L.kp_job_schedule.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]; L.kp_job_schedule.restype = C.c_int
    L.kp_sim_launch_cost.argtypes = [P, C.c_void_p]; L.kp_sim_launch_cost.restype = C.c_int
    L.kp_sim_lean_state.argtypes = [P, C.c_void_p]; L.kp_sim_lean_state.restype = C.c_int
    L.kp_sim_set_objects.argtypes = [P, F, U8]; L.kp_sim_set_objects.restype = C.c_int
    L.kp_sim_set_obj_state.argtypes = [P, F, F, U8]; L.kp_sim_set_obj_state.restype = C.c_int
    L.kp_sim_fk.argtypes = [P, C.c_int, F, F, F, F, F, F]; L.kp_sim_fk.restype = C.c_int
    L.kp_sim_fk_backward.argtypes = [P, C.c_int, F, F, F, F, F]; L.kp_sim_fk_backward.restype = C.c_int
    L.kp_sim_set_stream.argtypes = [P, C.c_void_p]; L.kp_sim_set_stream.restype = C.c_int
    L.kp_sim_status_device.argtypes = [P]; L.kp_sim_status_device.restype = C.c_void_p
    L.kp_sim_mass_matrix.argtypes = [P, F, F]; L.kp_sim_mass_matrix.restype = C.c_int
    L.kp_sim_contacts.argtypes = [P, C.c_void_p]; L.kp_sim_contacts.restype = C.c_int
    L.kp_last_error.restype = C.c_char_p
    L.kp_version.restype = C.c_char_p
    _lib = L
    return L


def _check(rc, what):
    if rc != 0:
        raise KinPolyNativeError(f"{what}: {load_library().kp_last_error().decode()}")


def compile_model_native(xml_path: str, uhc_yml: str | None, out_kpm: str):
    """kp_model_compile: the XML + STL + uhc.yml -> blob compiler behind the C ABI (kinpoly_amd/csrc/kp_compile.hpp; needs no GPU)."""
    L = load_library()
    _check(L.kp_model_compile(xml_path.encode(), None if uhc_yml is None else uhc_yml.encode(), out_kpm.encode()), "kp_model_compile")


class KpModel:
    def __init__(self, kpm_path: str = DEFAULT_KPM, xml: tuple | None = None, **options):
        """kpm_path: a compiled blob; or xml=(xml_path, uhc_yml_path or None): compile the reference's scene on the spot (kp_model_load_xml)."""
        self.L = load_library()
        if xml is not None:
            self.h = self.L.kp_model_load_xml(xml[0].encode(), None if xml[1] is None else xml[1].encode())
        else:
            self.h = self.L.kp_model_load(kpm_path.encode())
        if not self.h:
            raise KinPolyNativeError(f"kp_model_load: {self.L.kp_last_error().decode()}")
        for k, v in options.items():
            self.set_option(k, v)

    def set_option(self, name, value):
        _check(self.L.kp_model_set_option(self.h, name.encode(), float(value)), "kp_model_set_option")

    def get_option(self, name):
        return self.L.kp_model_get_option(self.h, name.encode())

    def __del__(self):
        try:
            self.L.kp_model_free(self.h)
        except Exception:
            pass


def _ptr(t: torch.Tensor | None, n, dim, dtype=torch.float32):
    if t is None:
        return None
    if not t.is_cuda or t.dtype != dtype or not t.is_contiguous() or tuple(t.shape) != (n, dim):
        raise ValueError(f"expected contiguous {dtype} device tensor of shape ({n}, {dim}), got {t.dtype} {tuple(t.shape)} on {t.device}")
    return C.c_void_p(t.data_ptr())


def _mask_ptr(m: torch.Tensor | None, n):
    if m is None:
        return None
    if not m.is_cuda or m.dtype != torch.uint8 or tuple(m.shape) != (n,) or not m.is_contiguous():
        raise ValueError("env_mask must be a contiguous uint8 device tensor of shape (n_envs,)")
    return C.c_void_p(m.data_ptr())


class KpSim:
    """N batched environments on one GPU (one kp_sim handle)."""

    def __init__(self, model: KpModel, n_envs: int, device: int | torch.device = 0):
        if not torch.cuda.is_available():
            raise KinPolyNativeError("no HIP device visible: the simulator has no CPU fallback")
        self.model = model
        self.L = model.L
        self.n = int(n_envs)
        self.device = torch.device("cuda", device if isinstance(device, int) else (device.index or 0))
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream(self.device).cuda_stream
            self.h = self.L.kp_sim_create(model.h, self.n, self.device.index, C.c_void_p(stream))
        if not self.h:
            raise KinPolyNativeError(f"kp_sim_create: {self.L.kp_last_error().decode()}")
        self._stream = stream

    def use_current_stream(self):
        """Enqueue all later calls on torch's current stream of this device (the caller orders the old and the new stream)."""
        stream = torch.cuda.current_stream(self.device).cuda_stream
        if stream != self._stream:
            _check(self.L.kp_sim_set_stream(self.h, C.c_void_p(stream)), "kp_sim_set_stream")
            self._stream = stream

    def status_tensor(self) -> torch.Tensor:
        """int32 [4] device view of the launch status words (kp_sim_status_device): [2] != 0 = a queue launch stalled."""
        if getattr(self, "_status", None) is None:
            ptr = self.L.kp_sim_status_device(self.h)
            iface = {"shape": (4,), "typestr": "<i4", "data": (int(ptr), False), "version": 3, "strides": None}
            holder = type("_KpStatus", (), {"__cuda_array_interface__": iface})()
            self._status = torch.as_tensor(holder, device=self.device)
        return self._status

    def queue_counters(self) -> dict:
        """Counters of the last job-queue launch (host read): jobs claimed / published, jobs a wave kept instead of queueing, and the jobs the lean
        layout handed to kp_step_overflow_kernel (more contacts than EnvLdsLean::MAXCON)."""
        if getattr(self, "_qctr", None) is None:
            ptr = self.L.kp_sim_status_device(self.h)
            iface = {"shape": (128,), "typestr": "<i4", "data": (int(ptr), False), "version": 3, "strides": None}
            holder = type("_KpQueueCounters", (), {"__cuda_array_interface__": iface})()
            self._qctr = torch.as_tensor(holder, device=self.device)
        c = self._qctr.cpu().numpy()
        st = (C.c_int32 * 3)()
        _check(self.L.kp_sim_lean_state(self.h, C.cast(st, C.c_void_p)), "kp_sim_lean_state")
        return {"claimed": int(c[0]), "published": int(c[1]), "stalled": int(c[2]), "kept_by_their_wave": int(c[16]), "lean_overflow_jobs": int(c[64]),
                "lean_layout_next_launch": bool(st[0]), "fallbacks_to_full_layout": int(st[1]), "control_step_launches": int(st[2])}

    def record_contacts(self):
        """Arm the contact read-out (kp_sim_contacts): later step_ctrl launches keep the contact set of their last collision pass."""
        _check(self.L.kp_sim_contacts(self.h, None), "kp_sim_contacts")

    def contacts(self):
        """List over envs of dict(body, b2, dist [n], pos [n,3], normal [n,3]) of the last collision pass (after record_contacts())."""
        buf = np.zeros((self.n, 1 + 64 * 9), np.float32)
        _check(self.L.kp_sim_contacts(self.h, buf.ctypes.data_as(C.c_void_p)), "kp_sim_contacts")
        out = []
        for e in range(self.n):
            n = int(buf[e, 0]); r = buf[e, 1:1 + 9 * n].reshape(n, 9).astype(np.float64)
            out.append(dict(body=r[:, 0].astype(int), b2=r[:, 1].astype(int), dist=r[:, 2], pos=r[:, 3:6], normal=r[:, 6:9]))
        return out

    def mass_matrix(self):
        """(M [N,75,75], qfrc_bias [N,75]) of the state the derived quantities belong to (mj_fullM / data.qfrc_bias)."""
        M = torch.empty((self.n, 75, 75), dtype=torch.float32, device=self.device); b = self._new(75)
        _check(self.L.kp_sim_mass_matrix(self.h, C.c_void_p(M.data_ptr()), C.c_void_p(b.data_ptr())), "kp_sim_mass_matrix")
        return M, b

    def __del__(self):
        try:
            self.L.kp_sim_destroy(self.h)
        except Exception:
            pass

    def _new(self, dim):
        return torch.empty((self.n, dim), dtype=torch.float32, device=self.device)

    def set_state(self, qpos, qvel, env_mask=None):
        _check(self.L.kp_sim_set_state(self.h, _ptr(qpos, self.n, NQ), _ptr(qvel, self.n, NV), _mask_ptr(env_mask, self.n)), "kp_sim_set_state")

    def set_objects(self, obj_qpos, env_mask=None):
        _check(self.L.kp_sim_set_objects(self.h, _ptr(obj_qpos, self.n, 35), _mask_ptr(env_mask, self.n)), "kp_sim_set_objects")

    def set_obj_state(self, obj_qpos, obj_qvel, env_mask=None):
        _check(self.L.kp_sim_set_obj_state(self.h, _ptr(obj_qpos, self.n, 35), _ptr(obj_qvel, self.n, 30), _mask_ptr(env_mask, self.n)), "kp_sim_set_obj_state")

    def set_target(self, target_qpos, env_mask=None):
        _check(self.L.kp_sim_set_target(self.h, _ptr(target_qpos, self.n, NQ), _mask_ptr(env_mask, self.n)), "kp_sim_set_target")

    def step_ctrl(self, cc_action, n_substeps=15, env_mask=None):
        _check(self.L.kp_sim_step_ctrl(self.h, _ptr(cc_action, self.n, CC_ACTION_DIM), int(n_substeps), _mask_ptr(env_mask, self.n)), "kp_sim_step_ctrl")

    def step_head(self, kin_action):
        """step_begin + step_kin + set_target(step_kin's result) in one launch (kp_sim_step_head)"""
        _check(self.L.kp_sim_step_head(self.h, _ptr(kin_action, self.n, KIN_ACTION_DIM)), "kp_sim_step_head")

    def step_kin(self, kin_action, out=None):
        out = self._new(NQ) if out is None else out
        _check(self.L.kp_sim_step_kin(self.h, _ptr(kin_action, self.n, KIN_ACTION_DIM), _ptr(out, self.n, NQ)), "kp_sim_step_kin")
        return out

    def obs_cc(self, out=None, zf_mean=None, zf_std=None, clip=0.0):
        out = self._new(CC_OBS_DIM) if out is None else out
        zm = None if zf_mean is None else C.c_void_p(zf_mean.data_ptr())
        zs = None if zf_std is None else C.c_void_p(zf_std.data_ptr())
        _check(self.L.kp_sim_obs_cc(self.h, _ptr(out, self.n, CC_OBS_DIM), zm, zs, float(clip)), "kp_sim_obs_cc")
        return out

    def get(self, field: str, out=None):
        fid = FIELDS[field]
        dim = self.L.kp_field_dim(fid)
        out = self._new(dim) if out is None else out
        _check(self.L.kp_sim_get(self.h, fid, _ptr(out, self.n, dim)), "kp_sim_get")
        return out

    def view(self, field: str) -> torch.Tensor:
        """Zero-copy [N, dim] device view of a STORED field (kp_sim_field_device: qpos, qvel, xpos, ..., obj_qpos; not the derived read-outs).  The
        view follows the simulator: it shows the state as of the work enqueued before the reader on the same stream, and is overwritten by the next step --
        for a consumer that copies the rows it needs in its own launch (the sampler's record kernel), not for keeping."""
        cache = self.__dict__.setdefault("_views", {})
        if field not in cache:
            fid = FIELDS[field]
            ptr = self.L.kp_sim_field_device(self.h, fid)
            if not ptr:
                raise KinPolyNativeError(f"kp_sim_field_device: '{field}' is not a stored field")
            dim = self.L.kp_field_dim(fid)
            iface = {"shape": (self.n, dim), "typestr": "<f4", "data": (int(ptr), False), "version": 3, "strides": None}
            cache[field] = torch.as_tensor(type("_KpField", (), {"__cuda_array_interface__": iface})(), device=self.device)
        return cache[field]

    def set_full_state(self, qpos, qvel, qpos_d, qvel_d, env_mask=None):
        _check(self.L.kp_sim_set_full_state(self.h, _ptr(qpos, self.n, NQ), _ptr(qvel, self.n, NV), _ptr(qpos_d, self.n, NQ),
                                            _ptr(qvel_d, self.n, NV), _mask_ptr(env_mask, self.n)), "kp_sim_set_full_state")

    def fk(self, qpos_rows: torch.Tensor):
        """qpos_fk_batch on [R,76] rows -> dict(qpos, wbpos, wbquat, bquat, body_com) of device tensors."""
        R = qpos_rows.shape[0]
        if not qpos_rows.is_cuda or qpos_rows.dtype != torch.float32 or not qpos_rows.is_contiguous() or qpos_rows.shape[1] != NQ:
            raise ValueError("fk: expected contiguous float32 device tensor [R,76]")
        out = {k: torch.empty((R, d), dtype=torch.float32, device=self.device) for k, d in
               (("qpos", 76), ("wbpos", 72), ("wbquat", 96), ("bquat", 96), ("body_com", 72))}
        _check(self.L.kp_sim_fk(self.h, R, C.c_void_p(qpos_rows.data_ptr()), *[C.c_void_p(out[k].data_ptr()) for k in
                                ("qpos", "wbpos", "wbquat", "bquat", "body_com")]), "kp_sim_fk")
        return out

    def fk_backward(self, qpos_rows, wbpos, wbquat, grad_wbpos):
        """(d wbpos / d qpos)^T grad_wbpos for the rows of an fk() call -> [R,76]."""
        R = qpos_rows.shape[0]
        for t, d in ((qpos_rows, NQ), (wbpos, 72), (wbquat, 96), (grad_wbpos, 72)):
            if not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous() or tuple(t.shape) != (R, d):
                raise ValueError("fk_backward: expected contiguous float32 device tensors [R,76], [R,72], [R,96], [R,72]")
        out = torch.empty((R, NQ), dtype=torch.float32, device=self.device)
        _check(self.L.kp_sim_fk_backward(self.h, R, *[C.c_void_p(t.data_ptr()) for t in (qpos_rows, wbpos, wbquat, grad_wbpos, out)]), "kp_sim_fk_backward")
        return out

    def step_begin(self):
        _check(self.L.kp_sim_step_begin(self.h), "kp_sim_step_begin")

    def make_ctx(self, T, head_pose, head_vels, obj_rel, action_one_hot, gt_bquat, gt_wbpos, cur_t, obj_qpos=None, row=None) -> "KpCtx":
        n = self.n
        R = head_pose.shape[0] if row is not None else n        # context rows (>= n_envs with the row indirection)
        if row is not None and (row.dtype != torch.int32 or tuple(row.shape) != (n,) or not row.is_cuda):
            raise ValueError("row must be an int32 device tensor [n_envs]")
        for t, shp in ((head_pose, (R, T, 7)), (head_vels, (R, T, 6)), (obj_rel, (R, T, 7)), (action_one_hot, (R, 4)),
                       (gt_bquat, (R, T, 96)), (gt_wbpos, (R, T, 72))):
            if not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous() or tuple(t.shape) != shp:
                raise ValueError(f"context tensor must be contiguous float32 on device with shape {shp}, got {tuple(t.shape)}")
        if cur_t.dtype != torch.int32 or tuple(cur_t.shape) != (n,) or not cur_t.is_cuda:
            raise ValueError("cur_t must be an int32 device tensor [n_envs]")
        ctx = KpCtx(int(T), head_pose.data_ptr(), head_vels.data_ptr(), obj_rel.data_ptr(), action_one_hot.data_ptr(), gt_bquat.data_ptr(),
                    gt_wbpos.data_ptr(), None if obj_qpos is None else obj_qpos.data_ptr(), cur_t.data_ptr(), None if row is None else row.data_ptr())
        ctx._keep = (head_pose, head_vels, obj_rel, action_one_hot, gt_bquat, gt_wbpos, obj_qpos, cur_t, row)
        return ctx

    def obs_ar(self, ctx: "KpCtx", out=None):
        out = self._new(AR_OBS_DIM) if out is None else out
        _check(self.L.kp_sim_obs_ar(self.h, C.byref(ctx), _ptr(out, self.n, AR_OBS_DIM)), "kp_sim_obs_ar")
        return out

    def term_reward(self, ctx: "KpCtx", cfg: "KpRewardCfg", reward=None, info=None, fail=None, diffs=None):
        reward = torch.empty(self.n, device=self.device) if reward is None else reward
        info = self._new(6) if info is None else info
        fail = torch.empty(self.n, dtype=torch.uint8, device=self.device) if fail is None else fail
        diffs = self._new(2) if diffs is None else diffs
        _check(self.L.kp_sim_term_reward(self.h, C.byref(ctx), C.byref(cfg), C.c_void_p(reward.data_ptr()), _ptr(info, self.n, 6),
                                         C.c_void_p(fail.data_ptr()), _ptr(diffs, self.n, 2)), "kp_sim_term_reward")
        return reward, info, fail, diffs

    def post_step(self, ctx: "KpCtx", cfg: "KpRewardCfg", cur_t, row_len, episode_len, reward, info, fail, diffs, done, end, percent, done_count=None, obj7=None):
        """cur_t += 1; termination + reward; end / done / percent -- the tail of HumanoidAREnv.step in one launch (kp_sim_post_step).
        obj7 [N,7]: refreshed with the simulated pose of every env's action object (get_obj_qpos(action_one_hot))."""
        _check(self.L.kp_sim_post_step(self.h, C.byref(ctx), C.byref(cfg), C.c_void_p(cur_t.data_ptr()), C.c_void_p(row_len.data_ptr()), int(episode_len),
                                       C.c_void_p(reward.data_ptr()), _ptr(info, self.n, 6), C.c_void_p(fail.data_ptr()), _ptr(diffs, self.n, 2),
                                       C.c_void_p(done.data_ptr()), C.c_void_p(end.data_ptr()), C.c_void_p(percent.data_ptr()),
                                       None if done_count is None else C.c_void_p(done_count.data_ptr()), _ptr(obj7, self.n, 7)), "kp_sim_post_step")

    def reset_rows(self, init_qpos, init_qvel, row=None, env_mask=None, cur_t=None, set_target=True, aux_rows=None, row_obj_qpos=None, row_one_hot=None, obj7=None):
        """masked reset from context rows: state <- init rows, cur_t = 0, sim.forward(), target = FK(init) (kp_sim_reset_rows).
        aux_rows [N, C]: caller-owned per-env rows zeroed for the same envs (the policy's GRU state).  row_obj_qpos [R, 35] (+ row_one_hot [R, 4],
        obj7 [N, 7]): the object block of reset_model from the env's context row, and get_obj_qpos(action_one_hot) of it."""
        R = init_qpos.shape[0]
        for t, d in ((row_obj_qpos, 35), (row_one_hot, 4)):
            if t is not None and not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and tuple(t.shape) == (R, d)):
                raise ValueError(f"object row tables must be contiguous float32 device tensors [R, {d}]")
        if aux_rows is not None and not (aux_rows.is_cuda and aux_rows.dtype == torch.float32 and aux_rows.is_contiguous() and aux_rows.dim() == 2 and aux_rows.shape[0] == self.n):
            raise ValueError("aux_rows must be a contiguous float32 device tensor [N, C]")
        for t, d in ((init_qpos, NQ), (init_qvel, NV)):
            if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.dim() == 2 and t.shape[1] == d):
                raise ValueError("init rows must be contiguous float32 device tensors [R, dim]")
        _check(self.L.kp_sim_reset_rows(self.h, C.c_void_p(init_qpos.data_ptr()), C.c_void_p(init_qvel.data_ptr()), None if row is None else C.c_void_p(row.data_ptr()),
                                        _mask_ptr(env_mask, self.n), None if cur_t is None else C.c_void_p(cur_t.data_ptr()), int(bool(set_target)),
                                        None if aux_rows is None else C.c_void_p(aux_rows.data_ptr()), 0 if aux_rows is None else int(aux_rows.shape[1]),
                                        None if row_obj_qpos is None else C.c_void_p(row_obj_qpos.data_ptr()), None if row_one_hot is None else C.c_void_p(row_one_hot.data_ptr()),
                                        _ptr(obj7, self.n, 7)), "kp_sim_reset_rows")

    def diag(self) -> np.ndarray:
        out = np.zeros((self.n, 4), np.int32)
        _check(self.L.kp_sim_diag(self.h, out.ctypes.data_as(C.c_void_p)), "kp_sim_diag")
        return out

    def timing_reset(self):
        _check(self.L.kp_sim_timing_reset(self.h), "kp_sim_timing_reset")

    def timing_mean_seconds(self):
        n = C.c_int(0)
        t = self.L.kp_sim_timing_mean_seconds(self.h, C.byref(n))
        return t, n.value

    def phase_cycles(self):
        out = (C.c_double * 8)()
        _check(self.L.kp_sim_phase_cycles(self.h, out), "kp_sim_phase_cycles")
        return dict(zip(("spd", "kin_bias", "collide", "constraint", "smooth", "contact", "integrate", "total"), list(out)))

    def phase_cycles_env(self):
        """per-env shader-clock cycles of the last control step's phases, numpy [N, 8] (KP_PROFILE=1)."""
        import numpy as np
        out = np.zeros((self.n, 8), np.float64)
        _check(self.L.kp_sim_phase_cycles_env(self.h, out.ctypes.data_as(C.c_void_p)), "kp_sim_phase_cycles_env")
        return out

    def launch_cost(self):
        """shader-clock cycles every env took in the last control-step launch (numpy uint64 [N])."""
        import numpy as np
        out = np.zeros(self.n, np.uint32)
        _check(self.L.kp_sim_launch_cost(self.h, out.ctypes.data_as(C.c_void_p)), "kp_sim_launch_cost")
        return out.astype(np.uint64) << 10

    def last_step_seconds(self) -> float:
        return self.L.kp_sim_last_step_seconds(self.h)


def job_schedule(n_substeps: int, substeps_per_job: int = 4, taper: int = 1) -> list:
    """Job sizes of the queue-scheduled control step (kp_job_schedule; host arithmetic only)."""
    L = load_library()
    out = (C.c_int * 16)()
    n = L.kp_job_schedule(int(n_substeps), int(substeps_per_job), int(taper), out)
    if n < 0:
        raise KinPolyNativeError(f"kp_job_schedule: {L.kp_last_error().decode()}")
    return list(out[:n])


def _dptr(t, dtype=None, name="tensor"):
    """device pointer of a contiguous tensor (None -> NULL); bool tensors are passed as their uint8 storage"""
    if t is None:
        return None
    if t.dtype == torch.bool:
        t = t.view(torch.uint8)
    if not (t.is_cuda and t.is_contiguous()) or (dtype is not None and t.dtype != dtype):
        raise ValueError(f"{name}: expected a contiguous {dtype} device tensor, got {t.dtype} {'contiguous' if t.is_contiguous() else 'strided'} on {t.device}")
    return t.data_ptr()


def _want(name, x, *shape):
    """the record kernels index their buffers with fixed row widths (kp_rollout_kernels.hpp): a tensor of another shape would be read / written out of bounds"""
    if x is not None and tuple(x.shape) != tuple(shape):
        raise ValueError(f"{name}: expected shape {tuple(shape)}, got {tuple(x.shape)}")


def record_pre(t: int, T: int, obs=None, fresh=None, qpos=None, ctx_qpos=None, row=None, cur_t=None, row_len=None, row_meta=None,
               states=None, episode_start=None, curr_qpos=None, gt_target_qpos=None, meta=None):
    """kp_rollout_record_pre: the before-the-step half of the sampler's per-step record, one launch (see include/kinpoly_sim.h)."""
    L = load_library()
    first = next(x for x in (obs, qpos, fresh) if x is not None)
    f32, i32, u8 = torch.float32, torch.int32, torch.uint8
    n = first.shape[0]
    _want("obs", obs, n, AR_OBS_DIM); _want("fresh", fresh, n); _want("qpos", qpos, n, 76); _want("row", row, n); _want("cur_t", cur_t, n)
    _want("states", states, n, T, AR_OBS_DIM); _want("episode_start", episode_start, n, T); _want("curr_qpos", curr_qpos, n, T, 76)
    _want("gt_target_qpos", gt_target_qpos, n, T, 76); _want("meta", meta, n, T, 2)
    if ctx_qpos is not None and (ctx_qpos.dim() != 3 or ctx_qpos.shape[2] != 76):
        raise ValueError(f"ctx_qpos: expected [R, T_ctx, 76], got {tuple(ctx_qpos.shape)}")
    r = KpRecordPre(first.shape[0], int(T), int(t), 0 if ctx_qpos is None else int(ctx_qpos.shape[1]),
                    _dptr(obs, f32, "obs"), _dptr(fresh, u8, "fresh"), _dptr(qpos, f32, "qpos"), _dptr(ctx_qpos, f32, "ctx_qpos"), _dptr(row, i32, "row"), _dptr(cur_t, i32, "cur_t"),
                    _dptr(row_len, i32, "row_len"), _dptr(row_meta, f32, "row_meta"), _dptr(states, f32, "states"), _dptr(episode_start, u8, "episode_start"),
                    _dptr(curr_qpos, f32, "curr_qpos"), _dptr(gt_target_qpos, f32, "gt_target_qpos"), _dptr(meta, f32, "meta"))
    _check(L.kp_rollout_record_pre(C.byref(r), C.c_void_p(torch.cuda.current_stream(first.device).cuda_stream)), "kp_rollout_record_pre")


def record_post(t: int, T: int, fr_num=0.0, action=None, reward=None, fail=None, done=None, percent=None, c_info=None, obs=None, qpos=None, cc_action=None, cc_state=None, meta=None,
                actions=None, rewards=None, fails=None, dones=None, percents=None, c_infos=None, next_states=None, res_qpos=None, cc_actions=None, cc_states=None, v_metas=None):
    """kp_rollout_record_post: the after-the-step half (one launch)."""
    L = load_library()
    first = next(x for x in (action, reward, done) if x is not None)
    f32, u8 = torch.float32, torch.uint8
    n = first.shape[0]
    _want("action", action, n, 80); _want("reward", reward, n); _want("fail", fail, n); _want("done", done, n); _want("percent", percent, n); _want("c_info", c_info, n, 6)
    _want("obs", obs, n, AR_OBS_DIM); _want("qpos", qpos, n, 76); _want("cc_action", cc_action, n, CC_ACTION_DIM); _want("cc_state", cc_state, n, CC_OBS_DIM); _want("meta", meta, n, T, 2)
    _want("actions", actions, n, T, 80); _want("rewards", rewards, n, T); _want("fails", fails, n, T); _want("dones", dones, n, T); _want("percents", percents, n, T)
    _want("c_infos", c_infos, n, T, 6); _want("next_states", next_states, n, T, AR_OBS_DIM); _want("res_qpos", res_qpos, n, T, 76); _want("cc_actions", cc_actions, n, T, CC_ACTION_DIM)
    _want("cc_states", cc_states, n, T, CC_OBS_DIM); _want("v_metas", v_metas, n, T, 3)
    r = KpRecordPost(first.shape[0], int(T), int(t), float(fr_num), _dptr(action, f32, "action"), _dptr(reward, f32, "reward"), _dptr(fail, u8, "fail"), _dptr(done, u8, "done"),
                     _dptr(percent, f32, "percent"), _dptr(c_info, f32, "c_info"), _dptr(obs, f32, "obs"), _dptr(qpos, f32, "qpos"), _dptr(cc_action, f32, "cc_action"),
                     _dptr(cc_state, f32, "cc_state"), _dptr(meta, f32, "meta"), _dptr(actions, f32, "actions"), _dptr(rewards, f32, "rewards"), _dptr(fails, u8, "fails"),
                     _dptr(dones, u8, "dones"), _dptr(percents, f32, "percents"), _dptr(c_infos, f32, "c_infos"), _dptr(next_states, f32, "next_states"), _dptr(res_qpos, f32, "res_qpos"),
                     _dptr(cc_actions, f32, "cc_actions"), _dptr(cc_states, f32, "cc_states"), _dptr(v_metas, f32, "v_metas"))
    _check(L.kp_rollout_record_post(C.byref(r), C.c_void_p(torch.cuda.current_stream(first.device).cuda_stream)), "kp_rollout_record_post")


def pool_advance(done: torch.Tensor, head: torch.Tensor, ahead: torch.Tensor, row: torch.Tensor, n_slots: int):
    """Episode turnover on a ring of n_slots context rows per env (kp_pool_advance): done uint8 / bool [N]; head, ahead, row int32 [N], in place."""
    L = load_library()
    n = done.shape[0]
    if done.dtype == torch.bool:
        done = done.view(torch.uint8)
    for t in (head, ahead, row):
        if not (t.is_cuda and t.dtype == torch.int32 and t.is_contiguous() and tuple(t.shape) == (n,)):
            raise ValueError("pool_advance: head / ahead / row must be contiguous int32 device tensors [N]")
    if not (done.is_cuda and done.dtype == torch.uint8 and done.is_contiguous()):
        raise ValueError("pool_advance: done must be a contiguous uint8 / bool device tensor [N]")
    stream = torch.cuda.current_stream(done.device).cuda_stream
    _check(L.kp_pool_advance(n, int(n_slots), C.c_void_p(done.data_ptr()), C.c_void_p(head.data_ptr()), C.c_void_p(ahead.data_ptr()), C.c_void_p(row.data_ptr()),
                             C.c_void_p(stream)), "kp_pool_advance")


def mcp_compose(logits: torch.Tensor, prim: torch.Tensor, noise: torch.Tensor | None = None, std: torch.Tensor | None = None, out: torch.Tensor | None = None):
    """sum_k softmax(logits)_k * prim[k] (+ std * noise): PolicyMCP's mixing stage in one launch (kp_mcp_compose).  logits [N, K], prim [K, N, A]
    contiguous float32 device tensors; noise [N, A] may be a column slice of a wider buffer (row stride = noise.stride(0))."""
    L = load_library()
    K, n, A = prim.shape
    if not (logits.is_cuda and logits.dtype == torch.float32 and logits.is_contiguous() and tuple(logits.shape) == (n, K) and prim.dtype == torch.float32 and prim.is_contiguous()):
        raise ValueError("mcp_compose: logits [N, K] and prim [K, N, A] must be contiguous float32 device tensors")
    nz, stride = None, 0
    if noise is not None:
        if not (noise.is_cuda and noise.dtype == torch.float32 and tuple(noise.shape) == (n, A) and noise.stride(1) == 1 and std is not None and std.is_contiguous() and std.numel() == A):
            raise ValueError("mcp_compose: noise [N, A] (unit column stride) needs std [A]")
        nz, stride = C.c_void_p(noise.data_ptr()), int(noise.stride(0))
    out = torch.empty((n, A), device=prim.device, dtype=torch.float32) if out is None else out
    stream = torch.cuda.current_stream(prim.device).cuda_stream
    _check(L.kp_mcp_compose(n, K, A, C.c_void_p(logits.data_ptr()), C.c_void_p(prim.data_ptr()), nz, stride, None if std is None else C.c_void_p(std.data_ptr()),
                            C.c_void_p(out.data_ptr()), C.c_void_p(stream)), "kp_mcp_compose")
    return out


def mcp_tail(h2: torch.Tensor, b2: torch.Tensor, w3: torch.Tensor, b3: torch.Tensor, logits: torch.Tensor, noise: torch.Tensor | None = None,
             std: torch.Tensor | None = None, out: torch.Tensor | None = None):
    """PolicyMCP's last layer + mixing stage (kp_mcp_tail, fp32 MFMA): h2 [K, N, J] raw second-GEMM output, b2 [K, J], w3 [K, J, A] or its
    rows zero-padded to [K, J, 80] (16-byte operand loads), b3 [K, A], logits [N, K] (composer output before the softmax); noise [N, A] may
    be a column slice of a wider buffer."""
    L = load_library()
    K, n, J = h2.shape
    A, ldw = b3.shape[1], w3.shape[2]
    ok = all(t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() for t in (h2, b2, w3, b3, logits))
    if not (ok and tuple(b2.shape) == (K, J) and tuple(w3.shape) == (K, J, ldw) and ldw >= A and tuple(b3.shape) == (K, A) and tuple(logits.shape) == (n, K)):
        raise ValueError("mcp_tail: h2 [K, N, J], b2 [K, J], w3 [K, J, >= A], b3 [K, A], logits [N, K] must be contiguous float32 device tensors")
    nz, stride = None, 0
    if noise is not None:
        if not (noise.is_cuda and noise.dtype == torch.float32 and tuple(noise.shape) == (n, A) and noise.stride(1) == 1 and std is not None and std.is_contiguous() and std.numel() == A):
            raise ValueError("mcp_tail: noise [N, A] (unit column stride) needs std [A]")
        nz, stride = C.c_void_p(noise.data_ptr()), int(noise.stride(0))
    out = torch.empty((n, A), device=h2.device, dtype=torch.float32) if out is None else out
    stream = torch.cuda.current_stream(h2.device).cuda_stream
    _check(L.kp_mcp_tail(n, K, J, A, C.c_void_p(h2.data_ptr()), C.c_void_p(b2.data_ptr()), C.c_void_p(w3.data_ptr()), int(ldw), C.c_void_p(b3.data_ptr()),
                         C.c_void_p(logits.data_ptr()), nz, stride,
                         None if std is None else C.c_void_p(std.data_ptr()), C.c_void_p(out.data_ptr()), C.c_void_p(stream)), "kp_mcp_tail")
    return out


def kin_advance(qpos: torch.Tensor, action: torch.Tensor, dt: float = 1.0 / 30.0, next_qpos: torch.Tensor | None = None, qvel: torch.Tensor | None = None):
    """One frame of TrajARNet's kinematic roll-out (kp_kin_advance): (next_qpos [N,76] with a unit root quaternion, finite-difference qvel [N,75])."""
    L = load_library()
    n = qpos.shape[0]
    if not all(t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() for t in (qpos, action)) or tuple(qpos.shape) != (n, NQ) or tuple(action.shape) != (n, 80):
        raise ValueError("kin_advance: qpos [N,76] and action [N,80] must be contiguous float32 device tensors")
    next_qpos = torch.empty((n, NQ), device=qpos.device) if next_qpos is None else next_qpos
    qvel = torch.empty((n, NV), device=qpos.device) if qvel is None else qvel
    if not (next_qpos.is_contiguous() and qvel.is_contiguous()):
        raise ValueError("kin_advance: outputs must be contiguous")
    stream = torch.cuda.current_stream(qpos.device).cuda_stream
    _check(L.kp_kin_advance(n, C.c_void_p(qpos.data_ptr()), C.c_void_p(action.data_ptr()), float(dt), C.c_void_p(next_qpos.data_ptr()), C.c_void_p(qvel.data_ptr()),
                            C.c_void_p(stream)), "kp_kin_advance")
    return next_qpos, qvel


def gru_cell_step(gi: torch.Tensor, gh: torch.Tensor, b_ih: torch.Tensor, b_hh: torch.Tensor, h: torch.Tensor, state: torch.Tensor | None = None,
                  h_out: torch.Tensor | None = None, xcat: torch.Tensor | None = None):
    """GRUCell gate math after the two bias-free gate GEMMs (kp_gru_cell_step): returns h' [N, H]; xcat [N, D + H] <- [state | h'] when given."""
    L = load_library()
    n, H = h.shape
    ok = all(t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() for t in (gi, gh, b_ih, b_hh, h))
    if not (ok and tuple(gi.shape) == (n, 3 * H) and tuple(gh.shape) == (n, 3 * H) and b_ih.numel() == 3 * H and b_hh.numel() == 3 * H):
        raise ValueError("gru_cell_step: gi / gh [N, 3H], b_ih / b_hh [3H], h [N, H] must be contiguous float32 device tensors")
    D = 0
    if xcat is not None:
        D = state.shape[1]
        if not (state.is_cuda and state.dtype == torch.float32 and state.is_contiguous() and xcat.is_contiguous() and xcat.dtype == torch.float32 and tuple(xcat.shape) == (n, D + H)):
            raise ValueError("gru_cell_step: xcat must be a contiguous float32 [N, D + H] device tensor next to state [N, D]")
    h_out = torch.empty_like(h) if h_out is None else h_out
    stream = torch.cuda.current_stream(h.device).cuda_stream
    _check(L.kp_gru_cell_step(n, H, D, *(C.c_void_p(t.data_ptr()) for t in (gi, gh, b_ih, b_hh, h)), None if xcat is None else C.c_void_p(state.data_ptr()),
                              C.c_void_p(h_out.data_ptr()), None if xcat is None else C.c_void_p(xcat.data_ptr()), C.c_void_p(stream)), "kp_gru_cell_step")
    return h_out


def gae(rewards: torch.Tensor, masks: torch.Tensor, values: torch.Tensor, gamma: float, tau: float, last_values: torch.Tensor | None = None):
    """estimate_advantages before normalisation on env-major [N, T] float32 device tensors (k_gae); last_values [N] bootstraps
    episodes the horizon cut (kp_gae_bootstrap)."""
    L = load_library()
    n, T = rewards.shape
    for t in (rewards, masks, values):
        if not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous() or tuple(t.shape) != (n, T):
            raise ValueError("gae: expected contiguous float32 device tensors [N, T]")
    adv = torch.empty_like(rewards); ret = torch.empty_like(rewards)
    stream = torch.cuda.current_stream(rewards.device).cuda_stream
    if last_values is not None and (not last_values.is_cuda or last_values.dtype != torch.float32 or not last_values.is_contiguous() or tuple(last_values.shape) != (n,)):
        raise ValueError("gae: last_values must be a contiguous float32 device tensor [N]")
    _check(L.kp_gae_bootstrap(n, T, C.c_void_p(rewards.data_ptr()), C.c_void_p(masks.data_ptr()), C.c_void_p(values.data_ptr()),
                              None if last_values is None else C.c_void_p(last_values.data_ptr()), float(gamma), float(tau),
                              C.c_void_p(adv.data_ptr()), C.c_void_p(ret.data_ptr()), C.c_void_p(stream)), "kp_gae_bootstrap")
    return adv, ret
