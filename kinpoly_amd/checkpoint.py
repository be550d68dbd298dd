"""Reference checkpoint layout (kin_poly/core/agent_ar.py:316-364): a pickle of
    {'policy_dict': PolicyAR.state_dict(), 'value_dict': Value.state_dict(), 'running_state': ZFilter | None [, 'cc_dict']}
`policy_dict` keys carry the `traj_ar_net.` prefix of PolicyAR (policy_ar.py:30-37); the UHC checkpoint
(`results/motion_im/uhc/models/iter_XXXX.p`, humanoid_ar_v1.py:70-81) holds PolicyMCP / Value / ZFilter the same way.
The ZFilter / RunningStat objects are unpickled into light stand-ins (the reference remaps their module path with a
CustomUnpickler, uhc/utils/tools.py:6-17), so no reference code is needed to read or write these files.
"""
from __future__ import annotations

import io
import pickle

import numpy as np
import torch


class RunningStat:
    def __init__(self, shape=()):
        self._n, self._M, self._S = 0, np.zeros(shape), np.zeros(shape)

    @property
    def mean(self):
        return self._M

    @property
    def std(self):
        return np.sqrt(self._S / (self._n - 1) if self._n > 1 else np.square(self._M))


class ZFilter:
    def __init__(self, shape=(), demean=True, destd=True, clip=10.0):
        self.demean, self.destd, self.clip, self.rs = demean, destd, clip, RunningStat(shape)


# pickles written by save_checkpoint must name the classes by the reference's module path
RunningStat.__module__ = ZFilter.__module__ = "uhc.khrylib.utils.zfilter"


class _Unpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if name == "ZFilter":
            return ZFilter
        if name == "RunningStat":
            return RunningStat
        return super().find_class(module, name)


class _RefModulePath:
    """pickle writes classes by module path and checks that path at dump time: provide it for the duration of a dump
    (only if the real reference package is not importable in this process)."""
    NAMES = ("uhc", "uhc.khrylib", "uhc.khrylib.utils", "uhc.khrylib.utils.zfilter")

    def __enter__(self):
        import sys
        import types
        self.added = []
        for n in self.NAMES:
            if n not in sys.modules:
                sys.modules[n] = types.ModuleType(n); self.added.append(n)
        self.prev = {k: getattr(sys.modules[self.NAMES[-1]], k, None) for k in ("ZFilter", "RunningStat")}
        sys.modules[self.NAMES[-1]].ZFilter = ZFilter
        sys.modules[self.NAMES[-1]].RunningStat = RunningStat
        return self

    def __exit__(self, *a):
        import sys
        for k, v in self.prev.items():
            if v is not None:
                setattr(sys.modules[self.NAMES[-1]], k, v)
        for n in self.added:
            sys.modules.pop(n, None)


def load_checkpoint(path_or_bytes):
    f = io.BytesIO(path_or_bytes) if isinstance(path_or_bytes, (bytes, bytearray)) else open(path_or_bytes, "rb")
    with f:
        cp = _Unpickler(f).load()
    return cp


def split_policy_dict(policy_dict: dict):
    """PolicyAR state_dict -> state_dict for kinpoly_amd.context.TrajARNet (strip 'traj_ar_net.', drop the frozen copy old_arnet)."""
    out = {}
    for k, v in policy_dict.items():
        if k.startswith("traj_ar_net."):
            out[k[len("traj_ar_net."):]] = v
        elif k == "action_log_std":
            out[k] = v
    return out


def running_state_arrays(rs):
    """ZFilter -> (mean, std, clip) for kinpoly_amd.env.RunningState."""
    if rs is None:
        return None
    return np.asarray(rs.rs.mean, np.float64), np.asarray(rs.rs.std, np.float64), float(rs.clip)


def save_checkpoint(path, traj_ar_net: torch.nn.Module, value_net: torch.nn.Module, running_state=None, cc_policy=None):
    pd = {"traj_ar_net." + k: v.detach().cpu() for k, v in traj_ar_net.state_dict().items() if k != "action_log_std"}
    pd["action_log_std"] = traj_ar_net.state_dict()["action_log_std"].detach().cpu()
    cp = {"policy_dict": pd, "value_dict": {k: v.detach().cpu() for k, v in value_net.state_dict().items()}, "running_state": running_state}
    if cc_policy is not None:
        cp["cc_dict"] = {k: v.detach().cpu() for k, v in cc_policy.state_dict().items()}
    with _RefModulePath(), open(path, "wb") as f:
        pickle.dump(cp, f)
    return cp


# Keys a reference checkpoint may legitimately lack when it is loaded into this repository's modules (everything else missing or unexpected is an error: a
# silently half-loaded policy trains and evaluates without complaint).  PolicyAR.state_dict() = traj_ar_net.* + action_log_std (+ old_arnet.*, which
# split_policy_dict drops): TrajARNet here holds exactly those parameters (tests/golden/traj_ar_net.npz: the reference's own key list), so nothing is allowed to be absent.
ALLOWED_MISSING_POLICY_KEYS = frozenset()


def load_state_strict(module: torch.nn.Module, state: dict, allow_missing=ALLOWED_MISSING_POLICY_KEYS, what="checkpoint"):
    """load_state_dict that names what does not fit: keys missing from `state` (beyond `allow_missing`) or not known to `module` raise."""
    state = {k: (v if torch.is_tensor(v) else torch.as_tensor(v)) for k, v in state.items()}
    own = set(module.state_dict().keys())
    missing = sorted(own - set(state) - set(allow_missing))
    unexpected = sorted(set(state) - own)
    if missing or unexpected:
        raise KeyError(f"{what}: does not match {type(module).__name__} -- missing {missing[:8]}{' ...' if len(missing) > 8 else ''}, unexpected {unexpected[:8]}{' ...' if len(unexpected) > 8 else ''}")
    module.load_state_dict(state, strict=False)          # strict=False only for the allow-listed keys, checked above
    return module


def load_bench_policies(policy_ckpt: str, cc_ckpt: str | None = None, device="cuda"):
    """The networks of a finished training run, as the reference's evaluation builds them (eval_ar_policy.py: policy checkpoint `iter_XXXX.p` of the
    kinematic policy + the UHC checkpoint the env is built around, humanoid_ar_v1.py:60-81): returns {"kin_policy": TrajARNet, "cc_policy": PolicyMCP,
    "cc_running_state": RunningState | None}.  policy_ckpt: {'policy_dict', 'value_dict', 'running_state' [, 'cc_dict']} (AgentAR.save_checkpoint);
    cc_ckpt: the UHC's {'policy_dict', 'value_dict', 'running_state'} (scripts/train_uhc.py --save) -- its ZFilter always, its weights unless the policy
    checkpoint carries the jointly trained controller (cc_dict)."""
    from .context import TrajARNet
    from .env import RunningState
    from .nets import PolicyMCP
    cp = load_checkpoint(policy_ckpt)
    net = load_state_strict(TrajARNet(), split_policy_dict(cp["policy_dict"]), what=policy_ckpt).to(device).float()
    mcp, rs = PolicyMCP(), None
    cc_weights = cp.get("cc_dict")
    if cc_ckpt:
        ccp = load_checkpoint(cc_ckpt)
        arr = running_state_arrays(ccp.get("running_state"))
        if arr is not None:
            rs = RunningState(arr[0], arr[1], arr[2], device)
        if cc_weights is None:
            cc_weights = ccp["policy_dict"]
    if cc_weights is None:
        raise KeyError(f"{policy_ckpt} holds no controller weights (cc_dict) and no UHC checkpoint was given")
    load_state_strict(mcp, cc_weights, allow_missing=(), what=cc_ckpt or policy_ckpt)
    return {"kin_policy": net.eval(), "cc_policy": mcp.to(device).float().eval(), "cc_running_state": rs}
