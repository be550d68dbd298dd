"""Policy / value networks of the rollout path, batched over environments on the GPU.

Parameter names mirror the reference modules so the reference's checkpoints (`policy_dict`,
`value_dict`, `cc_dict`; kin_poly/core/agent_ar.py:341-364) load with `load_state_dict`:

  MLP          uhc/khrylib/models/mlp.py:5-25          (affine_layers.N.{weight,bias})
  Value        uhc/khrylib/rl/core/critic.py:5-18      (net, value_head)
  PolicyMCP    uhc/core/policy_mcp.py:9-38             (nets.K.0 = MLP, nets.K.1 = Linear, composer.0 = MLP)
  KinPolicy    kin_poly/models/traj_ar_smpl_net.py:48-53,333-343 + policy_ar.py:317-320
               (action_rnn.rnn_f = GRUCell, action_mlp, action_fc; fixed log_std, kin_poly.yml:38)

The large GEMMs run through rocBLAS / hipBLASLt (MFMA) via torch; on the device's inference path PolicyMCP's 8 primitive MLPs are one
wide GEMM + one batched GEMM + kp_mcp_tail (this repo's fp32 MFMA kernel: the 8 output layers, the composer's softmax and the weighted sum),
and the kinematic policy's GRU step is two gate GEMMs + kp_gru_cell_step.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn


TUNED_GEMMS = __import__("os").environ.get("KP_TUNED_GEMMS") or __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), "assets", "tunableop_gfx950.csv")


def enable_tuned_gemms(path: str = TUNED_GEMMS) -> bool:
    """Use the rocBLAS / hipBLASLt solutions recorded for the rollout's GEMM shapes (4096 envs: GRU gates, the two policy MLPs,
    PolicyMCP's batched primitives) by torch's TunableOp on MI355X (`tools/tune_gemms.py` writes the file).  Selection only:
    no tuning happens at run time, shapes that are not in the file (or a file recorded with other library versions) fall back
    to the default heuristics.  Returns whether the file was taken."""
    import os
    if not (torch.cuda.is_available() and os.path.exists(path) and hasattr(torch.cuda, "tunable")):
        return False
    try:
        torch.cuda.tunable.enable(True)
        torch.cuda.tunable.tuning_enable(False)
        torch.cuda.tunable.set_filename(os.path.join(os.environ.get("TMPDIR", "/tmp"), "kp_tunableop_unused.csv"))   # never write next to the asset
        if hasattr(torch.cuda.tunable, "write_file_on_exit"):
            torch.cuda.tunable.write_file_on_exit(False)
        return bool(torch.cuda.tunable.read_file(path))
    except Exception:
        torch.cuda.tunable.enable(False)
        return False


class MLP(nn.Module):
    def __init__(self, input_dim, hidden_dims=(128, 128), activation="tanh"):
        super().__init__()
        self.activation = {"tanh": torch.tanh, "relu": torch.relu, "sigmoid": torch.sigmoid}[activation]
        self.out_dim = hidden_dims[-1]
        self.affine_layers = nn.ModuleList()
        last = input_dim
        for nh in hidden_dims:
            self.affine_layers.append(nn.Linear(last, nh))
            last = nh

    def forward(self, x):
        fused = self.activation is torch.relu and x.is_cuda and x.dim() == 2 and not torch.is_grad_enabled()
        for affine in self.affine_layers:
            # rollout path: bias + relu in the GEMM epilogue (hipBLASLt) instead of a separate pass over the activations
            x = torch._addmm_activation(affine.bias, x, affine.weight.t()) if fused else self.activation(affine(x))
        return x


class Value(nn.Module):
    def __init__(self, net, net_out_dim=None):
        super().__init__()
        self.net = net
        self.value_head = nn.Linear(net.out_dim if net_out_dim is None else net_out_dim, 1)
        self.value_head.weight.data.mul_(0.1)
        self.value_head.bias.data.mul_(0.0)

    def forward(self, x):
        return self.value_head(self.net(x))


class PolicyMCP(nn.Module):
    """Multiplicative-compositional UHC controller: sum_k softmax(composer(x))_k * net_k(x)."""

    def __init__(self, state_dim=784, action_dim=75, policy_hsize=(512, 256), policy_htype="relu", num_primitive=8,
                 composer_dim=(300, 200), log_std=-2.3, fix_std=True):
        super().__init__()
        self.type = "gaussian"
        self.num_primitive = num_primitive
        self.nets = nn.ModuleList()
        for _ in range(num_primitive):
            action_mean = nn.Linear(policy_hsize[-1], action_dim)
            action_mean.weight.data.mul_(0.1)
            action_mean.bias.data.mul_(0.0)
            self.nets.append(nn.Sequential(MLP(state_dim, policy_hsize, policy_htype), action_mean))
        self.composer = nn.Sequential(MLP(state_dim, list(composer_dim) + [num_primitive], policy_htype), nn.Softmax(dim=1))
        self.action_log_std = nn.Parameter(torch.ones(1, action_dim) * log_std, requires_grad=not fix_std)
        self._fused = None

    def _fuse(self):
        """[K*h1, in], [K, h1, h2], [K, h2, A] stacked weights (rebuilt when parameters change version)."""
        ver = tuple((p._version, p.data_ptr(), p.dtype, p.device) for p in self.parameters())
        if self._fused is None or self._fused[0] != ver:
            w1 = torch.cat([n[0].affine_layers[0].weight for n in self.nets], 0)
            b1 = torch.cat([n[0].affine_layers[0].bias for n in self.nets], 0)
            w2 = torch.stack([n[0].affine_layers[1].weight.t() for n in self.nets], 0)
            b2 = torch.stack([n[0].affine_layers[1].bias for n in self.nets], 0)
            w3 = torch.stack([n[1].weight.t() for n in self.nets], 0)
            b3 = torch.stack([n[1].bias for n in self.nets], 0)
            w3p = torch.nn.functional.pad(w3, (0, 80 - w3.shape[2])) if 48 < w3.shape[2] < 80 else w3      # rows of 80: kp_mcp_tail's 16-byte operand loads
            self._fused = (ver, tuple(t.detach().contiguous() for t in (w1, b1, w2, b2, w3, b3, w3p)))
        return self._fused[1]

    def _primitives_fused(self, x):
        """the K primitive MLPs as three (batched) GEMMs; returns their outputs in the GEMM's own layout [K, N, A]"""
        w1, b1, w2, b2, w3, b3, _ = self._fuse()
        K = self.num_primitive
        h = torch._addmm_activation(b1, x, w1.t())                   # [N, K*h1], relu in the GEMM epilogue
        h = h.view(x.shape[0], K, -1).transpose(0, 1)                # [K, N, h1]
        h = torch.relu(torch.baddbmm(b2.unsqueeze(1), h, w2))        # [K, N, h2]
        return torch.baddbmm(b3.unsqueeze(1), h, w3)                 # [K, N, A]

    def _inference(self, x):
        return not (torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()))

    def action_mean(self, x, noise=None):
        """sum_k softmax(composer(x))_k * net_k(x).  Inference path on the device in fp32: two (batched) library GEMMs for the primitives' hidden
        layers, the composer's GEMMs, and ONE fp32 MFMA kernel (kp_mcp_tail) for the rest -- second-layer bias + relu, the K output layers,
        softmax, weighted sum, optional mean + std * noise -- instead of a bias broadcast, a relu pass, a third batched GEMM with 75 columns,
        a transpose view, a softmax, a broadcast multiply, a sum and an addcmul."""
        if not self._inference(x):
            x_all = torch.stack([net(x) for net in self.nets], dim=1)  # training path: plain modules
            return torch.sum(self.composer(x)[:, :, None] * x_all, dim=1)
        if x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and self.num_primitive <= 16 and self.action_log_std.shape[1] <= 80:
            from . import sim as kpsim
            w1, b1, w2, b2, _, b3, w3 = self._fuse()
            K = self.num_primitive
            if w2.shape[2] % 64 == 0:
                h = torch._addmm_activation(b1, x, w1.t()).view(x.shape[0], K, -1).transpose(0, 1)     # [K, N, h1], relu in the GEMM epilogue
                h2 = torch.bmm(h, w2)                                                              # [K, N, h2]: raw, its bias and relu are applied by the tail
                logits = self.composer[0](x)                                                       # the composer MLP before its softmax
                return kpsim.mcp_tail(h2, b2, w3, b3, logits.contiguous(), noise, None if noise is None else self.std())
        prim = self._primitives_fused(x)
        mean = torch.sum(self.composer(x)[:, :, None] * prim.transpose(0, 1), dim=1)
        return mean if noise is None else torch.addcmul(mean, self.std().to(mean.dtype), noise)

    def std(self):
        """exp(action_log_std) [A], cached until the parameter changes (one launch less per env-step)"""
        p = self.action_log_std
        key = (p._version, p.data_ptr())
        if getattr(self, "_std_key", None) != key:
            self._std_key, self._std = key, torch.exp(p.detach()).reshape(-1).contiguous()
        return self._std

    def forward(self, x):
        mean = self.action_mean(x)
        return mean, self.action_log_std.expand_as(mean)

    def select_action(self, x, mean_action=False, generator=None, noise=None):
        """mean action, or a sample mean + exp(log_std) * eps; `noise` [N, A]: standard-normal draws made ahead by the caller (a rollout draws
        the whole horizon's noise in one launch) instead of a randn here."""
        if mean_action:
            return self.action_mean(x)
        if noise is None:
            noise = torch.randn((x.shape[0], self.action_log_std.shape[1]), device=x.device, dtype=x.dtype, generator=generator)
        if self._inference(x):
            return self.action_mean(x, noise)
        mean, log_std = self.forward(x)
        return torch.addcmul(mean, torch.exp(log_std), noise)


class _StepRNN(nn.Module):
    """RNN(cell_type='gru') in 'step' mode (uhc/khrylib/models/rnn.py:5-36) with a batched hidden state."""

    def __init__(self, input_dim, out_dim):
        super().__init__()
        self.rnn_f = nn.GRUCell(input_dim, out_dim)


class KinPolicy(nn.Module):
    """The per-step part of PolicyAR / TrajARNet: h <- GRUCell(x, h); MLP([x, h]); fc -> 80-d action."""

    def __init__(self, state_dim=105, action_dim=80, rnn_hdim=1024, mlp_hsize=(1024, 512, 256), htype="relu", log_std=-3.2):
        super().__init__()
        self.state_dim, self.action_dim, self.rnn_hdim = state_dim, action_dim, rnn_hdim
        self.action_rnn = _StepRNN(state_dim, rnn_hdim)
        self.action_mlp = MLP(rnn_hdim + state_dim, mlp_hsize, htype)
        self.action_fc = nn.Linear(mlp_hsize[-1], action_dim)
        self.action_log_std = nn.Parameter(torch.ones(1, action_dim) * log_std, requires_grad=False)
        self.log_std_init = float(log_std)

    @torch.no_grad()
    def refresh_log_std(self):
        """After a change of dtype: the reference builds `action_log_std = ones * log_std` directly in its training dtype (fp64: exactly -3.2), while a
        module built in fp32 and cast up carries fp32(-3.2) = -3.2000000477, a 1e-7 relative error of the variance that the PPO ratio sees
        (8e-9 in the surrogate of tests/golden/update_params.npz).  Re-evaluates the constant in the current dtype -- only while it still is the
        constructor's value (a checkpoint's own action_log_std is left alone)."""
        p = self.action_log_std
        if bool((p.float() == torch.tensor(self.log_std_init, dtype=torch.float32, device=p.device)).all()):
            p.fill_(self.log_std_init)
        return self

    def init_hidden(self, n, device=None):
        return torch.zeros((n, self.rnn_hdim), device=device or self.action_fc.weight.device, dtype=self.action_fc.weight.dtype)

    def get_action(self, state, hx):
        cell = self.action_rnn.rnn_f
        if state.is_cuda and state.dtype == torch.float32 and state.dim() == 2 and not torch.is_grad_enabled() and state.shape[1] <= self.rnn_hdim:
            # roll-out step on the device: the two gate GEMMs (library, MFMA) + one kernel for the gate math that also writes [state | h'],
            # the action MLP's input row, in place of gru_cell_forward (+ its backward workspace) and torch.cat
            from . import sim as kpsim
            state, hx = state.contiguous(), hx.contiguous()
            gi = torch.nn.functional.linear(state, cell.weight_ih)
            gh = torch.nn.functional.linear(hx, cell.weight_hh)
            x = torch.empty((state.shape[0], state.shape[1] + self.rnn_hdim), device=state.device, dtype=state.dtype)
            hx = kpsim.gru_cell_step(gi, gh, cell.bias_ih, cell.bias_hh, hx, state, None, x)
            return self.action_fc(self.action_mlp(x)), hx
        hx = cell(state, hx)
        x = torch.cat((state, hx), dim=1)
        return self.action_fc(self.action_mlp(x)), hx

    def std(self):
        p = self.action_log_std
        key = (p._version, p.data_ptr(), p.dtype)
        if getattr(self, "_std_key", None) != key:
            self._std_key, self._std = key, torch.exp(p.detach())
        return self._std

    def select_action(self, state, hx, mean_action=False, generator=None, noise=None):
        """noise [N, A]: standard-normal draws made ahead by the caller (see PolicyMCP.select_action)"""
        mean, hx = self.get_action(state, hx)
        if mean_action:
            return mean, hx
        if noise is None:
            noise = torch.randn(mean.shape, device=mean.device, dtype=mean.dtype, generator=generator)
        return torch.addcmul(mean, self.std(), noise), hx

    def log_prob(self, mean, action):
        """DiagGaussian.log_prob summed over the action dims (uhc/khrylib/rl/core/distributions.py:22-23)."""
        log_std = self.action_log_std
        var = torch.exp(2 * log_std)
        return (-(action - mean) ** 2 / (2 * var) - 0.5 * math.log(2 * math.pi) - log_std).sum(1, keepdim=True)

    def unroll(self, states, episode_start, hx0=None):
        """Training-time forward over an env-major rollout [N, T, state_dim]: re-runs the GRU through time,
        zeroing the hidden state where `episode_start[n, t]` (what initialize_rnn + the padded [T_max,
        n_episodes] re-pack do in the reference, policy_ar.py:104-122,216-234).  hx0 [N, H]: hidden state the behaviour
        policy held before row 0 (episodes that continue from the previous sample() call; default zeros).  Returns means [N, T, A].
        On the device the recurrence is the fused node of kinpoly_amd/gru_unroll.py (HIP gate kernels + one GEMM per step, weight
        gradients as one GEMM) and the MLP runs once over all N * T rows; on the CPU (fp64 fixtures) it is the plain GRUCell loop."""
        if not (states.is_cuda and states.dtype == torch.float32):
            return self.unroll_reference(states, episode_start, hx0)
        from .gru_unroll import gru_unroll
        N, T, _ = states.shape
        h = gru_unroll(self.action_rnn.rnn_f, states, episode_start, hx0)                      # [N, T, H]
        x = torch.cat((states, h), dim=2).reshape(N * T, -1)
        return self.action_fc(self.action_mlp(x)).view(N, T, -1)

    def unroll_reference(self, states, episode_start, hx0=None):
        """The same computation as a Python loop of T GRUCell + MLP steps (the shape of the reference's own loop; parity baseline)."""
        N, T, _ = states.shape
        hx = self.init_hidden(N, states.device) if hx0 is None else hx0.to(states.dtype)
        outs = []
        for t in range(T):
            hx = hx * (~episode_start[:, t]).to(hx.dtype).unsqueeze(1)
            mean, hx = self.get_action(states[:, t], hx)
            outs.append(mean)
        return torch.stack(outs, 1)
