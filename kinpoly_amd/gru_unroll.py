"""Fused re-unroll of the kinematic policy's GRU for the PPO / supervised updates (reference: PolicyAR.initialize_rnn +
forward(mode="train"), kin_poly/models/policy_ar.py:104-122, 216-240 -- a Python loop over T_max GRUCell steps on the padded
[T_max, n_episodes] layout, re-run 10 + 20 times per iteration).

Here the batch stays env-major [N, T, .] with an episode-start mask, and the T-step recurrence is one autograd node:

    forward   gi = x W_ih^T + b_ih for ALL steps in one GEMM (outside this node);  per step: gh = hm W_hh^T + b_hh (library GEMM, MFMA) and one
              HIP kernel (k_gru_gates_fwd) for the gate math, the new hidden state and its masked copy for the next step;
    backward  per step (reverse): one HIP kernel (k_gru_gates_bwd) + one GEMM for the gradient reaching the previous hidden state;
              dW_hh / db_hh as ONE [3H, N T] x [N T, H] GEMM at the end instead of T small ones.

The result equals torch.nn.GRUCell stepped in a Python loop (`KinPolicy.unroll_reference`) to fp32 rounding; without a HIP device (CPU
tests, fp64) `unroll` falls back to that loop -- it is the same arithmetic, not a different product path.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import sim as kpsim


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


class _GruUnroll(torch.autograd.Function):
    """H_all [T, N, H] = GRU(gi [T, N, 3H], hx0 [N, H]) with hidden state zeroed where start[t, n]; W_hh [3H, H], b_hh [3H]."""

    @staticmethod
    def forward(ctx, gi, w_hh, b_hh, hx0, keep):
        L = kpsim.load_library()
        T, N, H3 = gi.shape
        H = H3 // 3
        stream = C.c_void_p(torch.cuda.current_stream(gi.device).cuda_stream)
        gh = torch.empty_like(gi)                                   # saved for backward: gates are recomputed from gi + gh
        h_all = torch.empty((T, N, H), device=gi.device, dtype=gi.dtype)
        hm_all = torch.empty((T, N, H), device=gi.device, dtype=gi.dtype)       # hm_all[t] = GEMM input of step t (masked previous state)
        hm_all[0] = hx0 * keep[0].unsqueeze(1)
        w_t = w_hh.t()
        for t in range(T):
            torch.addmm(b_hh, hm_all[t], w_t, out=gh[t])
            nxt = t + 1 < T
            kpsim._check(L.kp_gru_gates_forward(N, H, _p(gi[t]), _p(gh[t]), _p(hm_all[t]), _p(keep[t + 1]) if nxt else None,
                                                _p(h_all[t]), _p(hm_all[t + 1]) if nxt else None, stream), "kp_gru_gates_forward")
        ctx.save_for_backward(gi, gh, hm_all, w_hh, keep)
        return h_all

    @staticmethod
    def backward(ctx, dh_all):
        gi, gh, hm_all, w_hh, keep = ctx.saved_tensors
        L = kpsim.load_library()
        T, N, H3 = gi.shape
        H = H3 // 3
        stream = C.c_void_p(torch.cuda.current_stream(gi.device).cuda_stream)
        dh_all = dh_all.contiguous()
        dgi = torch.empty_like(gi); dgh = torch.empty_like(gi)
        dhz = torch.empty((N, H), device=gi.device, dtype=gi.dtype)
        carry = None
        for t in range(T - 1, -1, -1):
            kpsim._check(L.kp_gru_gates_backward(N, H, _p(gi[t]), _p(gh[t]), _p(hm_all[t]), _p(dh_all[t]), _p(carry), _p(keep[t + 1]) if carry is not None else None,
                                                 _p(dgi[t]), _p(dgh[t]), _p(dhz), stream), "kp_gru_gates_backward")
            if t > 0 or ctx.needs_input_grad[3]:
                carry = torch.addmm(dhz, dgh[t], w_hh)                # gradient w.r.t. hm_all[t]; masked by keep[t] when step t - 1 consumes it
        dw = torch.mm(dgh.view(T * N, H3).t(), hm_all.view(T * N, H))
        db = dgh.view(T * N, H3).sum(0)
        dhx0 = (carry * keep[0].unsqueeze(1)) if ctx.needs_input_grad[3] else None
        return dgi, dw, db, dhx0, None


def gru_unroll(cell: torch.nn.GRUCell, states: torch.Tensor, episode_start: torch.Tensor, hx0: torch.Tensor | None = None) -> torch.Tensor:
    """Hidden states [N, T, H] of `cell` stepped over states [N, T, D] (env-major), zeroing the state where episode_start[n, t]."""
    N, T, _ = states.shape
    H = cell.hidden_size
    x_tm = states.transpose(0, 1).contiguous()                       # [T, N, D]
    gi = torch.nn.functional.linear(x_tm.view(T * N, -1), cell.weight_ih, cell.bias_ih).view(T, N, 3 * H)
    keep = (~episode_start).to(states.dtype).t().contiguous()        # [T, N]
    h0 = torch.zeros((N, H), device=states.device, dtype=states.dtype) if hx0 is None else hx0.to(states.dtype).contiguous()
    h_all = _GruUnroll.apply(gi.contiguous(), cell.weight_hh, cell.bias_hh, h0, keep)
    return h_all.transpose(0, 1)                                     # [N, T, H] view
