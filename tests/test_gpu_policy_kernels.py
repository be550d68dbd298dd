"""Round-3 GPU tests of the policy-side kernels between the library GEMMs (csrc/kp_policy_kernels.hpp): PolicyMCP's MFMA tail against an
fp64 evaluation of uhc/core/policy_mcp.py:30-38, the GRU roll-out step against torch.nn.GRUCell, the reset launch's policy-state rows."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _mcp_reference(h2, b2, w3, b3, logits, noise=None, std=None):
    """fp64: sum_k softmax(logits)_k (b3_k + relu(h2_k + b2_k) w3_k) (+ std * noise)"""
    d = lambda t: t.double().cpu()  # noqa: E731
    h2, b2, w3, b3, logits = map(d, (h2, b2, w3, b3, logits))
    prim = torch.baddbmm(b3.unsqueeze(1), torch.relu(h2 + b2.unsqueeze(1)), w3)          # [K, N, A]
    out = (torch.softmax(logits, 1).t().unsqueeze(2) * prim).sum(0)
    return out if noise is None else out + d(std) * d(noise)


@pytest.mark.parametrize("n,K,J,A", [(4096, 8, 256, 75), (37, 8, 256, 75), (1, 3, 64, 10), (50, 5, 128, 40), (16, 16, 64, 80)])
def test_mcp_tail_matches_fp64(n, K, J, A):
    from kinpoly_amd import sim as kpsim
    g = torch.Generator(device="cuda").manual_seed(n + K)
    r = lambda *s: torch.randn(s, device="cuda", generator=g)  # noqa: E731
    h2, b2, w3, b3, logits = r(K, n, J), r(K, J), r(K, J, A) * 0.1, r(K, A), r(n, K) * 2
    # asymmetric operands (a transposed or permuted fragment cannot pass): every primitive, hidden unit and column weighted differently
    w3 = w3 * (1 + torch.arange(A, device="cuda") * 0.01) * (1 + torch.arange(J, device="cuda")[:, None] * 0.003)
    ref = _mcp_reference(h2, b2, w3, b3, logits)
    out = kpsim.mcp_tail(h2, b2, w3.contiguous(), b3, logits)
    scale = float(ref.abs().max())
    assert float((out.double().cpu() - ref).abs().max()) < 3e-6 * max(scale, 1.0) * np.sqrt(J * K / 64)
    if 48 < A < 80:         # rows padded to 80 columns: the 16-byte operand path, other column-to-tile map
        outp = kpsim.mcp_tail(h2, b2, torch.nn.functional.pad(w3, (0, 80 - A), value=float("nan")).contiguous(), b3, logits)
        assert float((outp.double().cpu() - ref).abs().max()) < 3e-6 * max(scale, 1.0) * np.sqrt(J * K / 64)
    wide = r(n, A + 13)
    std = torch.rand(A, device="cuda", generator=g)
    out2 = kpsim.mcp_tail(h2, b2, w3.contiguous(), b3, logits, wide[:, 5:5 + A], std)
    ref2 = _mcp_reference(h2, b2, w3, b3, logits, wide[:, 5:5 + A], std)
    assert float((out2.double().cpu() - ref2).abs().max()) < 3e-6 * max(scale, 1.0) * np.sqrt(J * K / 64)


def test_policy_mcp_inference_path_is_the_module_math():
    """PolicyMCP.select_action on the device (two library GEMMs + kp_mcp_tail) against the plain nn.Module evaluation in fp64"""
    from kinpoly_amd.nets import PolicyMCP
    torch.manual_seed(3)
    pol = PolicyMCP().cuda()
    for p in pol.parameters():
        p.data.add_(torch.randn_like(p) * 0.05)
    x = torch.randn(300, 784, device="cuda")
    noise = torch.randn(300, 155, device="cuda")
    with torch.no_grad():
        got = pol.select_action(x, False, None, noise[:, 80:])
        ref_pol = PolicyMCP().double()
        ref_pol.load_state_dict({k: v.double().cpu() for k, v in pol.state_dict().items()})
        xd = x.double().cpu()
        mean = torch.sum(ref_pol.composer(xd)[:, :, None] * torch.stack([net(xd) for net in ref_pol.nets], 1), 1)
        ref = mean + torch.exp(ref_pol.action_log_std) * noise[:, 80:].double().cpu()
    assert float((got.double().cpu() - ref).abs().max()) < 2e-5


@pytest.mark.parametrize("n,H,D", [(4096, 1024, 105), (7, 32, 5)])
def test_gru_cell_step_matches_grucell(n, H, D):
    from kinpoly_amd import sim as kpsim
    torch.manual_seed(n)
    cell = torch.nn.GRUCell(D, H).cuda()
    x, h = torch.randn(n, D, device="cuda"), torch.randn(n, H, device="cuda")
    with torch.no_grad():
        ref = torch.nn.GRUCell(D, H).double()
        ref.load_state_dict({k: v.double().cpu() for k, v in cell.state_dict().items()})
        want = ref(x.double().cpu(), h.double().cpu())
        gi, gh = torch.nn.functional.linear(x, cell.weight_ih), torch.nn.functional.linear(h, cell.weight_hh)
        xcat = torch.full((n, D + H), float("nan"), device="cuda")
        got = kpsim.gru_cell_step(gi, gh, cell.bias_ih, cell.bias_hh, h, x, None, xcat)
        assert float((got.double().cpu() - want).abs().max()) < 5e-6
        assert torch.equal(xcat[:, :D], x) and torch.equal(xcat[:, D:], got)
        h2 = h.clone()
        kpsim.gru_cell_step(gi, gh, cell.bias_ih, cell.bias_hh, h2, None, h2, None)          # in place, no [state | h] row
        assert torch.equal(h2, got)


def test_kin_policy_rollout_step_is_the_module_math():
    """KinPolicy.get_action on the device (gate GEMMs + kp_gru_cell_step, no torch.cat) against GRUCell + cat + MLP in fp64"""
    from kinpoly_amd.nets import KinPolicy
    torch.manual_seed(5)
    pol = KinPolicy().cuda()
    s, h = torch.randn(129, 105, device="cuda"), torch.randn(129, 1024, device="cuda") * 0.3
    with torch.no_grad():
        mean, h1 = pol.get_action(s, h)
        ref = KinPolicy().double()
        ref.load_state_dict({k: v.double().cpu() for k, v in pol.state_dict().items()})
        mean_r, h1_r = ref.get_action(s.double().cpu(), h.double().cpu())
    assert float((h1.double().cpu() - h1_r).abs().max()) < 5e-6 and float((mean.double().cpu() - mean_r).abs().max()) < 2e-5


def test_reset_rows_zeroes_the_policy_state_of_the_reset_envs_only():
    from kinpoly_amd import sim as kpsim
    std = np.load(os.path.join(ROOT, "tests", "golden", "standing_neutral.npz"))
    n = 9
    sim = kpsim.KpSim(kpsim.KpModel(), n, 0)
    q = torch.tensor(std["qpos"], dtype=torch.float32, device="cuda").repeat(n, 1).contiguous()
    v = torch.zeros((n, 75), device="cuda")
    hx = torch.ones((n, 1024), device="cuda")
    mask = torch.tensor([1, 0, 0, 1, 0, 0, 0, 0, 1], dtype=torch.uint8, device="cuda")
    cur_t = torch.full((n,), 7, dtype=torch.int32, device="cuda")
    sim.reset_rows(q, v, None, mask, cur_t, True, hx)
    torch.cuda.synchronize()
    assert torch.equal(hx.sum(1).cpu(), torch.tensor([0, 1024, 1024, 0, 1024, 1024, 1024, 1024, 0.0]))
    assert cur_t.cpu().tolist() == [0, 7, 7, 0, 7, 7, 7, 7, 0]


def test_kin_advance_is_step_ar_plus_finite_difference_velocity():
    """kp_kin_advance against the fp64 oracle: HumanoidAREnv.step_ar (kin_poly/envs/humanoid_ar_v1.py:216-241), the root-quaternion
    normalisation of TrajARNet.step (traj_ar_smpl_net.py:323-327) and get_qvel_fd_batch (torch_utils.py:315-331) in one launch."""
    import sys
    sys.path.insert(0, ROOT)
    from kinpoly_amd import sim as kpsim
    from kinpoly_amd.context import get_qvel_fd_batch
    from oracle import np_oracle as O
    rng = np.random.default_rng(5)
    n, dt = 300, 1.0 / 30.0
    qpos = rng.normal(0, 0.4, (n, 76)); qpos[:, 2] += 0.9
    qpos[:, 3:7] = rng.normal(0, 1, (n, 4)); qpos[:, 3:7] /= np.linalg.norm(qpos[:, 3:7], axis=1, keepdims=True)
    act = rng.normal(0, 0.5, (n, 80))
    act[:100, 77:80] *= 10.0 ** rng.uniform(-3, -1, (100, 1))             # slow turns: the kernel takes sin / angle from |xyz| and atan2, not from 1 - w^2 and acos;
    # the slowest of them (1e-5 rad per frame) are below the reference's own `sin < 1e-5` test, which its clamped acos never lets it take: 2 xyz / dt there and here
    act[-1, 77:80] = 0.0                                                 # the 'small' branch: exactly no rotation
    nxt, qv = kpsim.kin_advance(torch.tensor(qpos, dtype=torch.float32, device="cuda"), torch.tensor(act, dtype=torch.float32, device="cuda"), dt)
    torch.cuda.synchronize()
    q32 = torch.tensor(qpos, dtype=torch.float32).double().numpy(); a32 = torch.tensor(act, dtype=torch.float32).double().numpy()
    want = np.stack([O.step_ar(q32[i], a32[i], dt) for i in range(n)])
    want[:, 3:7] /= np.linalg.norm(want[:, 3:7], axis=1, keepdims=True)
    # the reference holds unit quaternions to 1e-16; the fp32 rows are unit to 3e-8, which moves w of a 1e-4 rad turn by more than 1 - w itself in ITS
    # formula.  Give the fp64 evaluation the renormalised rows.
    q32n = q32.copy(); q32n[:, 3:7] /= np.linalg.norm(q32n[:, 3:7], axis=1, keepdims=True)
    wantv = get_qvel_fd_batch(torch.tensor(q32n), torch.tensor(want), dt).numpy()
    assert np.abs(nxt.cpu().numpy() - want).max() < 2e-6
    got = qv.cpu().numpy()
    assert np.abs(got[:, :3] - wantv[:, :3]).max() < 1e-4 and np.abs(got[:, 6:] - wantv[:, 6:]).max() < 1e-4
    assert np.abs(got[:-1, 3:6] - wantv[:-1, 3:6]).max() < 3e-5         # rotation vector / dt: 1e-7 of the quaternion product x 2 / dt
    assert np.abs(got[-1, 3:6]).max() < 6e-6           # no rotation asked for: what is left is the fp32 rounding of next (x) cur^-1 (1e-7) x 2 / dt, as 1e-15 is in the reference's fp64
    # in-place record layout of the roll-out: outputs are rows of time-major buffers
    Q = torch.zeros((2, n, 76), device="cuda"); V = torch.zeros((2, n, 75), device="cuda")
    Q[0].copy_(torch.tensor(qpos, dtype=torch.float32))
    kpsim.kin_advance(Q[0], torch.tensor(act, dtype=torch.float32, device="cuda"), dt, Q[1], V[1])
    assert torch.equal(Q[1], nxt) and torch.equal(V[1], qv)
