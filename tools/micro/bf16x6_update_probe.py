"""bf16x6 (fp32 emulation on the bf16 matrix pipe, see bf16x6_probe.py) on the GEMM shapes of one GRU re-unroll of the update: the per-step
recurrent GEMMs forward and backward, the single dW_hh GEMM, the MLP over all N T rows (forward, dX, dW) -- against the tuned fp32 library
GEMMs, with the cost of the operand-split passes a stand-alone implementation would pay."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from kinpoly_amd import nets
print("tuned fp32 solutions:", nets.enable_tuned_gemms(), flush=True)
torch.manual_seed(0)
dev = "cuda"


def split3(x):
    a1 = x.bfloat16(); r = x - a1.float()
    a2 = r.bfloat16(); r = r - a2.float()
    return a1, a2, r.bfloat16()


def timeit(f, n=10):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def pad8(x):            # K (last dim) to a multiple of 8 bf16 = 16 bytes
    k = x.shape[1]; p = (-k) % 8
    return x if p == 0 else torch.nn.functional.pad(x, (0, p))


NT = 4096 * 24
# (name, M, K, N, count per unroll): y[M, N] = A[M, K] B[N, K]^T
shapes = [("recurrent fwd  gh = hm W_hh^T", 4096, 1024, 3072, 24), ("recurrent bwd  dhm = dgh W_hh", 4096, 3072, 1024, 24),
          ("dW_hh = dgh^T hm", 3072, NT, 1024, 1), ("mlp1 fwd", NT, 1129, 1024, 1), ("mlp1 dX", NT, 1024, 1129, 1), ("mlp1 dW", 1024, NT, 1129, 1),
          ("mlp2 fwd", NT, 1024, 512, 1), ("mlp2 dX", NT, 512, 1024, 1), ("mlp2 dW", 512, NT, 1024, 1),
          ("mlp3 fwd", NT, 512, 256, 1), ("gi fwd", NT, 105, 3072, 1), ("dW_ih", 3072, NT, 105, 1)]
tot32 = tot6 = tots = 0.0
for name, M, K, N, cnt in shapes:
    A = torch.randn(M, K, device=dev); B = torch.randn(N, K, device=dev) * 0.05
    t32 = timeit(lambda: A @ B.t())
    y32 = A @ B.t()
    a = split3(pad8(A)); b = split3(pad8(B))
    A6 = torch.cat([a[0], a[0], a[0], a[1], a[1], a[2]], 1).contiguous(); B6 = torch.cat([b[0], b[1], b[2], b[0], b[1], b[0]], 1).contiguous()
    t6 = timeit(lambda: torch.mm(A6, B6.t(), out_dtype=torch.float32))
    y6 = torch.mm(A6, B6.t(), out_dtype=torch.float32)
    # split passes: one read of fp32 + one write of 6 bf16 per element, priced at 5 TB/s for A only (weights are split once per update step)
    ts = (M * K * (4 + 12)) / 5e12 * 1e6
    if M * K * N <= 4096 * 3072 * 1024 * 2:
        ref = A.double() @ B.double().t(); sc = float(ref.abs().max())
        e32 = float((y32.double() - ref).abs().max()) / sc; e6 = float((y6.double() - ref).abs().max()) / sc
        es = f"err {e32:.1e} / {e6:.1e}"
    else:
        es = f"x6 vs fp32 {float((y6 - y32).abs().max()) / float(y32.abs().max()):.1e}"
    fl = 2.0 * M * K * N
    print(f"{name:32s} [{M},{K}]x[{K},{N}] x{cnt}: fp32 {t32:8.1f} us ({fl / t32 / 1e6:4.0f} TF) | bf16x6 {t6:8.1f} us ({6 * fl / t6 / 1e6:5.0f} TF) | split(A) ~{ts:6.1f} us | {es}", flush=True)
    tot32 += cnt * t32; tot6 += cnt * t6; tots += cnt * ts
    del A, B, A6, B6, a, b, y32, y6
print(f"one unroll forward + backward: fp32 {tot32 / 1e3:.2f} ms, bf16x6 {tot6 / 1e3:.2f} ms (+ {tots / 1e3:.2f} ms of stand-alone split passes)")
