"""Feasibility probe: fp32-equivalent GEMM out of six bf16 MFMA products (a = a1 + a2 + a3 in bf16 pieces, the six products with i + j <= 4),
as ONE K-concatenated bf16 GEMM with fp32 output, against the library's fp32 GEMM: time and error vs fp64."""
import torch
torch.manual_seed(0)
dev = "cuda"


def split3(x):
    a1 = x.bfloat16(); r = x - a1.float()
    a2 = r.bfloat16(); r = r - a2.float()
    a3 = r.bfloat16()
    return a1, a2, a3


def timeit(f, n=20):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for (M, K, N) in ((4096, 784, 4096), (4096, 1024, 3072), (4096, 1129, 1024), (4096, 1024, 512)):
    A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) * 0.05      # y = A W^T
    ref = (A.double() @ W.double().t())
    y32 = A @ W.t()
    t32 = timeit(lambda: A @ W.t())
    a1, a2, a3 = split3(A); b1, b2, b3 = split3(W)
    A6 = torch.cat([a1, a1, a1, a2, a2, a3], 1).contiguous()
    B6 = torch.cat([b1, b2, b3, b1, b2, b1], 1).contiguous()            # [N, 6K]
    y6 = torch.mm(A6, B6.t(), out_dtype=torch.float32)
    t6 = timeit(lambda: torch.mm(A6, B6.t(), out_dtype=torch.float32))
    A3 = torch.cat([a1, a1, a2], 1).contiguous(); B3 = torch.cat([b1, b2, b1], 1).contiguous()
    y3 = torch.mm(A3, B3.t(), out_dtype=torch.float32)
    t3 = timeit(lambda: torch.mm(A3, B3.t(), out_dtype=torch.float32))
    t1 = timeit(lambda: torch.mm(a1, b1.t(), out_dtype=torch.float32))
    sc = float(ref.abs().max())
    err = lambda y: float((y.double() - ref).abs().max()) / sc   # noqa: E731
    fl = 2.0 * M * K * N
    print(f"M{M} K{K} N{N}: fp32 {t32:.1f} us ({fl / t32 / 1e6:.0f} TF) err {err(y32):.2e} | bf16x6 one GEMM {t6:.1f} us ({6 * fl / t6 / 1e6:.0f} TF bf16) err {err(y6):.2e} | "
          f"bf16x3 {t3:.1f} us err {err(y3):.2e} | plain bf16 {t1:.1f} us ({fl / t1 / 1e6:.0f} TF)", flush=True)
