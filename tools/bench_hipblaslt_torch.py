#!/usr/bin/env python
"""torch.matmul (hipBLASLt / rocBLAS under PyTorch-ROCm) on the training step's projection shapes, for comparison with
tools/bench_gemm.py: how far is the hand-written GEMM from the vendor library on the same GPU?"""
import torch

SHAPES = [(32768, 8192, 1024, "NT"), (32768, 1024, 8192, "NN"), (8192, 1024, 32768, "TN"), (32768, 3072, 1024, "NT"),
          (262144, 1024, 1024, "NT"), (1024, 1024, 262144, "TN"), (32768, 1024, 4096, "NT")]
for M, N, K, mode in SHAPES:
    if mode == "NT":      # x[M,K] @ w[N,K]^T   (forward)
        a, b = torch.randn(M, K, device="cuda", dtype=torch.bfloat16), torch.randn(N, K, device="cuda", dtype=torch.bfloat16)
        f = lambda: a @ b.t()
    elif mode == "NN":    # dy[M,K] @ w[K,N]    (dgrad)
        a, b = torch.randn(M, K, device="cuda", dtype=torch.bfloat16), torch.randn(K, N, device="cuda", dtype=torch.bfloat16)
        f = lambda: a @ b
    else:                 # dy[K,M]^T @ x[K,N]  (wgrad)
        a, b = torch.randn(K, M, device="cuda", dtype=torch.bfloat16), torch.randn(K, N, device="cuda", dtype=torch.bfloat16)
        f = lambda: a.t() @ b
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        f()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"M={M:6d} N={N:5d} K={K:6d} {mode}: {ms * 1e3:8.1f} us  {2.0 * M * N * K / ms / 1e9:7.1f} TF/s")
