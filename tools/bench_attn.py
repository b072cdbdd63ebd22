#!/usr/bin/env python
"""Micro-benchmark of the attention kernels at the training step's shapes (tv2o-medium, B=16, S=2048):
event-level flash attention forward/backward (H=16, head_dim 64) and token-level attention (N=32768 octets,
H=4, head_dim 256), bf16, timed with HIP events."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from midi_model_amd import ops  # noqa: E402


def timeit(fn, iters):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    iters = int(os.environ.get("MH_BENCH_ITERS", "5"))
    dev, dt = "cuda", torch.bfloat16
    g = torch.Generator(device=dev).manual_seed(0)
    B, S, H = 16, int(os.environ.get("MH_BENCH_S", "2048")), 16
    D = H * 64
    qkv = torch.randn((B * S, 3 * D), device=dev, generator=g).to(dt)
    do = torch.randn((B * S, D), device=dev, generator=g).to(dt)
    o = torch.empty((B * S, D), device=dev, dtype=dt)
    Sp = (S + 63) // 64 * 64
    lse = torch.zeros(B * H * Sp, device=dev)
    dqkv = torch.empty_like(qkv)
    fl = 4.0 * B * H * S * (S + 1) / 2 * 64  # QK^T + PV on the causal triangle
    ms = timeit(lambda: ops.attn_fwd(qkv, o, lse, B, S, H, 0.125), iters)
    print(f"attn_fwd  (+V^T prep) B={B} S={S} H={H}: {ms * 1e3:8.1f} us  {fl / ms / 1e9:7.1f} TF/s")
    ms = timeit(lambda: ops.attn_bwd(qkv, o, do, lse, dqkv, B, S, H, 0.125), iters)
    print(f"attn_bwd  (+prep)     B={B} S={S} H={H}: {ms * 1e3:8.1f} us  {2.5 * fl / ms / 1e9:7.1f} TF/s")
    N, T, Ht = B * S, 8, 4
    Dt = Ht * 256
    qkv_t = torch.randn((N * T, 3 * Dt), device=dev, generator=g).to(dt)
    do_t = torch.randn((N * T, Dt), device=dev, generator=g).to(dt)
    o_t = torch.empty((N * T, Dt), device=dev, dtype=dt)
    dq_t = torch.empty_like(qkv_t)
    by = N * T * Dt * 2
    ms = timeit(lambda: ops.tokattn_fwd(qkv_t, o_t, N, T, Ht, 256 ** -0.5), iters)
    print(f"tokattn_fwd N={N}: {ms * 1e3:8.1f} us  {4 * by / ms / 1e6:7.1f} GB/s (algorithmic: qkv in, o out)")
    ms = timeit(lambda: ops.tokattn_bwd(qkv_t, do_t, dq_t, N, T, Ht, 256 ** -0.5), iters)
    print(f"tokattn_bwd N={N}: {ms * 1e3:8.1f} us  {7 * by / ms / 1e6:7.1f} GB/s (algorithmic: qkv + do in, dqkv out)")


if __name__ == "__main__":
    main()
