#!/usr/bin/env python
"""Time per DEPENDENT launch of every projection class of a decode step (mh_gemm_skinny), the way generate() runs them: chains
of launches in a captured hipGraph, each launch reading what the previous one wrote (64 rows, bf16).  `cold` chains walk through
~600 MB of distinct weight matrices (the event-level stack: 403 MB per decode step, more than the Infinity Cache keeps), `warm`
chains cycle through three (the token-level stack: 51 MB re-read eight times per event).
Usage: python tools/skinny_chain.py [launches=96]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from midi_model_amd import ops  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 96
B = 64
dev = "cuda"
torch.manual_seed(0)


def weights(rows, K, n):
    return [(torch.randn((rows, K), device=dev) * (K ** -0.5)).to(torch.bfloat16) for _ in range(n)]


def chain(name, N, K, mode, rstd, res, cold, ld_out=None):
    """a chain of L launches out[64, N] = f(in[64, :K] . W^T); the next launch reads the first K columns of out"""
    rows = 2 * N if mode == ops.SKINNY_GATEUP else N
    per = rows * K * 2
    nmat = max(3, min(L, int(600e6 // per))) if cold else 3
    Ws = weights(rows, K, nmat)
    width = max(N, K) if ld_out is None else ld_out
    bufs = [torch.zeros((B, width), dtype=torch.bfloat16, device=dev) for _ in range(3)]
    bufs[0][:, :K] = torch.randn((B, K), device=dev).to(torch.bfloat16)

    def body():
        for i in range(L):
            src, dst = bufs[i % 3], bufs[(i + 1) % 3]
            a = src[:, :K]
            ops.gemm_skinny(a, Ws[i % nmat], dst[:, :N], mode=mode, norm_eps=1e-6 if rstd else 0.0,
                            res=(src[:, :N] if res and N <= src.shape[1] else None))
            if N < K:  # keep the next launch's input well defined (and dependent): widen by re-reading what was written
                pass

    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        body()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            body()
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(15):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            g.replay()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
    print(f"{name:58s} N={N:5d} K={K:5d} {'cold' if cold else 'warm'} ({nmat:3d} matrices of {per / 1e6:5.1f} MB): "
          f"{1e3 * best / L:6.2f} us per dependent launch", flush=True)
    del Ws, bufs


P, G = ops.SKINNY_PLAIN, ops.SKINNY_GATEUP
for cold in (False, True):
    chain("o / token down: plain + residual", 1024, 1024, P, False, True, cold)
    chain("q|k|v: folded RMSNorm", 3072, 1024, P, True, False, cold)
    chain("lm_head: folded RMSNorm, ragged N", 3406, 1024, P, True, False, cold, ld_out=3456)
    chain("token gate|up + SwiGLU: folded RMSNorm", 1024, 1024, G, True, False, cold)
    chain("event gate|up + SwiGLU: folded RMSNorm", 4096, 1024, G, True, False, cold)
# the K = 4096 down projection needs a 4096-wide input: alternate it with the event gate|up and subtract
def pair(cold):
    N1, K1, N2, K2 = 4096, 1024, 1024, 4096
    n1 = max(3, min(L // 2, int(300e6 // (2 * N1 * K1 * 2)))) if cold else 3
    n2 = max(3, min(L // 2, int(300e6 // (N2 * K2 * 2)))) if cold else 3
    W1, W2 = weights(2 * N1, K1, n1), weights(N2, K2, n2)
    x = torch.randn((B, 1024), device=dev).to(torch.bfloat16)
    a = torch.zeros((B, 4096), dtype=torch.bfloat16, device=dev)
    y = [x, torch.zeros_like(x)]

    def body():
        for i in range(L // 2):
            ops.gemm_skinny(y[i % 2], W1[i % n1], a, mode=G, norm_eps=1e-6)
            ops.gemm_skinny(a, W2[i % n2], y[(i + 1) % 2], res=y[i % 2])

    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        body()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            body()
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(15):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            g.replay()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
    print(f"{'event MLP pair: gate|up + SwiGLU, then down K = 4096 + residual':58s} {'cold' if cold else 'warm'}: "
          f"{1e3 * best / (L // 2):6.2f} us per pair", flush=True)


pair(False)
pair(True)

if os.environ.get("SWEEP"):
    print("-- tilings (skinny_mb = 16-row blocks per workgroup, skinny_nbt = 16-column blocks; 0 = the launcher's choice)")
    for name, N, K, mode, rstd in (("q|k|v", 3072, 1024, P, True), ("lm_head", 3406, 1024, P, True), ("o / down", 1024, 1024, P, False),
                                   ("token gate|up", 1024, 1024, G, True), ("event gate|up", 4096, 1024, G, True)):
        for nbt in ((0, 1, 2) if mode == P else (0,)):
            for mb in (0, 1, 2, 4):
                ops.set_option("skinny_mb", mb)
                ops.set_option("skinny_nbt", nbt)
                chain(f"{name} mb={mb} nbt={nbt}", N, K, mode, rstd, mode == P and not rstd, True, ld_out=3456 if N == 3406 else None)
    ops.set_option("skinny_mb", 0)
    ops.set_option("skinny_nbt", 0)
