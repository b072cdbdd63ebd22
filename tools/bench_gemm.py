#!/usr/bin/env python
"""Micro-benchmark of the projection GEMM kernels on the training step's shapes (tv2o-medium, B=16, S=2048):
for every (M, N, K, transA, transB) the step launches, time each kernel variant (argument 1: comma list, variant +
10 x ablation build) with HIP events on random bf16 data and cross-check against the first variant listed.  Output: one line per shape/variant."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from midi_model_amd import ops  # noqa: E402

_AB = ops.ab_library()  # a measurement tool: the compared kernel forms live in libmidihip_ab.so (build.py, -DMH_AB_BUILDS)
_AB.__enter__()

SHAPES = [  # (M, N, K, ta, tb, calls per step)
    (32768, 3072, 1024, 0, 0, 12), (32768, 1024, 1024, 0, 0, 12), (32768, 8192, 1024, 0, 0, 12), (32768, 1024, 4096, 0, 0, 12),
    (262144, 3072, 1024, 0, 0, 3), (262144, 1024, 1024, 0, 0, 6), (262144, 2048, 1024, 0, 0, 3), (32768, 3406, 1024, 0, 0, 8),
    (32768, 1024, 3072, 0, 1, 12), (32768, 1024, 1024, 0, 1, 12), (32768, 1024, 8192, 0, 1, 12), (32768, 4096, 1024, 0, 1, 12),
    (262144, 1024, 3072, 0, 1, 3), (262144, 1024, 1024, 0, 1, 6), (262144, 1024, 2048, 0, 1, 3), (32768, 1024, 3406, 0, 1, 8),
    (3072, 1024, 32768, 1, 1, 12), (1024, 1024, 32768, 1, 1, 12), (8192, 1024, 32768, 1, 1, 12), (1024, 4096, 32768, 1, 1, 12),
    (3072, 1024, 262144, 1, 1, 3), (1024, 1024, 262144, 1, 1, 6), (2048, 1024, 262144, 1, 1, 3), (3406, 1024, 32768, 1, 1, 8),
]


def main():
    variants = [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["0", "1"])]
    splitks = [int(v) for v in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["0"])]
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    tot = {(v, s): 0.0 for v in variants for s in splitks}
    shapes = SHAPES
    iters = 5
    if os.environ.get("MH_BENCH_SHAPES") == "quick":  # PMC passes: three representative shapes, one timed launch
        shapes = [(32768, 8192, 1024, 0, 0, 12), (32768, 1024, 8192, 0, 1, 12), (8192, 1024, 32768, 1, 1, 12)]
        iters = 1
    if os.environ.get("MH_BENCH_SHAPES") == "nn":  # ablation runs (row-major operands only)
        shapes = [(32768, 1024, 512, 0, 0, 0), (32768, 1024, 1024, 0, 0, 12), (32768, 1024, 2048, 0, 0, 0), (32768, 1024, 4096, 0, 0, 12),
                  (32768, 3072, 1024, 0, 0, 12), (262144, 1024, 1024, 0, 0, 6), (32768, 8192, 1024, 0, 0, 12),
                  (8192, 1024, 32768, 1, 1, 12), (1024, 1024, 262144, 1, 1, 6), (1024, 1024, 32768, 1, 1, 12)]
    if os.environ.get("MH_BENCH_SHAPES") == "nnq":  # ablation builds of the row-major main loops
        shapes = [(32768, 1024, 4096, 0, 0, 12), (32768, 8192, 1024, 0, 0, 12), (32768, 1024, 16384, 0, 0, 0)]
    if os.environ.get("MH_BENCH_SHAPES") == "mixed":   # one row-major, one contraction-major operand
        shapes = [sh for sh in SHAPES if sh[3] != sh[4]] + [(8192, 1024, 32768, 1, 0, 0), (3072, 1024, 32768, 1, 0, 0)]
    if os.environ.get("MH_BENCH_SHAPES") == "wgrad":
        shapes = [sh for sh in SHAPES if sh[3] and sh[4]]
    if os.environ.get("MH_BENCH_SHAPES") == "few":
        shapes = [(32768, 8192, 1024, 0, 0, 12), (32768, 1024, 8192, 0, 1, 12), (8192, 1024, 32768, 1, 1, 12)]
    for (M, N, K, ta, tb, calls) in shapes:
        Kp = (K + 7) // 8 * 8
        Mp, Np = (M + 63) // 64 * 64, (N + 63) // 64 * 64   # contraction-major operands keep an aligned row stride
        a = torch.randn((K, Mp) if ta else (M, Kp), device=dev, generator=g).to(torch.bfloat16)
        b = torch.randn((K, Np) if tb else (N, Kp), device=dev, generator=g).to(torch.bfloat16)
        if not ta and Kp != K:
            a[:, K:] = 0
        if not tb and Kp != K:
            b[:, K:] = 0
        ref = None
        for v in variants:
            ops.set_option("gemm", min(v % 10, 1))            # 0 = gemm.hip, 1 = gemm_pp256.hip as shipped,
            ops.set_option("gemm_k64", 0 if v % 10 == 2 else 1)  # 2 = gemm_pp256.hip with the K-step-32 loop everywhere
            ops.set_option("gemm_ablate", v // 10)  # e.g. 41 = variant 1 built without MFMA (see gemm_pp256.hip ABL)
            for sk in splitks:
                out = torch.empty((M, N), device=dev, dtype=torch.bfloat16)
                ops.gemm_nt(a, b, out, K=K, ta=bool(ta), tb=bool(tb), splitk=sk)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(iters):
                    ops.gemm_nt(a, b, out, K=K, ta=bool(ta), tb=bool(tb), splitk=sk)
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / iters
                tf = 2.0 * M * N * K / (ms * 1e-3) / 1e12
                err = ""
                if ref is None:
                    ref = out.float()
                else:
                    d = (out.float() - ref).abs().max().item()
                    err = f" maxdiff_vs_first {d:.3e} (ref max {ref.abs().max().item():.1f})"
                tot[(v, sk)] += ms * calls
                print(f"M={M:6d} N={N:5d} K={K:6d} ta={ta} tb={tb} variant={v} splitk={sk or 'auto'}: {ms * 1e3:9.1f} us {tf:7.1f} TF/s{err}",
                      flush=True)
        del a, b
    for k, v in tot.items():
        print(f"variant {k[0]} splitk {k[1] or 'auto'}: GEMM time per training step {v:.2f} ms")


if __name__ == "__main__":
    main()
