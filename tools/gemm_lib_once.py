#!/usr/bin/env python
"""projection GEMM timing of ONE build (MH_LIB_PATH selects it; the product library, not the A/B one): the forward (row-major x
row-major) and dgrad (row-major x contraction-major) shapes of the tv2o-medium step + the fused epilogues, bf16, HIP events, random
data.  For interleaved same-box A/B runs of two libraries (tools/gpu_r05_gemm_sbase_ab.sh)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from midi_model_amd import ops  # noqa: E402
from midi_model_amd.engine import RopeTable  # noqa: E402

SHAPES = [(32768, 1024, 1024, 0), (32768, 3072, 1024, 0), (32768, 8192, 1024, 0), (32768, 1024, 4096, 0), (32768, 1024, 16384, 0),
          (262144, 1024, 1024, 0), (32768, 3406, 1024, 0),
          (32768, 1024, 3072, 1), (32768, 1024, 8192, 1), (32768, 4096, 1024, 1), (262144, 1024, 2048, 1)]
g = torch.Generator(device="cuda").manual_seed(0)
tag = os.environ.get("MH_LIB_PATH", "tree")[-22:]
out_line, chk = [], 0.0


def timeit(fn, flops):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(6):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return flops * 6 / (e0.elapsed_time(e1) * 1e-3) / 1e12


for (M, N, K, tb) in SHAPES:
    a = torch.randn((M, K), device="cuda", generator=g).to(torch.bfloat16)
    b = (torch.randn((K, (N + 63) // 64 * 64) if tb else (N, K), device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    c = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
    tf = timeit(lambda: ops.gemm_nt(a, b, c, K=K, tb=bool(tb)), 2.0 * M * N * K)
    chk += float(c.float().abs().mean())
    out_line.append(f"{'dgrad' if tb else 'fwd'}[{M}x{N}x{K}] {tf:6.0f}")
# fused epilogues (forward-only block shapes)
M, D, I, S = 65536, 1024, 4096, 4096
x = torch.randn((M, D), device="cuda", generator=g).to(torch.bfloat16)
wq = (torch.randn((3 * D, D), device="cuda", generator=g) * 0.05).to(torch.bfloat16)
wg = (torch.randn((2 * I, D), device="cuda", generator=g) * 0.05).to(torch.bfloat16)
qkv = torch.empty((M, 3 * D), device="cuda", dtype=torch.bfloat16)
act = torch.empty((M, I), device="cuda", dtype=torch.bfloat16)
rope = RopeTable(64, 10000.0, torch.device("cuda"), S)
out_line.append(f"rope[{M}x{3 * D}x{D}] {timeit(lambda: ops.gemm_rope(x, wq, qkv, rope.fused(), S, 0, 64), 2.0 * M * 3 * D * D):6.0f}")
out_line.append(f"swiglu[{M}x{2 * I}x{D}] {timeit(lambda: ops.gemm_swiglu(x, wg, None, act), 2.0 * M * 2 * I * D):6.0f}")
chk += float(qkv.float().abs().mean()) + float(act.float().abs().mean())
# r06: the row-scaled forms of the folded block, the training forms (gate|up stored; SwiGLU backward), the statistics producer
rs = (0.5 + torch.rand((M,), device="cuda", generator=g)).float()
out_line.append(f"rope_scaled {timeit(lambda: ops.gemm_rope(x, wq, qkv, rope.fused(), S, 0, 64, rowscale=rs), 2.0 * M * 3 * D * D):6.0f}")
out_line.append(f"swiglu_scaled {timeit(lambda: ops.gemm_swiglu(x, wg, None, act, rowscale=rs), 2.0 * M * 2 * I * D):6.0f}")
chk += float(qkv.float().abs().mean()) + float(act.float().abs().mean())
Mt = 32768
xt = x[:Mt]
gu = torch.empty((Mt, 2 * I), device="cuda", dtype=torch.bfloat16)
out_line.append(f"swiglu_train[{Mt}x{2 * I}x{D}] {timeit(lambda: ops.gemm_swiglu(xt, wg, gu, act[:Mt]), 2.0 * Mt * 2 * I * D):6.0f}")
wd = (torch.randn((D, I), device="cuda", generator=g) * 0.05).to(torch.bfloat16)
dgu = torch.empty_like(gu)
out_line.append(f"dswiglu[{Mt}x{I}x{D}] {timeit(lambda: ops.gemm_dswiglu(xt, wd, gu, dgu), 2.0 * Mt * I * D):6.0f}")
chk += float(gu.float().abs().mean()) + float(dgu.float().abs().mean())
parts = torch.empty((D // 64, M), device="cuda")
wo = (torch.randn((D, D), device="cuda", generator=g) * 0.05).to(torch.bfloat16)
x2 = torch.empty((M, D), device="cuda", dtype=torch.bfloat16)
out_line.append(f"rowss[{M}x{D}x{D}] {timeit(lambda: ops.gemm_rowss(x, wo, x2, parts, res=x), 2.0 * M * D * D):6.0f}")
wdn = (torch.randn((D, I), device="cuda", generator=g) * 0.05).to(torch.bfloat16)
out_line.append(f"rowss[{M}x{D}x{I}] {timeit(lambda: ops.gemm_rowss(act, wdn, x2, parts, res=x), 2.0 * M * D * I):6.0f}")
chk += float(x2.float().abs().mean()) + float(parts.mean())
print(f"{tag:>22s} TF: " + " | ".join(out_line) + f" | checksum {chk:.6f}", flush=True)
