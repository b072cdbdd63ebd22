#!/usr/bin/env python
"""rmsnorm forward / backward timing of ONE build (MH_LIB_PATH selects it), event-level and token-level row counts, bf16"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from midi_model_amd import ops  # noqa: E402
tag = os.environ.get("MH_LIB_PATH", "tree")[-22:]
res = []
for M in (32768, 262144):
    D = 1024
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn((M, D), device="cuda", generator=g).to(torch.bfloat16)
    dy = torch.randn((M, D), device="cuda", generator=g).to(torch.bfloat16)
    dres = torch.randn((M, D), device="cuda", generator=g).to(torch.bfloat16)
    w = torch.ones(D, device="cuda", dtype=torch.bfloat16)
    y = torch.empty_like(x); dx = torch.empty_like(x); rstd = torch.empty(M, device="cuda"); dw = torch.zeros(D, device="cuda", dtype=torch.bfloat16)
    for name, fn, nbytes in (("fwd", lambda: ops.rmsnorm_fwd(x, w, y, rstd, 1e-6), 2 * M * D * 2), ("bwd", lambda: ops.rmsnorm_bwd(x, w, rstd, dy, dres, dx, dw, False), 4 * M * D * 2)):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 50
        res.append(f"{name}[{M}] {us:7.1f} us {nbytes / us / 1e6:5.2f} TB/s")
print(f"{tag:>22s} " + " | ".join(res) + f" | chk {float(dx.float().abs().mean()):.6f}", flush=True)
