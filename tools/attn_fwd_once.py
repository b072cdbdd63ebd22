#!/usr/bin/env python
"""forward attention timing of ONE build (MH_LIB_PATH selects it): B=16, H=16, S in (2048, 4096), bf16, HIP events; prints us"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from midi_model_amd import ops  # noqa: E402

B, H = 16, 16
D = H * 64
for S in (2048, 4096):
    g = torch.Generator(device="cuda").manual_seed(0)
    qkv = torch.randn((B * S, 3 * D), device="cuda", generator=g).to(torch.bfloat16)
    do = torch.randn((B * S, D), device="cuda", generator=g).to(torch.bfloat16)
    o = torch.empty((B * S, D), device="cuda", dtype=torch.bfloat16)
    lse = torch.zeros(B * H * S, device="cuda")
    dqkv = torch.empty_like(qkv)
    res = []
    for fn in (lambda: ops.attn_fwd(qkv, o, lse, B, S, H, 0.125), lambda: ops.attn_bwd(qkv, o, do, lse, dqkv, B, S, H, 0.125)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) * 100)
    print(f"{os.environ.get('MH_LIB_PATH', 'tree')[-24:]:>24s} S={S} fwd {res[0]:7.1f} us  bwd {res[1]:7.1f} us", flush=True)
