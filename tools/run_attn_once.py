#!/usr/bin/env python
"""A few launches of the event-level attention kernels at B=16, H=16, S=MH_BENCH_S (default 4096), bf16 -- the command the
PMC passes of tools/gpu_pmc.sh are run over (MH_ATTN_V3 / MH_ATTN_V3_WPS choose the form; MH_RUN_BWD=1 adds the backward)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from midi_model_amd import ops  # noqa: E402

B, H, S = 16, 16, int(os.environ.get("MH_BENCH_S", "4096"))
D = H * 64
g = torch.Generator(device="cuda").manual_seed(0)
qkv = torch.randn((B * S, 3 * D), device="cuda", generator=g).to(torch.bfloat16)
do = torch.randn((B * S, D), device="cuda", generator=g).to(torch.bfloat16)
o = torch.empty((B * S, D), device="cuda", dtype=torch.bfloat16)
lse = torch.zeros(B * H * ((S + 63) // 64 * 64), device="cuda")
dqkv = torch.empty_like(qkv)
for _ in range(3):
    ops.attn_fwd(qkv, o, lse, B, S, H, 0.125)
    if os.environ.get("MH_RUN_BWD") == "1":
        ops.attn_bwd(qkv, o, do, lse, dqkv, B, S, H, 0.125)
torch.cuda.synchronize()
