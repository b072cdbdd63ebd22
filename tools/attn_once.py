#!/usr/bin/env python
"""Event-level attention forward + backward at the step's shape (B=16, H=16, S=2048 and 4096, head_dim 64, bf16): time per call
(HIP events) and a digest of o / lse / dqkv, for A/B runs of two builds of the library (MH_LIB_PATH=... python tools/attn_once.py)."""
import hashlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from midi_model_amd import ops
from midi_model_amd.engine import RopeTable
B, H = 16, 16
D = H * 64
for S in (2048, 4096, 1000):
    g = torch.Generator(device="cuda").manual_seed(S)
    qkv = torch.randn((B * S, 3 * D), generator=g, device="cuda").to(torch.bfloat16)
    do = (torch.randn((B * S, D), generator=g, device="cuda") * 0.1).to(torch.bfloat16)
    Sp = (S + 63) // 64 * 64
    o = torch.empty((B * S, D), dtype=torch.bfloat16, device="cuda"); lse = torch.zeros(B * H * Sp, device="cuda")
    dq = torch.empty_like(qkv)
    rope = RopeTable(64, 10000.0, torch.device("cuda"), S)
    fwd = lambda: ops.attn_fwd(qkv, o, lse, B, S, H, 0.125)
    bwd = lambda: ops.attn_bwd(qkv, o, do, lse, dq, B, S, H, 0.125, rope.cos, rope.sin)
    t = {}
    for name, fn in (("fwd", fwd), ("bwd", bwd)):
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        t[name] = e0.elapsed_time(e1) / 10 * 1e3
    h = hashlib.sha256(o.cpu().view(torch.int16).numpy().tobytes() + lse.cpu().numpy().tobytes() + dq.cpu().view(torch.int16).numpy().tobytes()).hexdigest()[:16]
    print(f"S={S}: fwd {t['fwd']:7.1f} us   bwd {t['bwd']:7.1f} us   digest {h}", flush=True)
