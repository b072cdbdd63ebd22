#!/usr/bin/env python
"""Same-box A/B of forms of the event-level attention FORWARD (mh_set_option("attn_v3") values given on the command line):
numerics against the default form on the same inputs (O max / rms difference, lse max difference) and interleaved timing with
HIP events at B=16, H=16, S in {2048, 4096}.  A form is "bits" or "bits:wps" (attn_v3_wps: register budget / stages variant).
Usage: python tools/attn_fwd_ab.py 127 255 127:4 255:4"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from midi_model_amd import ops  # noqa: E402

forms = sys.argv[1:] or ["127", "255"]


def select(f):  # "bits[:wps[:passes]]"
    parts = f.split(":") + ["", ""]
    ops.set_option("attn_v3", int(parts[0]))
    ops.set_option("attn_v3_wps", int(parts[1]) if parts[1] else 0)
    ops.set_option("attn_passes", int(parts[2]) if parts[2] else 5)


B, H = 16, 16
D = H * 64
for S in (2048, 4096, 1000):
    g = torch.Generator(device="cuda").manual_seed(S)
    qkv = torch.randn((B * S, 3 * D), generator=g, device="cuda", dtype=torch.float32).to(torch.bfloat16)
    Sp = (S + 63) // 64 * 64
    outs = {}
    for f in forms:
        select(f)
        o = torch.empty((B * S, D), dtype=torch.bfloat16, device="cuda")
        lse = torch.zeros(B * H * Sp, device="cuda")
        ops.attn_fwd(qkv, o, lse, B, S, H, 0.125)
        torch.cuda.synchronize()
        outs[f] = (o.float(), lse.view(B, H, Sp)[:, :, :S].clone())
    ref = outs[forms[0]]
    for f in forms[1:]:
        d = (outs[f][0] - ref[0]).abs()
        print(f"S={S} form {f} vs {forms[0]}: O max diff {d.max().item():.3e} rms {d.pow(2).mean().sqrt().item():.3e} "
              f"(|O| rms {ref[0].pow(2).mean().sqrt().item():.3e}); lse max diff {(outs[f][1] - ref[1]).abs().max().item():.3e}; "
              f"finite {bool(torch.isfinite(outs[f][0]).all())}")
    if S == 1000:
        continue
    times = {f: [] for f in forms}
    o = torch.empty((B * S, D), dtype=torch.bfloat16, device="cuda")
    lse = torch.zeros(B * H * Sp, device="cuda")
    for rep in range(12):
        for f in forms:
            select(f)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                ops.attn_fwd(qkv, o, lse, B, S, H, 0.125)
            e1.record()
            torch.cuda.synchronize()
            if rep >= 2:
                times[f].append(e0.elapsed_time(e1) / 5 * 1e3)
    fl = 4.0 * 64 * S * (S + 1) / 2 * B * H
    # the backward pair of the same form (same work order option), interleaved likewise
    do = torch.randn((B * S, D), generator=g, device="cuda", dtype=torch.float32).to(torch.bfloat16)
    dqkv = torch.empty_like(qkv)
    btimes = {f: [] for f in forms}
    for rep in range(8):
        for f in forms:
            select(f)
            ops.attn_fwd(qkv, o, lse, B, S, H, 0.125)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                ops.attn_bwd(qkv, o, do, lse, dqkv, B, S, H, 0.125)
            e1.record()
            torch.cuda.synchronize()
            if rep >= 2:
                btimes[f].append(e0.elapsed_time(e1) / 3 * 1e3)
    for f in forms:
        t, bt = sorted(times[f]), sorted(btimes[f])
        print(f"S={S} form {f}: fwd median {t[len(t) // 2]:.1f} us (min {t[0]:.1f}, max {t[-1]:.1f}) = {fl / (t[len(t) // 2] * 1e-6) / 1e12:.0f} TFLOP/s; "
              f"bwd median {bt[len(bt) // 2]:.1f} us (min {bt[0]:.1f})")
select("127")
