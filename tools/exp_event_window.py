#!/usr/bin/env python
"""What a HIP-event window around one launch contains when its neighbours carry no events (bench.py's gemm_profile):
rmsnorm -> [e0] q|k|v+RoPE [e1] -> attention -> [e0] o [e1], 12 times, M = 16 x 2048."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from midi_model_amd import ops  # noqa: E402
from midi_model_amd.engine import RopeTable  # noqa: E402


def main():
    dev, dt = "cuda", torch.bfloat16
    B, S, H = 16, 2048, 16
    D, M = H * 64, B * S
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn((M, D), device=dev, generator=g).to(dt)
    w = torch.ones(D, device=dev, dtype=dt)
    wqkv = (torch.randn((3 * D, D), device=dev, generator=g) * 0.02).to(dt)
    wo = (torch.randn((D, D), device=dev, generator=g) * 0.02).to(dt)
    h1 = torch.empty_like(x)
    rstd = torch.empty(M, device=dev)
    qkv = torch.empty((M, 3 * D), device=dev, dtype=dt)
    o = torch.empty_like(x)
    y = torch.empty_like(x)
    lse = torch.zeros(B * H * S, device=dev)
    rope = RopeTable(64, 10000.0, dev, S)
    tab = rope.fused()

    def run(mode):
        prof = []
        ops.gemm_profile = prof
        extra = []
        for _ in range(12):
            ops.rmsnorm_fwd(x, w, h1, rstd, 1e-5)
            if mode == "sync":
                torch.cuda.synchronize()
            ops.gemm_rope(h1, wqkv, qkv, tab, S, 0, 64)
            if mode == "event_after_attn":
                ea = torch.cuda.Event(enable_timing=True)
                ops.attn_fwd(qkv, o, lse, B, S, H, 0.125)
                ea.record()
                extra.append(ea)
            elif mode == "no_attn":
                pass
            else:
                ops.attn_fwd(qkv, o, lse, B, S, H, 0.125)
            ops.gemm_nt(o, wo, y)
        ops.gemm_profile = None
        torch.cuda.synchronize()
        by = {}
        for e0, e1, f, shp in prof:
            by.setdefault(shp, []).append(e0.elapsed_time(e1) * 1e3)
        for shp, v in by.items():
            print(f"{mode:18s} {str(shp):44s} n={len(v)} min {min(v):7.1f} med {sorted(v)[len(v) // 2]:7.1f} max {max(v):7.1f} us", flush=True)

    for mode in ("plain", "plain", "event_after_attn", "no_attn", "sync", "plain"):
        run(mode)
    ops.set_option("attn_v3", 0)
    run("plain(first form)")
    ops.set_option("attn_v3", 127)


if __name__ == "__main__":
    main()
