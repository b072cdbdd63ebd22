#!/usr/bin/env python
"""Same-process A/B of mh_set_option("tokattn_bwd_batched", v) on the step's token-level attention backward (262144 octets x 4
heads of 256, bf16, RoPE on the way, with and without the folded norm's row scale): time per launch (HIP events, interleaved),
largest difference between the two forms against the gradient's range, both against an fp32 torch reference on a slice."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from midi_model_amd import ops
from midi_model_amd.engine import RopeTable
g = torch.Generator(device="cuda").manual_seed(0)
N, T, H, hd = (int(sys.argv[1]) if len(sys.argv) > 1 else 32768), 8, 4, 256
D = H * hd
qkv = (torch.randn((N * T, 3 * D), device="cuda", generator=g)).to(torch.bfloat16)
dout = (torch.randn((N * T, D), device="cuda", generator=g) * 0.1).to(torch.bfloat16)
rope = RopeTable(hd, 10000.0, torch.device("cuda"), T)
rs = (0.5 + torch.rand((N * T,), device="cuda", generator=g)).float()
scale = hd ** -0.5
for rowscale in (None, rs):
    outs, ts = {}, {0: [], 1: []}
    for rep in range(6):
        for v in (0, 1):
            ops.set_option("tokattn_bwd_batched", v)
            dq = torch.empty_like(qkv)
            ops.tokattn_bwd(qkv, dout, dq, N, T, H, scale, rope.cos, rope.sin, rowscale=rowscale)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                ops.tokattn_bwd(qkv, dout, dq, N, T, H, scale, rope.cos, rope.sin, rowscale=rowscale)
            e1.record(); torch.cuda.synchronize()
            outs[v] = dq
            if rep >= 1:
                ts[v].append(e0.elapsed_time(e1) / 3 * 1e3)
    med = {v: sorted(ts[v])[len(ts[v]) // 2] for v in ts}
    d = (outs[0].float() - outs[1].float()).abs()
    nbytes = qkv.numel() * 2 * 2 + dout.numel() * 2
    print(f"rowscale={'yes' if rowscale is not None else 'no '}: single {med[0]:7.1f} us ({nbytes / med[0] * 1e-6:.2f} TB/s)   batched {med[1]:7.1f} us "
          f"({nbytes / med[1] * 1e-6:.2f} TB/s)   elements that differ {int((d > 0).sum())} of {d.numel()}, max |diff| {d.max().item():.3e} "
          f"(gradient amax {outs[0].float().abs().max().item():.3e})", flush=True)
ops.set_option("tokattn_bwd_batched", 1)
