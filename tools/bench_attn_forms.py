#!/usr/bin/env python
"""A/B of the event-level attention kernel forms on one box, interleaved: first form (attention_mfma.hip) against the third
form (attention_mfma3.hip) per kernel -- mh_set_option("attn_v3", bits): 1 forward, 2 dQ, 4 dK/dV -- and its register
budgets ("attn_v3_wps").  tv2o-medium shapes: B=16, H=16, head_dim 64, S = 2048 and 4096, bf16, HIP events.
The backward numbers include the prep pass (transposed copies + delta), the same for every form."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from midi_model_amd import ops  # noqa: E402

_AB = ops.ab_library()  # a measurement tool: the compared kernel forms live in libmidihip_ab.so (build.py, -DMH_AB_BUILDS)
_AB.__enter__()


def timeit(fn, iters):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    iters = int(os.environ.get("MH_BENCH_ITERS", "10"))
    dev, dt = "cuda", torch.bfloat16
    g = torch.Generator(device=dev).manual_seed(0)
    B, H = 16, 16
    D = H * 64
    for S in (2048, 4096):
        qkv = torch.randn((B * S, 3 * D), device=dev, generator=g).to(dt)
        do = torch.randn((B * S, D), device=dev, generator=g).to(dt)
        o = torch.empty((B * S, D), device=dev, dtype=dt)
        Sp = (S + 63) // 64 * 64
        lse = torch.zeros(B * H * Sp, device=dev)
        dqkv = torch.empty_like(qkv)
        fl = 4.0 * B * H * S * (S + 1) / 2 * 64  # QK^T + PV on the causal triangle
        ref = {}
        for rnd in range(2):
            for name, v3, wps in (("first form", 0, 0), ("v3 fwd (3 waves/SIMD)", 1, 3), ("v3 fwd (2 waves/SIMD)", 1, 2),
                                  ("v3 fwd tr reads, no V^T copy (3)", 17, 3), ("v3 fwd tr reads, no V^T copy (2)", 17, 2),
                                  ("v3 fwd tr reads, three stages (3)", 81, 3), ("v3 fwd tr reads, three stages (2)", 81, 2)):
                ops.set_option("attn_v3", v3)
                ops.set_option("attn_v3_wps", wps)
                us = timeit(lambda: ops.attn_fwd(qkv, o, lse, B, S, H, 0.125), iters)
                print(f"S={S} fwd  {name:28s}: {us:8.1f} us  {fl / us / 1e6:7.1f} TF/s (V^T prep included)", flush=True)
            for name, v3, wps in (("first form", 0, 0), ("v3 dQ narrow (3 w/SIMD)", 2, 3), ("v3 dQ wide (2 w/SIMD)", 2, 2),
                                  ("v3 dK/dV", 4, 0), ("v3 dQ wide + dK/dV", 6, 2), ("v3 dQ narrow + dK/dV", 6, 3),
                                  ("v3 tr reads, no copies", 14, 3), ("v3 tr reads (dQ 2 w/SIMD)", 14, 2),
                                  ("v3 tr reads, delta inside dQ", 46, 3)):
                ops.set_option("attn_v3", v3)
                ops.set_option("attn_v3_wps", wps)
                us = timeit(lambda: ops.attn_bwd(qkv, o, do, lse, dqkv, B, S, H, 0.125), iters)
                if v3 == 0:
                    ref[rnd] = us
                print(f"S={S} bwd  {name:28s}: {us:8.1f} us  {2.5 * fl / us / 1e6:7.1f} TF/s  ({us - ref[rnd]:+8.1f} us vs first form; prep included)",
                      flush=True)
        # the prep passes alone
        vt = torch.empty((B * H * 64 * Sp,), dtype=dt, device=dev)
        from midi_model_amd.lib import lib
        st = torch.cuda.current_stream().cuda_stream
        us = timeit(lambda: lib().call("mh_attn_prep_fwd", qkv.data_ptr(), vt.data_ptr(), B, S, H, 1, st), iters)
        print(f"S={S} prep_fwd alone: {us:8.1f} us")
        buf = torch.empty((3, B * H * 64 * Sp), dtype=dt, device=dev)
        delta = torch.empty((B * H * Sp,), dtype=torch.float32, device=dev)
        us = timeit(lambda: lib().call("mh_attn_prep_bwd", qkv.data_ptr(), o.data_ptr(), do.data_ptr(), delta.data_ptr(), buf[0].data_ptr(),
                                       buf[1].data_ptr(), buf[2].data_ptr(), B, S, H, 1, st), iters)
        print(f"S={S} prep_bwd alone: {us:8.1f} us")
    ops.set_option("attn_v3", 127)
    ops.set_option("attn_v3_wps", 0)


if __name__ == "__main__":
    main()
