#!/usr/bin/env python
"""Probe: what does a kernel boundary cost on a plain stream against inside a captured hipGraph?  Chains of N dependent launches of the
step's own kernels (a [32768 x 1024 x 1024] projection, the folded norm backward on 32768 rows), timed as a whole with HIP events;
the per-launch kernel time comes from a chain that is long enough to amortise the ends."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from midi_model_amd import ops
g = torch.Generator(device="cuda").manual_seed(0)
mk = lambda *s: (torch.randn(s, device="cuda", generator=g) * 0.1).to(torch.bfloat16)
M, D = 32768, 1024
x = mk(M, D); w = mk(D, D); y = torch.empty_like(x); y2 = torch.empty_like(x)
big = mk(3 * D, D); yq = torch.empty((M, 3 * D), device="cuda", dtype=torch.bfloat16)
rstd = torch.rand((M,), device="cuda") + 0.5
def chain(n):
    for i in range(n // 2):
        ops.gemm_nt(x, w, y)            # ~70 us
        ops.rmsnorm_bwd_folded(x, rstd, y, None, y2)   # ~45 us
def chain_big(n):
    for i in range(n):
        ops.gemm_nt(x, big, yq)         # ~190 us
def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[len(ts) // 2]
for name, body, n in (("gemm 1024 + norm backward", chain, 200), ("gemm 3072", chain_big, 100)):
    t_stream = timed(lambda: body(n))
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        body(4); torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s):
            body(n)
    t_graph = timed(lambda: gr.replay())
    print(f"{name}: {n} launches   stream {t_stream / n:7.2f} us per launch   graph {t_graph / n:7.2f} us per launch   "
          f"difference {(t_stream - t_graph) / n:5.2f} us per boundary", flush=True)
