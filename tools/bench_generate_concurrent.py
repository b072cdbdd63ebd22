#!/usr/bin/env python
"""Aggregate generation throughput of concurrent generate() calls (threads x streams, one decode session each) next to a
single batch-64 call: the serving situation of app.py (up to 10 generators on one model, app.py:496).  r01: one call of
64 sequences 29.2 k events/s; two concurrent calls of 64 sequences 44.7 k events/s aggregate (a decode step is bound by
dependent-launch latency, so a second stream fills the gaps)."""
import sys, os, time, threading
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import midi_model_amd as mm
torch.manual_seed(0)
model = mm.MIDIModel(mm.MIDIModelConfig.from_name("tv2o-medium")).to("cuda", torch.bfloat16).eval()
N = 256
def run(B, seed, stream=None, out=None):
    ctx = torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream())
    with ctx:
        g = torch.Generator(device="cuda").manual_seed(seed)
        o = model.generate(None, batch_size=B, max_len=1 + N, generator=g, ban_eos=True)
        torch.cuda.current_stream().synchronize()
    if out is not None: out.append(o.shape)
# single
run(64, 1); torch.cuda.synchronize()
t0 = time.perf_counter(); run(64, 2); torch.cuda.synchronize(); t1 = time.perf_counter()
print("single B=64:", 64 * N / (t1 - t0), "events/s")
for nthr, B in ((2, 32), (4, 16), (2, 64)):
    streams = [torch.cuda.Stream() for _ in range(nthr)]
    # warm (sessions per stream captured)
    ths = [threading.Thread(target=run, args=(B, 10 + i, streams[i])) for i in range(nthr)]
    [t.start() for t in ths]; [t.join() for t in ths]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ths = [threading.Thread(target=run, args=(B, 20 + i, streams[i])) for i in range(nthr)]
    [t.start() for t in ths]; [t.join() for t in ths]
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"{nthr} threads x B={B}:", nthr * B * N / (t1 - t0), "events/s aggregate")
