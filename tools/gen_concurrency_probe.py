#!/usr/bin/env python
"""Probe: is a generate() of batch 64 faster as ONE chain of dependent launches, or as 2 / 4 independent chains of batch 32 / 16 that run
side by side on their own streams (host threads, one decode session each)?  A decode step is ~210 dependent launches of 4-10 us
whose fixed costs (ramp, tail) a second chain could fill.  Reports aggregate events/s."""
import os, sys, threading, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import midi_model_amd as mm

n_new = int(sys.argv[1]) if len(sys.argv) > 1 else 512
torch.manual_seed(0)
model = mm.MIDIModel(mm.MIDIModelConfig.from_name("tv2o-medium")).to("cuda", torch.bfloat16).eval()

def run(B, seed, stream):
    gen = torch.Generator(device="cuda"); gen.manual_seed(seed)
    with torch.cuda.stream(stream), torch.no_grad():
        out = model.generate(None, batch_size=B, max_len=1 + n_new, temp=1.0, top_p=0.98, top_k=20, generator=gen, ban_eos=True)
        stream.synchronize()
    return out

for nthr in (1, 2, 4, 1, 2):
    B = 64 // nthr
    streams = [torch.cuda.Stream() for _ in range(nthr)]
    def work(i):
        run(B, 7 + i, streams[i])
    for rep in range(3):  # rep 0, 1: warm-up (sessions, graphs)
        th = [threading.Thread(target=work, args=(i,)) for i in range(nthr)]
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for t in th: t.start()
        for t in th: t.join()
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{nthr} chain(s) x batch {B}: {64 * n_new / dt:9.0f} events/s  ({1e3 * dt / n_new:.3f} ms per event step)", flush=True)
