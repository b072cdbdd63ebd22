#!/usr/bin/env python
"""Experiment: the training micro-batch (forward + backward) replayed from one hipGraph against the eager launches
(tv2o-medium, bf16, B=16 x S=2048, one GPU; the optimiser step stays eager: its lr / bias corrections are launch arguments)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import midi_model_amd as mm  # noqa: E402
from midi_model_amd.data import synthetic_events  # noqa: E402
from midi_model_amd.train import TrainMIDIModel  # noqa: E402

torch.manual_seed(0)
cfg = mm.MIDIModelConfig.from_name("tv2o-medium")
model = TrainMIDIModel(cfg, lr=2e-4, weight_decay=0.01, warmup=1e3, max_step=1e6, accumulate_grad_batches=1)
model = model.to(torch.device("cuda", 0), torch.bfloat16)
model.configure_optimizers()
B, S = 16, 2048
batches = [synthetic_events(model.tokenizer, B, S + 1, seed=1000 + i, device="cuda") for i in range(2)]


def timed(fn, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        out = fn(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, out


for i in range(3):
    model.fit_step(batches[i % 2])
ms, loss = timed(lambda i: model.fit_step(batches[i % 2]), 8)
print(f"eager: {ms:.2f} ms/step  {B * S / ms * 1e3:.0f} events/s  loss {float(loss):.4f}", flush=True)

static = batches[0].clone()
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):  # warm-up on the capture stream (first-use allocations, attributes)
    model.training_step(static)
    model.optimizer_step()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    gloss = model.training_step(static)
model._micro = 0
torch.cuda.synchronize()


def gstep(i):
    static.copy_(batches[i % 2])
    g.replay()
    model._micro = 1
    model.optimizer_step()
    return gloss


for i in range(2):
    gstep(i)
ms2, loss2 = timed(gstep, 8)
print(f"graph: {ms2:.2f} ms/step  {B * S / ms2 * 1e3:.0f} events/s  loss {float(loss2):.4f}   ({(ms / ms2 - 1) * 100:+.2f} %)", flush=True)
ms3, loss3 = timed(lambda i: model.fit_step(batches[i % 2]), 8)
print(f"eager again: {ms3:.2f} ms/step", flush=True)
