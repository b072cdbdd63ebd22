#!/usr/bin/env python
"""One training step + a short generation on the BASELINE-worded "2x hidden" shape (24 layers, D=2048, 32 heads, I=8192;
SURVEY.md 8(d) config 5) to make sure nothing in the path assumes D = 1024.  Prints loss and throughput."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import midi_model_amd as mm  # noqa: E402
from midi_model_amd.data import synthetic_events  # noqa: E402
from midi_model_amd.train import TrainMIDIModel  # noqa: E402

cfg = mm.MIDIModelConfig.get_config("v2", True, 24, 32, 2048, 8192)
torch.manual_seed(0)
model = TrainMIDIModel(cfg, accumulate_grad_batches=1).to("cuda", torch.bfloat16)
model.configure_optimizers()
print("params", sum(p.numel() for p in model.parameters()) / 1e6, "M")
B, S = 4, 2048
batch = synthetic_events(model.tokenizer, B, S + 1, seed=1, device="cuda")
for i in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loss = model.training_step(batch)
    model.optimizer_step()
    torch.cuda.synchronize()
    print(f"step {i}: loss {float(loss):.4f}  {B * S / (time.perf_counter() - t0):.0f} events/s")
model.eval()
gen = torch.Generator(device="cuda").manual_seed(0)
t0 = time.perf_counter()
out = model.generate(None, batch_size=8, max_len=65, generator=gen, ban_eos=True)
torch.cuda.synchronize()
print("generate", out.shape, f"{8 * 64 / (time.perf_counter() - t0):.0f} events/s (incl. graph capture)")
assert all(model.tokenizer.tokens2event(r.tolist()) != [] for b in range(8) for r in out[b, 1:])
print("ok")
