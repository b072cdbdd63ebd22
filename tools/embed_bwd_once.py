#!/usr/bin/env python
"""mh_embed_segment_bwd on the training step's two calls (token-level: 32768 x 7 occurrences reading their own rows of d seq;
event-level: 32768 x 8 occurrences, each reading its event's row): time per launch (HIP events) and the result against a
torch index_add in fp32."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import midi_model_amd as mm
from midi_model_amd import ops
from midi_model_amd.data import synthetic_events
tok = mm.MIDITokenizerV2()
B, S, T, D, V = 16, 2048, 8, 1024, tok.vocab_size
batch = synthetic_events(tok, B, S + 1, seed=1000, device=torch.device("cuda"))
x, y = batch[:, :-1].contiguous(), batch[:, 1:].contiguous()
M = B * S
g = torch.Generator(device="cuda").manual_seed(0)
dseq = (torch.randn((M * T, D), device="cuda", generator=g) * 0.01).to(torch.bfloat16)
dx = (torch.randn((M, D), device="cuda", generator=g) * 0.01).to(torch.bfloat16)
y_t = y.view(M, T)
for name, ids, rows, dout, kw in (("token-level", y_t[:, : T - 1], None, dseq, dict(row_mul=T, col_mul=1, add=1)),
                                  ("event-level", x.view(-1, T), None, dx, dict(row_mul=1, col_mul=0, add=0))):
    src, seg = ops.token_segments(ids, V, **kw)
    acc = torch.zeros((V, D), dtype=torch.float32, device="cuda")
    ops.embed_segment_bwd(src, seg, dout, D, acc, tok.pad_id)
    idv = ids.reshape(-1)
    rr = (torch.arange(idv.numel(), device="cuda") // ids.shape[1]) * kw["row_mul"] + (torch.arange(idv.numel(), device="cuda") % ids.shape[1]) * kw["col_mul"] + kw["add"]
    ref = torch.zeros((V, D), dtype=torch.float32, device="cuda").index_add_(0, idv, dout[rr].float())
    ref[tok.pad_id] = 0
    err = (acc - ref).abs().max().item() / ref.abs().max().item()
    for _ in range(3):
        ops.embed_segment_bwd(src, seg, dout, D, acc, tok.pad_id)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    pid = tok.pad_id
    e0.record()
    for _ in range(20):
        ops.embed_segment_bwd(src, seg, dout, D, acc, pid)
    e1.record(); torch.cuda.synchronize()
    print(f"{name}: {idv.numel()} occurrences, {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us per launch, rel err vs index_add {err:.2e}", flush=True)
