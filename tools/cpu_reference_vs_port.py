#!/usr/bin/env python
"""The CPU baseline of bench.py is the oracle (`cpu_baseline.kind = "port"`): the GPU box has no /root/reference.  This tool runs
in the BUILD container, where the reference is present, and times the REAL reference (midi_model.py driven through train.py's
step as tests/gen_golden.py drives it: train.py:168-188 loss, clip_grad_norm_(1.0), AdamW of train.py:121-151) beside the
oracle port on the same host cores, the same weights, the same batch and the same thread count -- so that the "port" numbers of
the bench lines can be read as the reference's (the two run the same torch CPU kernels; the ratio is reported).
Usage: python tools/cpu_reference_vs_port.py [S=2048] [gen_events=8]   (writes nothing; redirect into profiles/)"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gen_golden as G  # noqa: E402  (import_reference / load_oracle / build_ref)

S = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
GEN = int(sys.argv[2]) if len(sys.argv) > 2 else 8
cores = min(os.cpu_count() or 1, 32)
torch.set_num_threads(cores)
orc = G.load_oracle()
ref_model, ref_tok = G.import_reference()
tok = ref_tok.MIDITokenizerV2()
tok.set_optimise_midi(True)
shp = orc.Shape(vocab=tok.vocab_size)  # tv2o-medium
sd = orc.make_state_dict(shp, seed=0)
batch = orc.synthetic_events(tok, 1, S + 1, seed=0)
print(f"host: {os.cpu_count()} cores, {cores} torch threads; torch {torch.__version__}; tv2o-medium fp32; batch 1 x {S} events")


def port_step():
    p = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    m = {k: torch.zeros_like(v) for k, v in p.items()}
    v2 = {k: torch.zeros_like(v) for k, v in p.items()}
    t0 = time.perf_counter()
    loss, _ = orc.training_loss(p, shp, batch)
    loss.backward()
    coef, _ = orc.clip_coef([q.grad for q in p.values()], 1.0)
    with torch.no_grad():
        for k, q in p.items():
            orc.adamw_step(q, q.grad * coef, m[k], v2[k], 1, 2e-4, 0.01 if orc.decays(k) else 0.0)
    return time.perf_counter() - t0, float(loss.detach())


def reference_step():
    model = G.build_ref(ref_model, shp, sd)
    model.train()
    # train.py:121-151 (configure_optimizers): AdamW, betas (0.9, 0.99), eps 1e-8, no decay on names with "bias" / "norm"
    decay = [q for n, q in model.named_parameters() if not any(s in n for s in ("bias", "norm"))]
    no_decay = [q for n, q in model.named_parameters() if any(s in n for s in ("bias", "norm"))]
    opt = torch.optim.AdamW([{"params": decay, "weight_decay": 0.01}, {"params": no_decay, "weight_decay": 0.0}], lr=2e-4,
                            betas=(0.9, 0.99), eps=1e-8)
    t0 = time.perf_counter()
    x, y = batch[:, :-1].contiguous(), batch[:, 1:].contiguous()  # train.py:168-188
    hidden = model.forward(x)
    hidden = hidden.reshape(-1, hidden.shape[-1])
    y = y.reshape(-1, y.shape[-1])
    logits = model.forward_token(hidden, y[:, :-1])
    loss = torch.nn.functional.cross_entropy(logits.view(-1, tok.vocab_size), y.view(-1), reduction="mean", ignore_index=tok.pad_id)
    loss.backward()
    torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
    opt.step()
    return time.perf_counter() - t0, float(loss.detach())


def port_fused():
    orc.FUSED_SDPA = True  # what bench.py's cpu_baseline times: attention through torch SDPA, as the reference
    try:
        return port_step()
    finally:
        orc.FUSED_SDPA = False


for rep in range(3):
    for name, fn in (("oracle, explicit attention  ", port_step), ("oracle, torch SDPA (bench.py)", port_fused), ("REAL reference               ", reference_step)):
        dt, loss = fn()
        print(f"training step  {name}: {dt:6.1f} s -> {S / dt:7.1f} events/s   (loss {loss:.6f})", flush=True)

with torch.no_grad():
    model = G.build_ref(ref_model, shp, sd).eval()
    for rep in range(2):
        t0 = time.perf_counter()
        out_p = orc.generate(sd, shp, tok, None, batch_size=64, max_len=1 + GEN, generator=torch.Generator().manual_seed(0), ban_eos=True)
        dp = time.perf_counter() - t0
        t0 = time.perf_counter()
        out_r = model.generate(None, batch_size=64, max_len=1 + GEN, generator=torch.Generator().manual_seed(0))  # (tqdm bar -> stderr)
        dr = time.perf_counter() - t0
        n_r = out_r.shape[1] - 1
        print(f"generate b=64  oracle port   : {dp:6.1f} s -> {64 * GEN / dp:7.1f} events/s ({GEN} new events, EOS masked)\n"
              f"generate b=64  REAL reference: {dr:6.1f} s -> {64 * n_r / dr:7.1f} events/s ({n_r} new events before every row ended)", flush=True)
