"""Run-to-run determinism of the training step (same weights, same batch -> the same gradient bits) under kernel-form options,
to locate a racing kernel: prints, per option set, how many gradient elements differ between repeated steps and in which tensors."""
import os, sys, itertools
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import midi_model_amd as mm
from midi_model_amd import ops
from midi_model_amd.train import TrainMIDIModel
from midi_model_amd.data import synthetic_events

torch.manual_seed(0)
model = TrainMIDIModel(mm.MIDIModelConfig.from_name("tv2o-medium"), accumulate_grad_batches=1).to("cuda", torch.bfloat16)
B, S = int(os.environ.get("B", 8)), 2048
batch = synthetic_events(model.tokenizer, B, S + 1, seed=3, device="cuda")
names = [n for n, _ in model.named_parameters()]

def grads():
    model.zero_grad()
    loss = model.training_step(batch)
    torch.cuda.synchronize()
    return loss.item(), model.grad_buffer().clone()

sets = [dict(), dict(gemm_lean_epi=0), dict(gemm_k64=0), dict(gemm_lean_epi=0, gemm_k64=0), dict(attn_passes=1), dict(attn_v3=127),
        dict(gemm_lean_epi=0, gemm_k64=0, attn_passes=1, attn_v3=127)]
defaults = dict(gemm_lean_epi=1, gemm_k64=1, attn_passes=5, attn_v3=255)
for st in sets:
    for k, v in defaults.items():
        ops.set_option(k, st.get(k, v))
    l0, g0 = grads()
    worst = 0
    bad_names = {}
    for rep in range(int(os.environ.get("REPS", 6))):
        l1, g1 = grads()
        d = (g0 != g1)
        nd = int(d.sum())
        worst = max(worst, nd)
        if nd:
            for n in names:
                off, cnt, _ = model._offsets[n]
                c = int(d[off:off + cnt].sum())
                if c:
                    bad_names[n] = max(bad_names.get(n, 0), c)
    top = sorted(bad_names.items(), key=lambda kv: -kv[1])[:8]
    print(f"{st or 'defaults'}: loss {l0:.6f}; max differing gradient elements between repeats {worst}; tensors: {top}", flush=True)
for k, v in defaults.items():
    ops.set_option(k, v)
