"""Golden vectors of the REAL reference for BASELINE.json configs[4] AS WORDED ("2x hidden"): ``MIDIModelConfig.get_config("v2",
True, 4, 32, 2048, 8192)`` (midi_model.py:63-76: D = 2048, 32 heads of 64, MLP 8192; token-level net 1 layer, 8 heads of 256, MLP
2048 -- 4 + 1 layers keep the real reference at a few minutes on 8 cores; the layer count changes no kernel shape), batch 1,
S = 4096 events, the full training step of train.py:168-188 with its backward -- in fp32 and in bf16 (the reference's own
``bf16-true`` drift on the same weights / inputs).  Runs ``/root/reference/midi_model.py`` on CPU (this container only) and
commits ``tests/golden/large2x_S4096.npz``.  tests/test_parity_long_gpu.py::test_two_times_hidden_step_at_S4096 bounds the
production kernels -- 32-head attention at S = 4096, K = 8192 contractions, the ``lean_activations`` form bench.py runs at this
shape -- by multiples of the reference's own bf16 errors.

Usage:  python tests/gen_golden_large2x.py      (a few minutes on 8 cores, ~12 GB of host memory)
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from gen_golden import OUT, import_reference, load_oracle, ref_train_loss  # noqa: E402
from gen_golden_long import slice_of  # noqa: E402

S, BATCH_SEED, WEIGHT_SEED = 4096, 23, 5
GRAD_SLICES = ("net.embed_tokens.weight", "net.layers.0.self_attn.q_proj.weight", "net.layers.2.self_attn.v_proj.weight",
               "net.layers.3.mlp.down_proj.weight", "net.layers.1.mlp.gate_proj.weight", "net_token.embed_tokens.weight",
               "net_token.layers.0.self_attn.o_proj.weight", "net_token.layers.0.mlp.up_proj.weight", "lm_head.weight",
               "net.layers.2.input_layernorm.weight", "net.norm.weight", "net_token.layers.0.post_attention_layernorm.weight")


def main():
    ref_model, ref_tok = import_reference()
    orc = load_oracle()
    torch.set_num_threads(os.cpu_count())
    tok = ref_tok.MIDITokenizer("v2")
    shp = orc.Shape(n_layer=4, n_head=32, n_embd=2048, n_inner=8192, vocab=tok.vocab_size)
    sd = orc.make_state_dict(shp, seed=WEIGHT_SEED)
    cfg = ref_model.MIDIModelConfig.get_config("v2", True, 4, 32, 2048, 8192)
    batch = orc.synthetic_events(tok, 1, S + 1, seed=BATCH_SEED)
    g = {"S": np.int64(S), "batch_seed": np.int64(BATCH_SEED), "weight_seed": np.int64(WEIGHT_SEED)}
    t0 = time.time()
    model = ref_model.MIDIModel(cfg)
    model.load_state_dict(sd, strict=True)
    model.eval()
    loss, logits, hidden = ref_train_loss(model, batch)
    loss.backward()
    named = dict(model.named_parameters())
    g["grad_names"] = np.array(list(named.keys()))
    g["grad_norms"] = np.array([p.grad.norm().item() for p in named.values()], dtype=np.float64)
    for k in GRAD_SLICES:
        g["grad:" + k] = slice_of(named[k].grad)
    grads32 = {k: named[k].grad.detach().clone() for k in GRAD_SLICES}
    flat32 = torch.cat([p.grad.reshape(-1) for p in named.values()])
    loss, logits, hidden = loss.detach(), logits.detach(), hidden.detach()
    del model, named
    g["loss"] = np.float64(loss.item())
    g["hidden_sub"] = hidden[::32, ::8].numpy().copy()
    g["logits_sub"] = logits[::64, :, ::16].numpy().copy()
    g["logits_lse"] = torch.logsumexp(logits, -1).numpy()
    g["logits_argmax"] = logits.argmax(-1).numpy()
    top2 = logits.topk(2, -1).values
    g["logits_margin"] = (top2[..., 0] - top2[..., 1]).numpy()
    print(f"fp32 reference done in {time.time() - t0:.0f} s, loss {loss.item():.6f}", flush=True)

    t0 = time.time()
    mb = ref_model.MIDIModel(cfg)
    mb.load_state_dict(sd, strict=True)
    mb = mb.to(torch.bfloat16).eval()
    lb, lgb, hb = ref_train_loss(mb, batch)
    lb.backward()
    nb = dict(mb.named_parameters())
    for k in GRAD_SLICES:
        d = nb[k].grad.float() - grads32[k]
        g["ref_bf16_grad_relerr:" + k] = np.float64((d.norm() / grads32[k].norm()).item())
    flatb = torch.cat([p.grad.float().reshape(-1) for p in nb.values()])
    fb, f32_ = flatb.double(), flat32.double()
    g["ref_bf16_grad_cosine"] = np.float64((torch.dot(fb, f32_) / (fb.norm() * f32_.norm())).item())
    g["ref_bf16_grad_norm_ratio"] = np.float64((fb.norm() / f32_.norm()).item())
    lb, lgb, hb = lb.detach(), lgb.detach(), hb.detach()
    dh, dl = hb.float() - hidden, lgb.float() - logits
    g["ref_bf16_loss"] = np.float64(lb.item())
    g["ref_bf16_hidden_maxerr"] = np.float64(dh.abs().max().item())
    g["ref_bf16_hidden_rmserr"] = np.float64(dh.pow(2).mean().sqrt().item())
    g["ref_bf16_logits_maxerr"] = np.float64(dl.abs().max().item())
    g["ref_bf16_logits_rmserr"] = np.float64(dl.pow(2).mean().sqrt().item())
    g["ref_bf16_lse_maxerr"] = np.float64((torch.logsumexp(lgb.float(), -1) - torch.logsumexp(logits, -1)).abs().max().item())
    g["ref_bf16_argmax_agree"] = np.float64((lgb.argmax(-1) == logits.argmax(-1)).float().mean().item())
    print(f"bf16 reference done in {time.time() - t0:.0f} s: loss {lb.item():.4f}, hidden max/rms "
          f"{g['ref_bf16_hidden_maxerr']:.4f}/{g['ref_bf16_hidden_rmserr']:.5f}, logits max/rms {g['ref_bf16_logits_maxerr']:.4f}/"
          f"{g['ref_bf16_logits_rmserr']:.5f}, grad cosine {g['ref_bf16_grad_cosine']:.6f}, norm ratio {g['ref_bf16_grad_norm_ratio']:.4f}",
          flush=True)
    path = os.path.join(OUT, "large2x_S4096.npz")
    np.savez_compressed(path, **g)
    print("wrote", path, os.path.getsize(path), flush=True)


if __name__ == "__main__":
    main()
