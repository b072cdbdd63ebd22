"""Golden vectors of the REAL reference at the BENCHMARKED sequence lengths: tv2o-medium, batch 1, S = 2048 and 4096 events
(BASELINE.json configs[1] / configs[4] lengths).  Runs ``/root/reference/midi_model.py`` on CPU (this container only; the
GPU box has no /root/reference) and commits ``tests/golden/medium_long_S{2048,4096}.npz``:

  fp32 reference   loss, sub-sampled hidden states / logits, log-sum-exp + arg-max + top-2 margin of every logits row, and
                   (S = 2048) the norm of all 140 parameter gradients + slices of eight named ones (train.py:168-188 + backward)
  bf16 reference   the reference's OWN drift when it runs in bf16 (``precision="bf16-true"``, train.py:365-371) against its
                   fp32 run on the same inputs: max / rms error of hidden states and logits, arg-max agreement, loss, and
                   (S = 2048) the relative error of the named gradients.  The GPU tests bound the production bf16 kernels by
                   multiples of these numbers (tests/test_parity_long_gpu.py).

Usage:  python tests/gen_golden_long.py      (about ten minutes on 8 cores)
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from gen_golden import OUT, import_reference, load_oracle, ref_train_loss  # noqa: E402

GRAD_SLICES = ("net.embed_tokens.weight", "net.layers.0.self_attn.q_proj.weight", "net.layers.5.self_attn.v_proj.weight",
               "net.layers.11.mlp.down_proj.weight", "net.layers.6.mlp.gate_proj.weight", "net_token.embed_tokens.weight",
               "net_token.layers.1.self_attn.o_proj.weight", "net_token.layers.2.mlp.up_proj.weight", "lm_head.weight",
               "net.layers.3.input_layernorm.weight", "net.norm.weight", "net_token.layers.0.post_attention_layernorm.weight")
SEEDS = {2048: 6, 4096: 7}


def slice_of(t: torch.Tensor) -> np.ndarray:
    t = t.detach().float()
    return (t if t.dim() == 1 else t[:64:3, ::5]).numpy().copy()


def main():
    ref_model, ref_tok = import_reference()
    orc = load_oracle()
    torch.set_num_threads(os.cpu_count())
    tok = ref_tok.MIDITokenizer("v2")
    shp = orc.Shape(vocab=tok.vocab_size)
    sd = orc.make_state_dict(shp, seed=0)
    cfg = ref_model.MIDIModelConfig.from_name("tv2o-medium")
    for S, seed in SEEDS.items():
        t0 = time.time()
        batch = orc.synthetic_events(tok, 1, S + 1, seed=seed)
        g = {"S": np.int64(S), "batch_seed": np.int64(seed), "weight_seed": np.int64(0)}
        model = ref_model.MIDIModel(cfg)
        model.load_state_dict(sd, strict=True)
        model.eval()
        want_grads = S == 2048
        if want_grads:
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
        else:
            with torch.no_grad():
                loss, logits, hidden = ref_train_loss(model, batch)
        g["loss"] = np.float64(loss.item())
        g["hidden_sub"] = hidden[::32, ::4].numpy().copy()
        g["logits_sub"] = logits[::64, :, ::16].numpy().copy()
        g["logits_lse"] = torch.logsumexp(logits, -1).numpy()
        g["logits_argmax"] = logits.argmax(-1).numpy()
        top2 = logits.topk(2, -1).values
        g["logits_margin"] = (top2[..., 0] - top2[..., 1]).numpy()
        g["hidden_absmax"] = np.float64(hidden.abs().max().item())
        g["logits_absmax"] = np.float64(logits.abs().max().item())
        print(f"S={S}: fp32 reference done in {time.time() - t0:.0f} s, loss {loss.item():.6f}", flush=True)

        # the reference's own bf16-true run on the same weights / inputs
        t0 = time.time()
        mb = ref_model.MIDIModel(cfg)
        mb.load_state_dict(sd, strict=True)
        mb = mb.to(torch.bfloat16).eval()
        if want_grads:
            lb, lgb, hb = ref_train_loss(mb, batch)
            lb.backward()
            nb = dict(mb.named_parameters())
            for k in GRAD_SLICES:
                d = nb[k].grad.float() - grads32[k]
                g["ref_bf16_grad_relerr:" + k] = np.float64((d.norm() / grads32[k].norm()).item())
            flatb = torch.cat([p.grad.float().reshape(-1) for p in nb.values()])
            fb, f32_ = flatb.double(), flat32.double()  # (234 M elements: fp32 accumulation is not good enough for a cosine)
            g["ref_bf16_grad_cosine"] = np.float64((torch.dot(fb, f32_) / (fb.norm() * f32_.norm())).item())
            g["ref_bf16_grad_norm_ratio"] = np.float64((fb.norm() / f32_.norm()).item())
            lb, lgb, hb = lb.detach(), lgb.detach(), hb.detach()
        else:
            with torch.no_grad():
                lb, lgb, hb = ref_train_loss(mb, batch)
        dh, dl = hb.float() - hidden, lgb.float() - logits
        g["ref_bf16_loss"] = np.float64(lb.item())
        g["ref_bf16_hidden_maxerr"] = np.float64(dh.abs().max().item())
        g["ref_bf16_hidden_rmserr"] = np.float64(dh.pow(2).mean().sqrt().item())
        g["ref_bf16_logits_maxerr"] = np.float64(dl.abs().max().item())
        g["ref_bf16_logits_rmserr"] = np.float64(dl.pow(2).mean().sqrt().item())
        g["ref_bf16_lse_maxerr"] = np.float64((torch.logsumexp(lgb.float(), -1) - torch.logsumexp(logits, -1)).abs().max().item())
        g["ref_bf16_argmax_agree"] = np.float64((lgb.argmax(-1) == logits.argmax(-1)).float().mean().item())
        print(f"S={S}: bf16 reference done in {time.time() - t0:.0f} s: loss {lb.item():.4f}, hidden max/rms "
              f"{g['ref_bf16_hidden_maxerr']:.4f}/{g['ref_bf16_hidden_rmserr']:.5f}, logits max/rms "
              f"{g['ref_bf16_logits_maxerr']:.4f}/{g['ref_bf16_logits_rmserr']:.5f}, argmax agree {g['ref_bf16_argmax_agree']:.4f}",
              flush=True)
        path = os.path.join(OUT, f"medium_long_S{S}.npz")
        np.savez_compressed(path, **g)
        print("wrote", path, os.path.getsize(path), flush=True)


if __name__ == "__main__":
    main()
