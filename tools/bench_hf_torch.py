#!/usr/bin/env python
"""What the reference's own stack does on this GPU: the architecture of midi_model.py restated on the PUBLIC HuggingFace
classes it uses (two `LlamaModel`s + `nn.Linear`, sdpa attention, midi_model.py:99-150) and the training step of
train.py:168-188 with `torch.optim.AdamW`, `clip_grad_norm_(1.0)` and bf16-true parameters (Lightning precision
"bf16-true"), timed with PyTorch-ROCm eager on the same synthetic batch as bench.py.  A measurement aid: nothing in the
package imports this, and the reference repository itself is not needed."""
import argparse
import os
import sys
import time

import torch
import torch.nn as nn
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--seq", type=int, default=2048)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--gen-batch", type=int, default=64)
    ap.add_argument("--gen-events", type=int, default=64)
    args = ap.parse_args()
    from transformers import LlamaConfig, LlamaModel

    import midi_model_amd as mm
    from midi_model_amd.data import synthetic_events

    tok = mm.MIDITokenizerV2()
    V = tok.vocab_size

    def cfg(layers, heads, inter):
        return LlamaConfig(vocab_size=V, hidden_size=1024, num_attention_heads=heads, num_hidden_layers=layers,
                           intermediate_size=inter, pad_token_id=tok.pad_id, max_position_embeddings=4096, use_cache=False,
                           attn_implementation="sdpa")

    torch.manual_seed(0)
    net, net_token = LlamaModel(cfg(12, 16, 4096)), LlamaModel(cfg(3, 4, 1024))
    lm_head = nn.Linear(1024, V, bias=False)
    model = nn.ModuleDict({"net": net, "net_token": net_token, "lm_head": lm_head}).to("cuda", torch.bfloat16)
    decay = [p for n, p in model.named_parameters() if "norm" not in n and "bias" not in n]
    nodecay = [p for n, p in model.named_parameters() if "norm" in n or "bias" in n]
    opt = torch.optim.AdamW([{"params": decay, "weight_decay": 0.01}, {"params": nodecay, "weight_decay": 0.0}], lr=2e-4,
                            betas=(0.9, 0.99), eps=1e-8)
    B, S = args.batch, args.seq
    batch = synthetic_events(tok, B, S + 1, seed=1000, device="cuda")

    def step():
        x, y = batch[:, :-1].contiguous(), batch[:, 1:].contiguous()
        h = net.embed_tokens(x).sum(dim=-2)
        hidden = net(inputs_embeds=h).last_hidden_state.reshape(-1, 1024)
        y = y.reshape(-1, y.shape[-1])
        emb = net_token.embed_tokens(y[:, :-1])
        seq = torch.cat([hidden.unsqueeze(1), emb], dim=1)
        logits = lm_head(net_token(inputs_embeds=seq).last_hidden_state)
        loss = F.cross_entropy(logits.view(-1, V), y.reshape(-1), reduction="mean", ignore_index=tok.pad_id)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        opt.step()
        return loss

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"HF/PyTorch-eager restatement, bf16-true, B={B} S={S}: {B * S * args.steps / dt:.0f} events/s, "
          f"{1e3 * dt / args.steps:.1f} ms/step, loss {float(loss.detach()):.4f}, "
          f"peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")

    # ---- generation, the loop of midi_model.py:167-250 on the same classes (DynamicCache, one forward per event and per
    #      token, masked softmax, sort / cumsum / multinomial sampler); the grammar masks come from range tables instead
    #      of the reference's per-row Python loop, which only makes this baseline faster
    from transformers import DynamicCache
    model.eval()
    Bg, n_new, T = args.gen_batch, args.gen_events, tok.max_token_seq
    first, lo_t, hi_t, arity = tok.grammar_tables()
    first_m = torch.tensor(first, device="cuda", dtype=torch.float32)
    first_m[tok.eos_id] = 0  # every row runs the full length, as bench.py --mode generate
    lo_t, hi_t = torch.tensor(lo_t, device="cuda"), torch.tensor(hi_t, device="cuda")
    ids = torch.arange(V, device="cuda")[None, :]
    gen = torch.Generator(device="cuda").manual_seed(0)

    def sample(probs, p=0.98, k=20):
        ps, pi = torch.sort(probs, dim=-1, descending=True)
        cs = torch.cumsum(ps, dim=-1)
        ps[cs - ps > p] = 0.0
        ps[:, k:] = 0.0
        ps.div_(ps.sum(dim=-1, keepdim=True))
        return torch.gather(pi, -1, torch.multinomial(ps, 1, generator=gen))

    @torch.inference_mode()
    def generate(n_events):
        seq = torch.full((Bg, 1, T), tok.pad_id, dtype=torch.long, device="cuda")
        seq[:, 0, 0] = tok.bos_id
        cache1, new = DynamicCache(), seq
        for _ in range(n_events):
            h = net(inputs_embeds=net.embed_tokens(new).sum(dim=-2), past_key_values=cache1, use_cache=True).last_hidden_state[:, -1]
            cache2, ev_tok, nxt = DynamicCache(), None, torch.full((Bg, T), tok.pad_id, dtype=torch.long, device="cuda")
            for i in range(T):
                x = h.unsqueeze(1) if i == 0 else net_token.embed_tokens(ev_tok)
                out = net_token(inputs_embeds=x, past_key_values=cache2, use_cache=True).last_hidden_state[:, -1]
                probs = torch.softmax(lm_head(out).float(), dim=-1)
                if i == 0:
                    mask = first_m[None, :]
                else:
                    mask = ((ids >= lo_t[nxt[:, 0], i][:, None]) & (ids < hi_t[nxt[:, 0], i][:, None])).float()
                ev_tok = sample(probs * mask)
                nxt[:, i] = ev_tok[:, 0]
                if i == 0:
                    ar = [arity[t] for t in nxt[:, 0].tolist()]  # the reference's break rule needs the ids on the host
                    stop = ar[0] + 1 if all(a == ar[0] for a in ar) else T
                if i + 1 >= stop:
                    break
            new = nxt.unsqueeze(1)
        return new

    generate(4)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    generate(n_new)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"HF/PyTorch-eager generation loop, bf16, B={Bg}, {n_new} new events: {Bg * n_new / dt:.0f} events/s "
          f"({1e3 * dt / n_new:.1f} ms/event)")


if __name__ == "__main__":
    main()
