"""Generate tests/golden/tiny_trained.npz: a TRAINED tiny model, so that parity is also pinned on PEAKED distributions.

Every other model-level fixture uses N(0, 0.02) random-init weights (loss ~ ln 3406): logits nearly flat, top-2 margins inside the
bf16 drift, top-p / top-k truncation never binding.  The reference is used with trained weights (app.py:299-320); there is no
network for a real checkpoint, so this script trains the REAL reference model (``/root/reference/midi_model.py`` +
``torch.optim.AdamW``, CPU fp32) for a few hundred steps on a fixed synthetic corpus with structure (motifs repeated with small
variations: the next event inside a motif is nearly certain, the motif that follows and the jittered parameters are not), snaps
the weights to a grid that fp32, bf16 and fp16 all hold exactly (2-D tensors: int8 x a power of two per tensor; 1-D: bf16), and
records what the reference then does with them: its loss / logits statistics on held-out corpus rows, its own bf16 drift, and its
seeded ``generate`` ids (sampled top_p 0.98 / top_k 20, greedy, and continuing a corpus prompt).

Runs only in the build container (the GPU box has no /root/reference); the outputs are committed.
Usage:  python tests/gen_golden_trained.py
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from gen_golden import OUT, build_ref, import_reference, load_oracle, ref_train_loss  # noqa: E402

SHAPE = dict(n_layer=4, n_head=4, n_embd=256, n_inner=1024)   # (token-level inner 256: the decode path takes its production kernels)


def motif_corpus(tok, n_seq: int, length: int, seed: int, n_motif: int = 12) -> torch.Tensor:
    """(n_seq, length, 8) int64.  Row 0 = BOS octet; then motifs drawn uniformly from a fixed set of ``n_motif`` (each 6-14
    events, fixed by seed 1000 -- the SAME motifs for every call), every instance transposed by one of three pitch offsets and
    its velocities jittered over four neighbouring values.  Event grammar as tok.events / tok.parameter_ids."""
    gm = torch.Generator().manual_seed(1000)
    names = list(tok.events.keys())
    motifs = []
    for _ in range(n_motif):
        n_ev = int(torch.randint(6, 15, (1,), generator=gm))
        rows = torch.full((n_ev, tok.max_token_seq), tok.pad_id, dtype=torch.long)
        for e in range(n_ev):
            name = "note" if torch.rand((), generator=gm) < 0.85 else names[int(torch.randint(0, len(names), (1,), generator=gm))]
            rows[e, 0] = tok.event_ids[name]
            for pos, pname in enumerate(tok.events[name], start=1):
                ids = tok.parameter_ids[pname]
                if pname in ("time1", "time2", "track", "channel"):   # small values: a motif advances slowly, few tracks
                    rows[e, pos] = ids[int(torch.randint(0, min(4, len(ids)), (1,), generator=gm))]
                else:
                    rows[e, pos] = ids[int(torch.randint(0, len(ids), (1,), generator=gm))]
        motifs.append(rows)
    g = torch.Generator().manual_seed(seed)
    note_id = tok.event_ids["note"]
    ppos = 1 + tok.events["note"].index("pitch")
    vpos = 1 + tok.events["note"].index("velocity")
    pitch, vel = tok.parameter_ids["pitch"], tok.parameter_ids["velocity"]
    out = torch.full((n_seq, length, tok.max_token_seq), tok.pad_id, dtype=torch.long)
    out[:, 0, 0] = tok.bos_id
    for s in range(n_seq):
        at = 1
        while at < length:
            m = motifs[int(torch.randint(0, n_motif, (1,), generator=g))].clone()
            shift = int(torch.randint(0, 3, (1,), generator=g)) * 2
            is_note = m[:, 0] == note_id
            m[is_note, ppos] = (m[is_note, ppos] - pitch[0] + shift).clamp_(max=len(pitch) - 1) + pitch[0]
            jit = torch.randint(0, 4, (int(is_note.sum()),), generator=g)
            m[is_note, vpos] = (m[is_note, vpos] - vel[0] + jit).clamp_(max=len(vel) - 1) + vel[0]
            n = min(len(m), length - at)
            out[s, at:at + n] = m[:n]
            at += n
    return out


def snap(sd):
    """weights every dtype holds exactly: 2-D -> int8 * 2^e per tensor, 1-D -> bf16.  Returns (snapped fp32 sd, storage dict)."""
    out, store = {}, {}
    for k, v in sd.items():
        v = v.detach().float()
        if v.ndim == 2:
            e = int(np.ceil(np.log2(max(float(v.abs().max()), 1e-12) / 127.0)))
            q = torch.round(v / 2.0 ** e).clamp_(-127, 127).to(torch.int8)
            out[k] = q.float() * 2.0 ** e
            store["q8:" + k] = q.numpy()
            store["e:" + k] = np.int32(e)
        else:
            b = v.to(torch.bfloat16)
            out[k] = b.float()
            store["b16:" + k] = b.view(torch.int16).numpy()
    return out, store


def load_snapped(npz):
    """inverse of snap()'s storage (the tests use tests/conftest.py:load_trained, the same five lines)"""
    sd = {}
    for k in npz.files:
        if k.startswith("q8:"):
            sd[k[3:]] = torch.from_numpy(npz[k].astype(np.float32)) * 2.0 ** int(npz["e:" + k[3:]])
        elif k.startswith("b16:"):
            sd[k[4:]] = torch.from_numpy(npz[k].copy()).view(torch.bfloat16).float()
    return sd


def main():
    ref_model, ref_tok = import_reference()
    orc = load_oracle()
    torch.set_num_threads(os.cpu_count())
    tok = ref_tok.MIDITokenizer("v2")
    shp = orc.Shape(vocab=tok.vocab_size, **SHAPE)
    model = build_ref(ref_model, shp, orc.make_state_dict(shp, seed=5))
    model.train()
    params = list(model.named_parameters())
    no_decay = ["bias", "norm"]   # train.py:121-151
    groups = [{"params": [p for n, p in params if not any(nd in n for nd in no_decay)], "weight_decay": 0.01},
              {"params": [p for n, p in params if any(nd in n for nd in no_decay)], "weight_decay": 0.0}]
    opt = torch.optim.AdamW(groups, lr=2e-3, betas=(0.9, 0.99), eps=1e-8)
    t0, hist = time.time(), []
    for step in range(400):
        batch = motif_corpus(tok, 8, 65, seed=step)
        opt.zero_grad(set_to_none=True)
        loss, _, _ = ref_train_loss(model, batch)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        opt.step()
        hist.append(loss.item())
        if step % 20 == 0:
            print(f"step {step}: loss {loss.item():.3f}  ({time.time() - t0:.0f} s)", flush=True)
        if step >= 150 and np.mean(hist[-10:]) < 1.6:
            break
    model.eval()
    sd, store = snap(model.state_dict())
    model = build_ref(ref_model, shp, sd)

    g = dict(store)
    g["train_loss_history"] = np.array(hist, dtype=np.float32)
    ev = motif_corpus(tok, 4, 49, seed=10_001)
    g["eval_batch"] = ev.numpy()
    with torch.no_grad():
        loss, logits, hidden = ref_train_loss(model, ev)
        g["eval_loss"] = np.float64(loss.item())
        assert loss.item() < 4.0, loss.item()
        g["eval_logits_lse"] = torch.logsumexp(logits, -1).numpy()
        g["eval_logits_argmax"] = logits.argmax(-1).numpy()
        top2 = logits.topk(2, -1).values
        g["eval_logits_margin"] = (top2[..., 0] - top2[..., 1]).numpy()
        g["eval_logits_sub"] = logits[:, :, ::16].numpy()
        g["eval_hidden"] = hidden.numpy()
        pmax = torch.softmax(logits, -1).amax(-1)
        g["eval_pmax"] = pmax.numpy()
        # the reference's own bf16 drift on these weights (the yardstick of the bf16 tests, as in gen_golden_long.py)
        mb = build_ref(ref_model, shp, sd).to(torch.bfloat16)
        lb, logits_b, hidden_b = ref_train_loss(mb, ev)
        g["ref_bf16_loss"] = np.float64(lb.item())
        g["ref_bf16_hidden_maxerr"] = np.float64((hidden_b.float() - hidden).abs().max().item())
        g["ref_bf16_logits_maxerr"] = np.float64((logits_b.float() - logits).abs().max().item())
        g["ref_bf16_logits_meanerr"] = np.float64((logits_b.float() - logits).abs().mean().item())
    # generation (midi_model.py:167-250) on the trained weights
    g["sampled_b4"] = model.generate(None, batch_size=4, max_len=40, temp=1.0, top_p=0.98, top_k=20,
                                     generator=torch.Generator().manual_seed(4321))
    g["greedy_b4"] = model.generate(None, batch_size=4, max_len=40, temp=1.0, top_p=0.98, top_k=1,
                                    generator=torch.Generator().manual_seed(0))
    prompt = motif_corpus(tok, 1, 13, seed=10_002)[0].numpy()
    g["prompt"] = prompt
    g["prompt_sampled_b3"] = model.generate(prompt, batch_size=3, max_len=36, temp=0.9, top_p=0.9, top_k=8,
                                            generator=torch.Generator().manual_seed(99))
    g["prompt_greedy_b2"] = model.generate(prompt, batch_size=2, max_len=36, top_k=1, generator=torch.Generator().manual_seed(0))
    np.savez_compressed(os.path.join(OUT, "tiny_trained.npz"), **g)
    sz = os.path.getsize(os.path.join(OUT, "tiny_trained.npz"))
    print(f"tiny_trained.npz: {sz / 1e6:.2f} MB; steps {len(hist)}, train loss {np.mean(hist[-10:]):.3f}, eval loss (snapped) "
          f"{g['eval_loss']:.3f}; rows with p_max > 0.9: {(pmax > 0.9).float().mean().item():.2f}, < 0.5: "
          f"{(pmax < 0.5).float().mean().item():.2f}; median top-2 margin {np.median(g['eval_logits_margin']):.2f}; reference bf16 "
          f"drift: logits {g['ref_bf16_logits_maxerr']:.4f} hidden {g['ref_bf16_hidden_maxerr']:.4f}")


if __name__ == "__main__":
    main()
