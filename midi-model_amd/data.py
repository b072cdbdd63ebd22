"""Synthetic MIDI-event batches of the shape the reference's collate produces (train.py:69-90): (B, L, 8) int64,
row 0 of every sequence the BOS octet, then well-formed events [event_id, params..., pad...] — event type
~ Categorical(note 0.90, the other types sharing 0.10), every parameter uniform over its id range
(SURVEY.md §8(d)).  Used by bench.py / smoke (no dataset or checkpoint is reachable offline)."""
from __future__ import annotations

import torch


def synthetic_events(tok, batch: int, length: int, seed: int = 0, note_p: float = 0.90, device="cpu") -> torch.Tensor:
    g = torch.Generator().manual_seed(seed)
    names = list(tok.events.keys())
    probs = torch.full((len(names),), (1.0 - note_p) / max(1, len(names) - 1))
    probs[names.index("note")] = note_p
    T = tok.max_token_seq
    n = batch * (length - 1)
    rows = torch.full((n, T), tok.pad_id, dtype=torch.long)
    kinds = torch.multinomial(probs, n, replacement=True, generator=g)
    u = torch.rand((n, T - 1), generator=g)
    for ki, name in enumerate(names):
        sel = (kinds == ki).nonzero().flatten()
        if sel.numel() == 0:
            continue
        rows[sel, 0] = tok.event_ids[name]
        for pos, pname in enumerate(tok.events[name], start=1):
            ids = tok.parameter_ids[pname]
            rows[sel, pos] = ids[0] + (u[sel, pos - 1] * len(ids)).long().clamp_(max=len(ids) - 1)
    out = torch.full((batch, length, T), tok.pad_id, dtype=torch.long)
    out[:, 0, 0] = tok.bos_id
    out[:, 1:] = rows.view(batch, length - 1, T)
    return out.to(device)
