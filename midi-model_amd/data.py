"""Batches for the training step.

``TokenCorpus`` + ``WindowSampler``: the step that precedes the hot path.  The reference tokenises
MIDI files in 4 Python workers per step (train.py:31-90); at ~260 k events/s per GPU that starves the device, so here the
corpus is tokenised ONCE (by the reference's tokenizer, out of scope) into int16 octets, kept whole in HBM (2 bytes x 8 per
event: a billion events fit in 16 GB of the 288), and a batch is assembled by one kernel from B (start, length) windows --
the slicing rule of ``MidiDataset.__getitem__`` (train.py:73-88) and the pad-to-longest of ``collate_fn`` (:84-90).

``synthetic_events``: synthetic MIDI-event batches of the shape the reference's collate produces (train.py:69-90): (B, L, 8) int64,
row 0 of every sequence the BOS octet, then well-formed events [event_id, params..., pad...] — event type
~ Categorical(note 0.90, the other types sharing 0.10), every parameter uniform over its id range
(SURVEY.md §8(d)).  Used by bench.py / smoke (no dataset or checkpoint is reachable offline)."""
from __future__ import annotations

import random
from typing import List, Sequence, Tuple

import numpy as np
import torch

from . import ops


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


class TokenCorpus:
    """Pre-tokenised pieces, int16 (n_i, 8) each, concatenated in device memory with their offsets on the host."""

    def __init__(self, pieces: Sequence[np.ndarray], device="cuda"):
        assert len(pieces) > 0 and all(p.ndim == 2 and p.shape[1] == pieces[0].shape[1] for p in pieces)
        self.T = int(pieces[0].shape[1])
        self.offsets = np.concatenate([[0], np.cumsum([len(p) for p in pieces])]).astype(np.int64)
        self.tokens = torch.from_numpy(np.ascontiguousarray(np.concatenate(pieces, 0).astype(np.int16))).to(device)

    def __len__(self) -> int:
        return len(self.offsets) - 1

    def piece_len(self, i: int) -> int:
        return int(self.offsets[i + 1] - self.offsets[i])

    def save(self, path: str) -> None:
        np.savez(path, tokens=self.tokens.cpu().numpy(), offsets=self.offsets)

    @classmethod
    def load(cls, path: str, device="cuda") -> "TokenCorpus":
        z = np.load(path)
        off = z["offsets"]
        return cls([z["tokens"][off[i]:off[i + 1]] for i in range(len(off) - 1)], device)


class WindowSampler:
    """The window rule of MidiDataset.__getitem__ (train.py:73-83): with ``rand_start`` a uniformly random start or 0
    (a coin flip), else a start that depends on the index; at most ``max_len`` events per window."""

    def __init__(self, corpus: TokenCorpus, max_len: int = 2048, rand_start: bool = True, seed: int = 0):
        self.corpus, self.max_len, self.rand_start = corpus, max_len, rand_start
        self.rng = random.Random(seed)

    def window(self, index: int) -> Tuple[int, int]:
        n = self.corpus.piece_len(index)
        if self.rand_start:
            start = self.rng.randrange(0, max(1, n - self.max_len))
            start = self.rng.choice([0, start])
        else:
            max_start = max(1, n - self.max_len)
            start = (index * (max_start // 8)) % max_start
        return int(self.corpus.offsets[index]) + start, min(self.max_len, n - start)

    def batch(self, indices: Sequence[int], pad_id: int = 0) -> torch.Tensor:
        """(B, longest window, 8) int64 on the corpus' device: ``collate_fn`` in one launch"""
        wins: List[Tuple[int, int]] = [self.window(i) for i in indices]
        dev = self.corpus.tokens.device
        start = torch.tensor([w[0] for w in wins], dtype=torch.int64).to(dev, non_blocking=True)
        length = torch.tensor([w[1] for w in wins], dtype=torch.int64).to(dev, non_blocking=True)
        out = torch.empty((len(wins), max(w[1] for w in wins), self.corpus.T), dtype=torch.int64, device=dev)
        return ops.collate_windows(self.corpus.tokens, start, length, out, pad_id)
