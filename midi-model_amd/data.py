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
    """Pre-tokenised pieces, int16 (n_i, 8) each, concatenated in device memory with their offsets on the host (and, for the
    augmentation kernels, on the device)."""

    def __init__(self, pieces: Sequence[np.ndarray], device="cuda"):
        assert len(pieces) > 0 and all(p.ndim == 2 and p.shape[1] == pieces[0].shape[1] for p in pieces)
        self.T = int(pieces[0].shape[1])
        self.offsets = np.concatenate([[0], np.cumsum([len(p) for p in pieces])]).astype(np.int64)
        host = torch.from_numpy(np.ascontiguousarray(np.concatenate(pieces, 0).astype(np.int16)))
        dev = torch.device(device)
        if dev.type == "cuda":  # one pinned staging copy: the upload runs at the link's rate without a pageable bounce
            host = host.pin_memory()
        self.tokens = host.to(dev, non_blocking=True)
        self.offsets_dev = torch.from_numpy(self.offsets).to(dev)
        self._aug = None  # (tokenizer table, per-piece stats) once augmentation has been asked for

    def __len__(self) -> int:
        return len(self.offsets) - 1

    def piece_len(self, i: int) -> int:
        return int(self.offsets[i + 1] - self.offsets[i])

    def augment_state(self, tok):
        """(tab int32[40], stats int32[P, 130]) on the corpus' device: the tokenizer tables and the per-file facts the
        device-side ``MIDITokenizer.augment`` needs (``mh_augment_piece_stats``, once per corpus)"""
        if self._aug is None:
            from .tokenizer import AUG_STATS, augment_table
            dev = self.tokens.device
            tab = torch.tensor(augment_table(tok), dtype=torch.int32, device=dev)
            if int(tab[0]) != self.T:
                raise ValueError(f"the corpus holds {self.T}-token events, the tokenizer {int(tab[0])}")
            stats = torch.empty((len(self), AUG_STATS), dtype=torch.int32, device=dev)
            ops.augment_piece_stats(self.tokens, self.offsets_dev, tab, stats)
            self._aug = (tab, stats)
        return self._aug

    def save(self, path: str) -> None:
        np.savez(path, tokens=self.tokens.cpu().numpy(), offsets=self.offsets)

    @classmethod
    def load(cls, path: str, device="cuda") -> "TokenCorpus":
        z = np.load(path)
        off = z["offsets"]
        return cls([z["tokens"][off[i]:off[i + 1]] for i in range(len(off) - 1)], device)


# MIDITokenizer.augment's default maxima (midi_tokenizer.py:1023-1024): pitch, velocity, cc value, bpm, track, channel
AUG_MAXIMA = (4, 10, 10, 10, 0, 16)


class WindowSampler:
    """``MidiDataset.load_midi`` + ``__getitem__`` (train.py:48-83) over a pre-tokenised corpus: with ``aug`` (the reference's
    default) every served file is augmented by ``tokenizer.augment`` -- here on the device, fused into the batch assembly, with
    the six shifts drawn on the host in the reference's order (midi_tokenizer.py:1025-1030) BEFORE the window draws, as in
    ``load_midi`` -> ``__getitem__``; then the window rule: with ``rand_start`` a uniformly random start or 0 (a coin flip),
    else a start that depends on the index; at most ``max_len`` events per window."""

    def __init__(self, corpus: TokenCorpus, max_len: int = 2048, rand_start: bool = True, seed: int = 0, aug: bool = False,
                 tokenizer=None, aug_maxima: Sequence[int] = AUG_MAXIMA):
        self.corpus, self.max_len, self.rand_start = corpus, max_len, rand_start
        self.rng = random.Random(seed)
        self.aug, self.aug_maxima = bool(aug), tuple(int(x) for x in aug_maxima)
        if self.aug:
            if tokenizer is None:
                raise ValueError("WindowSampler(aug=True) needs the tokenizer whose tables the augmentation rules read")
            self.tab, self.stats = corpus.augment_state(tokenizer)
        self._stage = None  # pinned host staging for the per-batch window descriptors

    def draw_shifts(self) -> Tuple[int, ...]:
        m = self.aug_maxima
        r = self.rng
        return (r.randint(-m[0], m[0]), r.randint(-m[1], m[1]), r.randint(-m[2], m[2]), r.randint(-m[3], m[3]),
                r.randint(0, m[4]), r.randint(0, m[5]))

    def window(self, index: int) -> Tuple[int, int]:
        n = self.corpus.piece_len(index)
        if self.rand_start:
            start = self.rng.randrange(0, max(1, n - self.max_len))
            start = self.rng.choice([0, start])
        else:
            max_start = max(1, n - self.max_len)
            start = (index * (max_start // 8)) % max_start
        return int(self.corpus.offsets[index]) + start, min(self.max_len, n - start)

    def _to_device(self, cols: np.ndarray) -> torch.Tensor:
        """[k, B] int64 window descriptors -> device, through a pinned staging buffer when the corpus is on a GPU"""
        dev = self.corpus.tokens.device
        host = torch.from_numpy(cols)
        if dev.type != "cuda":
            return host.to(dev)
        if self._stage is None or self._stage.shape != host.shape:
            self._stage = torch.empty(host.shape, dtype=torch.int64).pin_memory()
            self._stage_free = torch.cuda.Event()
        else:
            self._stage_free.synchronize()  # the previous batch's upload has left the staging buffer
        self._stage.copy_(host)
        out = self._stage.to(dev, non_blocking=True)
        self._stage_free.record()
        return out

    def batch(self, indices: Sequence[int], pad_id: int = 0) -> torch.Tensor:
        """(B, longest window, 8) int64 on the corpus' device: ``collate_fn`` (+ ``augment``) in one launch"""
        shifts, wins = [], []
        for i in indices:
            if self.aug:
                shifts.append(self.draw_shifts())  # load_midi -> tokenizer.augment draws first (train.py:62-63) ...
            wins.append(self.window(i))            # ... then __getitem__ draws the window (train.py:73-77)
        B = len(wins)
        dev = self.corpus.tokens.device
        cols = np.zeros((9, B), dtype=np.int64)
        cols[0] = [w[0] for w in wins]
        cols[1] = [w[1] for w in wins]
        cols[2] = list(indices)
        if self.aug:
            cols[3:9] = np.asarray(shifts, dtype=np.int64).T
        d = self._to_device(cols)
        out = torch.empty((B, max(w[1] for w in wins), self.corpus.T), dtype=torch.int64, device=dev)
        if not self.aug:
            return ops.collate_windows(self.corpus.tokens, d[0], d[1], out, pad_id)
        sh = d[3:9].t().to(torch.int32).contiguous()
        return ops.augment_collate_windows(self.corpus.tokens, d[0], d[1], d[2], sh, self.stats, self.tab, out, pad_id)
