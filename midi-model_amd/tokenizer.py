"""Vocabulary tables of the MIDI event tokenizers (v1 / v2) that the hot path reads.

Only the *static tables* live here: the model and the decode loop need
``vocab_size, max_token_seq, pad_id, bos_id, eos_id, event_ids, id_events, events,
parameter_ids`` (reference: midi_tokenizer.py:8-36 for v1, :506-535 for v2; consumed at
midi_model.py:169-237).  The MIDI-file <-> token codecs (tokenize / detokenize / augment,
midi_tokenizer.py:608-1186) are CPU data-format code outside the accelerated path
(SURVEY.md §8(f) "next").

The id layout is: [pad, bos, eos] + one id per event type (schema order) + one contiguous id
range per parameter (parameter-table order).  ``tests/test_tokenizer.py`` checks every table
against a fixture dumped from the reference class.
"""
from __future__ import annotations

from typing import Any, Dict, List

# event name -> ordered parameter names (the token octet is [event_id, *params, pad...])
_SCHEMA = {
    "v1": (
        ("note", ("time1", "time2", "track", "duration", "channel", "pitch", "velocity")),
        ("patch_change", ("time1", "time2", "track", "channel", "patch")),
        ("control_change", ("time1", "time2", "track", "channel", "controller", "value")),
        ("set_tempo", ("time1", "time2", "track", "bpm")),
    ),
    "v2": (
        ("note", ("time1", "time2", "track", "channel", "pitch", "velocity", "duration")),
        ("patch_change", ("time1", "time2", "track", "channel", "patch")),
        ("control_change", ("time1", "time2", "track", "channel", "controller", "value")),
        ("set_tempo", ("time1", "time2", "track", "bpm")),
        ("time_signature", ("time1", "time2", "track", "nn", "dd")),
        ("key_signature", ("time1", "time2", "track", "sf", "mi")),
    ),
}

# parameter name -> number of distinct values, in id-allocation order
_PARAM_CARD = {
    "v1": (("time1", 128), ("time2", 16), ("duration", 2048), ("track", 128), ("channel", 16),
           ("pitch", 128), ("velocity", 128), ("patch", 128), ("controller", 128), ("value", 128),
           ("bpm", 256)),
    "v2": (("time1", 128), ("time2", 16), ("duration", 2048), ("track", 128), ("channel", 16),
           ("pitch", 128), ("velocity", 128), ("patch", 128), ("controller", 128), ("value", 128),
           ("bpm", 384), ("nn", 16), ("dd", 4), ("sf", 15), ("mi", 2)),
}


class _VocabTables:
    """Shared builder: walks the schema once and hands out consecutive ids."""

    version = "?"

    def __init__(self) -> None:
        self.optimise_midi = False
        cursor = 0
        self.pad_id, self.bos_id, self.eos_id = 0, 1, 2
        cursor = 3
        self.events: Dict[str, List[str]] = {n: list(ps) for n, ps in _SCHEMA[self.version]}
        self.event_parameters: Dict[str, int] = dict(_PARAM_CARD[self.version])
        self.event_ids: Dict[str, int] = {}
        for name in self.events:
            self.event_ids[name] = cursor
            cursor += 1
        self.id_events: Dict[int, str] = {i: n for n, i in self.event_ids.items()}
        self.parameter_ids: Dict[str, List[int]] = {}
        for pname, card in self.event_parameters.items():
            self.parameter_ids[pname] = list(range(cursor, cursor + card))
            cursor += card
        self.vocab_size = cursor
        self.max_token_seq = 1 + max(len(ps) for ps in self.events.values())

    # -- reference-compatible helpers used around the model -------------------------------
    def set_optimise_midi(self, optimise_midi: bool = True) -> None:
        self.optimise_midi = optimise_midi

    def to_dict(self) -> Dict[str, Any]:
        return {
            "version": self.version,
            "optimise_midi": self.optimise_midi,
            "vocab_size": self.vocab_size,
            "events": self.events,
            "event_parameters": self.event_parameters,
            "max_token_seq": self.max_token_seq,
            "pad_id": self.pad_id,
            "bos_id": self.bos_id,
            "eos_id": self.eos_id,
        }

    def event2tokens(self, event) -> List[int]:
        """[name, p0, p1, ...] -> padded token octet ([] when a value is out of range)."""
        name, values = event[0], event[1:]
        pnames = self.events[name]
        for v, p in zip(values, pnames):
            if not 0 <= v < self.event_parameters[p]:
                return []
        toks = [self.event_ids[name]] + [self.parameter_ids[p][v] for v, p in zip(values, pnames)]
        return toks + [self.pad_id] * (self.max_token_seq - len(toks))

    def tokens2event(self, tokens) -> list:
        """token octet -> [name, p0, p1, ...] ([] when it is not a well-formed event)."""
        name = self.id_events.get(int(tokens[0]))
        if name is None:
            return []
        pnames = self.events[name]
        if len(tokens) <= len(pnames):
            return []
        out = [name]
        for t, p in zip(tokens[1:], pnames):
            v = int(t) - self.parameter_ids[p][0]
            if not 0 <= v < self.event_parameters[p]:
                return []
            out.append(v)
        return out

    # -- dense tables for the device-side grammar masks (ours) -----------------------------
    def grammar_tables(self):
        """Return (first_mask, param_lo, param_hi, arity).

        first_mask : list[vocab] 0/1 — ids legal as token 0 of an event (event ids + EOS)
        param_lo/hi: [n_ids][max_token_seq] inclusive-exclusive id range legal at position i
                     (i>=1) for an event whose token 0 is that id; (pad,pad+1) past its arity
        arity      : [n_ids] number of parameters of the event with that id (0 for non-events)
        Mirrors the mask construction of midi_model.py:202-214.
        """
        V, T = self.vocab_size, self.max_token_seq
        first = [0] * V
        for i in self.event_ids.values():
            first[i] = 1
        first[self.eos_id] = 1
        lo = [[self.pad_id] * T for _ in range(V)]
        hi = [[self.pad_id + 1] * T for _ in range(V)]
        arity = [0] * V
        for name, eid in self.event_ids.items():
            pnames = self.events[name]
            arity[eid] = len(pnames)
            for pos, p in enumerate(pnames, start=1):
                ids = self.parameter_ids[p]
                lo[eid][pos], hi[eid][pos] = ids[0], ids[-1] + 1
        return first, lo, hi, arity


class MIDITokenizerV1(_VocabTables):
    version = "v1"


class MIDITokenizerV2(_VocabTables):
    version = "v2"


class MIDITokenizer:
    """Factory with the reference's call shape: ``MIDITokenizer("v2")`` (midi_tokenizer.py:1189-1196)."""

    def __new__(cls, version: str = "v2"):
        if version == "v1":
            return MIDITokenizerV1()
        if version == "v2":
            return MIDITokenizerV2()
        raise ValueError(f"Unsupported version: {version}")
