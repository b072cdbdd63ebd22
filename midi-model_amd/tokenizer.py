"""Vocabulary tables of the MIDI event tokenizers (v1 / v2) that the hot path reads.

The model and the decode loop need only the *static tables*: ``vocab_size, max_token_seq, pad_id, bos_id,
eos_id, event_ids, id_events, events, parameter_ids`` (reference: midi_tokenizer.py:8-36 for v1, :506-535 for
v2; consumed at midi_model.py:169-237).  The MIDI-file <-> token codecs (``tokenize / detokenize / augment /
check_quality / midi2img``, midi_tokenizer.py:608-1186) are CPU data-format code outside the accelerated path, but
``train.py:60-64,214-228,397-406`` and ``app.py:185,250`` reach them through ``config.tokenizer`` /
``model.tokenizer``.  So the boundary is:

  * when the reference's own ``midi_tokenizer`` module is importable (the drop-in deployment: our ``midi_model.py``
    next to the reference's ``midi_tokenizer.py`` / ``MIDI.py``), ``MIDITokenizer(version)`` returns an instance of
    THE REFERENCE'S class, codecs intact -- nothing of it is re-implemented here;
  * otherwise (the GPU box, the tests) the tables-only classes below stand in; they carry no codec and say so;
  * the dense grammar tables the device-side masks need come from the free function ``grammar_tables(tok)``, which
    reads public attributes only and therefore works on either kind of tokenizer object.

``MH_TABLES_ONLY_TOKENIZER=1`` forces the tables-only classes.

The id layout is: [pad, bos, eos] + one id per event type (schema order) + one contiguous id
range per parameter (parameter-table order).  ``tests/test_tokenizer.py`` checks every table
against a fixture dumped from the reference class.
"""
from __future__ import annotations

import os
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

    def grammar_tables(self):
        return grammar_tables(self)

    def _no_codec(self, *a, **k):
        raise NotImplementedError(
            "this is the tables-only tokenizer: the MIDI-file codecs (tokenize / detokenize / augment / check_quality / "
            "midi2img) are the reference's midi_tokenizer.py, which is used as-is when it is importable (put it, with MIDI.py, "
            "next to midi_model.py); they are CPU data-format code outside the accelerated path")

    tokenize = detokenize = augment = check_quality = midi2img = _no_codec


def grammar_tables(tok):
    """Dense tables for the device-side grammar masks, from the PUBLIC attributes of any tokenizer object (the
    reference's MIDITokenizerV1/V2 or the tables-only classes here).  Returns (first_mask, param_lo, param_hi, arity):

    first_mask : list[vocab] 0/1 — ids legal as token 0 of an event (event ids + EOS)
    param_lo/hi: [n_ids][max_token_seq] inclusive-exclusive id range legal at position i
                 (i>=1) for an event whose token 0 is that id; (pad,pad+1) past its arity
    arity      : [n_ids] number of parameters of the event with that id (0 for non-events)
    Mirrors the mask construction of midi_model.py:202-214.  The id ranges of a parameter must be contiguous (they are:
    ``allocate_ids`` hands out consecutive ids, midi_tokenizer.py:14-17, 512-515); that is checked."""
    V, T = tok.vocab_size, tok.max_token_seq
    first = [0] * V
    for i in tok.event_ids.values():
        first[i] = 1
    first[tok.eos_id] = 1
    lo = [[tok.pad_id] * T for _ in range(V)]
    hi = [[tok.pad_id + 1] * T for _ in range(V)]
    arity = [0] * V
    for name, eid in tok.event_ids.items():
        pnames = tok.events[name]
        arity[eid] = len(pnames)
        for pos, p in enumerate(pnames, start=1):
            ids = list(tok.parameter_ids[p])
            if ids != list(range(ids[0], ids[0] + len(ids))):
                raise ValueError(f"parameter {p!r}: ids are not one contiguous range")
            lo[eid][pos], hi[eid][pos] = ids[0], ids[-1] + 1
    return first, lo, hi, arity


AUG_TAB_SIZE = 40
AUG_STATS = 2 + 128  # per piece: lowest / highest pitch off the drum channel, channel mask per track (csrc/augment.hip)


def augment_table(tok):
    """int32[40] for mh_augment_*: what MIDITokenizer.augment (midi_tokenizer.py:364-417, :1023-1102) reads of the tokenizer --
    event ids, the octet columns of the parameters it touches, id bases and sizes -- from PUBLIC attributes, so the reference's
    own tokenizer object works as well as the tables-only one.  Layout: csrc/augment.hip."""
    names = ["note", "patch_change", "control_change", "set_tempo", "time_signature", "key_signature"]
    tab = [0] * AUG_TAB_SIZE
    tab[0] = tok.max_token_seq

    def col(ev, pn):
        return 1 + tok.events[ev].index(pn) if ev in tok.events and pn in tok.events[ev] else 0

    for e, n in enumerate(names):
        tab[1 + e] = tok.event_ids.get(n, -1)
        tab[7 + e] = col(n, "track")
        tab[13 + e] = col(n, "channel")
    tab[19], tab[20] = col("note", "pitch"), col("note", "velocity")
    tab[21], tab[22] = col("control_change", "controller"), col("control_change", "value")
    tab[23] = col("set_tempo", "bpm")
    tab[24], tab[25] = col("key_signature", "sf"), col("key_signature", "mi")
    pid = tok.parameter_ids

    def base(pn):
        return pid[pn][0] if pn in pid else 0

    tab[26], tab[27] = base("track"), len(pid["track"])
    tab[28], tab[29] = base("channel"), len(pid["channel"])
    tab[30], tab[31], tab[32], tab[33] = base("pitch"), base("velocity"), base("controller"), base("value")
    tab[34], tab[35] = base("bpm"), len(pid["bpm"])
    tab[36], tab[37] = base("sf"), base("mi")
    if tab[27] > 128 or tab[29] > 32 or len(pid["pitch"]) != 128 or len(pid["velocity"]) != 128 or len(pid["value"]) != 128:
        raise ValueError("augment_table: the device augmentation is built for <= 128 tracks, <= 32 channels, 128 pitches / velocities / values")
    return tab


class MIDITokenizerV1(_VocabTables):
    version = "v1"


class MIDITokenizerV2(_VocabTables):
    version = "v2"


def reference_tokenizer_module():
    """The reference's ``midi_tokenizer`` module if it is importable and looks like it (both codec classes present), else
    None.  Never imported from a fixed path: it is whatever the deployment put on sys.path."""
    if os.environ.get("MH_TABLES_ONLY_TOKENIZER", "0") == "1":
        return None
    try:
        import midi_tokenizer as ref  # noqa: the reference's module (midi_tokenizer.py), NOT part of this package
    except Exception:  # absent, or one of its own imports (PIL) is
        return None
    ok = all(hasattr(ref, n) for n in ("MIDITokenizer", "MIDITokenizerV1", "MIDITokenizerV2")) and \
        all(hasattr(ref.MIDITokenizerV2, m) for m in ("tokenize", "detokenize", "augment", "check_quality"))
    return ref if ok else None


class MIDITokenizer:
    """Factory with the reference's call shape: ``MIDITokenizer("v2")`` (midi_tokenizer.py:1189-1196).  Hands out the
    reference's own class when its module is importable (codec methods intact), the tables-only class otherwise."""

    def __new__(cls, version: str = "v2"):
        if version not in ("v1", "v2"):
            raise ValueError(f"Unsupported version: {version}")
        ref = reference_tokenizer_module()
        if ref is not None:
            tok = ref.MIDITokenizer(version)
            mine = MIDITokenizerV1() if version == "v1" else MIDITokenizerV2()
            for attr in ("vocab_size", "max_token_seq", "pad_id", "bos_id", "eos_id", "event_ids", "events"):
                if getattr(tok, attr) != getattr(mine, attr):
                    raise RuntimeError(f"the importable midi_tokenizer disagrees with the {version} vocabulary on {attr}")
            # the id ranges the device-side grammar masks are built from, and the id -> event-name map of the break rule
            if {k: list(v) for k, v in tok.parameter_ids.items()} != mine.parameter_ids:
                raise RuntimeError(f"the importable midi_tokenizer disagrees with the {version} vocabulary on parameter_ids")
            if dict(tok.id_events) != dict(mine.id_events):
                raise RuntimeError(f"the importable midi_tokenizer disagrees with the {version} vocabulary on id_events")
            return tok
        return MIDITokenizerV1() if version == "v1" else MIDITokenizerV2()
