"""Vocabulary tables vs the fixture dumped from the reference tokenizer classes
(midi_tokenizer.py:8-36, :506-535; tests/gen_golden.py)."""
import json
import os

import pytest

import midi_model_amd as mm
from conftest import GOLDEN


@pytest.mark.parametrize("ver", ["v1", "v2"])
def test_tables_match_reference(ver):
    with open(os.path.join(GOLDEN, f"tokenizer_{ver}.json")) as f:
        ref = json.load(f)
    tok = mm.MIDITokenizer(ver)
    d = tok.to_dict()
    for k in ("version", "vocab_size", "events", "event_parameters", "max_token_seq", "pad_id", "bos_id", "eos_id"):
        assert d[k] == ref[k], k
    assert tok.event_ids == ref["event_ids"]
    assert list(tok.parameter_ids.keys()) == ref["parameter_order"]
    for p, (lo, hi) in ref["parameter_ranges"].items():
        ids = tok.parameter_ids[p]
        assert ids[0] == lo and ids[-1] == hi and ids == list(range(lo, hi + 1))
    assert tok.id_events == {v: k for k, v in ref["event_ids"].items()}


def test_v2_known_values():
    tok = mm.MIDITokenizerV2()
    assert tok.vocab_size == 3406 and tok.max_token_seq == 8
    assert tok.event_ids == dict(note=3, patch_change=4, control_change=5, set_tempo=6, time_signature=7, key_signature=8)
    assert tok.parameter_ids["time1"][0] == 9 and tok.parameter_ids["mi"][-1] == 3405


def test_event_roundtrip_and_grammar_tables():
    tok = mm.MIDITokenizerV2()
    ev = ["note", 5, 3, 1, 9, 60, 100, 480]
    toks = tok.event2tokens(ev)
    assert len(toks) == 8 and toks[0] == 3 and tok.tokens2event(toks) == ev
    assert tok.event2tokens(["note", 500, 0, 0, 0, 0, 0, 0]) == []
    assert tok.tokens2event([0] * 8) == []
    first, lo, hi, arity = tok.grammar_tables()
    assert sum(first) == 7 and first[tok.eos_id] == 1
    assert arity[3] == 7 and arity[6] == 4 and arity[0] == 0
    assert (lo[6][4], hi[6][4]) == (tok.parameter_ids["bpm"][0], tok.parameter_ids["bpm"][-1] + 1)
    assert (lo[6][5], hi[6][5]) == (tok.pad_id, tok.pad_id + 1)


def test_config_surface():
    for name in mm.config_name_list:
        cfg = mm.MIDIModelConfig.from_name(name)
        assert cfg.n_embd == 1024
        assert cfg.net_token_config.num_hidden_layers == cfg.net_config.num_hidden_layers // 4
    cfg = mm.MIDIModelConfig.from_name("tv2o-large")
    assert cfg.net_config.num_hidden_layers == 24 and cfg.net_config.hidden_size == 1024
    with pytest.raises(ValueError):
        mm.MIDIModelConfig.from_name("tv3-medium")
    with pytest.raises(ValueError):
        mm.MIDIModelConfig.from_name("tv2-huge")
    d = json.loads(json.dumps(cfg.to_dict()))
    cfg2 = mm.MIDIModelConfig.from_dict(d)
    assert cfg2.to_dict() == cfg.to_dict()
    assert cfg2.tokenizer.optimise_midi is True


REF_DIR = "/root/reference"


@pytest.mark.skipif(not os.path.exists(os.path.join(REF_DIR, "midi_tokenizer.py")), reason="reference tree not present")
def test_reference_tokenizer_is_used_as_is_when_importable(monkeypatch):
    """The boundary the reference's callers see: ``config.tokenizer`` / ``model.tokenizer`` is THE REFERENCE'S class when
    its module is importable (train.py:60-64,214-228,397-406 and app.py:185,250 call tokenize / detokenize / augment /
    check_quality / midi2img on that very object), and the device-side grammar tables come from its public attributes."""
    import sys
    import numpy as np
    from midi_model_amd.tokenizer import grammar_tables, reference_tokenizer_module
    monkeypatch.syspath_prepend(REF_DIR)
    monkeypatch.delitem(sys.modules, "midi_tokenizer", raising=False)
    try:
        ref = reference_tokenizer_module()
        assert ref is not None and ref.__file__.startswith(REF_DIR)
        cfg = mm.MIDIModelConfig.from_name("tv2o-medium")
        tok = cfg.tokenizer
        assert type(tok) is ref.MIDITokenizerV2 and tok.optimise_midi is True
        assert type(mm.MIDIModelConfig.from_name("tv1-medium").tokenizer) is ref.MIDITokenizerV1
        # codec methods are the reference's own, intact: events -> tokens -> MIDI score -> tokens
        events = [["set_tempo", 0, 0, 0, 120], ["patch_change", 0, 0, 1, 0, 5], ["note", 0, 0, 1, 0, 60, 100, 8],
                  ["note", 1, 4, 1, 0, 64, 90, 4]]
        seq = [[tok.bos_id] + [tok.pad_id] * 7] + [tok.event2tokens(e) for e in events] + [[tok.eos_id] + [tok.pad_id] * 7]
        score = tok.detokenize(seq)
        back = tok.tokenize(score)
        assert isinstance(score, list) and len(back) >= 3 and all(len(r) == tok.max_token_seq for r in back)
        assert all(0 <= t < tok.vocab_size for r in back for t in r)
        aug = tok.augment(np.array(seq).tolist())
        assert len(aug) == len(seq)
        # tables: identical to the tables-only class's, through the free function on public attributes
        mine = mm.MIDITokenizerV2()
        assert grammar_tables(tok) == grammar_tables(mine)
        assert tok.to_dict() == {**mine.to_dict(), "optimise_midi": True}
        # the model takes its masks from the reference object and round-trips it through the config JSON
        from midi_model_amd.model import MIDIModel
        model = MIDIModel(mm.MIDIModelConfig.get_config("v2", True, 4, 4, 64, 128))
        assert type(model.tokenizer) is ref.MIDITokenizerV2
        first, lo, hi, arity = model._grammar()
        assert int(first.sum()) == 7 and arity[tok.event_ids["note"]] == 7
        cfg2 = mm.MIDIModelConfig.from_dict(__import__("json").loads(__import__("json").dumps(cfg.to_dict())))
        assert type(cfg2.tokenizer) is ref.MIDITokenizerV2 and cfg2.tokenizer.optimise_midi is True
        monkeypatch.setenv("MH_TABLES_ONLY_TOKENIZER", "1")
        assert type(mm.MIDITokenizer("v2")) is mm.MIDITokenizerV2
    finally:
        sys.modules.pop("midi_tokenizer", None)


def test_tables_only_tokenizer_says_it_has_no_codec():
    tok = mm.MIDITokenizerV2()
    for m in ("tokenize", "detokenize", "augment", "check_quality", "midi2img"):
        with pytest.raises(NotImplementedError, match="reference's midi_tokenizer"):
            getattr(tok, m)([])
