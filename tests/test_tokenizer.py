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
