"""Generate tests/golden/augment_v{1,2}.npz by running the REAL reference tokenizers' ``augment``
(/root/reference/midi_tokenizer.py:364-417 for v1, :1023-1102 for v2) under fixed ``random.seed``s.
Runs only in the build container (the GPU box has no /root/reference); the outputs are committed.
Usage:  python tests/gen_golden_augment.py

Each case is one "file": a pre-tokenised event sequence (int16 [n, 8]) built here from the tokenizer's own
tables, the six shifts the reference drew for it (recovered by replaying ``random.randint`` in the reference's
order from the same seed -- asserted against a second run), and the reference's output.  The cases cover what
the restatement has to get right: drum-channel rules (channel 9 stays, a channel shifted ONTO 9 moves on),
velocity / controller-value / tempo clamps, the key-signature transposition and its second pass (a track whose
notes are all on channel 9 gets sf = 0), tracks with several channels, files whose shifted pitch leaves 0..127
(the reference returns the file UNCHANGED), files with no notes at all, and non-default maxima
(max_track_shift > 0, where the second pass looks the SHIFTED track up among the ORIGINAL note tracks).
"""
import os
import random
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "golden")
REF = "/root/reference"


def import_reference_tokenizer():
    root = os.path.dirname(HERE)
    sys.path[:] = [p for p in sys.path if os.path.abspath(p or ".") != root]
    sys.path.insert(0, REF)
    import midi_tokenizer as ref_tok
    assert ref_tok.__file__.startswith(REF)
    return ref_tok


def make_file(tok, rng: np.random.Generator, n: int, style: str):
    """a well-formed token sequence: BOS, n events, EOS (what MIDITokenizer.tokenize returns, midi_tokenizer.py:608+)"""
    T = tok.max_token_seq
    rows = [[tok.bos_id] + [tok.pad_id] * (T - 1)]
    names = list(tok.events.keys())
    drum_tracks = {3, 5} if style in ("drums", "mixed") else set()
    for _ in range(n):
        if style == "no_notes":
            name = names[int(rng.integers(1, len(names)))]
        else:
            name = "note" if rng.random() < 0.75 else names[int(rng.integers(0, len(names)))]
        row = [tok.event_ids[name]]
        track = int(rng.integers(0, 8))
        for pn in tok.events[name]:
            size = tok.event_parameters[pn]
            v = int(rng.integers(0, size))
            if pn == "track":
                v = track
            elif pn == "channel":
                if track in drum_tracks and name == "note":
                    v = 9
                elif style == "drums":
                    v = 9
                elif rng.random() < 0.15:
                    v = 9
            elif pn == "pitch":
                if style == "low":
                    v = int(rng.integers(0, 30))
                elif style == "high":
                    v = int(rng.integers(100, 128))
                elif style in ("mid", "mixed"):
                    v = int(rng.integers(12, 116))   # never leaves 0..127 under |shift| <= 4 ... 12
            elif pn == "velocity":
                v = int(rng.integers(1, size))
            elif pn == "controller" and rng.random() < 0.6:
                v = int(rng.choice([1, 2, 7, 11]))
            row.append(tok.parameter_ids[pn][v])
        rows.append(row + [tok.pad_id] * (T - len(row)))
    rows.append([tok.eos_id] + [tok.pad_id] * (T - 1))
    return rows


def replay_shifts(seed, mx):
    """the six draws augment makes, in its order (midi_tokenizer.py:1025-1030)"""
    random.seed(seed)
    return [random.randint(-mx[0], mx[0]), random.randint(-mx[1], mx[1]), random.randint(-mx[2], mx[2]),
            random.randint(-mx[3], mx[3]), random.randint(0, mx[4]), random.randint(0, mx[5])]


def main():
    ref_tok = import_reference_tokenizer()
    os.makedirs(OUT, exist_ok=True)
    for ver, cls in (("v1", ref_tok.MIDITokenizerV1), ("v2", ref_tok.MIDITokenizerV2)):
        tok = cls()
        rng = np.random.default_rng(20260926 + (1 if ver == "v2" else 0))
        styles = ["mid", "mixed", "drums", "low", "high", "no_notes", "mid", "mixed", "low", "high", "mixed", "mid"]
        arrays, offsets, shifts, maxima, changed = [], [0], [], [], []
        outs = []
        for ci, style in enumerate(styles):
            n = int(rng.integers(40, 400))
            seq = make_file(tok, rng, n, style)
            # default maxima for most cases; the last three use wider ones (incl. max_track_shift > 0)
            mx = [4, 10, 10, 10, 0, 16] if ci < len(styles) - 3 else [12, 40, 60, 200, 5, 16]
            seed = 1000 + 7 * ci
            sh = replay_shifts(seed, mx)
            random.seed(seed)
            out = tok.augment([list(r) for r in seq], *mx)
            random.seed(seed)
            out2 = tok.augment([list(r) for r in seq], *mx)
            assert out == out2
            a_in, a_out = np.asarray(seq, dtype=np.int16), np.asarray(out, dtype=np.int16)
            assert a_in.shape == a_out.shape
            arrays.append(a_in)
            outs.append(a_out)
            offsets.append(offsets[-1] + len(seq))
            shifts.append(sh)
            maxima.append(mx)
            changed.append(int((a_in != a_out).any()))
            print(f"{ver} case {ci:2d} {style:8s} n={len(seq):4d} shifts={sh} rows changed={(a_in != a_out).any(1).sum()}")
        assert 0 in changed and 1 in changed, "want both augmented and returned-unchanged files among the cases"
        np.savez_compressed(os.path.join(OUT, f"augment_{ver}.npz"), tokens=np.concatenate(arrays, 0),
                            augmented=np.concatenate(outs, 0), offsets=np.asarray(offsets, dtype=np.int64),
                            shifts=np.asarray(shifts, dtype=np.int32), maxima=np.asarray(maxima, dtype=np.int32),
                            changed=np.asarray(changed, dtype=np.int32))


if __name__ == "__main__":
    main()
