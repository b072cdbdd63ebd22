"""CPU ORACLE (test infrastructure, NOT product code).

A plain fp32 restatement, on torch CPU tensors, of the arithmetic the reference's hot path
executes.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import this file; the product path (``midi-model_amd/``) never does and fails loudly when
its HIP library is missing.

Where the arithmetic lives.  The reference (``/root/reference``) delegates all transformer math to
the un-vendored third-party package ``transformers`` (reference pin ``transformers>=4.36``,
requirements.txt:6; the copy in this image is 5.15.0) and the optimiser to ``torch``.  Each
function below cites the reference call site (``midi_model.py`` / ``train.py``) and the
``transformers`` source it restates (``TF:`` = transformers/, 5.15.0):

  rmsnorm            TF:models/llama/modeling_llama.py:62-67
  rope_tables/rope   TF:models/llama/modeling_llama.py:113-127 (inv_freq, cos/sin), :130-160 (rotate_half)
  attention          TF:models/llama/modeling_llama.py:243-281 + TF:integrations/sdpa_attention.py:79-166
  mlp                TF:models/llama/modeling_llama.py:174-176
  llama_stack        TF:models/llama/modeling_llama.py:295-324 (layer), :367-417 (model)
  midi_forward       midi_model.py:137-150
  midi_forward_token midi_model.py:116-135
  training_loss      train.py:168-188
  accuracy           train.py:153-166
  sample_top_p_k     midi_model.py:152-165
  generate           midi_model.py:167-250
  adamw_step / lr    train.py:93-103,121-151 + torch.optim.AdamW (single-tensor path), clip = Trainer(gradient_clip_val=1.0) train.py:464
  augment            midi_tokenizer.py:364-417 (v1), :1023-1102 (v2): the data augmentation MidiDataset.load_midi applies
                     (train.py:62-63), integer arithmetic on token ids; pinned to tests/golden/augment_v{1,2}.npz
                     (tests/gen_golden_augment.py ran the reference's own method)

PINNING.  The reference ships no tests, fixtures or golden vectors, so the oracle
is pinned against outputs of the reference itself: ``tests/gen_golden.py`` imports the real
``MIDIModel`` from /root/reference in the build container, runs it in fp32 on CPU and commits the
vectors under ``tests/golden/``; ``tests/test_oracle_golden.py`` checks every function here against
them.  Weights are a ``dict[str, Tensor]`` keyed exactly like the reference ``state_dict()``.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

Tensor = torch.Tensor
SD = Dict[str, Tensor]


# --------------------------------------------------------------------------------------------
# LLaMA block pieces
# --------------------------------------------------------------------------------------------
def rmsnorm(x: Tensor, w: Tensor, eps: float = 1e-6) -> Tensor:
    xf = x.float()
    var = xf.pow(2).mean(-1, keepdim=True)
    return w * (xf * torch.rsqrt(var + eps)).to(x.dtype)


def rope_tables(positions: Tensor, head_dim: int, theta: float = 10000.0) -> Tuple[Tensor, Tensor]:
    """cos/sin of pos * theta^(-2i/hd), duplicated to the full head_dim ("cat(freqs, freqs)")."""
    inv_freq = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.int64).float() / head_dim))
    freqs = positions.float()[:, None] * inv_freq[None, :]
    emb = torch.cat([freqs, freqs], dim=-1)
    return emb.cos(), emb.sin()


def rope(x: Tensor, cos: Tensor, sin: Tensor) -> Tensor:
    """x: (..., T, hd); half-split pairing (i, i+hd/2)."""
    half = x.shape[-1] // 2
    rot = torch.cat([-x[..., half:], x[..., :half]], dim=-1)
    return x * cos + rot * sin


# bench.py's cpu_baseline leg sets this: the attention of an un-cached forward then runs through
# torch.nn.functional.scaled_dot_product_attention(is_causal=True) -- the very call the reference makes
# (TF:integrations/sdpa_attention.py:79-166) -- instead of the explicit scores / softmax / product below, which
# materialises B x H x T x T scores and made the TIMED step 2x slower than the reference on the same cores
# (tools/cpu_reference_vs_port.py, profiles/r04_cpu_reference_vs_port.txt).  Parity tests keep the explicit form;
# tests/test_oracle_golden.py pins the fused form to it.
FUSED_SDPA = False


def attention(q: Tensor, k: Tensor, v: Tensor, causal: bool) -> Tensor:
    """q: (B,H,Tq,hd), k/v: (B,H,Tk,hd).  With a cache Tk >= Tq and query i sits at absolute
    position Tk-Tq+i.  The reference passes no padding mask, so the only mask is the causal one,
    applied when Tq > 1 (sdpa_attention.py:120)."""
    hd = q.shape[-1]
    if FUSED_SDPA and q.shape[-2] == k.shape[-2]:
        return torch.nn.functional.scaled_dot_product_attention(q, k, v, is_causal=causal and q.shape[-2] > 1)
    s = torch.matmul(q, k.transpose(-1, -2)) * (hd ** -0.5)
    tq, tk = q.shape[-2], k.shape[-2]
    if causal and tq > 1:
        qpos = torch.arange(tk - tq, tk)[:, None]
        kpos = torch.arange(tk)[None, :]
        s = s.masked_fill(kpos > qpos, float("-inf"))
    return torch.matmul(torch.softmax(s, dim=-1), v)


def mlp(x: Tensor, wg: Tensor, wu: Tensor, wd: Tensor) -> Tensor:
    return (torch.nn.functional.silu(x @ wg.T) * (x @ wu.T)) @ wd.T


class KV:
    """Minimal per-layer K/V store (what DynamicCache does with torch.cat, cache_utils.py:127-147)."""

    def __init__(self) -> None:
        self.k: List[Optional[Tensor]] = []
        self.v: List[Optional[Tensor]] = []

    def length(self) -> int:
        return 0 if not self.k or self.k[0] is None else self.k[0].shape[2]

    def update(self, layer: int, k: Tensor, v: Tensor) -> Tuple[Tensor, Tensor]:
        while len(self.k) <= layer:
            self.k.append(None)
            self.v.append(None)
        if self.k[layer] is None:
            self.k[layer], self.v[layer] = k, v
        else:
            self.k[layer] = torch.cat([self.k[layer], k], dim=2)
            self.v[layer] = torch.cat([self.v[layer], v], dim=2)
        return self.k[layer], self.v[layer]


def llama_stack(sd: SD, prefix: str, n_layer: int, n_head: int, x: Tensor, cache: Optional[KV] = None,
                eps: float = 1e-6, theta: float = 10000.0) -> Tensor:
    """inputs_embeds (B,T,D) -> last_hidden_state (B,T,D): pre-norm residual blocks + final norm."""
    B, T, D = x.shape
    hd = D // n_head
    past = cache.length() if cache is not None else 0
    cos, sin = rope_tables(torch.arange(past, past + T), hd, theta)
    for i in range(n_layer):
        p = f"{prefix}.layers.{i}."
        h = rmsnorm(x, sd[p + "input_layernorm.weight"], eps)
        q = (h @ sd[p + "self_attn.q_proj.weight"].T).view(B, T, n_head, hd).transpose(1, 2)
        k = (h @ sd[p + "self_attn.k_proj.weight"].T).view(B, T, n_head, hd).transpose(1, 2)
        v = (h @ sd[p + "self_attn.v_proj.weight"].T).view(B, T, n_head, hd).transpose(1, 2)
        q, k = rope(q, cos, sin), rope(k, cos, sin)
        if cache is not None:
            k, v = cache.update(i, k, v)
        a = attention(q, k, v, causal=True).transpose(1, 2).reshape(B, T, D)
        x = x + a @ sd[p + "self_attn.o_proj.weight"].T
        h = rmsnorm(x, sd[p + "post_attention_layernorm.weight"], eps)
        x = x + mlp(h, sd[p + "mlp.gate_proj.weight"], sd[p + "mlp.up_proj.weight"], sd[p + "mlp.down_proj.weight"])
    return rmsnorm(x, sd[f"{prefix}.norm.weight"], eps)


# --------------------------------------------------------------------------------------------
# MIDIModel
# --------------------------------------------------------------------------------------------
class Shape:
    """The numbers MIDIModelConfig.get_config derives (midi_model.py:63-76)."""

    def __init__(self, n_layer=12, n_head=16, n_embd=1024, n_inner=4096, vocab=3406, octet=8):
        self.n_layer, self.n_head, self.n_embd, self.n_inner = n_layer, n_head, n_embd, n_inner
        self.tok_layer, self.tok_head, self.tok_inner = n_layer // 4, n_head // 4, n_inner // 4
        self.vocab, self.octet, self.pad_id = vocab, octet, 0


def midi_forward(sd: SD, shp: Shape, x: Tensor, cache: Optional[KV] = None) -> Tensor:
    """x (B,S,8) int64 -> hidden (B,S,D): sum the octet's embeddings, run the event-level net."""
    # nn.Embedding(padding_idx=pad): same lookup, but the pad row receives no gradient (modeling_llama.py:353)
    e = torch.nn.functional.embedding(x, sd["net.embed_tokens.weight"], padding_idx=shp.pad_id).sum(dim=-2)
    return llama_stack(sd, "net", shp.n_layer, shp.n_head, e, cache)


def midi_forward_token(sd: SD, shp: Shape, hidden: Optional[Tensor], x: Optional[Tensor],
                       cache: Optional[KV] = None) -> Tensor:
    """hidden (N,D) and/or x (N,t) int64 -> logits (N,[1]+t,V)."""
    parts = []
    if hidden is not None:
        parts.append(hidden.unsqueeze(1))
    if x is not None:
        parts.append(torch.nn.functional.embedding(x, sd["net_token.embed_tokens.weight"], padding_idx=shp.pad_id))
    seq = torch.cat(parts, dim=1)
    h = llama_stack(sd, "net_token", shp.tok_layer, shp.tok_head, seq, cache)
    return h @ sd["lm_head.weight"].T


def training_loss(sd: SD, shp: Shape, batch: Tensor, pad_id: int = 0) -> Tuple[Tensor, Tensor]:
    """batch (B,S+1,8) -> (mean NLL over non-pad targets, logits (B*S,8,V))."""
    x, y = batch[:, :-1], batch[:, 1:]
    hidden = midi_forward(sd, shp, x).reshape(-1, shp.n_embd)
    y = y.reshape(-1, y.shape[-1])
    logits = midi_forward_token(sd, shp, hidden, y[:, :-1])
    loss = torch.nn.functional.cross_entropy(logits.reshape(-1, shp.vocab), y.reshape(-1),
                                             reduction="mean", ignore_index=pad_id)
    return loss, logits


def accuracy(logits: Tensor, labels: Tensor, pad_id: int = 0) -> Tensor:
    out = logits.argmax(-1).flatten()
    labels = labels.flatten()
    keep = labels != pad_id
    return (out[keep] == labels[keep]).float().sum() / keep.sum()


# --------------------------------------------------------------------------------------------
# optimiser (torch.optim.AdamW single-tensor semantics) + schedule + clip
# --------------------------------------------------------------------------------------------
def lr_lambda(step: int, warmup: float, max_step: float) -> float:
    if step < warmup:
        return float(step) / float(max(1, warmup))
    return max(0.0, float(max_step - step) / float(max(1, max_step - warmup)))


def decays(name: str) -> bool:
    """weight decay applies unless the parameter NAME contains 'bias' or 'norm' (train.py:123-131)."""
    return not any(nd in name for nd in ("bias", "norm"))


def clip_coef(grads: List[Tensor], max_norm: float = 1.0) -> Tuple[Tensor, Tensor]:
    total = torch.sqrt(sum(g.float().pow(2).sum() for g in grads))
    return torch.clamp(max_norm / (total + 1e-6), max=1.0), total


def adamw_step(p: Tensor, g: Tensor, m: Tensor, v: Tensor, step: int, lr: float, wd: float,
               b1: float = 0.9, b2: float = 0.99, eps: float = 1e-8) -> None:
    """In-place; ``step`` is 1-based."""
    p.mul_(1 - lr * wd)
    m.lerp_(g, 1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-(lr / bc1))


# --------------------------------------------------------------------------------------------
# sampling + generation
# --------------------------------------------------------------------------------------------
def sample_top_p_k(probs: Tensor, p: float, k: int, generator=None, noise: Optional[Tensor] = None) -> Tensor:
    """midi_model.py:152-165.  ``noise`` (same shape as probs): draw with these Exp(1) variates instead of a generator --
    torch.multinomial(num_samples=1) is ``argmax(p / q)`` with ``q = empty_like(p).exponential_(1)`` (aten/native
    Distributions.cpp, multinomial without replacement), applied by the reference to the SORTED probabilities, so variate j
    belongs to sorted rank j.  A test can hand the device sampler and this chain the same variates."""
    ps, idx = torch.sort(probs, dim=-1, descending=True)
    csum = torch.cumsum(ps, dim=-1)
    ps = ps.masked_fill(csum - ps > p, 0.0)
    keep = torch.zeros(ps.shape[-1])
    keep[:k] = 1
    ps = ps * keep
    ps = ps / ps.sum(dim=-1, keepdim=True)
    shape = ps.shape
    if noise is not None:
        nxt = (ps / noise).argmax(-1, keepdim=True)
    else:
        nxt = torch.multinomial(ps.reshape(-1, shape[-1]), num_samples=1, generator=generator).reshape(*shape[:-1], 1)
    return torch.gather(idx, -1, nxt).reshape(*shape[:-1])


def grammar_mask(tok, i: int, names: List[str], end: List[bool], ban_eos: bool = False, disable_patch_change: bool = False,
                 disable_control_change: bool = False, disable_channels=None) -> Tensor:
    """The (B, vocab) 0/1 mask of token position i (midi_model.py:202-214) with the serving loop's options
    (app.py:73-86): finished rows -> PAD; position 0 -> the event ids (+ EOS) minus disabled event types; position i ->
    the id range of the event's (i-1)-th parameter, a ``channel`` parameter minus the disabled channels; past the event's
    arity -> PAD."""
    banned = [tok.parameter_ids["channel"][c] for c in (disable_channels or [])]
    mask = torch.zeros((len(names), tok.vocab_size), dtype=torch.int64)
    for b in range(len(names)):
        if end[b]:
            mask[b, tok.pad_id] = 1
        elif i == 0:
            ids = list(tok.event_ids.values()) + ([] if ban_eos else [tok.eos_id])
            if disable_patch_change:
                ids.remove(tok.event_ids["patch_change"])
            if disable_control_change:
                ids.remove(tok.event_ids["control_change"])
            mask[b, ids] = 1
        else:
            pn = tok.events[names[b]]
            if i > len(pn):
                mask[b, tok.pad_id] = 1
            else:
                ids = list(tok.parameter_ids[pn[i - 1]])
                if pn[i - 1] == "channel":
                    ids = [t for t in ids if t not in banned]
                mask[b, ids] = 1
    return mask


def generate(sd: SD, shp: Shape, tok, prompt=None, batch_size=1, max_len=512, temp=1.0, top_p=0.98,
             top_k=20, generator=None, ban_eos: bool = False, disable_patch_change: bool = False,
             disable_control_change: bool = False, disable_channels=None, crop: Optional[int] = None) -> np.ndarray:
    """``tok`` supplies the vocabulary tables.  ``ban_eos`` (ours, throughput runs only) removes
    EOS from the first-token mask so every row runs to ``max_len``.  ``disable_*`` are the mask options of the serving
    loop (app.py:27-31, 73-86) and ``crop`` its prompt crop (``input_tensor[:, -4096:]``, app.py:53)."""
    T = tok.max_token_seq
    if prompt is None:
        inp = torch.full((batch_size, 1, T), tok.pad_id, dtype=torch.long)
        inp[:, 0, 0] = tok.bos_id
    else:
        prompt = np.asarray(prompt)
        if prompt.ndim == 2:
            prompt = np.repeat(prompt[None], batch_size, axis=0)
        elif prompt.shape[0] == 1:
            prompt = np.repeat(prompt, batch_size, axis=0)
        elif prompt.ndim != 3 or prompt.shape[0] != batch_size:
            raise ValueError(f"invalid shape for prompt, {prompt.shape}")
        prompt = prompt[..., :T]
        if prompt.shape[-1] < T:
            prompt = np.pad(prompt, ((0, 0), (0, 0), (0, T - prompt.shape[-1])), constant_values=tok.pad_id)
        inp = torch.from_numpy(prompt).long()
    if crop is not None:
        inp = inp[:, -crop:]
    cur, past = inp.shape[1], 0
    cache1 = KV()
    while cur < max_len:
        end = [False] * batch_size
        hidden = midi_forward(sd, shp, inp[:, past:], cache1)[:, -1]
        names = [""] * batch_size
        cache2 = KV()
        seq = None
        for i in range(T):
            mask = grammar_mask(tok, i, names, end, ban_eos, disable_patch_change, disable_control_change, disable_channels)
            if i == 0:
                logits = midi_forward_token(sd, shp, hidden, None, cache2)[:, -1:]
            else:
                logits = midi_forward_token(sd, shp, None, seq[:, -1:], cache2)[:, -1:]
            scores = torch.softmax(logits / temp, dim=-1) * mask.unsqueeze(1)
            samples = sample_top_p_k(scores, top_p, top_k, generator)
            if i == 0:
                seq = samples
                for b in range(batch_size):
                    if end[b]:
                        continue
                    eid = int(samples[b])
                    if eid == tok.eos_id:
                        end[b] = True
                    else:
                        names[b] = tok.id_events[eid]
            else:
                seq = torch.cat([seq, samples], dim=1)
                if all(len(tok.events[names[b]]) == i for b in range(batch_size) if not end[b]):
                    break
        if seq.shape[1] < T:
            seq = torch.nn.functional.pad(seq, (0, T - seq.shape[1]), value=tok.pad_id)
        inp = torch.cat([inp, seq.unsqueeze(1)], dim=1)
        past, cur = cur, cur + 1
        if all(end):
            break
    return inp.numpy()


# --------------------------------------------------------------------------------------------
# deterministic weights + synthetic event batches shared by tests / bench / golden generation
# --------------------------------------------------------------------------------------------
def state_dict_keys(shp: Shape) -> List[Tuple[str, Tuple[int, ...]]]:
    D, V = shp.n_embd, shp.vocab
    out: List[Tuple[str, Tuple[int, ...]]] = []

    def stack(prefix, n_layer, inner):
        out.append((f"{prefix}.embed_tokens.weight", (V, D)))
        for i in range(n_layer):
            p = f"{prefix}.layers.{i}."
            for nm in ("q", "k", "v", "o"):
                out.append((p + f"self_attn.{nm}_proj.weight", (D, D)))
            out.append((p + "mlp.gate_proj.weight", (inner, D)))
            out.append((p + "mlp.up_proj.weight", (inner, D)))
            out.append((p + "mlp.down_proj.weight", (D, inner)))
            out.append((p + "input_layernorm.weight", (D,)))
            out.append((p + "post_attention_layernorm.weight", (D,)))
        out.append((f"{prefix}.norm.weight", (D,)))

    stack("net", shp.n_layer, shp.n_inner)
    stack("net_token", shp.tok_layer, shp.tok_inner)
    out.append(("lm_head.weight", (V, D)))
    return out


def make_state_dict(shp: Shape, seed: int = 0, std: float = 0.02, pad_id: int = 0) -> SD:
    """Seeded test weights: N(0, std) matrices (pad row of each embedding zeroed, as HF's
    padding_idx init does), norm weights 1 + N(0, 0.1) so the norm scale is exercised."""
    g = torch.Generator().manual_seed(seed)
    sd: SD = {}
    for name, shape in state_dict_keys(shp):
        if len(shape) == 1:
            sd[name] = 1.0 + 0.1 * torch.randn(shape, generator=g)
        else:
            sd[name] = std * torch.randn(shape, generator=g)
            if "embed_tokens" in name:
                sd[name][pad_id].zero_()
    return sd


def synthetic_events(tok, batch: int, length: int, seed: int = 0, note_p: float = 0.90) -> Tensor:
    """(batch, length, 8) int64: row 0 = BOS octet, then events drawn as SURVEY.md §8(d) describes
    (type ~ Categorical(note 0.90, rest uniform), every parameter uniform over its id range)."""
    g = torch.Generator().manual_seed(seed)
    names = list(tok.events.keys())
    probs = torch.full((len(names),), (1.0 - note_p) / max(1, len(names) - 1))
    probs[names.index("note")] = note_p
    out = torch.full((batch, length, tok.max_token_seq), tok.pad_id, dtype=torch.long)
    out[:, 0, 0] = tok.bos_id
    n = batch * (length - 1)
    kinds = torch.multinomial(probs, n, replacement=True, generator=g)
    u = torch.rand((n, tok.max_token_seq - 1), generator=g)
    rows = torch.full((n, tok.max_token_seq), tok.pad_id, dtype=torch.long)
    for ki, name in enumerate(names):
        sel = (kinds == ki).nonzero().flatten()
        if sel.numel() == 0:
            continue
        rows[sel, 0] = tok.event_ids[name]
        for pos, pname in enumerate(tok.events[name], start=1):
            ids = tok.parameter_ids[pname]
            rows[sel, pos] = ids[0] + (u[sel, pos - 1] * len(ids)).long().clamp_(max=len(ids) - 1)
    out[:, 1:] = rows.reshape(batch, length - 1, tok.max_token_seq)
    return out


# --------------------------------------------------------------------------------------------
# data augmentation (integer work on token ids)
# --------------------------------------------------------------------------------------------
def augment(tok, seq: np.ndarray, shifts) -> np.ndarray:
    """``tokenizer.augment(midi_seq)`` (midi_tokenizer.py:364-417 for v1, :1023-1102 for v2) for one file, with the six
    random draws handed in: ``shifts = (pitch, velocity, cc_value, bpm, track, channel)`` in the order the reference draws
    them (:1025-1030).  ``seq`` is the file's token rows [n, T]; returns the augmented rows (same dtype).  Whole-array
    numpy, one masked assignment per rule:

      * every event: track <- (track + track_shift) mod n_track; channel <- (channel + channel_shift) mod n_channel,
        except that channel 9 stays 9 and a channel that LANDS on 9 goes to (9 + channel_shift) mod n_channel (:1041-1056)
      * note: pitch += pitch_shift unless the note's ORIGINAL channel is 9; velocity clamped to 1..127 (:1058-1070)
      * any note whose pitch leaves 0..127: the file comes back UNCHANGED (:1065-1066)
      * control_change: controllers 1, 2, 7, 11 get value += cc shift, clamped to 1..127 (:1075-1081)
      * set_tempo: bpm += bpm shift, clamped to 1..(n_bpm - 1) (:1082-1086)
      * key_signature (v2): key transposed by pitch_shift through sf2key / key2sf (:568-579, :1087-1097); second pass
        (:1099-1104): when the key signature's SHIFTED track is, among the ORIGINAL tracks of the file's notes, one whose
        notes all sit on channel 9, sf <- 0 (token sf_ids[7])
    """
    pitch_s, vel_s, cc_s, bpm_s, track_s, chan_s = (int(x) for x in shifts)
    a = np.asarray(seq).astype(np.int64)
    out = a.copy()
    pid, npar = tok.parameter_ids, tok.event_parameters
    names = {n: (a[:, 0] == eid) for n, eid in tok.event_ids.items()}

    def col(name, pn):
        return 1 + tok.events[name].index(pn)

    for name, rows in names.items():
        if not rows.any():
            continue
        if "track" in tok.events[name]:
            c = col(name, "track")
            out[rows, c] = pid["track"][0] + (a[rows, c] - pid["track"][0] + track_s) % npar["track"]
        if "channel" in tok.events[name]:
            c = col(name, "channel")
            c0 = a[rows, c] - pid["channel"][0]
            c1 = (c0 + chan_s) % npar["channel"]
            c1 = np.where(c0 == 9, 9, np.where(c1 == 9, (9 + chan_s) % npar["channel"], c1))
            out[rows, c] = pid["channel"][0] + c1
    note = names["note"]
    n_ch = a[note, col("note", "channel")] - pid["channel"][0]
    p = a[note, col("note", "pitch")] - pid["pitch"][0] + np.where(n_ch != 9, pitch_s, 0)
    if ((p < 0) | (p >= 128)).any():
        return np.asarray(seq).copy()
    out[note, col("note", "pitch")] = pid["pitch"][0] + p
    v = a[note, col("note", "velocity")] - pid["velocity"][0] + vel_s
    out[note, col("note", "velocity")] = pid["velocity"][0] + np.clip(v, 1, 127)
    cc = names["control_change"]
    ctrl = a[cc, col("control_change", "controller")] - pid["controller"][0]
    val = a[cc, col("control_change", "value")] - pid["value"][0]
    val = np.where(np.isin(ctrl, (1, 2, 7, 11)), np.clip(val + cc_s, 1, 127), val)
    out[cc, col("control_change", "value")] = pid["value"][0] + val
    st = names["set_tempo"]
    bpm = a[st, col("set_tempo", "bpm")] - pid["bpm"][0] + bpm_s
    out[st, col("set_tempo", "bpm")] = pid["bpm"][0] + np.clip(bpm, 1, npar["bpm"] - 1)
    if "key_signature" in names:
        ks = names["key_signature"]
        sf = a[ks, col("key_signature", "sf")] - pid["sf"][0] - 7
        mi = a[ks, col("key_signature", "mi")] - pid["mi"][0]
        k = ((sf * 7) % 12 + pitch_s) % 12                       # sf2key, transposed
        sf2 = (k * 7) % 12                                       # key2sf
        sf2 = np.where((sf2 > 6) | ((mi == 1) & (sf2 >= 5)), sf2 - 12, sf2) + 7
        # tracks (ORIGINAL numbering) whose notes all sit on the drum channel
        n_tr = a[note, col("note", "track")] - pid["track"][0]
        drum_only = np.array([t for t in np.unique(n_tr) if (n_ch[n_tr == t] == 9).all()], dtype=np.int64)
        ks_tr = out[ks, col("key_signature", "track")] - pid["track"][0]   # SHIFTED track of the key signature
        sf2 = np.where(np.isin(ks_tr, drum_only), 7, sf2)
        out[ks, col("key_signature", "sf")] = pid["sf"][0] + sf2
    return out.astype(np.asarray(seq).dtype)
