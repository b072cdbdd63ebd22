"""KV-cached decode driver for ``MIDIModel.generate`` (midi_model.py:167-250).

The reference spends a generated event in ~1000 launches and B host syncs.  Here one event is

    1 graph replay   net step: embedding sum of the previous event's 8 tokens -> 12 decoder layers on one position
                     per sequence -> final norm; K/V appended to preallocated caches at a position kept in DEVICE memory
    <=8 graph replays  token step i: (embedding of the token just sampled) -> 3 decoder layers at position i ->
                     lm_head -> grammar-masked softmax
    + the reference's own sampling ops after each token step (torch.sort / cumsum / multinomial, eager, so that a seeded
      torch.Generator is consumed exactly as the reference consumes it) and ONE device->host copy per event.

The graphs are hipGraphs captured through torch.cuda.CUDAGraph from the same Python schedule the eager path runs
(``engine.stack_decode``), so there is one implementation of the step.  A session owns every buffer the graphs touch and
is keyed by (parameter buffer, batch, capacity, temperature); ``MIDIModel`` keeps a pool of them and hands one to each
``generate`` call, so concurrent generators on one model (app.py:496) never share scratch.
"""
from __future__ import annotations

import os
from typing import List, Optional

import torch

from . import engine, ops
from .engine import KVState, RopeTable


def graphs_enabled(device: torch.device) -> bool:
    return device.type == "cuda" and os.environ.get("MH_DECODE_GRAPHS", "1") != "0"


class DecodeSession:
    def __init__(self, model, B: int, capacity: int, temp: float):
        tok = model.tokenizer
        self.model, self.B, self.cap, self.temp = model, B, capacity, float(temp)
        self.key = (model._flat.data_ptr(), model._flat.dtype, B, capacity, float(temp))
        dev, dt = model.device, model.dtype
        self.T, self.V, self.Vp = tok.max_token_seq, tok.vocab_size, model.vocab_padded
        spec, tspec = model._specs["net"], model._specs["net_token"]
        self.kv1 = KVState(spec, B, capacity, model._flat)
        self.kv2 = KVState(tspec, B, self.T, model._flat)
        # rope tables private to the session: a captured graph keeps their addresses
        self.rope1 = RopeTable(spec.hd, spec.theta, dev, capacity)
        self.rope2 = RopeTable(tspec.hd, tspec.theta, dev, self.T)
        first, lo, hi, _ = model._grammar()
        self.first_mask = first.clone()           # generate() overwrites it (ban_eos)
        self.lo_tab, self.hi_tab = lo, hi
        self.pos = torch.zeros(1, dtype=torch.int32, device=dev)           # cached events so far (device side)
        self.tokens_in = torch.zeros((B, self.T), dtype=torch.long, device=dev)
        self.hidden = torch.zeros((B, spec.D), dtype=dt, device=dev)
        self.samples_in = torch.zeros((B,), dtype=torch.long, device=dev)  # token sampled at the previous position
        self.ev = torch.zeros((B,), dtype=torch.long, device=dev)          # event id (token 0) of the current event
        self.neg1 = torch.full((B,), -1, dtype=torch.int32, device=dev)
        self.probs = torch.zeros((B, 1, self.V), dtype=torch.float32, device=dev)
        self.logits = torch.zeros((B, self.Vp), dtype=dt, device=dev)
        self.g_net = None
        self.g_tok: List[Optional[torch.cuda.CUDAGraph]] = [None] * self.T
        self.use_graphs = graphs_enabled(dev)
        if self.use_graphs:
            self._capture()

    # ---- the step bodies (run eagerly, or once under capture) ---------------------------------------------
    def _net_body(self):
        m = self.model
        spec = m._specs["net"]
        e = torch.empty((self.B, spec.D), dtype=m.dtype, device=m.device)
        ops.embed_sum_fwd(self.tokens_in, m._W["net"].embed, e)
        y = engine.stack_decode(spec, m._W["net"], e, self.rope1, self.kv1, pos_dev=self.pos)
        self.hidden.copy_(y)
        self.pos.add_(1)

    def _tok_body(self, i: int):
        m = self.model
        tspec, Wt = m._specs["net_token"], m._W["net_token"]
        x = self.hidden if i == 0 else Wt.embed.index_select(0, self.samples_in)
        self.kv2.len = i
        h = engine.stack_decode(tspec, Wt, x, self.rope2, self.kv2)
        ops.gemm_nt(h, m.lm_head.weight.data, self.logits[:, : self.V])
        if i == 0:
            lo, hi = self.neg1, self.neg1
        else:
            lo, hi = self.lo_tab[self.ev, i].contiguous(), self.hi_tab[self.ev, i].contiguous()
        ops.masked_softmax(self.logits, lo, hi, self.first_mask, self.probs.view(self.B, self.V), self.V, self.temp)

    def _capture(self):
        # one eager pass on a side stream first: lazy initialisation (kernel attributes, allocator pools) must not
        # happen inside a capture
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            self._net_body()
            for i in range(self.T):
                self._tok_body(i)
        cur.wait_stream(side)
        torch.cuda.synchronize()
        pool = torch.cuda.graph_pool_handle()
        self.g_net = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.g_net, pool=pool):
            self._net_body()
        for i in range(self.T):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=pool):
                self._tok_body(i)
            self.g_tok[i] = g
        self.reset()

    # ---- driver interface ---------------------------------------------------------------------------------
    def reset(self):
        self.kv1.len = 0
        self.kv2.len = 0
        self.pos.zero_()

    def prefill(self, tokens: torch.Tensor) -> None:
        """tokens [B, S, T]: causal forward over the prompt from an empty cache; leaves hidden = last position."""
        m = self.model
        spec = m._specs["net"]
        B, S, T = tokens.shape
        e = torch.empty((B * S, spec.D), dtype=m.dtype, device=m.device)
        ops.embed_sum_fwd(tokens.contiguous().view(B * S, T), m._W["net"].embed, e)
        self.kv1.len = 0
        y = engine.stack_prefill(spec, m._W["net"], e, B, S, self.rope1, self.kv1)
        self.hidden.copy_(y.view(B, S, spec.D)[:, -1])
        self.pos.fill_(S)

    def net_step(self, prev_event: torch.Tensor) -> None:
        """prev_event [B, T]: decode one position; hidden <- net output."""
        self.tokens_in.copy_(prev_event)
        if self.g_net is not None:
            self.g_net.replay()
        else:
            self._net_body()
        self.kv1.len += 1

    def tok_step(self, i: int) -> torch.Tensor:
        """token position i of the current event -> probs [B, 1, V] (a view of session memory)"""
        if self.g_tok[i] is not None:
            self.g_tok[i].replay()
        else:
            self._tok_body(i)
        return self.probs
